#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02l}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_edge_cases.py tests/test_gpu_configs_c4_c5.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
run() { name=$1; shift; env "$@" OPH_TRACE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder > $out/bench_$name.json 2> $out/bench_$name.err; echo "bench $name rc=$?"; }
run loop
run loop_nocone OPH_SKIP_CONE=1
run loop_legacycone OPH_NO_CONE_FUSED=1
run layers OPH_DECODE=layers
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/'+os.environ.get("OUT","r02l")+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],2), 'seq', d['config'].get('sequential_ms_per_step'), 'fp32', d['config'].get('all_fp32_ms_per_step'), 'h2h', d['config'].get('host_to_host_ms_per_step'))
        if 'kernel_classes' in d:
            print('   ', [(k['kernel'], k['launches'], k['avg_us']) for k in d['kernel_classes'] if k['launches']])
        if 'kernel_rooflines' in d:
            for k,v in d['kernel_rooflines'].items(): print('   ', k, round(v['avg_us'],1), 'us hbm_frac', round(v['hbm_frac'],3), 'mfma_frac', round(v['mfma_frac'],3))
    except Exception as e: print(f, 'ERR', e)
PY

#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02m}; mkdir -p $out
run() { name=$1; shift; env "$@" OPH_TRACE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder > $out/bench_$name.json 2> $out/bench_$name.err; echo "bench $name rc=$?"; }
run loop8
run loop8_nocone OPH_SKIP_CONE=1
run loop4 OPH_RUN_ROWS=4
run loop4_nocone OPH_RUN_ROWS=4 OPH_SKIP_CONE=1
run loop8_noskip_nocone OPH_LOOP_DBG=16 OPH_SKIP_CONE=1
run loop8_fusedcone OPH_CONE_FUSED=1
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/'+os.environ.get("OUT","r02m")+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],2), 'seq', round(d['config'].get('sequential_ms_per_step') or 0,2), 'fp32', round(d['config'].get('all_fp32_ms_per_step'),2), 'h2h', round(d['config'].get('host_to_host_ms_per_step'),2))
        if 'kernel_classes' in d:
            print('   ', [(k['kernel'], k['launches'], k['avg_us']) for k in d['kernel_classes'] if k['launches']])
    except Exception as e: print(f, 'ERR', e)
PY
OPH_SKIP_CONE=1 OPH_RUN_STAMPS=1 OPH_TRACE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline > $out/bench_stamps.json 2> $out/bench_stamps.err
grep "stamped step\|run 0 layer" $out/bench_stamps.err | tail -25
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -4

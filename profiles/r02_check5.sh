#!/bin/bash
# fused cone layers (cone_gemm_ln): parity + timing in loop / runs / layers decode modes
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02k}; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -8 $out/pytest.log
run() { name=$1; shift; env "$@" OPH_TRACE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder > $out/bench_$name.json 2> $out/bench_$name.err; echo "bench $name rc=$?"; }
run loop
run loop_nocone OPH_SKIP_CONE=1
run loop_legacycone OPH_NO_CONE_FUSED=1
run layers OPH_DECODE=layers
run layers_legacycone OPH_DECODE=layers OPH_NO_CONE_FUSED=1
OPH_RUN_STAMPS=1 OPH_TRACE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline > $out/bench_stamps.json 2> $out/bench_stamps.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/'+os.environ.get("OUT","r02k")+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],2), 'seq', d['config'].get('sequential_ms_per_step'), 'fp32', d['config'].get('all_fp32_ms_per_step'))
        if 'kernel_classes' in d:
            print('   ', [(k['kernel'], k['launches'], k['avg_us']) for k in d['kernel_classes'] if k['launches']])
    except Exception as e: print(f, 'ERR', e)
PY
grep -h "one launch" $out/bench_loop.err | tail -1
grep "signals of step\|stamped step" $out/bench_stamps.err | tail -4

#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02n}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_synthesize.py -m gpu -x -q 2>&1 | tail -4
run() { name=$1; shift; env "$@" OPH_TRACE=1 timeout 400 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder > $out/bench_$name.json 2> $out/bench_$name.err; echo "bench $name rc=$?"; }
run default
run nopreenc OPH_NO_PREENCODE=1
run layers OPH_DECODE=layers
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/'+os.environ.get("OUT","r02n")+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],2), 'seq', round(d['config'].get('sequential_ms_per_step') or 0,2), 'fp32', round(d['config'].get('all_fp32_ms_per_step'),2), 'h2h', round(d['config'].get('host_to_host_ms_per_step'),2))
    except Exception as e: print(f, 'ERR', e)
PY

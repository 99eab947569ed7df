#!/bin/bash
# round-2 first GPU check of the dec_run path
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
for v in run norun; do
  if [ $v = norun ]; then export OPH_NO_DECRUN=1; else unset OPH_NO_DECRUN; fi
  OPH_TRACE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder > gpurun_out/r02a/bench_$v.json 2> gpurun_out/r02a/bench_$v.err; echo "bench $v rc=$?"
  OPH_SKIP_CONE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > gpurun_out/r02a/bench_${v}_nocone.json 2> gpurun_out/r02a/bench_${v}_nocone.err; echo "bench $v nocone rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02a/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],2), 'seq', d['config'].get('sequential_ms_per_step'), 'fp32', d['config'].get('all_fp32_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
grep -h "decode loop" gpurun_out/r02a/bench_run.err | tail -3
grep -h "decode loop" gpurun_out/r02a/bench_norun.err | tail -3

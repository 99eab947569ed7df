#!/bin/bash
# dec_loop (whole decode in one launch): parity + timing
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02e}; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
for rows in 4 8; do
  export OPH_RUN_ROWS=$rows
  OPH_TRACE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > $out/bench_r$rows.json 2> $out/bench_r$rows.err; echo "bench rows=$rows rc=$?"
  OPH_SKIP_CONE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > $out/bench_r${rows}_nocone.json 2> $out/bench_r${rows}_nocone.err; echo "bench rows=$rows nocone rc=$?"
done
export OPH_RUN_ROWS=4
OPH_DECODE=runs timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > $out/bench_runs.json 2> $out/bench_runs.err; echo "bench runs rc=$?"
OPH_SKIP_CONE=1 OPH_RUN_STAMPS=1 OPH_TRACE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline > $out/bench_stamps.json 2> $out/bench_stamps.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/'+os.environ.get("OUT","r02e")+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],2), 'seq', d['config'].get('sequential_ms_per_step'), 'fp32', d['config'].get('all_fp32_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
grep -h "decode loop" $out/bench_r4.err | tail -2
grep -h "decode loop" $out/bench_r4_nocone.err | tail -2
grep "run [01] layer" $out/bench_stamps.err | tail -25

#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02i}; mkdir -p $out
for dbg in 0 15; do
  OPH_LOOP_DBG=$dbg OPH_SKIP_CONE=1 OPH_RUN_STAMPS=1 OPH_TRACE=1 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline > $out/b_$dbg.json 2> $out/b_$dbg.err
  echo "dbg=$dbg: $(grep -h 'stamped step' $out/b_$dbg.err | tail -1)  $(grep -h 'one launch' $out/b_$dbg.err | tail -1)"
  grep "run 0 layer  [79] \|run 0 layer 13\|run 0 layer 2[12] " $out/b_$dbg.err | tail -5
done
OPH_TRACE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > $out/bench_loop.json 2> $out/bench_loop.err; echo "bench loop rc=$?"
OPH_SKIP_CONE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > $out/bench_loop_nocone.json 2> $out/bench_loop_nocone.err; echo "bench nocone rc=$?"
OPH_DECODE=layers timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > $out/bench_layers.json 2> $out/bench_layers.err; echo "bench layers rc=$?"
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/'+os.environ.get("OUT","r02i")+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],2), 'seq', d['config'].get('sequential_ms_per_step'), 'fp32', d['config'].get('all_fp32_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
grep -h "one launch" $out/bench_loop.err | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -5

#!/bin/bash
# round 5: the pipelined cone (AQL queue) against the HIP-stream cone: same results (decode-mode / model tests), bench lines of both
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
OPH_HANG_DUMP_S=100 timeout 900 python -m pytest tests/test_gpu_decode_modes.py tests/test_gpu_model.py -m gpu -x -q --timeout 150 > gpurun_out/r05/pytest_aql.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05/pytest_aql.log
for v in aql:OPH_X=1 hip:OPH_NO_AQL=1; do
  name=${v%%:*}; envs=${v#*:}
  env $envs OPH_TRACE=1 timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > gpurun_out/r05/bench_$name.json 2> gpurun_out/r05/bench_$name.err; echo "bench $name rc=$?"
  grep -i "pipelined cone\|AQL" gpurun_out/r05/bench_$name.err | head -3
  python - $name <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "launch us", round(d["roofline"]["avg_launch_us"], 1), "host cores", d["config"]["rank_host_cores"], {k: round(v) for k, v in d["config"].items() if k.endswith("_value") and v})
PY
done

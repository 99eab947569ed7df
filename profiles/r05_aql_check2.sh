#!/bin/bash
# round 5: the cone through our own AQL queues -- OPH_AQL=1 (one lane, barrier bits, the plain kernels), OPH_AQL=2 (pipelined on two lanes) --
# against the HIP stream (OPH_AQL=0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for v in "aql1:OPH_AQL=1" "aql2:OPH_AQL=2" "hip:OPH_AQL=0" "aql1b:OPH_AQL=1" "hipb:OPH_AQL=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs OPH_TRACE=1 timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs > gpurun_out/r05/b_$name.json 2> gpurun_out/r05/b_$name.err; echo "bench $name rc=$?"
  python - $name <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05/b_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "launch us", round(d["roofline"]["avg_launch_us"], 1), "host cores", d["config"]["rank_host_cores"], "recov", d["config"]["recoveries"])
PY
done

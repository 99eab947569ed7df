#!/bin/bash
# round 5: the split cone (OPH_AQL=3: head + levels 1..2 on lane 0, levels 3..5 on lane 1 behind a packet-processor dependency)
# against the HIP stream (OPH_AQL=0) and the one-lane queue (OPH_AQL=1); optional: split points
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
LIST="hip:OPH_AQL=0 aql3:OPH_AQL=3 aql3s2:OPH_AQL=3,OPH_AQL_SPLIT=2 aql3s4:OPH_AQL=3,OPH_AQL_SPLIT=4 aql1:OPH_AQL=1 aql3b:OPH_AQL=3 hipb:OPH_AQL=0"
[ -n "$1" ] && LIST="$1"
for v in $LIST; do
  name=${v%%:*}; envs=${v#*:}; envs=${envs//,/ }
  env $envs OPH_TRACE=1 timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs > gpurun_out/r05/c_$name.json 2> gpurun_out/r05/c_$name.err; echo "bench $name rc=$?"
  python - $name <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r05/c_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "launch us", round(d["roofline"]["avg_launch_us"], 1), "host cores", d["config"]["rank_host_cores"], "recov", d["config"]["recoveries"])
except Exception as e:
    print(sys.argv[1], "no line:", e)
    print(open("gpurun_out/r05/c_%s.err" % sys.argv[1]).read()[-1500:])
PY
done

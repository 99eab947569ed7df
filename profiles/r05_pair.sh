#!/bin/bash
# round 5: hc_fused_pair (the cone's last two levels in one launch) against two hc_fused launches: bench lines alternated on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
LIST="pair:OPH_HC_PAIR=1 nopair:OPH_X=1 pair2:OPH_HC_PAIR=1 nopair2:OPH_X=1 pair3:OPH_HC_PAIR=1"
[ -n "$1" ] && LIST="$1"
for v in $LIST; do
  name=${v%%:*}; envs=${v#*:}; envs=${envs//,/ }
  env $envs OPH_TRACE=1 timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs > gpurun_out/r05/p_$name.json 2> gpurun_out/r05/p_$name.err; echo "bench $name rc=$?"
  python - $name <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r05/p_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "launch us", round(d["roofline"]["avg_launch_us"], 1), "host cores", d["config"]["rank_host_cores"], "recov", d["config"]["recoveries"])
except Exception as e:
    print(sys.argv[1], "no line:", e); print(open("gpurun_out/r05/p_%s.err" % sys.argv[1]).read()[-1500:])
PY
done

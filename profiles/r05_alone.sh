#!/bin/bash
# round 5: what bounds the step -- the whole-decode launch alone (no cone: OPH_SKIP_CONE, results wrong), without the streamed SSRN
# beside it, and the default; avg launch us / 200 = us per step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
LIST="default:OPH_X=1 nocone:OPH_SKIP_CONE=1 nocone_nossrn:OPH_SKIP_CONE=1,OPH_NO_STREAM_SSRN=1 nossrn:OPH_NO_STREAM_SSRN=1 default2:OPH_X=1"
[ -n "$1" ] && LIST="$1"
for v in $LIST; do
  name=${v%%:*}; envs=${v#*:}; envs=${envs//,/ }
  env $envs timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs > gpurun_out/r05/a_$name.json 2> gpurun_out/r05/a_$name.err; echo "bench $name rc=$?"
  python - $name <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r05/a_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "launch us", round(d["roofline"]["avg_launch_us"], 1), "per step", round(d["roofline"]["avg_launch_us"] / 200, 2), "recov", d["config"]["recoveries"])
except Exception as e:
    print(sys.argv[1], "no line:", e); print(open("gpurun_out/r05/a_%s.err" % sys.argv[1]).read()[-800:])
PY
done

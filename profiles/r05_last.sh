#!/bin/bash
# round 5: a bounded final SSRN piece (OPH_SSRN_LAST = frames left for after the decode; the frames in front of them go as one short chunk)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
LIST="base:OPH_X=1 last24:OPH_SSRN_LAST=24 last16:OPH_SSRN_LAST=16 last20:OPH_SSRN_LAST=20 last28:OPH_SSRN_LAST=28 base2:OPH_X=1 last12:OPH_SSRN_LAST=12"
[ -n "$1" ] && LIST="$1"
for v in $LIST; do
  name=${v%%:*}; envs=${v#*:}; envs=${envs//,/ }
  env $envs timeout 300 python bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > gpurun_out/r05/l_$name.json 2> gpurun_out/r05/l_$name.err; echo "bench $name rc=$?"
  python - $name <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05/l_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print(sys.argv[1], "value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "launch us", round(d["roofline"]["avg_launch_us"], 1), "gap", round(d["ms_per_step"] * 1e3 - d["roofline"]["avg_launch_us"]), {k: round(v) for k, v in c.items() if k.endswith("_value") and v})
PY
done

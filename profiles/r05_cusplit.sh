#!/bin/bash
# round 5: CU partition re-sweep with the faster chain (the cone is the step's bound now): chain,cone CUs (the rest: SSRN)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for sp in 64,128 64,144 64,152 64,160 64,128; do
  OPH_CU_SPLIT=$sp timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > gpurun_out/r05/cu_$sp.json 2> gpurun_out/r05/cu_$sp.err
  python - $sp <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05/cu_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "launch us", round(d["roofline"]["avg_launch_us"], 1), {k: round(v) for k, v in d["config"].items() if k.endswith("_value") and v})
PY
done

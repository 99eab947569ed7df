#!/bin/bash
# round 5: does a rocprofv3 kernel trace perturb the decode when the cone's ~1200 launches per batch go through our own AQL queue
# (OPH_AQL=1: packets written in one go, no host call per launch)?  traced vs un-traced, AQL vs HIP stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05
for m in 1 0; do
  rm -rf /tmp/tr_$m
  ( cd $R && OPH_AQL=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$m -o t -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs > gpurun_out/r05/traced_aql$m.json 2> gpurun_out/r05/traced_aql$m.err ); echo "traced aql=$m rc=$?"
  f=$(find /tmp/tr_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r05/kernel_stats_aql$m.csv
  ( cd $R && OPH_AQL=$m timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs > gpurun_out/r05/untraced_aql$m.json 2> gpurun_out/r05/untraced_aql$m.err ); echo "untraced aql=$m rc=$?"
  python - $m $R <<'PY'
import json, sys, csv
m, R = sys.argv[1], sys.argv[2]
for k in ("traced", "untraced"):
    try:
        d = json.loads(open("%s/gpurun_out/r05/%s_aql%s.json" % (R, k, m)).read().strip().splitlines()[-1])
        print("aql", m, k, "value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "launch us", round(d["roofline"]["avg_launch_us"], 1), "device clock", round(d["roofline"]["device_clock_us"], 1))
    except Exception as e:
        print("aql", m, k, "no line", e)
try:
    for r in csv.DictReader(open("%s/gpurun_out/r05/kernel_stats_aql%s.csv" % (R, m))):
        if "dec_chain" in r["Name"]: print("  rocprofv3 stats:", r["Name"][:40], "calls", r["Calls"], "avg ns", r["AverageNs"])
except Exception as e:
    print("  no stats", e)
PY
done

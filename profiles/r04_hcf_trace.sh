#!/bin/bash
# per-level hc_fused durations (rocprofv3 kernel trace, grouped by grid size) for the trees named in $1
cd /tmp && export TMPDIR=/tmp
for d in ${1:-"."}; do
  rm -rf /tmp/tr_$d; 
  ( cd $GRAFT_REPO_ROOT/$d && env $2 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$d -o t -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs > /dev/null 2>&1 )
  python - "$d" <<'PY'
import csv, sys, glob, collections, statistics
d = sys.argv[1]
f = glob.glob("/tmp/tr_%s/**/*kernel_trace.csv" % d, recursive=True)[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "hc_fused" in n or "cone_head" in n or "dec_chain" in n:
        g[(n.split("(")[0][-24:], r.get("Grid_Size_X", r.get("Grid_Size")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(g): print(d, k, len(g[k]), "avg %.2f us  med %.2f  min %.2f" % (statistics.mean(g[k]) / 1e3, statistics.median(g[k]) / 1e3, min(g[k]) / 1e3))
PY
done

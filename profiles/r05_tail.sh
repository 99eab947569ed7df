#!/bin/bash
# (round 5 copy: table kept in gpurun_out/r05/tail.txt)
# What runs between the end of a whole-decode launch and the start of the next one (host -> host batches): the SSRN final piece,
# copies, the next batch's set-up.  rocprofv3 kernel + memory-copy trace of a short bench run, listed for the third batch.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_tail
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr_tail -o t -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs > /dev/null 2>&1 )
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r05
python - > $GRAFT_REPO_ROOT/gpurun_out/r05/tail.txt <<'PY'
import csv, glob
kf = glob.glob("/tmp/tr_tail/**/*kernel_trace.csv", recursive=True)[0]
mf = glob.glob("/tmp/tr_tail/**/*memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(kf)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("oph::", "")[:44], r.get("Grid_Size_X", "")))
if mf:
    for r in csv.DictReader(open(mf[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), r.get("Size", r.get("Bytes", ""))))
ev.sort()
dec = [e for e in ev if e[2].startswith("dec_chain")]
d0, d1 = dec[2], dec[3]
print("decode launch %.1f us; gap to the next decode launch %.1f us" % ((d0[1] - d0[0]) / 1e3, (d1[0] - d0[1]) / 1e3))
last = d0[1]
for s, e, n, g in ev:
    if s >= d0[1] - 200000 and s < d1[0] and not n.startswith(("hc_fused", "cone_head", "dec_chain")):
        print("%9.1f us after decode end  +%7.1f us  %-46s %s" % ((s - d0[1]) / 1e3, (e - s) / 1e3, n, g))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/r05/tail.txt

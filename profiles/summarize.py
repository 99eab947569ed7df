#!/usr/bin/env python
"""Condenses the raw rocprofv3 outputs of profiles/collect_rNN.sh (gpurun_out/rNN/, scratch) into the
tracked summaries under profiles/:  rNN_kernel_stats.csv (rocprofv3 --stats, verbatim),
rNN_pmc_summary.csv (per-kernel mean FETCH_SIZE / WRITE_SIZE / MFMA counters) and rNN_traffic.json
(HBM bytes per launch per kernel, corrected as MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE is in KB
and reports 1/2 of the bytes of wide coalesced reads on gfx950 -> x2; WRITE_SIZE in KB, uncorrected).
usage: python profiles/summarize.py r01"""
import collections
import csv
import json
import os
import shutil
import statistics
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()


def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def find(sub, suffix):
    """rocprofv3 writes <dir>/<tag>_<suffix> or nests it under a host directory, depending on the version"""
    direct = os.path.join(src, sub, tag + "_" + suffix)
    if os.path.exists(direct):
        return direct
    for root, _, files in os.walk(os.path.join(src, sub)):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    raise SystemExit("no %s under %s" % (suffix, os.path.join(src, sub)))


shutil.copy(find("trace", "kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench.json")):
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
fetch = agg(find("pmc_fetch", "counter_collection.csv"))
write = agg(find("pmc_write", "counter_collection.csv"))
mfma = agg(find("pmc_mfma", "counter_collection.csv"))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(find("trace", "kernel_trace.csv"))):
    dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows, traffic = [], {}
for k in sorted(dur):
    f = statistics.mean(fetch[k]["FETCH_SIZE"]) if fetch[k]["FETCH_SIZE"] else 0.0
    w = statistics.mean(write[k]["WRITE_SIZE"]) if write[k]["WRITE_SIZE"] else 0.0
    hbm = 2.0 * f * 1024 + w * 1024
    traffic[k] = {"launches": len(dur[k]), "avg_us": statistics.mean(dur[k]) / 1e3, "fetch_size_kb_raw": f,
                  "write_size_kb_raw": w, "hbm_bytes_per_launch": hbm}
    mm = {c: statistics.mean(v) for c, v in mfma[k].items()}
    rows.append([k, len(dur[k]), "%.3f" % (statistics.mean(dur[k]) / 1e3), "%.1f" % f, "%.1f" % w, "%.0f" % hbm,
                 "%.0f" % mm.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), "%.0f" % mm.get("SQ_BUSY_CYCLES", 0),
                 "%.0f" % mm.get("GRBM_GUI_ACTIVE", 0), "%.0f" % mm.get("SQ_LDS_BANK_CONFLICT", 0),
                 "%.0f" % mm.get("SQ_LDS_IDX_ACTIVE", 0)])
with open(os.path.join(dst, tag + "_pmc_summary.csv"), "w") as fo:
    wr = csv.writer(fo)
    wr.writerow(["kernel", "dispatches", "avg_duration_us", "FETCH_SIZE_KB_raw_mean", "WRITE_SIZE_KB_raw_mean",
                 "hbm_bytes_per_launch(2*FETCH+WRITE)", "SQ_VALU_MFMA_BUSY_CYCLES_mean", "SQ_BUSY_CYCLES_mean", "GRBM_GUI_ACTIVE_mean",
                 "SQ_LDS_BANK_CONFLICT_mean", "SQ_LDS_IDX_ACTIVE_mean"])
    wr.writerows(rows)
cmd_file = os.path.join(src, "command.txt")
command = open(cmd_file).read().strip() if os.path.exists(cmd_file) else "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile"
json.dump({"tag": tag, "command": command,
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
           "kernels": traffic}, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
print("wrote", [f for f in os.listdir(dst) if f.startswith(tag)])

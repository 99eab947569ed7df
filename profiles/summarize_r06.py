#!/usr/bin/env python
"""Condenses the raw rocprofv3 outputs of profiles/collect_r06.sh (gpurun_out/r06c/, scratch) into the tracked summaries:
  r06_kernel_stats.csv      rocprofv3 --stats of pass A (the default bench command, every dispatch traced)
  r06_kernel_stats_all.csv  rocprofv3 --stats of pass B (every kernel, sequential batches)
  r06_witness.json          the whole-decode launch by three witnesses, TRACED (pass A's own line) and UN-TRACED (pass D's line)
  r06_pmc_summary.csv       per kernel: dispatches, mean duration, FETCH_SIZE / WRITE_SIZE (KB, raw means), corrected HBM bytes
  r06_sq_summary.csv        per kernel: the SQ busy / stall / LDS counters (means per dispatch) and derived fractions
  r06_traffic.json          HBM bytes per launch per kernel -- what bench.py quotes as roofline.traffic
Corrections as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE is in KB and reports 1/2 of the bytes of wide coalesced
reads on gfx950 -> x2; WRITE_SIZE in KB, uncorrected."""
import collections
import csv
import json
import os
import shutil
import statistics

tag = "r06"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", tag + "c")
dst = os.path.join(ROOT, "profiles")


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("oph::", "").strip()


def find(sub, suffix):
    for root, _, files in os.walk(os.path.join(src, sub)):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    return None


def agg(sub):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    path = find(sub, "counter_collection.csv")
    if path:
        for r in csv.DictReader(open(path)):
            d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def durations(sub):
    dur = collections.defaultdict(list)
    path = find(sub, "kernel_trace.csv")
    if path:
        rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))       # dispatch order
        for r in rows:
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return dur


for sub, out in (("traceA", "_kernel_stats.csv"), ("traceB", "_kernel_stats_all.csv")):
    p = find(sub, "kernel_stats.csv")
    if p:
        shutil.copy(p, os.path.join(dst, tag + out))
durB = durations("traceB")
durA = durations("traceA")
fetch = {"loop": agg("loop_fetch"), "runs": agg("runs_fetch")}
write = {"loop": agg("loop_write"), "runs": agg("runs_write")}
sq, lds = agg("runs_sq"), agg("runs_lds")
kernels = sorted(set(durB) | set(durA))
rows, traffic = [], {}
for k in kernels:
    which = "loop" if k.startswith(("dec_loop", "dec_chain")) else "runs"
    f = statistics.mean(fetch[which][k]["FETCH_SIZE"]) if fetch[which][k]["FETCH_SIZE"] else 0.0
    w = statistics.mean(write[which][k]["WRITE_SIZE"]) if write[which][k]["WRITE_SIZE"] else 0.0
    d = durA[k] if k.startswith(("dec_loop", "dec_chain")) and durA[k] else durB[k]
    hbm = 2.0 * f * 1024 + w * 1024
    traffic[k] = {"launches": len(d), "avg_us": statistics.mean(d) / 1e3 if d else 0.0, "fetch_size_kb_raw": f, "write_size_kb_raw": w,
                  "hbm_bytes_per_launch": hbm, "counter_pass": "measurement build, option LOOP_ALONE (loop kernel alone)" if which == "loop" else "option DECODE=runs"}
    rows.append([k, len(d), "%.3f" % (statistics.mean(d) / 1e3 if d else 0.0), "%.1f" % f, "%.1f" % w, "%.0f" % hbm, traffic[k]["counter_pass"]])
with open(os.path.join(dst, tag + "_pmc_summary.csv"), "w") as fo:
    wr = csv.writer(fo)
    wr.writerow(["kernel", "dispatches(trace)", "avg_duration_us(trace)", "FETCH_SIZE_KB_raw_mean", "WRITE_SIZE_KB_raw_mean",
                 "hbm_bytes_per_launch(2*FETCH+WRITE)", "counter_pass"])
    wr.writerows(rows)
names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
         "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS"]
with open(os.path.join(dst, tag + "_sq_summary.csv"), "w") as fo:
    wr = csv.writer(fo)
    wr.writerow(["kernel", "dispatches"] + [n + "_mean" for n in names] +
                ["wait_any/wave_cycles", "wait_inst_any/wave_cycles", "active_inst/wave_cycles", "lds_conflict/lds_active"])
    for k in sorted(set(sq) | set(lds)):
        m = {n: (statistics.mean(sq[k][n]) if sq[k][n] else (statistics.mean(lds[k][n]) if lds[k][n] else 0.0)) for n in names}
        wc = m["SQ_WAVE_CYCLES"] or 1.0
        n_disp = len(sq[k]["SQ_WAVE_CYCLES"]) or len(lds[k]["SQ_LDS_IDX_ACTIVE"])
        wr.writerow([k, n_disp] + ["%.0f" % m[n] for n in names] +
                    ["%.3f" % (m["SQ_WAIT_ANY"] / wc), "%.3f" % (m["SQ_WAIT_INST_ANY"] / wc), "%.3f" % (m["SQ_ACTIVE_INST_ANY"] / wc),
                     "%.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"] if m["SQ_LDS_IDX_ACTIVE"] else 0.0)])
# the three witnesses of the whole-decode launch on the same runs: rocprofv3's kernel trace (all launches, and the launches of the
# timed region = the last `steps` of them: the warm-up batches come first), the HIP events and the kernel's own clock of the bench
# line that same traced run printed
def witness(sub, line_file):
    try:
        line = json.loads(open(os.path.join(src, line_file)).read().strip().splitlines()[-1])
    except Exception:
        return None
    d = [x for k, v in durations(sub).items() if k.startswith(("dec_chain", "dec_loop")) for x in v]
    if not d:
        return None
    steps = int(line["steps"])
    r = line["roofline"]
    w = {"rocprofv3_all_launches": len(d), "rocprofv3_avg_us_all_launches": statistics.mean(d) / 1e3,
         "rocprofv3_avg_us_timed_region": statistics.mean(d[-steps:]) / 1e3, "timed_launches": steps,
         "hip_events_avg_us": r.get("avg_launch_us"), "device_clock_avg_us": r.get("device_clock_us")}
    ref = w["rocprofv3_avg_us_timed_region"]
    w["max_relative_spread"] = max(abs(w[k] - ref) / ref for k in ("hip_events_avg_us", "device_clock_avg_us") if w[k])
    return w


def untraced(line_file):
    """the same build and box WITHOUT a profiler: HIP events and the kernel's own clock of the bench line (there is no third witness
    without a tracer: that is the point -- the tracer delays the ~1200 cone dispatches per batch the decode waits for)"""
    try:
        line = json.loads(open(os.path.join(src, line_file)).read().strip().splitlines()[-1])
    except Exception:
        return None
    r = line["roofline"]
    w = {"timed_launches": int(line["steps"]), "hip_events_avg_us": r.get("avg_launch_us"), "device_clock_avg_us": r.get("device_clock_us"),
         "ms_per_step": line.get("ms_per_step"), "value": line.get("value")}
    if w["hip_events_avg_us"] and w["device_clock_avg_us"]:
        w["relative_spread"] = abs(w["hip_events_avg_us"] - w["device_clock_avg_us"]) / w["device_clock_avg_us"]
    return w


json.dump({"traced_default_run": witness("traceA", "bench_traced.json"), "untraced_same_command": untraced("bench_untraced.json"),
           "untraced_default_bench": untraced("bench.json"),
           "note": "durations() keeps dispatch order per kernel; the bench's timed region is its last `steps` whole-decode launches.  The traced run is "
                   "slower than the un-traced one because rocprofv3 intercepts every dispatch of the cone's ~1200 launches per batch, which the "
                   "whole-decode launch waits for; within each run the witnesses agree"},
          open(os.path.join(dst, tag + "_witness.json"), "w"), indent=1)
command = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else ""
json.dump({"tag": tag, "command": command,
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
           "kernels": traffic}, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
print("wrote", sorted(f for f in os.listdir(dst) if f.startswith(tag)))

"""Stand-alone SSRN on the whole chip (oph_ssrn on a host mel array: no decode beside it, no streaming), for a kernel trace:
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <out> -o s -- python profiles/r03_ssrn_layers.py
   python profiles/r03_ssrn_layers.py --summarize <out>/.../s_kernel_trace.csv     (per-dispatch table of the LAST of the runs)"""
import csv
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    rows = [r for r in csv.DictReader(open(sys.argv[2]))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    runs, cur = [], []
    for r in rows:
        n = r["Kernel_Name"]
        if "embed" in n or "dec_" in n or "cone" in n:
            continue
        if "copy_rows" in n or ("conv_gemm" in n and cur and "ln_rows" not in cur[-1]["Kernel_Name"] and len(cur) > 30):
            pass
        cur.append(r)
    # the runs are separated by the H2D copy of Y: split where the gap between dispatches exceeds 200 us
    out, last_end = [[]], None
    for r in cur:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if last_end is not None and s - last_end > 200000:
            out.append([])
        out[-1].append(r)
        last_end = e
    run = out[-1]
    t0 = int(run[0]["Start_Timestamp"])
    tot = 0
    for r in run:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        tot += e - s
        print("%8.1f us  +%7.1f  %-60s grid %s" % ((e - s) / 1e3, (s - t0) / 1e3, r["Kernel_Name"][:60], r.get("Grid_Size", "?")))
    print("kernels %.3f ms, span %.3f ms, %d dispatches" % (tot / 1e6, (int(run[-1]["End_Timestamp"]) - t0) / 1e6, len(run)))
    sys.exit(0)

import numpy as np                              # noqa: E402
import bench                                    # noqa: E402
from ophelia_amd.engine import Engine           # noqa: E402
from ophelia_amd import weights as WT           # noqa: E402

hp = bench.load_hp()
eng = Engine(hp, device=0)
eng.load_weights(WT.random_weights(eng.inventory(), seed=2))
Y = np.random.default_rng(0).random((16, hp.max_T, hp.n_mels), dtype=np.float32)
for i in range(4):
    t0 = time.perf_counter()
    Z = eng.ssrn(Y)
    print("ssrn call %d: %.2f ms (with H2D of Y and D2H of Z)" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
    time.sleep(0.01)
eng.close()

"""One 16-utterance batch of the bench workload with the decode kernels' phase stamps (OPH_RUN_STAMPS=1 OPH_TRACE=1) and the
pipeline counters: which decode flavour ran, how long the chain waited for each cone level, per-layer phases.
usage: OPH_TRACE=1 OPH_RUN_STAMPS=1 python profiles/r03_probe.py [n_batches]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from ophelia_amd.engine import Engine           # noqa: E402
from ophelia_amd import weights as WT           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
hp = bench.load_hp()
eng = Engine(hp, device=0)
eng.load_weights(WT.random_weights(eng.inventory(), seed=2))
L, ends = bench.synth_text(hp, 16, seed=3)
eng.stage_text(L, ends)
eng.profile_enable(2)                       # one HIP event pair per whole-decode launch
for i in range(n):
    eng.profile_reset()
    t0 = time.perf_counter()
    steps = eng.run_resident(stop_mode=1, run_ssrn=True, pipelined=False)
    t1 = time.perf_counter()
    eng.synchronize()
    loop = next((p for p in eng.profile() if p["name"] == "dec_loop" and p["launches"] > 0), None)
    print("batch %d: %d steps, %.2f ms (call returned after %.2f ms; dec_loop %.2f ms by HIP events)"
          % (i, steps, (time.perf_counter() - t0) * 1e3, (t1 - t0) * 1e3, loop["total_ms"] / loop["launches"] if loop else -1), flush=True)
print("counters", eng.counters())
eng.close()

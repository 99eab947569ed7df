"""How many SSRN chunks stream under a host -> host decode, and what the call costs, per chunk size (OPH_SSRN_CHUNK)."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(float(os.environ.get("OPH_HANG_DUMP_S", "45")), exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as BN
from ophelia_amd.engine import Engine
from ophelia_amd import weights as WT
hp = BN.load_hp()
eng = Engine(hp, device=0)
eng.load_weights(WT.random_weights(eng.inventory(), seed=2))
texts = [BN.synth_text(hp, 16, seed=s) for s in (3, 4)]
eng.stage_text(*texts[0]); eng.stage_text_next(*texts[1])
for i in range(3):
    eng.run_host(stop_mode=1, want_kv=True); eng.stage_text_next(*texts[i & 1])
eng.synchronize()
c0 = eng.counters()
t0 = time.perf_counter()
n = 20
for i in range(n):
    eng.run_host(stop_mode=1, want_kv=True); eng.stage_text_next(*texts[(i + 1) & 1])
eng.synchronize()
dt = (time.perf_counter() - t0) / n * 1e3
c1 = eng.counters()
print("OPH_SSRN_CHUNK=%s: %.3f ms per call, %.2f chunks streamed per decode" % (os.environ.get("OPH_SSRN_CHUNK", "40"), dt, (c1["chunks_streamed"] - c0["chunks_streamed"]) / n))

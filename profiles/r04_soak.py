"""Repeatability soak of the host -> host call with plane_gemm in SSRN / TextEnc: the same text N times through oph_run_host (SSRN
streams in chunks on its CU partition under the decode, the TextEnc of the staged next text beside it); every K, V, Y, alignment
and Z must be bitwise equal to the first call's.  A race in the hand-written LDS ring would show as a differing Z."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench as BN
from ophelia_amd.engine import Engine
from ophelia_amd import weights as WT
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
hp = BN.load_hp()
eng = Engine(hp, device=0)
eng.load_weights(WT.random_weights(eng.inventory(), seed=2))
text = BN.synth_text(hp, 16, seed=3)
eng.stage_text(*text); eng.stage_text_next(*text)
ref = None
bad = 0
for i in range(N):
    out = eng.run_host(stop_mode=1, want_kv=True)
    eng.stage_text_next(*text)
    arrs = [np.asarray(out[k]) for k in sorted(out) if isinstance(out[k], np.ndarray)]
    h = [hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12] for a in arrs]
    if ref is None:
        ref = h
        print("call 0:", [a.shape for a in arrs], h, flush=True)
    elif h != ref:
        bad += 1
        print("call %d differs:" % i, h, flush=True)
print("%d calls, %d differing" % (N, bad))
sys.exit(1 if bad else 0)

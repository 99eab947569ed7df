"""The drop-in session calls on a batch of more than 16 utterances (tiles): encode_text -> text2mel -> ssrn wall times.
usage: python profiles/r03_api_batch.py [B]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from ophelia_amd.engine import Engine           # noqa: E402
from ophelia_amd import weights as WT           # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hp = bench.load_hp()
eng = Engine(hp, device=0)
eng.load_weights(WT.random_weights(eng.inventory(), seed=2))
for rep in range(3):
    L, ends = bench.synth_text(hp, B, seed=3 + rep)
    t0 = time.perf_counter()
    K, V = eng.encode_text(L)
    t1 = time.perf_counter()
    Y, t_ends, al, steps = eng.text2mel(K, V, ends, stop_mode=1)
    t2 = time.perf_counter()
    Z = eng.ssrn(Y)
    t3 = time.perf_counter()
    print("B=%d: encode %.2f ms, text2mel %.2f ms (%d steps), ssrn %.2f ms, total %.2f ms = %.0f frames/s" %
          (B, (t1 - t0) * 1e3, (t2 - t1) * 1e3, steps, (t3 - t2) * 1e3, (t3 - t0) * 1e3, B * steps / (t3 - t0)), flush=True)
eng.close()

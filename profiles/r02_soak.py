"""Soak of the whole-decode launch + side-stream cone protocol: many decodes on one handle with the reference's early stop
(random text lengths, so the stop step varies), alternating with fixed-length ones and with pipelined resident batches;
every decode is compared with the first decode of the same inputs (bitwise) -- a lost signal or a stale cone row shows up
as a mismatch, a protocol dead-lock as the bounded-spin error.  usage: python profiles/r02_soak.py [rounds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import hp_from_snapshot
from oracle import ophelia_oracle as O          # seeded weights / texts only
from ophelia_amd.engine import Engine

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
hp = hp_from_snapshot("lj_tutorial.cfg")
W = O.random_weights(hp, 2)
eng = Engine(hp, device=0)
eng.load_weights(W)
cases = []
for i, (B, lo, hi) in enumerate([(16, 6, 30), (16, 75, 149), (5, 10, 40), (11, 4, 12), (16, 20, 60)]):
    L = O.random_text(hp, B, 100 + i, min_len=lo, max_len=hi)
    ends = O.get_text_lengths(L)
    K, V = eng.encode_text(L)
    ref = {}
    for sm in (0, 1):
        ref[sm] = eng.text2mel(K, V, ends, stop_mode=sm)
    cases.append((L, ends, K, V, ref))
    print("case %d: B=%d stop step %d" % (i, B, ref[0][3]))
t0 = time.time()
n = 0
for r in range(rounds):
    for ci, (L, ends, K, V, ref) in enumerate(cases):
        sm = (r + ci) & 1
        Y, t_ends, al, steps = eng.text2mel(K, V, ends, stop_mode=sm)
        Y0, t0_, al0, s0 = ref[sm]
        assert steps == s0 and np.array_equal(t_ends, t0_), (r, ci, sm, steps, s0)
        assert np.array_equal(Y, Y0) and np.array_equal(al, al0), "decode %d of case %d differs from its first run" % (r, ci)
        n += 1
    if r % 8 == 7:                                   # pipelined resident batches in between (SSRN + pre-encode overlap)
        L, ends = cases[1][0], cases[1][1]
        eng.stage_text(L, ends)
        for _ in range(4):
            assert eng.run_resident(stop_mode=1, run_ssrn=True, pipelined=True) == hp.max_T
        eng.synchronize()
        Yp, tp, alp = eng.fetch_mel()
        Yr = cases[1][4][1][0]
        assert np.array_equal(Yp, Yr), "pipelined resident batch differs from the host-buffer decode of the same text"
print("soak: %d decodes identical to their first run, %.1f s" % (n, time.time() - t0))
eng.close()

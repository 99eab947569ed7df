"""Diagnostic: is an utterance's result independent of the ROW of the 16-row tile it occupies and of its batch mates?  (tests/
test_gpu_properties.py::test_deterministic_and_batch_independent holds the property; this prints where it breaks.)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import hp_from_snapshot
from oracle import ophelia_oracle as O
from ophelia_amd.engine import Engine
hp = hp_from_snapshot("lj_tutorial.cfg", max_T=int(os.environ.get("MT", "12")))
W = O.random_weights(hp, 2)
eng = Engine(hp, device=0); eng.load_weights(W)
L = O.random_text(hp, 16, 5, min_len=75, max_len=149); ends = O.get_text_lengths(L)
K, V = eng.encode_text(L); Y, _, al, _ = eng.text2mel(K, V, ends, stop_mode=1); Y = np.array(Y)
for sh in (1, 2, 4, 8, 3):
    P = np.roll(np.arange(16), sh)
    K2, V2 = eng.encode_text(L[P]); Y2, _, _, _ = eng.text2mel(K2, V2, ends[P], stop_mode=1); Y2 = np.array(Y2)
    d = np.abs(Y2 - Y[P])
    rows = (d.max(axis=(1, 2)) > 0).astype(int)
    first = [int(np.nonzero(d[r].max(axis=1))[0][0]) if rows[r] else -1 for r in range(16)]
    print("roll %d: rows that differ %s first step %s" % (sh, rows.tolist(), first))
    if sh == 1 and rows.any():
        r = int(np.nonzero(rows)[0][0]); t = first[r]
        ch = np.nonzero(d[r, t])[0]
        print("   row %d step %d: %d of 80 channels differ, e.g. %s, max %.2e" % (r, t, len(ch), ch[:12].tolist(), d[r, t].max()))

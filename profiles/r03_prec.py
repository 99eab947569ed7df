"""Accuracy of the split contractions on the bench workload (lj_tutorial, B = 16, 200 steps, seeded random weights):
SSRN in fp32 MFMA / split-bf16 x3 / split-fp16 x3 against the CPU oracle and against each other; Text2Mel with the cone's
two many-row contractions in the flavour OPH_CONE_PREC selects (one process per flavour; `compare` prints the differences).
usage: OPH_CONE_PREC=k python profiles/r03_prec.py run out.npz | python profiles/r03_prec.py compare a.npz b.npz ..."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402

if sys.argv[1] == "run":
    from ophelia_amd.engine import Engine
    from ophelia_amd import weights as WT
    hp = bench.load_hp()
    eng = Engine(hp, device=0)
    W = WT.random_weights(eng.inventory(), seed=2)
    eng.load_weights(W)
    L, ends = bench.synth_text(hp, 16, seed=3)
    K, V = eng.encode_text(L)
    Y, t_ends, al, steps = eng.text2mel(K, V, ends, stop_mode=1)
    Yc = np.array(Y)
    Z = {}
    for mode in (0, 1, 2, 3, 4):        # 3 / 4: split-fp16 with 2 products (weights hi only) / 1 product -- measurement only
        eng.set_ssrn_precision(mode)
        eng.ssrn(Yc)
        t0 = time.perf_counter()
        Z[mode] = eng.ssrn(Yc)
        print("ssrn mode %d: %.2f ms host-to-host" % (mode, (time.perf_counter() - t0) * 1e3))
    print("SSRN vs fp32 MFMA: bf16x3 %.3e  fp16x3 %.3e  fp16x2 %.3e  fp16x1 %.3e (max-abs on mag)" %
          tuple(np.abs(Z[k] - Z[0]).max() for k in (1, 2, 3, 4)))
    import ctypes as C
    for name, T in (("D4", hp.max_T), ("D7", 2 * hp.max_T)):
        for prec in (0, 2, 3, 4):
            us, by, fl = C.c_double(), C.c_double(), C.c_double()
            eng.lib.oph_bench_conv1d_transpose(0, 16, T, hp.c, hp.c, prec, 3, 20, C.byref(us), C.byref(by), C.byref(fl))
            print("conv1d_transpose %s precision mode %d: %.1f us = %.1f %% of the HBM roofline" % (name, prec, us.value, by.value / us.value / 1e3 / 8000 * 100))
    if len(sys.argv) > 3 and sys.argv[3] == "oracle":
        from oracle import cpu_oracle
        m = cpu_oracle.CpuModel(hp, W, threads=min(cpu_oracle.usable_cores(), 32))
        Z0 = m.ssrn(Yc[:4])
        for mode in (0, 1, 2, 3, 4):
            print("SSRN mode %d vs the CPU oracle (4 utterances): %.3e" % (mode, np.abs(Z[mode][:4] - Z0).max()))
    np.savez(sys.argv[2], Y=Y, al=al, trace=al.argmax(1))
    eng.close()
else:
    ref = np.load(sys.argv[2])
    for f in sys.argv[3:]:
        g = np.load(f)
        print("%s vs %s: mel max-abs %.3e, alignments %.3e, attention trace identical: %s" %
              (os.path.basename(f), os.path.basename(sys.argv[2]), np.abs(g["Y"] - ref["Y"]).max(), np.abs(g["al"] - ref["al"]).max(),
               bool(np.array_equal(g["trace"], ref["trace"]))))

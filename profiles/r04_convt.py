"""conv1d_transpose (SSRN D_4 / D_7) timing through oph_bench_conv1d_transpose: fp32-operand MFMA (prec 0), split-bf16 (1), split-fp16 (2: plane_gemm; 5: round-3 launches; 6-9: ablation builds without MFMAs / operand stream / stores / with three K blocks):
even- and odd-phase contractions in one launch + LayerNorm rows."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ophelia_amd import _lib
lib = _lib.load()
for name, T in (("D_4", 200), ("D_7", 400)):
    for prec in (0, 1, 2, 5, 6, 7, 8, 9):      # 2: planes in, planes out (plane_gemm + ln_rows); 5: the round-3 launches (rows split inside the paired contraction)
        us, by, fl = C.c_double(), C.c_double(), C.c_double()
        rc = lib.oph_bench_conv1d_transpose(0, 16, T, 512, 512, prec, 5, 50, C.byref(us), C.byref(by), C.byref(fl))
        if rc != 0:
            print(name, "prec", prec, "rc", rc, lib.oph_op_last_error().decode())
            continue
        print(name, "prec", prec, "rc", rc, "%.1f us  %.1f%% HBM  %.0f TFLOP/s (x%d products)" % (us.value, by.value / us.value / 1e3 / 8000 * 100, fl.value / us.value / 1e6, 1 if prec == 0 else 3), lib.oph_op_last_error().decode() if rc else "")

#!/bin/bash
# round 6: where the fused conv1d_transpose + LayerNorm launch spends its epilogue (measurement build of the library)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
OPH_HIPCC_FLAGS=-DOPH_ABLATE timeout 300 python - > gpurun_out/r06/convt_ablate.txt 2>&1 <<'PY'
import ctypes as C
from ophelia_amd import _lib
lib = _lib.load()
names = {2: "fused (shipping)", 10: "two launches (round 5)", 11: "fused, no wait for the partners", 12: "fused, no plane stores", 8: "plane_gemm alone, no stores", 6: "plane_gemm alone, no MFMAs"}
for T in (200, 400):
    for rep in range(2):
        for prec in (2, 11, 12, 10, 8):
            us, by, fl = C.c_double(), C.c_double(), C.c_double()
            rc = lib.oph_bench_conv1d_transpose(0, 16, T, 512, 512, prec, 10, 100, C.byref(us), C.byref(by), C.byref(fl))
            print("D_%d  %-34s rc %d  %.2f us" % (4 if T == 200 else 7, names[prec], rc, us.value))
PY
cat gpurun_out/r06/convt_ablate.txt

#!/bin/bash
# round 6: conv1d_transpose with its LayerNorm inside the launch -- parity legs, then the layer's timing (fused / two launches)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_pipeline.py tests/test_gpu_properties.py -m gpu -q -x --timeout 300 -k "transpose or ssrn or plane_gemm or c3 or stream or chunk or receptive or golden or determin" > gpurun_out/r06/convt_tests.log 2>&1; echo rc=$?; tail -8 gpurun_out/r06/convt_tests.log
timeout 300 python - > gpurun_out/r06/convt_time.txt 2>&1 <<'PY'
import ctypes as C
from ophelia_amd import _lib
lib = _lib.load()
for T in (200, 400):
    for prec, name in ((2, "fused"), (10, "two launches"), (2, "fused"), (10, "two launches")):
        us, by, fl = C.c_double(), C.c_double(), C.c_double()
        rc = lib.oph_bench_conv1d_transpose(0, 16, T, 512, 512, prec, 5, 50, C.byref(us), C.byref(by), C.byref(fl))
        print("D_%d  (16, %d, 512)  %-13s rc %d  %.2f us  %.1f%% HBM  %.1f%% of the 16-bit MFMA peak" % (4 if T == 200 else 7, T, name, rc, us.value,
              100 * by.value / (us.value * 1e-6) / 8e12, 100 * 3 * fl.value / (us.value * 1e-6) / 2.5e15))
PY
cat gpurun_out/r06/convt_time.txt

#!/bin/bash
# plane_gemm with 4 waves per workgroup (one per SIMD) against 8 (two per SIMD, OPH_PG_WAVES): stand-alone SSRN per-dispatch tables.
# The 8-wave forms of the k = 1 / 3-tap layers exist only in a -DOPH_ABLATE build (OPH_HIPCC_FLAGS=-DOPH_ABLATE, as profiles/r04_convt_ablate.sh builds one)
cd /root/repo; mkdir -p gpurun_out/planes; export TMPDIR=/tmp
for v in 4 8 0; do
  if [ $v = 0 ]; then unset OPH_PG_WAVES; else export OPH_PG_WAVES=$v; fi
  (cd /tmp && rm -rf /tmp/tw_$v && rocprofv3 --kernel-trace --output-format csv -d /tmp/tw_$v -o s -- python /root/repo/profiles/r03_ssrn_layers.py > /root/repo/gpurun_out/planes/waves_$v.log 2>&1)
  f=$(find /tmp/tw_$v -name 's_kernel_trace.csv' | head -1)
  python profiles/r03_ssrn_layers.py --summarize $f > gpurun_out/planes/waves_${v}_table.txt 2>&1
done
unset OPH_PG_WAVES
python profiles/r04_convt.py > gpurun_out/planes/convt.log 2>&1

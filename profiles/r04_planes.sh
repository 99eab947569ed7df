#!/bin/bash
# plane_gemm against the round-3 contraction: conv1d_transpose timing (precision 2 = planes, 5 = rows split in the kernel) and the
# stand-alone SSRN per-dispatch table of both paths (rocprofv3 kernel trace)
cd /root/repo
mkdir -p gpurun_out/planes
python profiles/r04_convt.py > gpurun_out/planes/convt.log 2>&1
export TMPDIR=/tmp
for v in planes rows; do
  if [ $v = rows ]; then export OPH_NO_PLANE_GEMM=1; else unset OPH_NO_PLANE_GEMM; fi
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -o s -- python /root/repo/profiles/r03_ssrn_layers.py > /root/repo/gpurun_out/planes/ssrn_$v.log 2>&1)
  f=$(find /tmp/tr_$v -name 's_kernel_trace.csv' | head -1)
  python profiles/r03_ssrn_layers.py --summarize $f > gpurun_out/planes/ssrn_${v}_table.txt 2>&1
done
unset OPH_NO_PLANE_GEMM

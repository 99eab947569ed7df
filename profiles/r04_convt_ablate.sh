#!/bin/bash
# conv1d_transpose timings including the ablation builds of plane_gemm (oph_bench_conv1d_transpose precisions 6..9: no MFMAs / no operand
# stream / no stores / three K blocks), which exist only in a library built with -DOPH_ABLATE: a side copy of the tree is built that way
# HERE (before gpurun: the GPU box only runs it) -- usage:  bash profiles/r04_convt_ablate.sh build ; gpurun -- 'bash profiles/r04_convt_ablate.sh run'
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  rm -rf _ab && mkdir _ab && git archive HEAD | tar -x -C _ab && (cd _ab && OPH_HIPCC_FLAGS=-DOPH_ABLATE python -c "from ophelia_amd import _lib; _lib.build()")
else
  mkdir -p gpurun_out/planes && (cd _ab && python profiles/r04_convt.py) > gpurun_out/planes/convt_ablate.log 2>&1; cat gpurun_out/planes/convt_ablate.log
fi

#!/bin/bash
# Profile collection for round 1 (run on the GPU box through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash profiles/collect_r01.sh'
# Writes raw outputs under gpurun_out/ (scratch); the summaries are copied into profiles/ afterwards.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-pipeline --no-vocoder"
# 1. kernel trace + stats (per-kernel durations)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r01 -- $CMD > $O/trace.log 2>&1
# 2./3. HBM traffic counters, each in its own pass (TCC slots: FETCH_SIZE costs 3, WRITE_SIZE 2)
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o r01 -- $CMD > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o r01 -- $CMD > $O/pmc_write.log 2>&1
# 4. MFMA / VALU utilisation of the GEMM kernels
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o r01 -- $CMD > $O/pmc_mfma.log 2>&1
ls -R $O | head -40
# the un-profiled bench line of the same build
cd $R && timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
# condense on the box (the raw traces exceed what gpurun copies back) and keep only the summaries
python profiles/summarize.py r01
mkdir -p $R/gpurun_out/r01_summary && cp profiles/r01_kernel_stats.csv profiles/r01_pmc_summary.csv profiles/r01_traffic.json profiles/r01_bench.json $R/gpurun_out/r01_summary/
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_mfma

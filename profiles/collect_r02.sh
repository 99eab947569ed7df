#!/bin/bash
# Profile collection for round 2 (run on the GPU box through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash profiles/collect_r02.sh'
# Raw outputs go under gpurun_out/r02/ (scratch); profiles/summarize.py condenses them into profiles/r02_*.
# Pass 1 traces the DEFAULT bench command (pipelined batches, whole-decode launch + cone side stream).  The counter passes
# serialise dispatches across queues, so there the decode launch runs without its side stream (OPH_BENCH_PMC=1 ->
# OPH_LOOP_DBG=32: the kernel's own reads and writes are unchanged, the cone launches are absent).
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs"
PMC="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile"
echo "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs   [counter passes: OPH_BENCH_PMC=1 rocprofv3 --pmc <counter> --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile]" > $O/command.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r02 -- $CMD > $O/trace.log 2>&1
echo "trace rc=$?"
grep '^{"metric"' $O/trace.log | tail -1 > $O/bench_traced.json      # the bench line of the traced run itself
OPH_BENCH_PMC=1 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o r02 -- $PMC > $O/pmc_fetch.log 2>&1
echo "fetch rc=$?"
OPH_BENCH_PMC=1 timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o r02 -- $PMC > $O/pmc_write.log 2>&1
echo "write rc=$?"
OPH_BENCH_PMC=1 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o r02 -- $PMC > $O/pmc_mfma.log 2>&1
echo "mfma rc=$?"
find $O -name "*.csv" | head -20
# condense the counter passes first, so that the un-profiled bench line below can quote the PMC traffic of this build
cd $R && python profiles/summarize.py r02
# the un-profiled bench line of the same build (default flags)
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json
cp $O/bench.json profiles/r02_bench.json
mkdir -p $R/gpurun_out/r02_summary && cp $O/bench_traced.json profiles/r02_bench_traced.json; cp profiles/r02_kernel_stats.csv profiles/r02_pmc_summary.csv profiles/r02_traffic.json profiles/r02_bench.json profiles/r02_bench_traced.json $R/gpurun_out/r02_summary/
tail -n 3 $O/trace.log
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_mfma

#!/bin/bash
# Profile collection for round 6 (run on the GPU box through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash profiles/collect_r06.sh'
# Raw outputs go under gpurun_out/r06c/ (scratch); profiles/summarize_r06.py condenses them into profiles/r06_*.
#  A  kernel trace of the DEFAULT bench command (every dispatch traced: the traced run is slower than the un-traced one, see
#     r06_witness.json); what must agree is rocprofv3's dec_chain average with the HIP events / device clock of the line THIS run prints.
#  D0 the SAME command un-traced, right behind A on the same box: the un-traced pair of witnesses.
#  B  kernel trace of every kernel (sequential batches, 5 steps): per-kernel durations of the cone / SSRN / TextEnc kernels.
#  C  counter passes.  They serialise dispatches across queues, so the whole-decode launch must run without its side stream there:
#     OPH_BENCH_PMC=1 makes bench.py ask for the option LOOP_ALONE, which only a MEASUREMENT build of the library accepts
#     (OPH_HIPCC_FLAGS=-DOPH_ABLATE -> ophelia_amd/lib/libophelia_hip.<hash>.so, built here if absent; the production library
#     refuses the option) -> dec_chain's own FETCH / WRITE; the cone's and the batched nets' kernels are counted in DECODE=runs
#     passes (two launches per step + the cone's launches, chained by events the profiler understands).
#  D  the un-profiled default bench line of the same build (all legs).
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c; rm -rf $O; mkdir -p $O $R/gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs"
SEQ="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs --no-pipeline"
PMC="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs --no-pipeline"
cat > $O/command.txt <<EOT
A: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs
D0: (the same command without rocprofv3)
B: rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs --no-pipeline
C: [OPH_HIPCC_FLAGS=-DOPH_ABLATE OPH_BENCH_PMC=1 | OPH_BENCH_OPTIONS="DECODE=runs"] rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs --no-pipeline
EOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/traceA -o r06 -- $CMD > $O/traceA.log 2>&1
echo "traceA rc=$?"
grep '^{"metric"' $O/traceA.log | tail -1 > $O/bench_traced.json
timeout 600 $CMD > $O/bench_untraced.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/traceB -o r06 -- $SEQ > $O/traceB.log 2>&1
echo "traceB rc=$?"
# the measurement build of the library for the loop-alone counter passes
( cd $R && OPH_HIPCC_FLAGS=-DOPH_ABLATE python -c "from ophelia_amd import _lib; print(_lib.build())" )
OPH_HIPCC_FLAGS=-DOPH_ABLATE OPH_BENCH_PMC=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/loop_fetch -o r06 -- $PMC > $O/loop_fetch.log 2>&1
echo "loop fetch rc=$?"
OPH_HIPCC_FLAGS=-DOPH_ABLATE OPH_BENCH_PMC=1 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/loop_write -o r06 -- $PMC > $O/loop_write.log 2>&1
echo "loop write rc=$?"
OPH_BENCH_OPTIONS="DECODE=runs" timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/runs_fetch -o r06 -- $PMC > $O/runs_fetch.log 2>&1
echo "runs fetch rc=$?"
OPH_BENCH_OPTIONS="DECODE=runs" timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/runs_write -o r06 -- $PMC > $O/runs_write.log 2>&1
echo "runs write rc=$?"
OPH_BENCH_OPTIONS="DECODE=runs" timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/runs_sq -o r06 -- $PMC > $O/runs_sq.log 2>&1
echo "runs sq rc=$?"
OPH_BENCH_OPTIONS="DECODE=runs" timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/runs_lds -o r06 -- $PMC > $O/runs_lds.log 2>&1
echo "runs lds rc=$?"
cd $R
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json
python profiles/summarize_r06.py
cp $O/bench.json profiles/r06_bench.json; cp $O/bench_traced.json profiles/r06_bench_traced.json; cp $O/bench_untraced.json profiles/r06_bench_untraced.json
# T  what runs between two whole-decode launches (host -> host batches)
sed 's#gpurun_out/r05#gpurun_out/r06#g; s#/tmp/tr_tail#/tmp/tr_tail6#g' profiles/r05_tail.sh > /tmp/r06_tail.sh; bash /tmp/r06_tail.sh > /dev/null 2>&1; cp $R/gpurun_out/r06/tail.txt $R/profiles/r06_tail.txt
# L  soak: ten bench processes one after the other on this box
( for i in 1 2 3 4 5 6 7 8 9 10; do OPH_HANG_DUMP_S=120 timeout 200 python bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i value %.0f launch_us %.1f recoveries %s' % (d['value'], d['roofline']['avg_launch_us'], d['config']['recoveries']))" || echo "run $i FAILED rc=$?"; done > profiles/r06_soak.txt )
# R  ten consecutive runs of the two-rank test on this box (VERDICT r05 #1 "Done")
( for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python -m pytest tests/test_gpu_bench_ranks.py -m gpu -q -s -k two_ranks 2>&1 | grep -h "rank_host_cores\|passed\|failed" | tr '\n' ' '; echo; done > profiles/r06_ranks_soak.txt )
mkdir -p $R/gpurun_out/r06_summary && cp $R/profiles/r06_* $R/gpurun_out/r06_summary/
rm -rf $O/traceA $O/traceB $O/loop_fetch $O/loop_write $O/runs_fetch $O/runs_write $O/runs_sq $O/runs_lds

#!/bin/bash
# Profile collection for round 5 (run on the GPU box through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash profiles/collect_r05.sh'
# Raw outputs go under gpurun_out/r05/ (scratch); profiles/summarize_r05.py condenses them into profiles/r05_*.
#  A  kernel trace of the DEFAULT bench command.  rocprofv3's kernel filter (--kernel-include-regex) applies to counter collection
#     only, so every dispatch is traced, and tracing the ~1900 cone dispatches per batch delays the cone the loop kernel waits
#     for: the traced run is slower than the un-traced one (D).  What must agree is the rocprofv3 average of dec_chain and the
#     roofline.avg_launch_us (HIP events) of the bench line THIS traced run prints (bench_traced.json).
#  A2 the same with the cone as one persistent launch (OPH_CONE_LOOP=1): two dispatches per decode, nothing for the tracer to
#     delay -- rocprofv3's dec_chain average, the HIP events of that run and its un-traced twin all agree (the method check).
#  B  kernel trace of every kernel (sequential batches, 5 steps): per-kernel durations of the cone / SSRN / TextEnc kernels.
#  C  counter passes.  They serialise dispatches across queues, so the whole-decode launch runs without its side stream
#     there (OPH_BENCH_PMC=1 -> OPH_LOOP_ALONE=1: its own reads and writes are unchanged) -> dec_chain's FETCH / WRITE;
#     the cone's and the batched nets' kernels are counted in OPH_DECODE=runs passes (two launches per step + the cone's
#     launches, chained by events the profiler understands): FETCH_SIZE, WRITE_SIZE (separate passes) and the SQ busy / stall
#     counters.
#  D  the un-profiled default bench line of the same build.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; rm -rf $O; mkdir -p $O $R/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs"
SEQ="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs --no-pipeline"
PMC="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs --no-pipeline"
cat > $O/command.txt <<EOT
A: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs
A2: OPH_CONE_LOOP=1 rocprofv3 --kernel-trace --stats -- (the same)
B: rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs --no-pipeline
C: [OPH_BENCH_PMC=1 | OPH_DECODE=runs] rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs --no-pipeline
EOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/traceA -o r05 -- $CMD > $O/traceA.log 2>&1
echo "traceA rc=$?"
grep '^{"metric"' $O/traceA.log | tail -1 > $O/bench_traced.json
OPH_CONE_LOOP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/traceA2 -o r05 -- $CMD > $O/traceA2.log 2>&1
echo "traceA2 rc=$?"
grep '^{"metric"' $O/traceA2.log | tail -1 > $O/bench_traced_coneloop.json
OPH_CONE_LOOP=1 timeout 600 $CMD > $O/bench_coneloop.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/traceB -o r05 -- $SEQ > $O/traceB.log 2>&1
echo "traceB rc=$?"
OPH_BENCH_PMC=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/loop_fetch -o r05 -- $PMC > $O/loop_fetch.log 2>&1
echo "loop fetch rc=$?"
OPH_BENCH_PMC=1 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/loop_write -o r05 -- $PMC > $O/loop_write.log 2>&1
echo "loop write rc=$?"
OPH_DECODE=runs timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/runs_fetch -o r05 -- $PMC > $O/runs_fetch.log 2>&1
echo "runs fetch rc=$?"
OPH_DECODE=runs timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/runs_write -o r05 -- $PMC > $O/runs_write.log 2>&1
echo "runs write rc=$?"
OPH_DECODE=runs timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/runs_sq -o r05 -- $PMC > $O/runs_sq.log 2>&1
echo "runs sq rc=$?"
OPH_DECODE=runs timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/runs_lds -o r05 -- $PMC > $O/runs_lds.log 2>&1
echo "runs lds rc=$?"
find $O -name "*.csv" | head -40
cd $R && python profiles/summarize_r05.py
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 1200 $O/bench.json
cp $O/bench.json profiles/r05_bench.json;  cp $O/bench_traced.json profiles/r05_bench_traced.json; cp $O/bench_traced_coneloop.json profiles/r05_bench_traced_coneloop.json; cp $O/bench_coneloop.json profiles/r05_bench_coneloop.json
# T  what runs between two whole-decode launches (host -> host batches): profiles/r05_tail.sh
bash $R/profiles/r05_tail.sh > /dev/null 2>&1; cp $R/gpurun_out/r05/tail.txt $R/profiles/r05_tail.txt
# S  where a step goes: the chain's per-layer stamps and the cone's level completion times (the stamping build of the kernels)
bash $R/profiles/r05_stamps.sh > $O/stamps.log 2>&1; grep -h "stamped step\|cone of step 100\|hc_fused level\|step 100\|run 0 layer" $R/gpurun_out/r05/stamps_chain.txt | head -60 > $R/profiles/r05_stamps.txt
# K  the decode beside a second client of the GPU (tests/test_gpu_second_client.py), with its counts
( cd $R && timeout 600 python -m pytest tests/test_gpu_second_client.py -m gpu -q -s 2>&1 | grep -h "second client\|passed\|failed" > profiles/r05_second_client.txt )
# L  soak: ten bench processes one after the other on this box (the two unexplained stalls of round 4 were runtime stalls between processes)
( cd $R && for i in 1 2 3 4 5 6 7 8 9 10; do OPH_HANG_DUMP_S=120 timeout 200 python bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i value %.0f launch_us %.1f recoveries %s' % (d['value'], d['roofline']['avg_launch_us'], d['config']['recoveries']))" || echo "run $i FAILED rc=$?"; done > profiles/r05_soak.txt )
mkdir -p $R/gpurun_out/r05_summary && cp $R/profiles/r05_* $R/gpurun_out/r05_summary/
rm -rf $O/traceA $O/traceA2 $O/traceB $O/loop_fetch $O/loop_write $O/runs_fetch $O/runs_write $O/runs_sq $O/runs_lds

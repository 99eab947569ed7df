#!/bin/bash
# round 6: the two cheap knobs once more with this round's kernels -- cones queued ahead of the chain, and the CU partition
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
run() { OPH_BENCH_OPTIONS="$1" timeout 300 python bench.py --steps 30 --warmup 2 --no-extra-legs --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('%-28s value %.0f  ms_per_step %.3f  launch_us %.1f  tail_ms %.3f  host cores %.2f' % ('$1' or 'default', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['ms_per_step'] - d['roofline']['avg_launch_us'] / 1e3, d['config']['rank_host_cores'][0]))"; }
for rep in 1 2; do
for o in "" "LOOP_LOOKAHEAD=4" "LOOP_LOOKAHEAD=16" "LOOP_LOOKAHEAD=32" "CU_SPLIT=64,136" "CU_SPLIT=56,136" "CU_SPLIT=72,120" "CU_SPLIT=64,120" "SSRN_CHUNK=32" "SSRN_CHUNK=50"; do run "$o"; done; done > gpurun_out/r06/knobs.txt 2>&1
cat gpurun_out/r06/knobs.txt

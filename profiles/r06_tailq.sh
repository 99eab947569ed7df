#!/bin/bash
# round 6: the final SSRN piece queued behind the running decode (default) against queued after it (option NO_TAIL_PREQUEUE), alternated
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_call_sequences.py tests/test_gpu_properties.py tests/test_gpu_edge_cases.py tests/test_gpu_synthesize.py -m gpu -q -x --timeout 300 > gpurun_out/r06/tailq_tests.log 2>&1; echo rc=$?; tail -4 gpurun_out/r06/tailq_tests.log
for i in 1 2 3; do
for o in "" "NO_TAIL_PREQUEUE=1"; do
OPH_BENCH_OPTIONS="$o" timeout 300 python bench.py --steps 30 --warmup 2 --no-extra-legs --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('%-20s value %.0f  ms_per_step %.3f  launch_us %.1f  tail_ms %.3f' % ('$o' or 'default', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['ms_per_step'] - d['roofline']['avg_launch_us'] / 1e3))"
done; done > gpurun_out/r06/tailq.txt 2>&1
cat gpurun_out/r06/tailq.txt

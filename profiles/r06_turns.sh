#!/bin/bash
# round 6: the inter-process GPU turn (two ranks on one GPU take turns instead of colliding) and the fused-LayerNorm time-out redo
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_second_client.py tests/test_gpu_bench_ranks.py tests/test_gpu_misuse.py -m gpu -q -x -s --timeout 900 > gpurun_out/r06/turns_tests.log 2>&1; echo rc=$?; grep -h "recoveries\|passed\|failed\|RESULT\|{'rec" gpurun_out/r06/turns_tests.log | cut -c1-400
for i in 1 2 3 4 5 6; do
OPH_BENCH_SHARED_GPU=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 --no-extra-legs --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); c=d['config']
print('run $i: cores', [round(x,2) for x in c['rank_host_cores']], 'recoveries', c['recoveries'], 'ms per step and rank', [round(x,1) for x in d['rank_ms_per_step']], 'value %.0f' % d['value'])"
done > gpurun_out/r06/turns_soak.txt 2>&1
cat gpurun_out/r06/turns_soak.txt

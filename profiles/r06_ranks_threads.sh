#!/bin/bash
# which host threads are busy in a two-ranks-on-one-GPU run (and at N = 1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for i in 1 2 3 4; do
OPH_BENCH_SHARED_GPU=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 --no-extra-legs --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); c=d['config']
print('cores', c['rank_host_cores'], 'recoveries', c['recoveries'], 'ms', d['rank_ms_per_step'])
for r in c['ranks']: print('   rank', r['rank'], 'degraded_left', r.get('degraded_left'), 'cores', r.get('cores'), 'threads', r.get('host_threads'))
"
done > gpurun_out/r06/ranks_threads.txt 2>&1
timeout 300 python bench.py --steps 10 --warmup 2 --no-extra-legs --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); c=d['config']
print('N=1 cores', c['rank_host_cores'], 'threads', c['ranks'][0].get('host_threads'))
" >> gpurun_out/r06/ranks_threads.txt 2>&1
OPH_BENCH_OPTIONS="DECODE=runs" timeout 300 python bench.py --steps 10 --warmup 2 --no-extra-legs --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); c=d['config']
print('N=1 DECODE=runs cores', c['rank_host_cores'], 'threads', c['ranks'][0].get('host_threads'), 'ms', d['ms_per_step'])
" >> gpurun_out/r06/ranks_threads.txt 2>&1
OPH_BENCH_OPTIONS="DECODE=runs STREAM_VALUE=1" timeout 300 python bench.py --steps 10 --warmup 2 --no-extra-legs --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); c=d['config']
print('N=1 DECODE=runs STREAM_VALUE=1 cores', c['rank_host_cores'], 'threads', c['ranks'][0].get('host_threads'), 'ms', d['ms_per_step'])
" >> gpurun_out/r06/ranks_threads.txt 2>&1
cat gpurun_out/r06/ranks_threads.txt

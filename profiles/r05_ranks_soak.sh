#!/bin/bash
# round 5: two ranks sharing ONE GPU (two processes, six CU-masked queues: the configuration in which 2 of 29 runs of round 4 ended in
# the 2 s co-residency time-out and an error).  Twelve runs: every run must print its line; recoveries are counted, not fatal.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  OPH_BENCH_SHARED_GPU=1 OPH_HANG_DUMP_S=200 timeout 400 python bench.py --gpus 2 --steps 4 --warmup 1 --no-extra-legs --no-cpu-baseline --no-vocoder --no-profile 2> gpurun_out/r05/ranks_$i.err | python -c "
import json,sys
try:
    d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); c=d['config']
    print('run $i value %.0f ms %.2f recoveries %s host cores %s' % (d['value'], d['ms_per_step'], c['recoveries'], [round(x,2) for x in c['rank_host_cores']]))
except Exception as e:
    print('run $i FAILED', e)
"
done | tee gpurun_out/r05/ranks_soak.txt

#!/bin/bash
# hc_fused shapes per level, measured on one box: OPH_HCF_SHAPES bit 0 = 128-row blocks for the 1312-row level, bit 1 = K split over 4 workgroups for levels <= 256 rows
B="--steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs"
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for sh in 0 1 2 3; do
  OPH_HCF_SHAPES=$sh python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('shapes=$sh', 'value %.0f  ms %.3f  dec launch %.1f us  clock %.1f us' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['device_clock_us'] or 0))"
done
done

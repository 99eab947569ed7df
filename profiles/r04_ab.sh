#!/bin/bash
# A/B on ONE box: side copies (_ab, _ab2: `git archive` of a commit, optionally with files replaced, built there) against the working tree.
# usage (through gpurun): bash profiles/r04_ab.sh "<dirs>" [env assignments]
B="--steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs"
DIRS=${1:-"_ab ."}
for rep in 1 2 3; do
for d in $DIRS; do
  ( cd $GRAFT_REPO_ROOT/$d && env $2 python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$d', 'value %.0f  ms %.3f  dec launch %.1f us  clock %.1f us' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['device_clock_us'] or 0))" )
done
done

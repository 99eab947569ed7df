#!/bin/bash
# CU partition sweep (chain,cone CUs; the rest is the SSRN partition) with the round-4 kernels: value (host to host), resident, decode launch
cd $GRAFT_REPO_ROOT
for sp in 64,128 64,144 64,152 56,136 56,144 48,144 64,168 72,128; do
  OPH_CU_SPLIT=$sp python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('split $sp', 'value %.0f  resident %.0f  sequential %.0f  dec launch %.1f us' % (d['value'], c['resident_value'], c['sequential_value'], r['avg_launch_us']))"
done

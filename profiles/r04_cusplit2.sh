#!/bin/bash
# CU partition A/B with the faster SSRN (plane_gemm): chain 64 CUs, cone 128..160, the rest SSRN; three alternations on one box
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for sp in 64,128 64,144 64,152 64,160; do
  OPH_CU_SPLIT=$sp python bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('split $sp', 'value %.0f  resident %.0f  sequential %.0f  api %.0f  dec launch %.1f us' % (d['value'], c['resident_value'], c['sequential_value'], c['api_value'], r['avg_launch_us']))"
done
done

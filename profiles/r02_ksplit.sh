#!/bin/bash
# split-K of the cone's GEMMs (big = many-row levels, small = the rest) with the round-2 cone
cd $GRAFT_REPO_ROOT
for ks in 3,4 2,4 4,4 3,2 3,3 2,2 2,3; do
  r=$(OPH_CONE_KSPLIT=$ks timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2))")
  echo "ksplit $ks: $r"
done

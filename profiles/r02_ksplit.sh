#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02r}; mkdir -p $out
for ks in 4,4 2,4 1,4 2,2 3,4; do
  r=$(OPH_CONE_KSPLIT=$ks OPH_TRACE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['config']['sequential_ms_per_step'],2))")
  echo "ksplit $ks: pipelined/sequential ms per batch: $r"
done
for split in 64,128 64,144 48,144 64,160; do
  r=$(OPH_CU_SPLIT=$split timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['config']['sequential_ms_per_step'],2))")
  echo "CU split $split: $r"
done

#!/bin/bash
# throughput vs utterances per GPU (the whole-decode launch needs all its workgroups resident: B <= 16 per handle; larger
# batches take the two-launches-per-step path)
cd $GRAFT_REPO_ROOT
for b in 8 16 32 64 128; do
  r=$(timeout 600 python bench.py --batch $b --steps 8 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), 'frames/s', round(d['ms_per_step'],2), 'ms per batch; sequential', round(d['config']['sequential_ms_per_step'],2))")
  echo "B=$b: $r"
done

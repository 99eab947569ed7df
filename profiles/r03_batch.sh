#!/bin/bash
# throughput vs utterances per GPU: batches beyond 16 decode in tiles of 16, each on the whole-decode launch
cd $GRAFT_REPO_ROOT
for b in 8 16 32 64 128; do
  r=$(timeout 900 python bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), 'frames/s', round(d['ms_per_step'],2), 'ms per batch')")
  echo "B=$b: $r"
done

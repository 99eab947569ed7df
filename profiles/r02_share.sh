#!/bin/bash
# experiment: overlapping CU masks (cone and/or SSRN on all 192 CUs outside the chain's partition)
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; r=$(env "$@" timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2), 'seq', round(d['config']['sequential_ms_per_step'],2), 'fp32', round(d['config']['all_fp32_ms_per_step'],2))"); echo "$name: $r"; }
run default
run ssrn_all OPH_SSRN_ALL=1
run both_all OPH_SSRN_ALL=1 OPH_CONE_ALL=1
run cone_all OPH_CONE_ALL=1

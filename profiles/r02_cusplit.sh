#!/bin/bash
# CU partition sweep (chain | cone | SSRN) for the loop-mode decode
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02x}; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > $out/bench_$name.json 2> $out/bench_$name.err; python - "$out/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d['value']), round(d['ms_per_step'],2), 'seq', round(d['config'].get('sequential_ms_per_step') or 0,2), 'fp32', round(d['config'].get('all_fp32_ms_per_step') or 0,2))
except Exception as e: print(sys.argv[2],'ERR',e)
PY
}
run default
run s64_136 OPH_CU_SPLIT=64,136
run s64_144 OPH_CU_SPLIT=64,144
run s64_152 OPH_CU_SPLIT=64,152
run s64_120 OPH_CU_SPLIT=64,120
run s64_112 OPH_CU_SPLIT=64,112

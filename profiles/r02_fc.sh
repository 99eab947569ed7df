#!/bin/bash
# cone_fc16 row threshold: which cone levels run as one fused launch
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r03f}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > $out/bench_$name.json 2> $out/bench_$name.err; python - "$out/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d['value']), round(d['ms_per_step'],2), 'seq', round(d['config'].get('sequential_ms_per_step') or 0,2))
except Exception as e: print(sys.argv[2],'ERR',e)
PY
}
run fc64
run fc64_in1 OPH_CONE_FC_INSPLIT=1
run fc0 OPH_CONE_FC_ROWS=0
run fc256 OPH_CONE_FC_ROWS=256
run fc256_in1 OPH_CONE_FC_ROWS=256 OPH_CONE_FC_INSPLIT=1
OPH_CONE_FC_ROWS=256 timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
OPH_CONE_FC_ROWS=${TRACE_FC:-256} OUT=${OUT:-r03f} bash profiles/r02_trace.sh 2>&1 | tail -14

#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02f}; mkdir -p $out
export OPH_RUN_ROWS=${ROWS:-4}
OPH_SKIP_CONE=1 OPH_RUN_STAMPS=1 OPH_TRACE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline > $out/bench_stamps.json 2> $out/bench_stamps.err
grep "stamped step\|run [01] layer\|decode loop" $out/bench_stamps.err | tail -28

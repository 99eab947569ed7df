#!/bin/bash
# phase stamps of the loop kernel WITH the cone running on the side stream (the production configuration)
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02u}; mkdir -p $out
OPH_RUN_STAMPS=1 OPH_TRACE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline > $out/bench_stamps.json 2> $out/bench_stamps.err
grep "stamped step\|run [01] layer\|decode loop\|spun for cone" $out/bench_stamps.err | tail -34

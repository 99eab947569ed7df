#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02j}; mkdir -p $out
for dbg in 15 0; do
  OPH_LOOP_DBG=$dbg OPH_SKIP_CONE=1 OPH_RUN_STAMPS=1 OPH_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline > $out/b_$dbg.json 2> $out/b_$dbg.err
  echo "dbg=$dbg: $(grep -h 'stamped step' $out/b_$dbg.err | tail -1)"
  grep "signals of step" $out/b_$dbg.err | tail -5
done

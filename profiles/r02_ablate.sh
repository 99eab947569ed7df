#!/bin/bash
# dec_loop ablations (timing only; results are wrong when a switch is set): where does a layer's time go?
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02h}; mkdir -p $out
for dbg in 0 1 2 4 8 3 15; do
  OPH_LOOP_DBG=$dbg OPH_SKIP_CONE=1 OPH_RUN_STAMPS=1 OPH_TRACE=1 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline > $out/b_$dbg.json 2> $out/b_$dbg.err
  echo "dbg=$dbg: $(grep -h 'stamped step' $out/b_$dbg.err | tail -1)  $(grep -h 'one launch' $out/b_$dbg.err | tail -1)"
  grep "run 0 layer  [79] \|run 0 layer 13\|run 0 layer 2[12] " $out/b_$dbg.err | tail -5
done
for split in 64,128 96,96 128,64; do
  echo "CU split $split: $(OPH_CU_SPLIT=$split OPH_SKIP_CONE=1 OPH_TRACE=1 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline 2>&1 >/dev/null | grep 'one launch' | tail -1)"
done

#!/bin/bash
# where does the ~1 us between the MFMAs and the second barrier go?  stamps with the next layer's weight requests (dbg 1) /
# tap requests (dbg 4) switched off (results are wrong, timing only)
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r03d}; mkdir -p $out
for dbg in 0 1 4 5; do
  OPH_LOOP_DBG=$((dbg+32)) OPH_RUN_STAMPS=1 OPH_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline > $out/b_$dbg.json 2> $out/b_$dbg.err
  echo "dbg=$dbg: $(grep -h 'stamped step' $out/b_$dbg.err | tail -1)"
  grep "run 0 layer  [5-7] \|run 0 layer 1[5-6]" $out/b_$dbg.err | tail -5
done

#!/bin/bash
# repeated short bench runs under a timeout, progress notes on: which leg does a stuck run sit in?
# usage: r04_hang_hunt.sh <runs> [alternate]   (alternate: every other run with OPH_STREAM_VALUE=1)
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-12}); do
  sv=0; if [ -n "$2" ] && [ $((i % 2)) -eq 0 ]; then sv=1; fi
  OPH_STREAM_VALUE=$sv OPH_BENCH_VERBOSE=1 timeout -s KILL 90 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > /tmp/hh.out 2> /tmp/hh.err
  rc=$?
  echo "run $i stream_value=$sv rc=$rc $(tail -1 /tmp/hh.err | cut -c1-120)"
  if [ $rc -ne 0 ]; then tail -15 /tmp/hh.err; fi
done

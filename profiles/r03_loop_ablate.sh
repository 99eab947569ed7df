#!/bin/bash
# dec_loop timing ablations (OPH_LOOP_DBG bits, oph_internal.h LoopArgs::dbg; results are wrong when 1/2/4/8 are set -- timing only):
# one sequential 16-utterance batch each.  usage: bash profiles/r03_loop_ablate.sh [bits ...]
for d in ${@:-0 1 4 8 13 64}; do echo "== OPH_LOOP_DBG=$d"; OPH_LOOP_DBG=$d python profiles/r03_probe.py 4 2>&1 | grep "batch [23]"; done

#!/bin/bash
# round 5: where a decode step goes -- the chain's per-layer phase stamps (dec_chain), alone and with the cone, and how long the
# generic kernel (dec_loop, which carries the signal stamps) waits for each cone level
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for v in "chain" "chain_alone:OPH_SKIP_CONE=1" "loop:OPH_NO_CHAIN=1"; do
  name=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs=""
  env $envs OPH_TRACE=1 OPH_RUN_STAMPS=1 timeout 300 python profiles/r03_probe.py 3 > gpurun_out/r05/stamps_$name.txt 2>&1; echo "$name rc=$?"
done
grep -h "batch\|stamped step\|spun for\|cone of step\|hc_fused level\|step 100" gpurun_out/r05/stamps_*.txt | head -80

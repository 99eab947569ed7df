#!/bin/bash
# round 6: the dec_chain hand-off confined to one / two XCDs, at the chain's geometry, with its weight stream
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-unused-result profiles/xcd_chain_probe.hip -o /tmp/xcd_chain_probe || exit 1
timeout 300 /tmp/xcd_chain_probe > gpurun_out/r06/xcd_chain_probe.txt 2>&1; echo rc=$?
cat gpurun_out/r06/xcd_chain_probe.txt
# host-side view of a batch (OPH_TRACE: where oph_run_host's time goes outside the decode)
OPH_TRACE=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs 2> gpurun_out/r06/trace_host.txt > /dev/null
grep -n "run_host:\|decode loop: host" gpurun_out/r06/trace_host.txt | tail -12

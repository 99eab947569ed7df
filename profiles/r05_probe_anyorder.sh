#!/bin/bash
# round 5: does hipExtAnyOrderLaunch overlap consecutive kernels of one stream on gfx950?  + the baseline bench line of this round's box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 profiles/anyorder_probe.hip -o /tmp/anyorder_probe 2>/dev/null
timeout 200 /tmp/anyorder_probe > gpurun_out/r05/anyorder_probe.txt 2>&1; echo "probe rc=$?"
cat gpurun_out/r05/anyorder_probe.txt

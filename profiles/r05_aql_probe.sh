#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
/opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 profiles/aql_probe_kernels.hip -o /tmp/aql_probe_kernels.co 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 profiles/aql_probe.cpp ophelia_amd/csrc/oph_aql.hip -L/opt/rocm/lib -lhsa-runtime64 -Wl,-rpath,/opt/rocm/lib -o /tmp/aql_probe 2>/dev/null || exit 2
timeout 120 /tmp/aql_probe /tmp/aql_probe_kernels.co 2>&1 | tee gpurun_out/r05/aql_probe.txt

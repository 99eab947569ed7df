#!/bin/bash
# hand-off micro-probe + per-phase stamps of dec_run (round 2 diagnostics)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 profiles/handoff_probe.hip -o /tmp/handoff_probe 2>/dev/null
timeout 120 /tmp/handoff_probe > gpurun_out/r02b/handoff_probe.txt 2>&1; echo "probe rc=$?"
cat gpurun_out/r02b/handoff_probe.txt
OPH_RUN_STAMPS=1 OPH_TRACE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline > gpurun_out/r02b/bench_stamps.json 2> gpurun_out/r02b/bench_stamps.err; echo "bench rc=$?"
grep "run [01] layer" gpurun_out/r02b/bench_stamps.err | tail -27

#!/bin/bash
# round 6: the whole GPU suite as the driver runs it (-x, parity files first), log kept; then smoke + the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
OPH_HANG_DUMP_S=100 timeout 1700 python -m pytest tests -m gpu -x -q --timeout 300 -rs > gpurun_out/r06/pytest_full.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r06/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py > gpurun_out/r06/bench0.json 2> gpurun_out/r06/bench0.err; echo "bench rc=$?"; cat gpurun_out/r06/bench0.json | cut -c1-600

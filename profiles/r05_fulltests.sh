#!/bin/bash
# round 5: the whole GPU suite as the driver runs it, log kept
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
OPH_HANG_DUMP_S=100 timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/r05/pytest_full.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r05/pytest_full.log

#!/bin/bash
# round 5 working check: the named GPU test files under a per-test time-out, log kept; then the bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
FILES=${1:-"tests/test_gpu_decode_modes.py tests/test_gpu_model.py"}
OPH_HANG_DUMP_S=100 timeout ${3:-1200} python -m pytest $FILES -m gpu -x -q -v --timeout 150 > gpurun_out/r05/pytest_${2:-check}.log 2>&1; echo "pytest rc=$?"
grep -c PASSED gpurun_out/r05/pytest_${2:-check}.log; grep -n "FAILED\|Timeout\|Error\|error" gpurun_out/r05/pytest_${2:-check}.log | head -20; tail -5 gpurun_out/r05/pytest_${2:-check}.log

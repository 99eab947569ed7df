#!/bin/bash
# round 6: A/B on one box -- _ab = HEAD before the fused tile reset (six stream operations + a host synchronisation in front of every
# decode), . = the working tree (one reset launch, no synchronisation); parity legs of the working tree first
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py tests/test_gpu_call_sequences.py tests/test_gpu_configs_c4_c5.py tests/test_gpu_decode_modes.py -m gpu -q -x --timeout 600 > gpurun_out/r06/reset_tests.log 2>&1; echo rc=$?; tail -3 gpurun_out/r06/reset_tests.log
bash profiles/r04_ab.sh "_ab . _ab ." > gpurun_out/r06/ab_reset.txt 2>&1; cat gpurun_out/r06/ab_reset.txt

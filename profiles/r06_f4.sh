cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_misuse.py -m gpu -q -x -k "golden or inventory or misuse or inherited or wrong_order" --timeout 300 > gpurun_out/r06/f4.log 2>&1; echo rc=$?; tail -30 gpurun_out/r06/f4.log

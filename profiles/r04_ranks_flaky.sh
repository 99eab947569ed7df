mkdir -p gpurun_out/planes
for v in planes rows; do
  if [ $v = rows ]; then export OPH_NO_PLANE_GEMM=1; else unset OPH_NO_PLANE_GEMM; fi
  for i in 1 2 3 4 5 6; do timeout 200 python -m pytest tests/test_gpu_bench_ranks.py::test_bench_two_ranks_on_one_gpu -x -q -m gpu 2>&1 | grep -E "passed|failed|OpheliaHipError" | cut -c1-250 | head -3; done > gpurun_out/planes/ranks_$v.log 2>&1
done

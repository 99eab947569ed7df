#!/bin/bash
# SSRN contraction kernel: tests that cover it + the conv1d_transpose roofline legs of the bench
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r03j}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder > $out/bench.json 2> $out/bench.err
python - "$out/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],2), 'seq', round(d['config']['sequential_ms_per_step'],2), 'fp32', round(d['config']['all_fp32_ms_per_step'],2))
for k,v in d['kernel_rooflines'].items(): print(k, round(v['avg_us'],1), 'us  mfma_frac', round(v['mfma_frac'],3), 'TF/s', round(v['mfma_TFLOP_per_s'],1), 'hbm_frac', round(v['hbm_frac'],3))
for k in d['kernel_classes']:
    if k['launches'] and ('gemm' in k['kernel'] or 'ln_rows' in k['kernel']): print(k)
PY

#!/bin/bash
# dec_loop with the 4x4x1 MFMA contraction: layout probe, parity, timing, CU splits
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02g}; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 profiles/mfma4x4_probe.hip -o /tmp/mfma4x4_probe 2>/dev/null && timeout 60 /tmp/mfma4x4_probe | tee $out/mfma4x4_probe.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
OPH_TRACE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > $out/bench_loop.json 2> $out/bench_loop.err; echo "bench loop rc=$?"
OPH_SKIP_CONE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > $out/bench_loop_nocone.json 2> $out/bench_loop_nocone.err; echo "bench nocone rc=$?"
OPH_CU_SPLIT=128,64 OPH_SKIP_CONE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > $out/bench_loop_nocone_cu128.json 2> $out/bench_loop_nocone_cu128.err; echo "bench nocone cu128 rc=$?"
OPH_CU_SPLIT=128,64 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > $out/bench_loop_cu128.json 2> $out/bench_loop_cu128.err; echo "bench cu128 rc=$?"
OPH_SKIP_CONE=1 OPH_RUN_STAMPS=1 OPH_TRACE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline > $out/bench_stamps.json 2> $out/bench_stamps.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/'+os.environ.get("OUT","r02g")+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],2), 'seq', d['config'].get('sequential_ms_per_step'), 'fp32', d['config'].get('all_fp32_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
grep -h "decode loop" $out/bench_loop.err | tail -2
grep "stamped step\|run [01] layer" $out/bench_stamps.err | tail -26

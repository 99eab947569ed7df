#!/bin/bash
# stall breakdown of the cone's kernels (one SQ counter pass; OPH_DECODE=layers: the per-layer decode is profiler-safe)
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r03x}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
OPH_DECODE=layers timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc -o c -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline --no-extra-legs > $GRAFT_REPO_ROOT/$out/pmc.log 2>&1
echo "rc=$?"
cd $GRAFT_REPO_ROOT
python - "$out/pmc/c_counter_collection.csv" <<'PY'
import csv,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'][:44]+' g'+r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(d.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES',[0]))):
    if 'conv_gemm_f32' not in k and 'cone' not in k and 'ln_rows<1>' not in k: continue
    m={c:sum(x)/len(x) for c,x in v.items()}
    wc=m.get('SQ_WAVE_CYCLES',1)
    print('%-60s n=%4d wave_cyc %.3g  wait_any %.0f%%  wait_inst %.0f%% (lds %.0f%%)  active %.0f%%  mfma_busy/wave_cyc %.2f  lds_conf/lds_active %.2f' % (
        k, len(v['SQ_WAVE_CYCLES']), wc, 100*m.get('SQ_WAIT_ANY',0)/wc, 100*m.get('SQ_WAIT_INST_ANY',0)/wc, 100*m.get('SQ_WAIT_INST_LDS',0)/wc,
        100*m.get('SQ_ACTIVE_INST_ANY',0)/wc, m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/wc, m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_LDS_IDX_ACTIVE',1),1)))
PY

#!/bin/bash
# stall breakdown of the SSRN contraction kernels (one SQ counter pass over stand-alone SSRN runs, full chip)
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r03r}; mkdir -p $out
cat > /tmp/ssrn_only.py <<'PY'
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from conftest import hp_from_snapshot
from oracle import ophelia_oracle as O
from ophelia_amd.engine import Engine
hp = hp_from_snapshot("lj_tutorial.cfg")
W = O.random_weights(hp, 2)
eng = Engine(hp, device=0); eng.load_weights(W)
Y = np.random.default_rng(0).random((16, hp.max_T, hp.n_mels), dtype=np.float32)
for _ in range(3): Z = eng.ssrn(Y)
eng.close()
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc -o s -- python /tmp/ssrn_only.py $GRAFT_REPO_ROOT > $GRAFT_REPO_ROOT/$out/pmc.log 2>&1
echo "rc=$?"
cd $GRAFT_REPO_ROOT
python - "$out/pmc/s_counter_collection.csv" <<'PY'
import csv,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'][:48]+' g'+r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items():
    if 'gemm' not in k and 'ln_rows' not in k: continue
    m={c:sum(x)/len(x) for c,x in v.items()}
    wc=m.get('SQ_WAVE_CYCLES',1)
    print('%-64s n=%3d wave_cyc %.3g  wait_any %.0f%%  wait_inst %.0f%% (lds %.0f%%)  active %.0f%%  mfma_busy/wave_cyc %.2f  lds_conf/lds_active %.2f' % (
        k, len(v['SQ_WAVE_CYCLES']), wc, 100*m.get('SQ_WAIT_ANY',0)/wc, 100*m.get('SQ_WAIT_INST_ANY',0)/wc, 100*m.get('SQ_WAIT_INST_LDS',0)/wc,
        100*m.get('SQ_ACTIVE_INST_ANY',0)/wc, m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/wc, m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_LDS_IDX_ACTIVE',1),1)))
PY

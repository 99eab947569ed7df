#!/bin/bash
# Counter passes over the stand-alone SSRN (profiles/r03_ssrn_layers.py: oph_ssrn on a host mel array, whole chip, nothing beside it):
# FETCH_SIZE / WRITE_SIZE (separate passes, MI355X_MICROARCH.md), the SQ busy / stall counters and the LDS counters, per dispatch.
# Output: gpurun_out/ssrn_pmc/summary.txt (one line per dispatch of the LAST SSRN evaluation: kernel, grid, counters)
cd /root/repo; O=/root/repo/gpurun_out/ssrn_pmc; mkdir -p $O
export TMPDIR=/tmp
run() { (cd /tmp && rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc_$1 -o s -- python /root/repo/profiles/r03_ssrn_layers.py > $O/$1.log 2>&1); }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS"
python - <<'PY' > $O/summary.txt
import csv, glob, collections
def load(tag):
    f = glob.glob('/tmp/pmc_%s/**/s_counter_collection.csv' % tag, recursive=True)
    rows = list(csv.DictReader(open(f[0]))) if f else []
    d = collections.OrderedDict()
    for r in rows:
        d.setdefault(int(r['Dispatch_Id']), {'k': r['Kernel_Name'].replace('void oph::', '').split('(')[0], 'g': r.get('Grid_Size', '?')})[r['Counter_Name']] = float(r['Counter_Value'])
    return d
tabs = {t: load(t) for t in ('fetch', 'write', 'sq', 'lds')}
n = 33                                             # dispatches of one SSRN evaluation
for t, d in tabs.items():
    ids = [i for i in d if 'embed' not in d[i]['k']][-n:]
    tabs[t] = [d[i] for i in ids]
for j in range(n):
    f, w, s, l = (tabs[t][j] if j < len(tabs[t]) else {} for t in ('fetch', 'write', 'sq', 'lds'))
    hbm = (2 * f.get('FETCH_SIZE', 0) + w.get('WRITE_SIZE', 0)) * 1024
    wc = s.get('SQ_WAVE_CYCLES', 0) or 1
    print('%-34s grid %-8s hbm %7.1f MB (fetch %7.1f x2, write %7.1f)  mfma_busy/gui %.3f  wait_any %.2f wait_inst %.2f active %.2f  lds_conflict/idx %.3f wait_lds %.2f' % (
        f.get('k', s.get('k', '?')), f.get('g', '?'), hbm / 1e6, f.get('FETCH_SIZE', 0) * 1024 / 1e6, w.get('WRITE_SIZE', 0) * 1024 / 1e6,
        s.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(s.get('GRBM_GUI_ACTIVE', 1), 1) / 1024.0 * 4,
        s.get('SQ_WAIT_ANY', 0) / wc, s.get('SQ_WAIT_INST_ANY', 0) / wc, s.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        l.get('SQ_LDS_BANK_CONFLICT', 0) / max(l.get('SQ_LDS_IDX_ACTIVE', 1), 1), l.get('SQ_WAIT_INST_LDS', 0) / max(l.get('SQ_WAVE_CYCLES', wc), 1)))
PY
cat $O/summary.txt

#!/bin/bash
# per-kernel durations of modules.conv1d_transpose (D_4, D_7) as the SSRN path launches it
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r03k}; mkdir -p $out
cat > /tmp/convT.py <<'PY'
import ctypes as C, sys
sys.path.insert(0, sys.argv[1])
from ophelia_amd import _lib
lib = _lib.load()
for T in (200, 400):
    for prec in (0, 1):
        us, by, fl = C.c_double(), C.c_double(), C.c_double()
        rc = lib.oph_bench_conv1d_transpose(0, 16, T, 512, 512, prec, 3, 20, C.byref(us), C.byref(by), C.byref(fl))
        print("T", T, "prec", prec, "rc", rc, "avg_us", round(us.value, 2))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/convT -o ct -- python /tmp/convT.py $GRAFT_REPO_ROOT > $GRAFT_REPO_ROOT/$out/convT.log 2>&1
cd $GRAFT_REPO_ROOT
grep "^T " $out/convT.log
python - "$out/convT/ct_kernel_trace.csv" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    key=(r['Kernel_Name'][:60], r['Grid_Size_X'], r['Workgroup_Size_X'])
    agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    v2=sorted(v)
    print('%-62s grid %8s wg %4s  n %4d  median %8.2f us  min %8.2f' % (k[0],k[1],k[2],len(v),v2[len(v2)//2],v2[0]))
PY

#!/bin/bash
# per-dispatch kernel trace of a short run (loop mode): durations of the cone's 13 launches, by position in the step
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r02o}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder --no-profile --no-pipeline > $GRAFT_REPO_ROOT/$out/trace_bench.json 2> $GRAFT_REPO_ROOT/$out/trace_bench.err
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $out/trace -name "*kernel_trace.csv" | head -1); echo $f; wc -l $f
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
# cone kernels are on their own queue: take the dispatches between two consecutive attn_rows launches in the middle of a decode
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'cone_head' in n or 'attn_rows' in n]
print(len(rows),'dispatches;',len(idx),'cone starts')
mid=idx[len(idx)//2+100] if len(idx)>300 else idx[len(idx)//2]
q=rows[mid]['Queue_Id']
seq=[r for r in rows[mid:mid+400] if r['Queue_Id']==q][:16]
t0=int(seq[0]['Start_Timestamp'])
for r in seq:
    print('%8.2f us  dur %7.2f us  %s' % ((int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Kernel_Name'][:70]))
# averages per position over steps 100..180
per=collections.defaultdict(list)
for k in range(120,180):
    if k+1>=len(idx): break
    a=idx[k]; qq=rows[a]['Queue_Id']
    s=[r for r in rows[a:a+200] if r['Queue_Id']==qq][:15]
    for j,r in enumerate(s): per[j].append(((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Kernel_Name'][:40], (int(r['Start_Timestamp'])-int(s[0]['Start_Timestamp']))/1e3))
for j in sorted(per):
    d=[x[0] for x in per[j]]; st=[x[2] for x in per[j]]
    print(j, per[j][0][1], 'avg dur %.2f  avg start %+.2f' % (sum(d)/len(d), sum(st)/len(st)))
PY

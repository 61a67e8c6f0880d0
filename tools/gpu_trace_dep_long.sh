#!/bin/bash
# per-kernel durations of the flow pass early and late in a 300-frame loop
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/trace_long
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 300 > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv,glob,sys,re,collections
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
def short(n):
    m=re.search(r'(\w+_kernel)(<[^>]*>)?',n.replace('th::(anonymous namespace)::','')); return (m.group(1)+(m.group(2) or ''))[:40] if m else n[:30]
draws=[]; cur=None
for r in rows:
    n=r['Kernel_Name']; s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if 'deposit_raster_kernel' in n: cur=collections.OrderedDict()
    if cur is not None:
        k=short(n); cur[k]=cur.get(k,0)+(e-s)/1e3
        if 'deposit_blend_kernel' in n: draws.append(cur); cur=None
flow=[d for d in draws if any('FlowTarget' in k for k in d)]
for name,idx in (('frames 8-12',range(8,13)),('frames 145-150',range(145,150)),('frames 295-300',range(len(flow)-5,len(flow)))):
    acc=collections.OrderedDict()
    for i in idx:
        for k,v in flow[i].items(): acc[k]=acc.get(k,0)+v/len(list(idx))
    print(name, 'sum %.0f'%sum(acc.values()), {k:round(v) for k,v in acc.items() if v>15})
PY

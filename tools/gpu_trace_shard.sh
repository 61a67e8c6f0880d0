#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/trace_shard
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/frame_bench_dist.py 12 > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $OUT.log
python3 - $OUT <<'PY'
import csv,glob,sys,re
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
mc=glob.glob(sys.argv[1]+'/*/*memory_copy_trace.csv')
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','').replace('th::(anonymous namespace)::','')[:60]) for r in rows]
if mc:
    for r in csv.DictReader(open(mc[0])): ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r.get('Direction','')+' '+r.get('Bytes', r.get('Size',''))))
ev.sort()
idx=[i for i,e in enumerate(ev) if 'deposit_raster_kernel' in e[2]]
i0=idx[-3]; i1=idx[-2]
t0=ev[i0][0]; prev=t0
for s,e,n in ev[i0-6:i1]:
    print("%-64s start %8.1f dur %7.1f gap %6.1f"%(n,(s-t0)/1e3,(e-s)/1e3,(s-prev)/1e3)); prev=e
PY

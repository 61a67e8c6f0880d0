"""Average duration (us) of the kernels whose names hold the given patterns, from a rocprofv3 --kernel-trace --stats output directory:
    python3 tools/kernel_avgs.py <dir> <pattern> [pattern ...]"""
import csv,sys,glob
d=sys.argv[1]; pats=sys.argv[2:]
f=glob.glob(d+'/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
out=[]
for pat in pats:
    for r in rows:
        if pat in r['Name']:
            out.append("%s %.1f"%(pat, float(r['AverageNs'])/1e3)); break
print(d, ' | '.join(out))

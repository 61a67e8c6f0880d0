"""Per-kernel averages of the counters of tools/gpu_pmc_draw.sh: python tools/pmc_summary.py gpurun_out/r3/pmc_TAG [name filter...]"""
import csv, glob, sys, collections
out = sys.argv[1]
want = sys.argv[2:] or ["bins_", "crowd_", "giant_", "long_sort", "run_walk", "deposit_", "radix_"]
dur, acc = {}, collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[(f.split('/')[-3], r['Dispatch_Id'])] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    run = f.split('/')[-3]
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').replace('th::', '').replace('(anonymous namespace)::', '').split('(')[0]
        if not any(s in k for s in want):
            continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        acc[k]['_ns_' + run].append(dur.get((run, r['Dispatch_Id']), 0))
for k, cs in sorted(acc.items()):
    print('#', k)
    for c, v in sorted(cs.items()):
        print('   %-24s %16.1f (n=%d)' % (c, sum(v) / len(v), len(v)))

#!/bin/bash
# round 4: the frame loop's steady state (crowded target), kernel trace of the last frames: which kernel ends a draw()
# usage: gpu_r4_steady_trace.sh [frames] [extra arguments of deposit_bench.py, e.g. --wall] [last N frames to average]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
FRAMES=${1:-600}; EXTRA=${2:-}; export LAST=${3:-100}
cd /tmp; export TMPDIR=/tmp
D=$OUT/steady_trace
rm -rf $D
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $ROOT/tools/deposit_bench.py $FRAMES --both $EXTRA > $D.log 2>&1
grep '^{' $D.log | cut -c1-330
python3 - $D <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')
rows = list(csv.DictReader(open(fs[0])))
short = lambda n: n.replace("th::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows))
# frames: from one bins_fused_kernel start to the next
starts = [i for i, e in enumerate(ev) if e[2].startswith("bins_fused_kernel")]
per = collections.defaultdict(list)
for a, b in list(zip(starts, starts[1:]))[-int(__import__("os").environ.get("LAST", "100")):]:
    t0 = ev[a][0]
    frame = ev[a:b]
    end = max(e[1] for e in frame)
    for e in frame:
        per[(e[2], e[4])].append(((e[0] - t0) / 1e3, (e[1] - t0) / 1e3))
    per[("FRAME", "-")].append((0.0, (ev[b][0] - t0) / 1e3))
print("%-46s %-4s %6s %9s %9s %9s" % ("kernel (last 100 frames)", "strm", "n", "start us", "end us", "dur us"))
for k, v in sorted(per.items(), key=lambda kv: sum(x[0] for x in kv[1]) / len(kv[1])):
    n = len(v)
    print("%-46s %-4s %6d %9.1f %9.1f %9.1f" % (k[0], k[1], n, sum(x[0] for x in v) / n, sum(x[1] for x in v) / n, sum(x[1] - x[0] for x in v) / n))
PY

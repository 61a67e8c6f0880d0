#!/usr/bin/env python3
"""Where a frame of the loop `tick(); step(); draw()` stands still: from a rocprofv3 --kernel-trace CSV of tools/frame_wall_probe.py,
per frame (from one integrator launch to the next) the span, the time some kernel runs (union over the streams), the idle
gaps, and - over all frames - which kernel boundaries the largest gaps sit at.   python tools/frame_timeline.py trace.csv [skip]"""
import csv
import sys
from collections import defaultdict


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace("th::", "")[:48]


rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 10
starts = [k for k, r in enumerate(rows) if r[2].startswith(("logic_kernel", "logic_sorted_kernel"))]
frames = []
for a, b in zip(starts[skip:-1], starts[skip + 1:]):
    frames.append(rows[a:b])
gaps_at = defaultdict(list)
spans, busys, idles = [], [], []
for fr in frames:
    t0, t1 = fr[0][0], max(e for _, e, _ in fr)
    busy, cur_end, last = 0, t0, fr[0][2]
    for s, e, n in fr:
        if s > cur_end:
            gaps_at[(last, n)].append(s - cur_end)
            cur_s = s
        else:
            cur_s = cur_end
        if e > cur_end:
            busy += e - max(cur_s, s) if s > cur_end else e - cur_end
            cur_end, last = e, n
    spans.append(t1 - t0); busys.append(busy); idles.append(t1 - t0 - busy)
n = len(frames)
print("%d frames: span %.1f us (first kernel of a frame to its last kernel's end), some kernel running %.1f us, idle inside the frame %.1f us" %
      (n, sum(spans) / n / 1e3, sum(busys) / n / 1e3, sum(idles) / n / 1e3))
# between frames: from a frame's last kernel to the next frame's first
between = [frames[k + 1][0][0] - max(e for _, e, _ in frames[k]) for k in range(n - 1)]
print("between frames (last kernel's end -> next step's start): %.1f us on average" % (sum(between) / max(len(between), 1) / 1e3))
print("largest idle gaps inside a frame, by the kernels on either side (us per frame, occurrences per frame):")
for (a, b), v in sorted(gaps_at.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("  %-48s -> %-48s %7.1f  x%.2f" % (a, b, sum(v) / n / 1e3, len(v) / n))
per = defaultdict(lambda: [0, 0])
for fr in frames:
    for s, e, nme in fr:
        per[nme][0] += e - s; per[nme][1] += 1
print("kernels (us per frame, launches per frame):")
for nme, (tot, cnt) in sorted(per.items(), key=lambda kv: -kv[1][0])[:24]:
    print("  %-48s %8.1f  x%.2f" % (nme, tot / n / 1e3, cnt / n))

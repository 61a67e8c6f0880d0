#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output of tools/gpu_profile.sh: per-kernel averages of every counter
and the kernel-trace duration statistics."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()


def main(out):
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        print("# kernel trace stats:", os.path.relpath(f, out))
        for row in csv.DictReader(open(f)):
            print("  %-70s calls %6s avg %12.1f ns  total%% %6s" % (short(row["Name"])[:70], row["Calls"],
                                                                    float(row["AverageNs"]), row["Percentage"]))
    counters = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row.get("Kernel_Name", ""))
            counters[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in counters.items():
        print("# counters (average per dispatch):", k[:90])
        for c, vals in sorted(cs.items()):
            print("  %-26s %18.1f   (n=%d)" % (c, sum(vals) / len(vals), len(vals)))


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python
"""Average rocprofv3 --pmc counters per kernel over the dispatches in pmc_counter_collection.csv files.
Dispatches of one kernel are kept apart by grid size (the plain 2-D step kernel runs with interpolation blocks
in the forward loop and without them as the first step of a pair in the gradient loop).
Usage: pmc_summary.py <dir-with-p*/pmc_counter_collection.csv> [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "step"
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "p*", "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if sub not in k:
            continue
        acc[k.split("(")[0][-60:] + "  grid=" + row.get("Grid_Size", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s n=%5d  mean=%.6g" % (c, len(v), sum(v) / len(v)))

#!/usr/bin/env python3
"""Summarise a tools_profile.sh output directory: per-kernel stats and PMC sums
per dispatch of the trace kernel (counter values are summed over the rows
rocprofv3 emits per dispatch/dimension)."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, d))
    print(open(f).read())
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    agg = defaultdict(lambda: defaultdict(float))
    for r in rows:
        k = r.get("Kernel_Name", "")
        if "trace_kernel" not in k:
            continue
        agg[(r.get("Dispatch_Id"), k[:60])][r.get("Counter_Name")] += float(r.get("Counter_Value", 0))
    print("== pmc:", os.path.relpath(f, d))
    for (disp, k), cs in sorted(agg.items(), key=lambda x: int(x[0][0])):
        print("  dispatch", disp, k, {c: v for c, v in cs.items()})

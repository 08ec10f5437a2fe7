#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel stats and PMC sums
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

# traffic of the dominant kernel per launch (MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 64-B
# requests of 128-B wide reads as 64 B -> doubled; WRITE_SIZE exact), KB -> bytes.
# "per launch" = mean over the launches above 10 % of the largest (a step is one launch per
# batch of <= 2^27 rays; the filter only matters for odd-sized last batches).
import json
vals = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(d, "pmc_%s" % name, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "trace_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == name:
                per[r.get("Dispatch_Id")] += float(r.get("Counter_Value", 0))
        if per:
            top = max(per.values())
            main = [v for v in per.values() if v > 0.1 * top]
            vals[name] = sum(main) / len(main)
if len(vals) == 2:
    out = {"kernel": "trace_kernel", "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
           "hbm_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
           "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per main trace_kernel launch; fabric requests incl. Infinity-Cache hits"}
    # VALU issue occupancy of the same launches: wave-instructions x 4 cycles (wave64 on a 16-lane
    # SIMD) / (duration x 2.4 GHz x 1024 SIMDs) — what actually bounds the sorted-ray kernel
    try:
        valu = []
        for f in glob.glob(os.path.join(d, "pmc_SQ_INSTS_VALU*", "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(float)
            for r in csv.DictReader(open(f)):
                if "trace_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == "SQ_INSTS_VALU":
                    per[r.get("Dispatch_Id")] += float(r.get("Counter_Value", 0))
            top = max(per.values())
            valu = [v for v in per.values() if v > 0.1 * top]
        dur = []
        for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "trace_kernel" in r.get("Kernel_Name", ""):
                    dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        dur = [x for x in dur if x > 0.1 * max(dur)]
        if valu and dur:
            out["valu_insts_per_launch"] = sum(valu) / len(valu)
            out["avg_launch_ns"] = sum(dur) / len(dur)
            out["valu_issue_frac"] = out["valu_insts_per_launch"] * 4 / (out["avg_launch_ns"] * 1e-9 * 2.4e9 * 1024)
    except Exception as e:
        print("no VALU occupancy:", e)
    # the workload the numbers belong to (bench.py only quotes them for the same one)
    try:
        line = [l for l in open(os.path.join(d, "stats.log")) if l.startswith("{")][-1]
        cfg = json.loads(line)["config"]
        out.update(grid=cfg["grid"], rays=cfg["rays_per_gpu"], sticking=cfg["sticking"])
    except Exception as e:
        print("no bench line in stats.log:", e)
    print("== traffic:", json.dumps(out))
    json.dump(out, open(os.path.join(d, "traffic.json"), "w"))

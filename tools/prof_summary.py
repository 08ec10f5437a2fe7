#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory.

Prints the rocprofv3 per-kernel stats and, for the two hot kernels (trace_kernel, gen_kernel), the
PMC counters PER LAUNCH (mean over the main launches: those above 10 % of the largest value of the
counter, which only matters for odd-sized last batches), and writes counters.json:

  {lib_sha256, grid, rays, sticking, trace_kernel: {avg_ms, SQ_INSTS_VALU, ..., hbm_bytes, ...},
   gen_kernel: {...}}

bench.py quotes these counts in its roofline block only when lib_sha256 is the library it runs.
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md §HBM: FETCH_SIZE tallies the
128-B requests of wide reads at 64 B; Infinity-Cache hits are included in both)."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("trace_kernel", "gen_kernel")

for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, d))
    print(open(f).read())


def main_mean(per_dispatch):
    vals = list(per_dispatch.values())
    if not vals:
        return None
    top = max(vals)
    main = [v for v in vals if v > 0.1 * top] if top > 0 else vals
    return sum(main) / len(main)


out = {k: {} for k in KERNELS}
names = {}
# durations from the kernel trace
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_trace.csv"), recursive=True):
    per = {k: {} for k in KERNELS}
    for r in csv.DictReader(open(f)):
        for k in KERNELS:
            if k in r.get("Kernel_Name", ""):
                per[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                names[k] = r["Kernel_Name"].split("(")[0].replace("void ", "")
    for k in KERNELS:
        m = main_mean(per[k])
        if m:
            out[k]["avg_ms"] = m * 1e-6
            out[k]["launches"] = len([v for v in per[k].values() if v > 0.1 * max(per[k].values())])
# counters
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    per = {k: defaultdict(lambda: defaultdict(float)) for k in KERNELS}
    for r in csv.DictReader(open(f)):
        for k in KERNELS:
            if k in r.get("Kernel_Name", ""):
                per[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r.get("Counter_Value", 0))
    for k in KERNELS:
        for cname, disp in per[k].items():
            out[k][cname] = main_mean(disp)
for k in KERNELS:
    o = out[k]
    if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
        o["hbm_bytes"] = (2 * o["FETCH_SIZE"] + o["WRITE_SIZE"]) * 1024
    if o.get("SQ_ACTIVE_INST_VALU") and o.get("SQ_THREAD_CYCLES_VALU"):
        o["lanes_per_valu_instr"] = o["SQ_THREAD_CYCLES_VALU"] / o["SQ_ACTIVE_INST_VALU"]
    if o.get("GRBM_GUI_ACTIVE") and o.get("avg_ms"):
        o["clock_ghz_profiled"] = o["GRBM_GUI_ACTIVE"] / 8 / (o["avg_ms"] * 1e-3) / 1e9
    if o.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if o.get(c):
                o[c + "_frac"] = o[c] / o["SQ_WAVE_CYCLES"]
    if k in names:
        o["name"] = names[k]
    print("== per launch,", k)
    for c, v in sorted(o.items()):
        print("   %-28s %s" % (c, ("%.6g" % v) if isinstance(v, float) else v))

res = dict(trace_kernel=out["trace_kernel"], gen_kernel=out["gen_kernel"])
try:
    h = hashlib.sha256(open(os.path.join(root, "viennaray_amd", "libviennaray_amd.so"), "rb").read()).hexdigest()
    res["lib_sha256"] = h
except OSError:
    pass
try:  # ... and the sources it was built from (a rebuild of the same sources may differ in its bytes)
    hs = hashlib.sha256()
    src = os.path.join(root, "viennaray_amd", "csrc")
    for fn in sorted(os.listdir(src)):
        if fn.endswith((".hip", ".hpp", ".cpp")) or fn == "Makefile":
            hs.update(fn.encode() + b"\0" + open(os.path.join(src, fn), "rb").read())
    res["src_sha256"] = hs.hexdigest()
except OSError:
    pass
try:  # the workload the numbers belong to (bench.py only quotes them for the same one)
    line = [l for l in open(os.path.join(d, "stats.log")) if l.startswith("{")][-1]
    cfg = json.loads(line)["config"]
    res.update(grid=cfg["grid"], rays=cfg["rays_per_gpu"], sticking=cfg["sticking"])
except Exception as e:
    print("no bench line in stats.log:", e)
json.dump(res, open(os.path.join(d, "counters.json"), "w"), indent=1)
print("== counters.json written")

#!/usr/bin/env python3
"""Profiles ONE workload on the GPU box with rocprofv3 and reduces the result to per-launch figures of the two
hot kernels (trace_kernel, gen_kernel):

    python3 tools/pmc_profile.py <outdir> <case-name> -- python3 bench.py --steps 3 ...
    python3 tools/pmc_profile.py <outdir> <case-name> -- python3 tools/case_bench.py trench3d 0.1 2000 2

Passes (each its own run of the program; counters are NEVER combined with a trace domain, and at most one
TCC-heavy derived counter per pass — MI355X_MICROARCH.md "HBM / rocprofv3"):
    kernel-trace + stats | FETCH_SIZE | WRITE_SIZE | TCC hit / miss | TCC EA requests | SQ wave cycles / waits |
    SQ instruction classes | clock + VALU lanes | vector L1
Output: <outdir>/<case>_summary.txt (the rocprofv3 stats table + per-launch counters) and <outdir>/<case>.json:

    {case, cmd, rays, segments, trace_kernel: {name, avg_ms, launches, <counters>, derived...}, gen_kernel: {...}}

Derived per kernel (all per launch):
    hbm_bytes         (2 * FETCH_SIZE + WRITE_SIZE) * 1024: fabric bytes of the L2s (gfx950 tallies the 128-byte read
                      requests of wide loads at 64 B: the guide's correction; Infinity-Cache hits are included)
    l2_hit_rate       TCC_HIT / (TCC_HIT + TCC_MISS)
    dram_read_share   TCC_EA0_RDREQ_DRAM / TCC_EA0_RDREQ: share of the L2s' read requests routed to the DRAM path
                      (the path the memory-side Infinity Cache sits on; rocprofv3 exposes no MALL hit counter on gfx950)
    lanes_per_valu_instr  SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64)
    useful_lane_frac  SQ_THREAD_CYCLES_VALU / (64 lanes * device cycles * 1024 SIMDs * 0.5 VALU instr per cycle):
                      the share of the chip's VALU lane-slots that did work on an active lane — executing MORE
                      instructions on idle lanes cannot raise it
    wave_instr        sum of SQ_INSTS_{VALU,SALU,SMEM,VMEM_RD,VMEM_WR,LDS}; wave_instr_per_segment when the program
                      reports its trace segments
    wait_frac         SQ_WAIT_ANY / SQ_WAVE_CYCLES;  clock_ghz  GRBM_GUI_ACTIVE / 8 XCDs / kernel time
The program's own stdout is searched for a JSON line with rays / segments ("vr_case" lines of tools/case_bench.py,
the bench line of bench.py)."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

KERNELS = ("trace_kernel", "gen_kernel")
PASSES = [
    ("fetch", ["FETCH_SIZE"]),
    ("write", ["WRITE_SIZE"]),
    ("l2", ["TCC_HIT_sum", "TCC_MISS_sum"]),
    ("ea", ["TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_ATOMIC_sum", "TCC_EA0_RDREQ_DRAM_sum"]),
    ("wave", ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]),
    ("insts", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"]),
    ("lanes", ["GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_SCA"]),
    ("l1", ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TCC_ATOMIC_WITH_RET_REQ_sum"]),
]
INST_CLASSES = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS")


def main_mean(per_dispatch):
    """mean over the main launches (those above 10 % of the largest: odd-sized last batches, warm-ups of 1 ray)"""
    vals = list(per_dispatch.values())
    if not vals:
        return None
    top = max(vals)
    main = [v for v in vals if v > 0.1 * top] if top > 0 else vals
    return sum(main) / len(main)


def run(cmd, log):
    with open(log, "w") as fh:
        return subprocess.call(cmd, stdout=fh, stderr=subprocess.STDOUT, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))


def derive(o, segments):
    if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
        o["hbm_bytes"] = (2 * o["FETCH_SIZE"] + o["WRITE_SIZE"]) * 1024
    if o.get("TCC_HIT_sum") is not None and o.get("TCC_MISS_sum") is not None and o["TCC_HIT_sum"] + o["TCC_MISS_sum"] > 0:
        o["l2_hit_rate"] = o["TCC_HIT_sum"] / (o["TCC_HIT_sum"] + o["TCC_MISS_sum"])
    if o.get("TCC_EA0_RDREQ_sum"):
        o["dram_read_share"] = (o.get("TCC_EA0_RDREQ_DRAM_sum") or 0.0) / o["TCC_EA0_RDREQ_sum"]
    if o.get("SQ_ACTIVE_INST_VALU") and o.get("SQ_THREAD_CYCLES_VALU"):
        o["lanes_per_valu_instr"] = o["SQ_THREAD_CYCLES_VALU"] / o["SQ_ACTIVE_INST_VALU"]
    if o.get("GRBM_GUI_ACTIVE") and o.get("avg_ms"):
        o["clock_ghz"] = o["GRBM_GUI_ACTIVE"] / 8 / (o["avg_ms"] * 1e-3) / 1e9
    if o.get("GRBM_GUI_ACTIVE") and o.get("SQ_THREAD_CYCLES_VALU"):
        o["useful_lane_frac"] = o["SQ_THREAD_CYCLES_VALU"] / (64.0 * (o["GRBM_GUI_ACTIVE"] / 8) * 1024 * 0.5)
    if o.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if o.get(c):
                o[c + "_frac"] = o[c] / o["SQ_WAVE_CYCLES"]
    if all(c in o for c in INST_CLASSES):
        o["wave_instr"] = sum(o[c] for c in INST_CLASSES)
        if segments:
            o["wave_instr_per_segment"] = o["wave_instr"] / segments
    if o.get("TCP_TOTAL_CACHE_ACCESSES_sum") and o.get("TCP_TCC_READ_REQ_sum") is not None:
        o["l1_miss_per_access"] = o["TCP_TCC_READ_REQ_sum"] / o["TCP_TOTAL_CACHE_ACCESSES_sum"]


def main():
    if "--" not in sys.argv or len(sys.argv) < 5:
        raise SystemExit(__doc__)
    k = sys.argv.index("--")
    outdir, case = sys.argv[1], sys.argv[2]
    prog = sys.argv[k + 1:]
    skip = set(a[7:] for a in sys.argv[3:k] if a.startswith("--skip="))
    outdir = os.path.abspath(outdir)
    work = os.path.join(outdir, case + "_raw")
    os.makedirs(work, exist_ok=True)
    rc = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(work, "stats"), "--"] + prog,
             os.path.join(work, "stats.log"))
    print(f"[{case}] stats rc={rc}", flush=True)
    for name, ctrs in PASSES:
        if name in skip:
            continue
        rc = run(["rocprofv3", "--pmc"] + ctrs + ["--output-format", "csv", "-d", os.path.join(work, "pmc_" + name), "--"] + prog,
                 os.path.join(work, f"pmc_{name}.log"))
        print(f"[{case}] pmc {name} rc={rc}", flush=True)

    # what the program says about itself
    rays = segments = None
    try:
        for line in open(os.path.join(work, "stats.log")):
            if line.startswith("{"):
                j = json.loads(line)
                if "vr_case" in j:
                    rays, segments = j.get("rays"), j.get("segments")
                elif "config" in j:
                    rays, segments = j["config"].get("rays_per_gpu"), j.get("segments_per_step")
    except Exception as e:  # noqa: BLE001
        print("no case line:", e)

    # A scene with relief runs TWO trace kernels per batch (the flat-scene kernel over the tight bins, the structured-scene
    # kernel over the loose ones): dispatches are grouped by their full kernel name, reduced to a per-launch mean each, and
    # the trace phase is the SUM over the instantiations (their times add up to the trace_kernel_ms the library reports).
    def short(name):
        return name.split("(")[0].replace("void ", "")

    out = {kk: {} for kk in KERNELS}
    parts = {kk: {} for kk in KERNELS}   # kind -> full kernel name -> {avg_ms, counters}
    lines = []
    for f in glob.glob(os.path.join(work, "stats", "**", "*kernel_stats.csv"), recursive=True):
        lines.append("== rocprofv3 --kernel-trace --stats: " + " ".join(prog))
        lines.append(open(f).read())
    for f in glob.glob(os.path.join(work, "stats", "**", "*kernel_trace.csv"), recursive=True):
        per = {kk: defaultdict(dict) for kk in KERNELS}
        for r in csv.DictReader(open(f)):
            for kk in KERNELS:
                if kk in r.get("Kernel_Name", ""):
                    per[kk][short(r["Kernel_Name"])][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for kk in KERNELS:
            for nm, disp in per[kk].items():
                m = main_mean(disp)
                if m:
                    parts[kk].setdefault(nm, {})["avg_ms"] = m * 1e-6
                    parts[kk][nm]["launches"] = len([v for v in disp.values() if v > 0.1 * max(disp.values())])
    for f in sorted(glob.glob(os.path.join(work, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        per = {kk: defaultdict(lambda: defaultdict(lambda: defaultdict(float))) for kk in KERNELS}
        for r in csv.DictReader(open(f)):
            for kk in KERNELS:
                if kk in r.get("Kernel_Name", ""):
                    per[kk][short(r["Kernel_Name"])][r["Counter_Name"]][r["Dispatch_Id"]] += float(r.get("Counter_Value", 0))
        for kk in KERNELS:
            for nm, ctrs in per[kk].items():
                for cname, disp in ctrs.items():
                    parts[kk].setdefault(nm, {})[cname] = main_mean(disp)
    for kk in KERNELS:
        # (instantiations that only ran warm-up-sized launches — under 3 % of the phase's time — are left out)
        tot = sum(v.get("avg_ms", 0.0) for v in parts[kk].values())
        names = sorted((nm for nm, v in parts[kk].items() if v.get("avg_ms", 0.0) >= 0.03 * tot and tot > 0), key=lambda nm: -parts[kk][nm]["avg_ms"])
        if not names:
            continue
        out[kk]["name"] = " + ".join(names)
        out[kk]["launches"] = parts[kk][names[0]].get("launches")
        keys = set().union(*(parts[kk][nm].keys() for nm in names)) - {"launches"}
        for key in keys:
            vals = [parts[kk][nm].get(key) for nm in names]
            if all(v is not None for v in vals):
                out[kk][key] = sum(vals)
        if len(names) > 1:
            out[kk]["parts"] = {nm: {"avg_ms": parts[kk][nm].get("avg_ms")} for nm in names}
    for kk in KERNELS:
        derive(out[kk], segments if kk == "trace_kernel" else None)
        lines.append(f"== per launch, {kk}")
        for c, v in sorted(out[kk].items()):
            lines.append("   %-30s %s" % (c, ("%.6g" % v) if isinstance(v, float) else v))
        if len(parts[kk]) > 1:
            for nm, v in parts[kk].items():
                lines.append("   per instantiation: %-60s avg %.4f ms x %s launches" % (nm, v.get("avg_ms", 0.0), v.get("launches")))
    res = dict(case=case, cmd=" ".join(prog), rays=rays, segments=segments, trace_kernel=out["trace_kernel"],
               gen_kernel=out["gen_kernel"])
    json.dump(res, open(os.path.join(outdir, case + ".json"), "w"), indent=1)
    open(os.path.join(outdir, case + "_summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-60:]))


if __name__ == "__main__":
    main()

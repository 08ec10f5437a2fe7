#!/usr/bin/env python3
"""Collects what tools/pmc_profile.py left in a directory (one <case>.json + <case>_summary.txt per workload) into
profiles/ (tracked):
    profiles/<name>_counters.json        every case, with the library / source hashes and the commit it was taken at
    profiles/<name>_<case>_summary.txt   the rocprofv3 --kernel-trace --stats table + the per-launch counters
    profiles/counters_latest.json        the copy bench.py reads (used only when the hashes match the running build)
usage: tools/publish_counters.py <dir> <name> [issue_ceiling.json]"""
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, name = sys.argv[1], sys.argv[2]
res = {"cases": {}}
for f in sorted(glob.glob(os.path.join(src, "*.json"))):
    j = json.load(open(f))
    if "case" in j and "trace_kernel" in j:
        res["cases"][j["case"]] = j
        s = os.path.join(src, j["case"] + "_summary.txt")
        if os.path.exists(s):
            shutil.copy(s, os.path.join(ROOT, "profiles", f"{name}_{j['case']}_summary.txt"))
lib = os.path.join(ROOT, "viennaray_amd", "libviennaray_amd.so")
res["lib_sha256"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()
hs = hashlib.sha256()
csrc = os.path.join(ROOT, "viennaray_amd", "csrc")
for fn in sorted(os.listdir(csrc)):
    if fn.endswith((".hip", ".hpp", ".cpp")) or fn == "Makefile":
        hs.update(fn.encode() + b"\0" + open(os.path.join(csrc, fn), "rb").read())
res["src_sha256"] = hs.hexdigest()
try:
    res["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    if subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "viennaray_amd/csrc"], text=True).strip():
        res["commit"] += "+uncommitted-csrc"
except Exception:  # noqa: BLE001 (no git on the GPU box)
    res["commit"] = None
if len(sys.argv) > 3:
    rows = json.load(open(sys.argv[3]))["rows"]
    res["issue_ceiling"] = {f'{r["kind"]}@{r["waves_per_simd"]}w': round(r.get("per_simd_cycle", r.get("per_cu_cycle")), 4)
                            for r in rows if r["waves_per_simd"] in (6, 8)}
    shutil.copy(sys.argv[3], os.path.join(ROOT, "profiles", f"{name}_issue_ceiling.json"))
json.dump(res, open(os.path.join(ROOT, "profiles", f"{name}_counters.json"), "w"), indent=1)
json.dump(res, open(os.path.join(ROOT, "profiles", "counters_latest.json"), "w"), indent=1)
print("published", name, "cases", list(res["cases"]), "lib", res["lib_sha256"][:16], "commit", res["commit"])

#!/usr/bin/env python3
"""Copies what a tools/profile.sh run left under gpurun_out/prof_<tag>/ into profiles/ (tracked):
    profiles/<name>_rocprof_summary.txt   rocprofv3 --kernel-trace --stats table + per-launch PMC counters
    profiles/<name>_counters.json         the same counters as JSON
    profiles/counters_latest.json         ... the copy bench.py reads (only used when its lib_sha256 is the
                                          sha256 of the library that runs), plus the issue ceilings of
                                          tools/issue_ceiling.py if a run of it is given
usage: tools/publish_profile.py <gpurun_out/prof_dir> <name> [issue_ceiling.json]"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, name = sys.argv[1], sys.argv[2]
cj = json.load(open(os.path.join(src, "counters.json")))
lib = os.path.join(ROOT, "viennaray_amd", "libviennaray_amd.so")
sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()
if cj.get("lib_sha256") != sha:
    print("WARNING: the profile was taken with another build of the library:", cj.get("lib_sha256"), "!=", sha)
try:
    cj["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    if subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "viennaray_amd/csrc"], text=True).strip():
        cj["commit"] += "+uncommitted-csrc"
except Exception:
    cj["commit"] = None
if len(sys.argv) > 3:
    rows = json.load(open(sys.argv[3]))["rows"]
    cj["issue_ceiling"] = {f'{r["kind"]}@{r["waves_per_simd"]}w': round(r.get("per_simd_cycle", r.get("per_cu_cycle")), 4)
                           for r in rows if r["waves_per_simd"] in (6, 8)}
    shutil.copy(sys.argv[3], os.path.join(ROOT, "profiles", f"{name}_issue_ceiling.json"))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(ROOT, "profiles", f"{name}_rocprof_summary.txt"))
json.dump(cj, open(os.path.join(ROOT, "profiles", f"{name}_counters.json"), "w"), indent=1)
json.dump(cj, open(os.path.join(ROOT, "profiles", "counters_latest.json"), "w"), indent=1)
print("published", name, "lib", sha[:16], "commit", cj["commit"])

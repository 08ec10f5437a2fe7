#!/usr/bin/env python3
"""At home, after `gpurun -- bash tools/run_final.sh <tag>`: copies the round's evidence from gpurun_out/<tag>/ into
profiles/ (tracked): counters (+ counters_latest.json), per-case rocprof summaries, issue ceilings, the bench line,
the GPU test log, the case timings and the list of counters rocprofv3 offers on gfx950."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", tag)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "publish_counters.py"), src, tag,
                       os.path.join(src, "issue_ceiling.json")])
for fn, dst in (("bench.json", f"{tag}_bench.json"), ("pytest_gpu.log", f"{tag}_pytest_gpu.log"), ("cases.txt", f"{tag}_cases.txt")):
    if os.path.exists(os.path.join(src, fn)):
        shutil.copy(os.path.join(src, fn), os.path.join(ROOT, "profiles", dst))
av = os.path.join(src, "rocprof_avail.txt")
if os.path.exists(av):  # only the counter names: the full list is 3000 lines
    names = [l.split(":", 1)[1].strip() for l in open(av) if l.startswith("Counter_Name")]
    open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_avail.txt"), "w").write(
        "# counters rocprofv3 --list-avail offers on gfx950 (MI355X, ROCm 7.2): no MALL / Infinity-Cache hit counter among them\n"
        + "\n".join(sorted(set(names))) + "\n")
print("collected", tag)

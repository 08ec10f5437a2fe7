cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp VR_WAVEFRONT=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/wfprof -- python3 tools/case_bench.py ${@:-trench3d 0.1 1000 1} > /dev/null 2>&1
f=$(find gpurun_out/wfprof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "gen_kernel" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"]); prev=None
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    if "trace_kernel" in name: name = "trace_kernel" + name[name.index("<"):name.index(">")+1]
    print(f'{name[:44]:44s} start {(s-t0)/1e3:9.1f} us dur {(e-s)/1e3:8.1f} us gap {((s-prev)/1e3 if prev else 0):6.1f}')
    prev = e
PY

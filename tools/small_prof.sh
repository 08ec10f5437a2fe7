# Kernel-level timeline of a 10^6-ray launch (what a level-set time step pays): rocprofv3 kernel trace of three applies.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
for c in "trench3d 0.1 35 3" "plane100 0.1 100 3"; do
  tag=$(echo $c | cut -d' ' -f1)
  python3 tools/case_bench.py $c | tail -1
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/small_$tag -- python3 tools/case_bench.py $c > /dev/null 2>&1
  f=$(find gpurun_out/small_$tag -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last apply: from the last gen_kernel on
idx = max(i for i, r in enumerate(rows) if "gen_kernel" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
prev_end = None
for r in rows[max(0, idx - 3):]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f'  {r["Kernel_Name"][:60]:60s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:7.1f} us  grid {r.get("Grid_Size_X", r.get("Grid_Size", "?"))}')
    prev_end = e
PY
done

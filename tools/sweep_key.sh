# Sort-plane sweep (VR_KEY_COORD) on the trench workloads: where on the tracing axis the rays are binned
cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { echo "$1 :: $( "${@:2}" 2>/dev/null | tail -1 | grep -oE 'device [0-9.]+ ms trace_kernel [0-9.]+ ms')"; }
for c in "trench3d 1.0 10000 2" "trench3d 0.1 4000 2"; do
t "$c default" python3 tools/case_bench.py $c
for k in 0 -5 -10 -15 -30; do VR_KEY_COORD=$k t "$c key$k" python3 tools/case_bench.py $c; done
done
for c in "mesh 1.0 4000 2" "C4 2"; do
t "$c default" python3 tools/case_bench.py $c
for k in 0 -10 -20 -40; do VR_KEY_COORD=$k t "$c key$k" python3 tools/case_bench.py $c; done
done

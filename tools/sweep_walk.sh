cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { echo "$1 :: $( "${@:2}" 2>/dev/null | tail -1 | grep -oE 'trace_kernel [0-9.]+ ms')"; }
for c in "trench3d 0.1 4000 2" "C4 2" "mesh 0.1 4000 2" "trench3d 1.0 10000 2"; do
for e in 12 16 20 24 28 36; do VR_WALK_EXIT=$e t "$c exit$e" python3 tools/case_bench.py $c; done
for k in 10 18 25 35 50; do VR_WALK_PARK=$k t "$c park$k" python3 tools/case_bench.py $c; done
done

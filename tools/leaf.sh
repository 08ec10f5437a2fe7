cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { echo "$1 :: $( "${@:2}" 2>/dev/null | tail -1 | grep -oE 'device [0-9.]+ ms trace_kernel [0-9.]+ ms')"; }
for c in "trench3d 0.1 4000 2" "trench3d 1.0 10000 2" "mesh 0.1 4000 2" "C4 2" "C5p 2" "plane100 0.1 10000 2"; do
for l in 4 2 3 6 8; do
VR_LEAF_MAX=$l t "$c leaf$l" python3 tools/case_bench.py $c
done; done

cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { echo "$1 :: $( "${@:2}" 2>&1 | tail -1 | grep -oE 'trace_kernel [0-9.]+ ms')"; }
for c in "trench3d 0.1 1000 2" "mesh 0.1 2000 2" "plane100 0.1 3000 2" "trench2d 0.1 30000 2"; do
for v in "" _lean "" _lean; do
VR_LIB_PATH=$PWD/viennaray_amd/libviennaray_amd$v.so VR_CASE_PARTICLE=cosine2 t "$c cosine2 [$v]" python3 tools/case_bench.py $c
done; done

# A/B of library variants on the bounce-heavy cases: usage (on the box) bash tools/abx.sh <suffix> [<suffix> ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { echo "$1 :: $( "${@:2}" 2>/dev/null | tail -1 | grep -oE 'trace_kernel [0-9.]+ ms')"; }
for c in "trench3d 0.1 4000 2" "mesh 0.1 4000 2" "C4 2" "C5p 2"; do
for v in _prev "$@" _prev "$@"; do
VR_LIB_PATH=$PWD/viennaray_amd/libviennaray_amd$v.so t "$c [$v]" python3 tools/case_bench.py $c
done; done

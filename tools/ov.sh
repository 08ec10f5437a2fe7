cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { echo "$1 :: $( "${@:2}" 2>/dev/null | tail -1 | grep -oE 'device [0-9.]+ ms trace_kernel [0-9.]+ ms')"; }
for c in "trench3d 0.1 4000 2" "C4 2" "C5p 2"; do
t "$c base" python3 tools/case_bench.py $c
VR_BATCH_RAYS=25000000 t "$c batch25M" python3 tools/case_bench.py $c
VR_BATCH_RAYS=25000000 VR_OVERLAP=1 t "$c overlap(blocks-2)" python3 tools/case_bench.py $c
VR_BATCH_RAYS=25000000 VR_OVERLAP=1 VR_TRACE_BLOCKS=5 t "$c overlap blocks5" python3 tools/case_bench.py $c
VR_BATCH_RAYS=25000000 VR_OVERLAP=1 VR_TRACE_BLOCKS=6 t "$c overlap blocks6" python3 tools/case_bench.py $c
done

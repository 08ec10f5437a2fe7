# Slab-packet give-up thresholds on the structured scenes (env knobs only)
cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { echo "$1 :: $( "${@:2}" 2>/dev/null | tail -1 | grep -oE 'trace_kernel [0-9.]+ ms')"; }
for c in "trench3d 1.0 10000 2" "trench3d 0.1 4000 2" "mesh 1.0 4000 2" "C4 2"; do
t "$c default" python3 tools/case_bench.py $c
for b in 32 64 256 512; do VR_PACKET_BUDGET=$b t "$c budget$b" python3 tools/case_bench.py $c; done
for r in 2 5 8; do VR_PACKET_RATIO=$r t "$c ratio$r" python3 tools/case_bench.py $c; done
VR_PACKET_BUDGET=512 VR_PACKET_RATIO=8 t "$c budget512 ratio8" python3 tools/case_bench.py $c
VR_PQ_CAND=12 t "$c pqcand12" python3 tools/case_bench.py $c
VR_PQ_FRONTIER=24 t "$c pqfrontier24" python3 tools/case_bench.py $c
done

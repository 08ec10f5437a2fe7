cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { echo "$1 :: $( "${@:2}" 2>&1 | tail -1 | cut -c1-170)"; }
for c in "trench3d 0.1 1000 2" "plane100 0.1 3000 2" "trench2d 0.1 30000 2"; do
t "$c diffuse" python3 tools/case_bench.py $c
VR_CASE_PARTICLE=cosine2 t "$c cosine2(EXT)" python3 tools/case_bench.py $c
VR_CASE_PARTICLE=coned t "$c coned(EXT)" python3 tools/case_bench.py $c
done

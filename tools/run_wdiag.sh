cd $GRAFT_REPO_ROOT
export VR_LIB_PATH=$GRAFT_REPO_ROOT/viennaray_amd/libviennaray_amd_diag.so
for c in "trench3d 0.1 1000 1" "C4 1" "C5p 1" "plane100 0.1 10000 1"; do
echo "== $c"; python3 tools/case_bench.py $c 2>&1 | grep -E "^phase|segments" | cut -c1-200
done

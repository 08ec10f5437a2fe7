cd $GRAFT_REPO_ROOT
export VR_LIB_PATH=$GRAFT_REPO_ROOT/viennaray_amd/libviennaray_amd_diag.so
export VR_PRINT_LAUNCHES=1
for c in "ripple1000a0 1.0 100 1" "ripple1000a0.5 1.0 100 1" "ripple1000a0 0.1 100 1" "ripple1000a0.5 0.1 100 1"; do
echo "== $c"; python3 tools/case_bench.py $c 2>&1 | grep -E "^phase|^diag|launch|segments|spill" | cut -c1-400
done

# -DVR_DIAG build (make -C viennaray_amd/csrc diag): lane occupancy and phase shares of the flat-scene kernels on the
# plane and on the rippled sheet (tight launch alone: VR_SKIP_LOOSE=1), both stickings
cd $GRAFT_REPO_ROOT
export VR_LIB_PATH=$GRAFT_REPO_ROOT/viennaray_amd/libviennaray_amd_diag.so
export VR_PRINT_LAUNCHES=1
for c in "ripple1000a0 1.0 100 1" "ripple1000a0.5 1.0 100 1" "ripple1000a0 0.1 100 1" "ripple1000a0.5 0.1 100 1"; do
echo "== $c (tight launch alone)"; VR_SKIP_LOOSE=1 python3 tools/case_bench.py $c 2>&1 | grep -E "^phase|^diag|launch|segments|spill" | cut -c1-400
done

# -DVR_SELFCHECK build (make -C viennaray_amd/csrc check) over the round's workloads: every finished segment's closest hit
# (geometry and walls) against the escape-link walk, inside the trace kernels; prints the disagreement count of each run
cd $GRAFT_REPO_ROOT
export VR_LIB_PATH=$GRAFT_REPO_ROOT/viennaray_amd/libviennaray_amd_check.so
for c in "ripple1000a0 1.0 30 1" "ripple1000a0 0.1 30 1" "ripple1000a0.5 1.0 30 1" "ripple1000a0.5 0.1 30 1" "ripple1000a0.3p0.05 0.1 30 1" "plane100 0.1 3000 1" "trench3d 0.1 1000 1" "trench3d 1.0 1000 1" "mesh 0.1 500 1" "C4 1" "C5p 1"; do
  echo "== $c"; timeout -k 10 300 python3 tools/case_bench.py $c 2>&1 | grep -iE "selfcheck|disagree|Mrays" | cut -c1-200
done

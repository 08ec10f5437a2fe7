# ordered-walk experiment: closest-hit fuzz against the escape-link walk, GPU suite, self-check build, timings
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/walk
for c in mesh trench3d trench2d sphere; do timeout -k 10 200 python3 tools/walk_fuzz.py $c 2000000 2>&1 | grep -v "mismatches 0" ; done
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/walk/t2.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/walk/t2.log
for c in "trench3d 0.1 1000 1" "mesh 0.1 1000 1" "C4 1" "C5p 1"; do VR_LIB_PATH=$GRAFT_REPO_ROOT/viennaray_amd/libviennaray_amd_check.so timeout -k 10 200 python3 tools/case_bench.py $c 2>&1 | cut -c1-200; done
bash tools/cases.sh 2>&1

cd $GRAFT_REPO_ROOT
for i in 1 2 3; do VR_DEBUG_FLAGS=$((256+160)) python3 tools/case_bench.py C4 1 2>&1 | cut -c1-250; done
VR_DEBUG_FLAGS=256 python3 tools/case_bench.py C4 1 2>&1 | cut -c1-250
VR_DEBUG_FLAGS=256 python3 tools/case_bench.py mesh 0.1 4000 1 2>&1 | cut -c1-250
VR_DEBUG_FLAGS=256 python3 tools/case_bench.py trench3d 0.1 1000 1 2>&1 | cut -c1-250

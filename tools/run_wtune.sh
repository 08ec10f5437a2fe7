cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/walk
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/walk/t3.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/walk/t3.log
for park in 15 25 40 60; do for ex in 12 20 28; do
echo "== park $park exit $ex"
VR_WALK_PARK=$park VR_WALK_EXIT=$ex python3 tools/case_bench.py trench3d 0.1 2000 2 | tail -1 | cut -c30-130
VR_WALK_PARK=$park VR_WALK_EXIT=$ex python3 tools/case_bench.py C4 2 | tail -1 | cut -c30-130
VR_WALK_PARK=$park VR_WALK_EXIT=$ex python3 tools/case_bench.py C5p 2 | tail -1 | cut -c30-130
done; done

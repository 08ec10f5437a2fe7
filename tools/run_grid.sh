# cell-grid experiment: closest-hit parity through the grid, full GPU suite, then A/B timings
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/grid
VR_DEBUG_GRID=1 VR_GRID_VERBOSE=1 timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "closest_hit or intersection_known" > gpurun_out/grid/t1.log 2>&1; echo "grid closest-hit rc=$?"; tail -5 gpurun_out/grid/t1.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/grid/t2.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/grid/t2.log
for g in 0 1; do
echo "== VR_GRID=$g"
export VR_GRID=$g
VR_GRID_VERBOSE=1 timeout -k 10 120 python3 tools/case_bench.py trench3d 0.1 4000 2 2>&1 | tail -2
timeout -k 10 120 python3 tools/case_bench.py trench3d 1.0 10000 2 | tail -1
VR_GRID_VERBOSE=1 timeout -k 10 120 python3 tools/case_bench.py mesh 0.1 4000 2 2>&1| tail -2
timeout -k 10 120 python3 tools/case_bench.py mesh 1.0 4000 2 | tail -1
VR_GRID_VERBOSE=1 timeout -k 10 120 python3 tools/case_bench.py C4 2 2>&1| tail -2
VR_GRID_VERBOSE=1 timeout -k 10 120 python3 tools/case_bench.py C5p 2 2>&1| tail -2
done

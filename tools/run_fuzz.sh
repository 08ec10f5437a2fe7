cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/grid
for c in mesh trench3d trench2d sphere; do timeout -k 10 280 python3 tools/grid_fuzz.py $c 3000000 2>&1 | tee -a gpurun_out/grid/fuzz.log; done

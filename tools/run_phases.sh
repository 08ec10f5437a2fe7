# Lane occupancy and wave-time share of the phases of a round (refill, packets, walk, leaf tests, walls, state
# machine, ...) for the bounce-heavy workloads and C2, from the -DVR_DIAG build.
# usage (on the GPU box, after `make -C viennaray_amd/csrc diag`): bash tools/run_phases.sh
cd $GRAFT_REPO_ROOT
export VR_LIB_PATH=$GRAFT_REPO_ROOT/viennaray_amd/libviennaray_amd_diag.so
for c in "trench3d 0.1 1000 1" "trench3d 1.0 2000 1" "C4 1" "C5p 1" "plane100 0.1 10000 1"; do
echo "== $c"; python3 tools/case_bench.py $c 2>&1 | grep -E "^phase|^diag|segments" | cut -c1-200
done
for s in 1.0 0.1; do echo "== C2 $s"; python3 bench.py --steps 1 --warmup 0 --cpu-rays 0 --no-secondary --no-parity --sticking $s 2>&1 | grep -E "^phase|^diag" | sort -u | cut -c1-200; done

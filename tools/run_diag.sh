cd $GRAFT_REPO_ROOT
export VR_LIB_PATH=$GRAFT_REPO_ROOT/viennaray_amd/libviennaray_amd_diag.so
for s in 1.0 0.1; do
echo "== C2 diag sticking $s"
python3 bench.py --steps 1 --warmup 0 --cpu-rays 0 --no-secondary --sticking $s 2>&1 | grep -E "^diag|value" | cut -c1-200
done

set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
python3 tools/issue_ceiling.py gpurun_out/r02b/issue_ceiling.json > gpurun_out/r02b/issue_ceiling.log 2>&1; echo "issue rc=$?"
grep independent_mix gpurun_out/r02b/issue_ceiling.log
for c in "trench3d 0.1 4000 1" "mesh 0.1 4000 1" "plane100 0.1 10000 1" "C4 1" "C5p 1"; do
  echo "== diag $c"
  VR_LIB_PATH=$GRAFT_REPO_ROOT/viennaray_amd/libviennaray_amd_diag.so python3 tools/case_bench.py $c 2>&1 | tail -14
done > gpurun_out/r02b/diag.log 2>&1
cat gpurun_out/r02b/diag.log
(time python3 -m pytest tests -m gpu -x -q) > gpurun_out/r02b/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r02b/pytest_gpu.log

set -e
cd $GRAFT_REPO_ROOT
for f in 0 1 4 32 33; do
echo "== flags $f"
VR_DEBUG_FLAGS=$f python3 tools_case_bench.py trench3d 1.0 10000 2 | tail -1
VR_DEBUG_FLAGS=$f python3 tools_case_bench.py trench3d 0.1 4000 2 | tail -1
VR_DEBUG_FLAGS=$f python3 tools_case_bench.py mesh 0.1 4000 2 | tail -1
done

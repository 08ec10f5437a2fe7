set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python3 -m pytest tests -m gpu -x -q -k "c2 or rng_stream or c1_plane or device_bvh" > gpurun_out/r02a/pytest_c2.log 2>&1; echo "pytest rc=$?" 
tail -3 gpurun_out/r02a/pytest_c2.log
python3 tools/issue_ceiling.py gpurun_out/r02a/issue_ceiling.json > gpurun_out/r02a/issue_ceiling.log 2>&1; echo "issue rc=$?"
python3 bench.py --steps 5 --warmup 1 > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"
cat gpurun_out/r02a/bench.json | head -c 6000

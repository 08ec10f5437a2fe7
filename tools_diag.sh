cd $GRAFT_REPO_ROOT
echo "=== trench3d 0.1"
python3 tools_case_bench.py trench3d 0.1 4000 1 2>&1 | tail -14
echo "=== C2 0.1"
python3 bench.py --cpu-rays 0 --sticking 0.1 --steps 1 --warmup 0 2>&1 | grep diag | tail -12
echo "=== mesh 0.1"
python3 tools_case_bench.py mesh 0.1 4000 1 2>&1 | tail -14

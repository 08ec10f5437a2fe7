set -e
cd $GRAFT_REPO_ROOT
for l in 2 3 4 6; do
echo "== leafMax $l"
export VR_LEAF_MAX=$l
python3 bench.py --cpu-rays 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', d['value'], d.get('trace_kernel_ms'), d.get('device_pipeline_ms'))"
python3 bench.py --cpu-rays 0 --sticking 0.1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 s=0.1', d['value'], d.get('trace_kernel_ms'), d.get('device_pipeline_ms'))"
python3 tools_case_bench.py trench3d 0.1 4000 2 | tail -1
python3 tools_case_bench.py trench3d 1.0 10000 2 | tail -1
python3 tools_case_bench.py mesh 0.1 4000 2 | tail -1
python3 tools_case_bench.py trench2d 0.1 100000 2 | tail -1
done

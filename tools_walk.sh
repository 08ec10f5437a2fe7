set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python3 tools_case_bench.py trench3d 0.1 4000 2 | tail -1
python3 tools_case_bench.py trench3d 1.0 10000 2 | tail -1
python3 tools_case_bench.py mesh 0.1 4000 2 | tail -1
python3 tools_case_bench.py mesh 1.0 4000 2 | tail -1
python3 tools_case_bench.py trench2d 0.1 100000 2 | tail -1
python3 tools_case_bench.py plane100 0.1 10000 2 | tail -1
python3 bench.py --cpu-rays 0 --sticking 0.1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 s=0.1', d['value'], d.get('trace_kernel_ms'), d.get('device_pipeline_ms'))"
python3 bench.py --cpu-rays 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', d['value'], d.get('trace_kernel_ms'), d.get('device_pipeline_ms'))"
python3 bench.py --cpu-rays 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', d['value'], d.get('trace_kernel_ms'), d.get('device_pipeline_ms'))"

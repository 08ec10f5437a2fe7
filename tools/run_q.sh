cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python3 bench.py --cpu-rays 0 --no-secondary --no-parity 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 1.0:', d['value'], 'Mrays/s, trace_kernel', d['trace_kernel_ms'], 'ms, gen', d['gen_kernel_ms'])"; done
bash tools/cases.sh
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3

# usage (on the GPU box): bash tools/run_quick.sh [pytest -k expression]
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/quick
(python3 -m pytest tests -m gpu -x -q ${1:+-k "$1"}) > gpurun_out/quick/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/quick/pytest.log
python3 bench.py --steps 5 --warmup 1 --cpu-rays 0 --no-secondary 2>gpurun_out/quick/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C2 s=1:', d['value'], 'Mrays/s  trace', d['trace_kernel_ms'], 'gen', d['gen_kernel_ms'], 'ms/step', d['ms_per_step'])"
bash tools/cases.sh 2>&1 | tee gpurun_out/quick/cases.txt

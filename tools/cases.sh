#!/bin/bash
# Times the non-headline workloads (tests/golden fixtures) plus both C2 variants on the GPU box.
# usage (on the box): bash tools/cases.sh
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 tools/case_bench.py trench3d 0.1 4000 2 | tail -1
python3 tools/case_bench.py trench3d 1.0 10000 2 | tail -1
python3 tools/case_bench.py mesh 0.1 4000 2 | tail -1
python3 tools/case_bench.py mesh 1.0 4000 2 | tail -1
python3 tools/case_bench.py trench2d 0.1 100000 2 | tail -1
python3 tools/case_bench.py plane100 0.1 10000 2 | tail -1
python3 tools/case_bench.py C4 2 | tail -1
python3 tools/case_bench.py C5p 2 | tail -1
python3 tools/case_bench.py C5r 2 | tail -1
for s in 0.1 1.0; do
python3 bench.py --cpu-rays 0 --sticking $s 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 sticking $s:', d['value'], 'Mrays/s, trace_kernel', d['trace_kernel_ms'], 'ms, device', d['device_pipeline_ms'], 'ms')"
done

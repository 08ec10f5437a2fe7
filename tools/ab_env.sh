#!/bin/bash
# Same-box A/B of two environment settings on the working tree's library, alternating A B A B.
# usage (on the GPU box): bash tools/ab_env.sh "VR_QUEUES=1" "VR_QUEUES=8" [quick]
cd ${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2
run() { # label, command...
  local label=$1; shift
  for rep in 1 2; do
    for cfg in "$A" "$B"; do
      echo "$label [$cfg] $(env $cfg "$@" 2>/dev/null | tail -1 | grep -oE 'device [0-9.]+ ms trace_kernel [0-9.]+ ms')"
    done
  done
}
for s in 1.0 0.1; do for rep in 1 2; do for cfg in "$A" "$B"; do
  echo "C2 $s [$cfg] $(env $cfg python3 bench.py --cpu-rays 0 --no-secondary --sticking $s 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('device', d['device_pipeline_ms'], 'ms trace_kernel', d['trace_kernel_ms'], 'ms gen', d['gen_kernel_ms'])")"
done; done; done
run "plane100 0.1" python3 tools/case_bench.py plane100 0.1 10000 2
run "trench3d 0.1" python3 tools/case_bench.py trench3d 0.1 4000 2
run "trench3d 1.0" python3 tools/case_bench.py trench3d 1.0 10000 2
if [ "$3" != quick ]; then
run "mesh 0.1    " python3 tools/case_bench.py mesh 0.1 4000 2
run "C4          " python3 tools/case_bench.py C4 2
run "C5p         " python3 tools/case_bench.py C5p 2
run "trench3d 1e6" python3 tools/case_bench.py trench3d 0.1 35 5
fi

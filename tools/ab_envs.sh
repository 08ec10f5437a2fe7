#!/bin/bash
# One library, several environment settings, on case_bench.py workloads, alternating.
# usage (on the GPU box): ENVS="VR_MORTON_ANISO=1 VR_MORTON_ANISO=2" bash tools/ab_envs.sh "ripple1000a1 0.1 30 2" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for c in "$@"; do for rep in 1 2; do for e in $ENVS; do
  echo "$c [$e] $(env $e python3 tools/case_bench.py $c 2>/dev/null | tail -2 | head -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('mode', d['mode'], 'device %.2f trace %.2f ms' % (d['device_ms'], d['trace_kernel_ms']))")"
done; done; done

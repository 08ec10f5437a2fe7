#!/bin/bash
# Same-box comparison of several builds of the library on case_bench.py workloads, alternating.
# usage (on the GPU box): LIBS="libviennaray_amd.so libviennaray_amd_o6.so" bash tools/ab_cases.sh "ripple1000a1 0.1 30 2" "ripple1000a3 0.1 30 2" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for c in "$@"; do for rep in 1 2; do for lib in $LIBS; do
  echo "$c [$lib] $(VR_LIB_PATH=$PWD/viennaray_amd/$lib python3 tools/case_bench.py $c 2>/dev/null | tail -2 | head -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('mode', d['mode'], 'segments', d['segments'], 'device %.2f trace %.2f ms' % (d['device_ms'], d['trace_kernel_ms']))")"
done; done; done

#!/bin/bash
# Same-box comparison of several builds of the library on one bench configuration, alternating.
# usage (on the GPU box): bash tools/ab_libs.sh "<sticking>" lib1.so lib2.so ...   (paths relative to viennaray_amd/)
cd ${GRAFT_REPO_ROOT:-/root/repo}
S=$1; shift
for rep in 1 2; do for lib in "$@"; do
  echo "C2 $S [$lib] $(VR_LIB_PATH=$PWD/viennaray_amd/$lib python3 bench.py --cpu-rays 0 --no-secondary --sticking $S 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('device', d['device_pipeline_ms'], 'ms trace_kernel', d['trace_kernel_ms'], 'ms gen', d['gen_kernel_ms'])")"
done; done

#!/bin/bash
# Same-box A/B of the working tree's library against libviennaray_amd_prev.so on chosen cases, alternating.
# usage (on the GPU box): bash tools/ab_lib.sh flat|bounce|all
cd ${GRAFT_REPO_ROOT:-/root/repo}
NEW=${NEW_LIB:-$PWD/viennaray_amd/libviennaray_amd.so}
OLD=${OLD_LIB:-$PWD/viennaray_amd/libviennaray_amd_prev.so}
run() { local label=$1; shift
  for rep in 1 2; do for lib in OLD NEW; do
    echo "$label [$lib] $(VR_LIB_PATH=${!lib} "$@" 2>/dev/null | tail -1 | grep -oE 'device [0-9.]+ ms trace_kernel [0-9.]+ ms')"
  done; done; }
c2() { for rep in 1 2; do for lib in OLD NEW; do
  echo "C2 $1 [$lib] $(VR_LIB_PATH=${!lib} python3 bench.py --cpu-rays 0 --no-secondary --sticking $1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('device', d['device_pipeline_ms'], 'ms trace_kernel', d['trace_kernel_ms'], 'ms gen', d['gen_kernel_ms'])")"
done; done; }
if [ "$1" = flat ] || [ "$1" = all ]; then
  c2 0.1; c2 1.0
  run "plane100 0.1" python3 tools/case_bench.py plane100 0.1 10000 2
  VR_CASE_PARTICLE=cosine2 run "plane100 cos2" python3 tools/case_bench.py plane100 0.1 3000 2
fi
if [ "$1" = bounce ] || [ "$1" = all ]; then
  run "trench3d 0.1" python3 tools/case_bench.py trench3d 0.1 2000 2
  run "trench3d 1.0" python3 tools/case_bench.py trench3d 1.0 10000 2
  run "mesh 0.1    " python3 tools/case_bench.py mesh 0.1 4000 2
  run "C4          " python3 tools/case_bench.py C4 2
  run "C5p         " python3 tools/case_bench.py C5p 2
  run "trench3d 1e6" python3 tools/case_bench.py trench3d 0.1 35 5
fi

#!/bin/bash
# Same-box A/B of the working tree's library against libviennaray_amd_prev.so (tools/build_prev.sh):
# the bounce-heavy cases and both C2 variants, alternating A B A B so drift of the box shows.
# usage (on the GPU box): bash tools/ab.sh [quick]
cd ${GRAFT_REPO_ROOT:-/root/repo}
NEW=$PWD/viennaray_amd/libviennaray_amd.so
OLD=$PWD/viennaray_amd/libviennaray_amd_prev.so
run() { # label, command...
  local label=$1; shift
  for rep in 1 2; do
    for lib in OLD NEW; do
      local path=${!lib}
      echo "$label [$lib] $(VR_LIB_PATH=$path "$@" 2>/dev/null | tail -1 | grep -oE 'device [0-9.]+ ms trace_kernel [0-9.]+ ms')"
    done
  done
}
run "trench3d 0.1" python3 tools/case_bench.py trench3d 0.1 4000 2
run "trench3d 1.0" python3 tools/case_bench.py trench3d 1.0 10000 2
run "mesh 0.1    " python3 tools/case_bench.py mesh 0.1 4000 2
run "C4          " python3 tools/case_bench.py C4 2
run "C5p         " python3 tools/case_bench.py C5p 2
run "trench3d 1e6" python3 tools/case_bench.py trench3d 0.1 35 5
if [ "$1" != quick ]; then
run "plane100 0.1" python3 tools/case_bench.py plane100 0.1 10000 2
for s in 0.1 1.0; do for rep in 1 2; do for lib in OLD NEW; do
  echo "C2 $s [$lib] $(VR_LIB_PATH=${!lib} python3 bench.py --cpu-rays 0 --no-secondary --sticking $s 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('device', d['device_pipeline_ms'], 'ms trace_kernel', d['trace_kernel_ms'], 'ms gen', d['gen_kernel_ms'])")"
done; done; done
fi

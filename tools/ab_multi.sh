#!/bin/bash
# Same-box comparison of several builds of the library (viennaray_amd/libviennaray_amd<suffix>.so) on the
# bounce-heavy and flat cases, two rounds.   usage (on the GPU box): bash tools/ab_multi.sh "" _vA _vB ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { local label=$1; shift
  for rep in 1 2; do for sfx in "${LIBS[@]}"; do
    echo "$label [${sfx:-base}] $(VR_LIB_PATH=$PWD/viennaray_amd/libviennaray_amd$sfx.so "$@" 2>/dev/null | tail -1 | grep -oE 'device [0-9.]+ ms trace_kernel [0-9.]+ ms')"
  done; done; }
LIBS=("$@")
run "trench3d 0.1" python3 tools/case_bench.py trench3d 0.1 2000 2
run "trench3d 1e6" python3 tools/case_bench.py trench3d 0.1 35 5
run "mesh 0.1    " python3 tools/case_bench.py mesh 0.1 4000 2
run "C4          " python3 tools/case_bench.py C4 2
run "C5p         " python3 tools/case_bench.py C5p 2
run "plane100 0.1" python3 tools/case_bench.py plane100 0.1 10000 2
for rep in 1 2; do for sfx in "${LIBS[@]}"; do
  echo "C2 0.1 [${sfx:-base}] $(VR_LIB_PATH=$PWD/viennaray_amd/libviennaray_amd$sfx.so python3 bench.py --cpu-rays 0 --no-secondary --sticking 0.1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('device', d['device_pipeline_ms'], 'ms trace_kernel', d['trace_kernel_ms'], 'ms gen', d['gen_kernel_ms'])")"
done; done

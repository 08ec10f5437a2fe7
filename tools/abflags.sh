# Library variants (compiler-flag experiments) on the headline and the bounce-heavy cases
cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { echo "$1 :: $( "${@:2}" 2>/dev/null | tail -1 | grep -oE 'trace_kernel [0-9.]+ ms')"; }
for v in "" "$@" ""; do
  L=$PWD/viennaray_amd/libviennaray_amd$v.so
  echo "== [$v]"
  VR_LIB_PATH=$L python3 bench.py --cpu-rays 0 --no-secondary --sticking 1.0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 1.0 trace', d['trace_kernel_ms'], 'gen', d['gen_kernel_ms'])"
  VR_LIB_PATH=$L python3 bench.py --cpu-rays 0 --no-secondary --sticking 0.1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 0.1 trace', d['trace_kernel_ms'], 'gen', d['gen_kernel_ms'])"
  VR_LIB_PATH=$L t "trench3d 0.1" python3 tools/case_bench.py trench3d 0.1 4000 2
  VR_LIB_PATH=$L t "trench3d 1.0" python3 tools/case_bench.py trench3d 1.0 10000 2
  VR_LIB_PATH=$L t "C4" python3 tools/case_bench.py C4 2
  VR_LIB_PATH=$L t "C5p" python3 tools/case_bench.py C5p 2
done

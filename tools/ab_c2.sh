# Same-box A/B of the headline workload (C2, both stickings) and plane100: working tree against libviennaray_amd_prev.so
cd ${GRAFT_REPO_ROOT:-/root/repo}
NEW=$PWD/viennaray_amd/libviennaray_amd.so
OLD=$PWD/viennaray_amd/libviennaray_amd_prev.so
for s in ${1:-1.0 0.1}; do for rep in 1 2 3; do for lib in OLD NEW; do
  echo "C2 $s [$lib] $(VR_LIB_PATH=${!lib} python3 bench.py --cpu-rays 0 --no-secondary --sticking $s 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Mrays/s', d['value'], 'device', d['device_pipeline_ms'], 'ms trace_kernel', d['trace_kernel_ms'], 'ms gen', d['gen_kernel_ms'])")"
done; done; done

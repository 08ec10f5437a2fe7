cd $GRAFT_REPO_ROOT
for r in 1 2; do for lib in libviennaray_amd_prev.so libviennaray_amd.so; do export VR_LIB_PATH=$GRAFT_REPO_ROOT/viennaray_amd/$lib; echo "== $lib"
python3 tools/case_bench.py plane100 0.1 10000 1 | tail -1 | cut -c1-140
python3 tools/case_bench.py plane100 0.1 100 1 | tail -1 | cut -c1-140
python3 bench.py --cpu-rays 0 --no-secondary --no-parity --sticking 0.1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 0.1:', d['value'], 'Mrays/s, trace_kernel', d['trace_kernel_ms'], 'ms, gen', d['gen_kernel_ms'])"
done; done

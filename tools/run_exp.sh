cd $GRAFT_REPO_ROOT
for i in 1 2; do
for lib in libviennaray_amd.so libviennaray_amd_exp.so; do
echo "== $lib"
export VR_LIB_PATH=$GRAFT_REPO_ROOT/viennaray_amd/$lib
python3 tools/case_bench.py trench3d 0.1 4000 2 | tail -1
python3 tools/case_bench.py C5p 2 | tail -1
python3 tools/case_bench.py trench3d 1.0 10000 2 | tail -1
python3 bench.py --steps 5 --warmup 1 --cpu-rays 0 --no-secondary --no-parity --sticking 0.1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C2 s=0.1', d['value'], 'Mrays/s  trace', d['trace_kernel_ms'])"
done; done

cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in viennaray_amd/lib_ab_old.so viennaray_amd/libviennaray_amd.so; do
echo "== $lib"
VR_LIB_PATH=$GRAFT_REPO_ROOT/$lib python3 bench.py --cpu-rays 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', d['value'], d.get('trace_kernel_ms'), d.get('device_pipeline_ms'))"
VR_LIB_PATH=$GRAFT_REPO_ROOT/$lib python3 bench.py --cpu-rays 0 --sticking 0.1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 s=0.1', d['value'], d.get('trace_kernel_ms'), d.get('device_pipeline_ms'))"
done
done
echo "== new, leaf 4"
VR_LEAF_MAX=4 python3 bench.py --cpu-rays 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', d['value'], d.get('trace_kernel_ms'), d.get('device_pipeline_ms'))"
echo "== new, no order"
VR_NO_CHILD_ORDER=1 python3 bench.py --cpu-rays 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', d['value'], d.get('trace_kernel_ms'), d.get('device_pipeline_ms'))"
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python3 tools_case_bench.py trench3d 0.1 4000 2 | tail -1
python3 tools_case_bench.py trench3d 1.0 10000 2 | tail -1
python3 tools_case_bench.py mesh 0.1 4000 2 | tail -1

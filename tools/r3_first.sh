# round 3, first GPU call: the new tests, host-time breakdown, available counters, overlap re-test
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
python3 -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3a/pytest.log
python3 tools/apply_wall.py > gpurun_out/r3a/apply_wall.txt 2>&1; cat gpurun_out/r3a/apply_wall.txt
(cd /tmp && TMPDIR=/tmp rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/r3a/avail.txt 2>&1); grep -i -c "counter" gpurun_out/r3a/avail.txt
for cfg in "" "VR_OVERLAP=1 VR_BATCH_RAYS=50000000" "VR_OVERLAP=1 VR_BATCH_RAYS=34000000" "VR_OVERLAP=1 VR_BATCH_RAYS=25000000"; do
  echo "== $cfg"; env $cfg python3 bench.py --steps 10 --warmup 2 --cpu-rays 0 --no-secondary 2>&1 | python3 -c "import sys,json; [print({k:j[k] for k in ('value','ms_per_step','device_pipeline_ms','trace_kernel_ms','gen_kernel_ms')}) for j in (json.loads(l) for l in sys.stdin if l.startswith('{'))]"
done

set -e
cd $GRAFT_REPO_ROOT
for k in probe 0 -1 -5 -30; do
echo "== key $k"
if [ $k = probe ]; then E=""; else E="VR_NO_PROBE=1 VR_KEY_COORD=$k"; fi
env $E python3 tools_case_bench.py trench3d 1.0 10000 2 | tail -1
env $E python3 tools_case_bench.py trench3d 0.1 4000 2 | tail -1
env $E python3 tools_case_bench.py mesh 0.1 4000 2 | tail -1
env $E python3 tools_case_bench.py mesh 1.0 4000 2 | tail -1
done

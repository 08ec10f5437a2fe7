# parameter sweep of the relief kernels on the rippled 10^6-disk sheet (tools/case_bench.py): env settings one per line
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; mkdir -p $out
st=${2:-1.0}
while read -r envs; do
  echo "== $envs"
  env $envs VR_PRINT_LAUNCHES=1 timeout -k 10 120 python3 tools/case_bench.py ripple1000a0.5 $st 100 2 2>&1 | grep -E "launch|Mrays" | tail -3
done > $out/sweep_$st.txt
cat $out/sweep_$st.txt

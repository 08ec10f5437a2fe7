# A/B of environment settings on one case of tools/case_bench.py: tools/env_sweep.sh <out dir> "<case args>" < file with one env setting per line
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; mkdir -p $out
tag=$(echo "$2" | tr ' ./' '___')
while read -r envs; do
  echo "== $envs"
  env $envs VR_PRINT_LAUNCHES=1 timeout -k 10 150 python3 tools/case_bench.py $2 2>&1 | grep -E "Mrays" | tail -2
done > $out/sweep_$tag.txt
cat $out/sweep_$tag.txt

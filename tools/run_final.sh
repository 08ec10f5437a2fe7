# Final evidence of a round, on the GPU box: profile (stats + PMC) of the headline workload, issue ceilings,
# the full bench line, the secondary cases.   usage: bash tools/run_final.sh <tag>
set -o pipefail
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
mkdir -p gpurun_out/$TAG
bash tools/profile.sh $TAG > gpurun_out/$TAG/profile.log 2>&1; echo "profile rc=$?"
python3 tools/issue_ceiling.py gpurun_out/$TAG/issue_ceiling.json > gpurun_out/$TAG/issue_ceiling.log 2>&1; echo "issue rc=$?"
bash tools/pmc_case.sh ${TAG}_c2s01 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --cpu-rays 0 --no-secondary --sticking 0.1 > gpurun_out/$TAG/pmc_c2_s0.1.txt 2>&1
bash tools/pmc_case.sh ${TAG}_t3d $GRAFT_REPO_ROOT/tools/case_bench.py trench3d 0.1 4000 1 > gpurun_out/$TAG/pmc_trench3d_s0.1.txt 2>&1
bash tools/pmc_case.sh ${TAG}_c4 $GRAFT_REPO_ROOT/tools/case_bench.py C4 1 > gpurun_out/$TAG/pmc_C4.txt 2>&1
cd $GRAFT_REPO_ROOT
# (on the box: bench.py then finds the counters of the very build it runs; publish again at home from gpurun_out/)
python3 tools/publish_profile.py gpurun_out/prof_$TAG ${TAG}_box gpurun_out/$TAG/issue_ceiling.json > gpurun_out/$TAG/publish.log 2>&1
python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"
bash tools/cases.sh > gpurun_out/$TAG/cases.txt 2>&1; echo "cases rc=$?"
cat gpurun_out/$TAG/cases.txt
head -c 1500 gpurun_out/$TAG/bench.json

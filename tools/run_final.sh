# Final evidence of a round, on the GPU box: PMC profiles (headline + secondary workloads), issue ceilings, the full
# bench line (with the counters of the very build it runs), the GPU test log, the per-workload device times.
#   usage: bash tools/run_final.sh <tag>          then, at home:  python3 tools/collect_final.py <tag>
set -o pipefail
cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
O=gpurun_out/$TAG
mkdir -p $O
bash tools/run_profiles.sh $TAG > $O/profiles.log 2>&1; echo "profiles rc=$?"
python3 tools/issue_ceiling.py $O/issue_ceiling.json > $O/issue_ceiling.log 2>&1; echo "issue rc=$?"
# (on the box: bench.py then finds the counters of the very build it runs; published again at home from gpurun_out/)
python3 tools/publish_counters.py $O ${TAG}_box $O/issue_ceiling.json > $O/publish.log 2>&1; echo "publish rc=$?"
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
bash tools/cases.sh > $O/cases.txt 2>&1; echo "cases rc=$?"
(cd /tmp && TMPDIR=/tmp rocprofv3 --list-avail > $GRAFT_REPO_ROOT/$O/rocprof_avail.txt 2>&1)
cat $O/cases.txt
head -c 1200 $O/bench.json

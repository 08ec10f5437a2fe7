# PMC profiles of the headline and the named secondary workloads with the build in the tree (on the GPU box):
#   bash tools/run_profiles.sh <tag> [cases...]      cases default: C2_s1.0 C2_s0.1 C1_trench3d C4 C5p C2_rippled_s1.0 C2_rippled_s0.1
# Leaves gpurun_out/<tag>/{<case>.json, <case>_summary.txt}; tools/publish_counters.py copies them into profiles/.
cd $GRAFT_REPO_ROOT
TAG=$1; shift
CASES=${@:-C2_s1.0 C2_s0.1 C1_trench3d C4 C5p C2_rippled_s1.0 C2_rippled_s0.1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
for c in $CASES; do
  case $c in
    C2_s1.0) CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-secondary" ;;
    C2_s0.1) CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-secondary --sticking 0.1" ;;
    C1_trench3d) CMD="python3 $GRAFT_REPO_ROOT/tools/case_bench.py trench3d 0.1 2000 3" ;;
    C4) CMD="python3 $GRAFT_REPO_ROOT/tools/case_bench.py C4 3" ;;
    C5p) CMD="python3 $GRAFT_REPO_ROOT/tools/case_bench.py C5p 3" ;;
    C2_rippled_s1.0) CMD="python3 $GRAFT_REPO_ROOT/tools/case_bench.py ripple1000a0.5 1.0 100 3" ;;
    C2_rippled_s0.1) CMD="python3 $GRAFT_REPO_ROOT/tools/case_bench.py ripple1000a0.5 0.1 100 3" ;;
    *) echo "unknown case $c"; continue ;;
  esac
  python3 tools/pmc_profile.py $OUT $c -- $CMD > $OUT/$c.log 2>&1; echo "$c rc=$?"; tail -4 $OUT/$c.log
done

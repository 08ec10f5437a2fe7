# A/B of library variants (tools/build_variant.sh) on cases of tools/case_bench.py, all in one call (boxes differ by +-8 %):
#   bash tools/ab_cases.sh "<tag> <tag> ..." "<case args>" ["<case args>" ...]      tag "prod" = the production library
cd $GRAFT_REPO_ROOT
tags=$1; shift
for c in "$@"; do
  for t in $tags; do
    lib=$GRAFT_REPO_ROOT/viennaray_amd/libviennaray_amd_$t.so; [ $t = prod ] && lib=$GRAFT_REPO_ROOT/viennaray_amd/libviennaray_amd.so
    echo -n "$t | "; VR_LIB_PATH=$lib timeout -k 10 200 python3 tools/case_bench.py $c 2>&1 | grep -E "Mrays" | tail -1
  done
done

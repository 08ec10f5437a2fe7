# The ordered per-lane walk against the escape-link walk it replaced: closest-hit fuzz (millions of rays per
# geometry through vr_debug_intersect), then whole traces with the -DVR_SELFCHECK build, which re-walks every
# finished segment with the reference walk inside the trace kernel and counts disagreements.
# usage (on the GPU box, after `make -C viennaray_amd/csrc check`): bash tools/check_walk.sh
cd $GRAFT_REPO_ROOT
for c in mesh trench3d trench2d sphere; do timeout -k 10 200 python3 tools/walk_fuzz.py $c 2000000 2>&1 | grep -v "mismatches 0" ; done
for c in "trench3d 0.1 1000 1" "mesh 0.1 1000 1" "C4 1" "C5p 1"; do VR_LIB_PATH=$GRAFT_REPO_ROOT/viennaray_amd/libviennaray_amd_check.so timeout -k 10 200 python3 tools/case_bench.py $c 2>&1 | cut -c1-200; done

cd $GRAFT_REPO_ROOT
for pk in 10 34 60 100; do
for w in 1 32; do
export VR_WALK_EXIT=$w VR_WALK_PARK=$pk
echo "=== park $pk walkExit $w: trench3d 0.1"
python3 tools_case_bench.py trench3d 0.1 4000 1 2>&1 | grep -E "walk steps|leaf prim|state mach|Mrays"
done
done

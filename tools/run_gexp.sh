cd $GRAFT_REPO_ROOT
for park in 25 60 100; do for h in 1.6 2.3 3.2; do
echo "== park $park H $h"; VR_GRID_VERBOSE=1 VR_WALK_PARK=$park VR_GRID_H=$h python3 tools/case_bench.py trench3d 0.1 2000 2 2>&1 | grep -E "cell grid|segments" | tail -2 | cut -c1-200
done; done
for park in 25 100; do for h in 1.0 1.5 2.0; do
echo "== mesh park $park H $h"; VR_GRID_VERBOSE=1 VR_WALK_PARK=$park VR_GRID_H=$h python3 tools/case_bench.py mesh 0.1 2000 2 2>&1 | grep -E "cell grid|segments" | tail -2 | cut -c1-200
done; done

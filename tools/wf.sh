cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { echo "$1 :: $( "${@:2}" 2>&1 | tail -1 | cut -c1-200)"; }
c="trench3d 0.1 4000 2"
for G in 4 12; do VR_WAVEFRONT=1 VR_WF_GENS=$G t "$c wf G$G" python3 tools/case_bench.py $c; done
VR_WAVEFRONT=1 VR_WF_GENS=3 t "C4 wf G3" python3 tools/case_bench.py C4 2
VR_WAVEFRONT=1 VR_WF_GENS=3 t "mesh wf G3" python3 tools/case_bench.py mesh 0.1 4000 2
VR_WAVEFRONT=1 VR_WF_MIN_RAYS=0 VR_WF_GENS=3 timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3

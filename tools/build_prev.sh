# Builds the COMMITTED state (HEAD) of the library as viennaray_amd/libviennaray_amd_prev.so, for same-box A/B
# runs against the working tree (boxes differ by +-3 %): VR_LIB_PATH=.../libviennaray_amd_prev.so python3 bench.py ...
# usage (in the dev container): bash tools/build_prev.sh [commit]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/vr_prev && git -C $ROOT worktree add -q /tmp/vr_prev ${1:-HEAD}
make -C /tmp/vr_prev/viennaray_amd/csrc OUT=$ROOT/viennaray_amd/libviennaray_amd_prev.so BUILD=/tmp/vr_prev/build RCCL_OUT=/tmp/vr_prev/rccl.so $ROOT/viennaray_amd/libviennaray_amd_prev.so 2>&1 | grep -E "error|Error" || true
git -C $ROOT worktree remove --force /tmp/vr_prev
ls -la $ROOT/viennaray_amd/libviennaray_amd_prev.so

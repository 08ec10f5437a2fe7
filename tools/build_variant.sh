# builds a variant of the library next to the production one: tools/build_variant.sh <tag> "<-D flags>"
# -> viennaray_amd/libviennaray_amd_<tag>.so (select it with VR_LIB_PATH); objects in csrc/build_<tag>
tag=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
make -C $here/viennaray_amd/csrc -j8 BUILD=$here/viennaray_amd/csrc/build_$tag OUT=$here/viennaray_amd/libviennaray_amd_$tag.so EXTRA="$*" $here/viennaray_amd/libviennaray_amd_$tag.so 2>&1 | grep -E "error|warning: failed" 
ls -la $here/viennaray_amd/libviennaray_amd_$tag.so

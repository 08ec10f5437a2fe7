// Host functions of the drop-in façade evaluated on hand-made inputs, dumped as raw floats for
// tests/test_facade_units.py to compare bit for bit with the oracle's restatement:
//   facade_samples <in.bin> <out.bin>
// in.bin : int32 n, int32 ncones, float cones[ncones], float rayDir[3n], float normal[3n]
// out.bin: ReflectionConedCosine<float,3> (3n floats), <float,2> (3n), then SourceGrid<float,3> and <float,2>
//          directions for cosine power 1 and 5 (4 x 3n floats); engine of sample i: RNG(seed0 + i)
#include <rayReflection.hpp>
#include <raySourceGrid.hpp>

#include <cstdio>
#include <vector>

using namespace viennaray;

int main(int argc, char **argv) {
  if (argc < 3)
    return 2;
  FILE *f = std::fopen(argv[1], "rb");
  int n = 0, nc = 0;
  if (!f || std::fread(&n, 4, 1, f) != 1 || std::fread(&nc, 4, 1, f) != 1)
    return 3;
  std::vector<float> cones(nc), rd(3 * (size_t)n), nn(3 * (size_t)n);
  if (std::fread(cones.data(), 4, nc, f) != (size_t)nc || std::fread(rd.data(), 4, rd.size(), f) != rd.size() ||
      std::fread(nn.data(), 4, nn.size(), f) != nn.size())
    return 4;
  std::fclose(f);
  const unsigned seed0 = 424242u;
  std::vector<float> out;
  out.reserve(18 * (size_t)n);
  auto put = [&](const Vec3D<float> &v) {
    for (int k = 0; k < 3; ++k)
      out.push_back(v[k]);
  };
  for (int D = 3; D >= 2; --D)
    for (int i = 0; i < n; ++i) {
      RNG rng(seed0 + (unsigned)i);
      const Vec3D<float> r{rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]}, m{nn[3 * i], nn[3 * i + 1], nn[3 * i + 2]};
      put(D == 3 ? ReflectionConedCosine<float, 3>(r, m, rng, cones[i % nc])
                 : ReflectionConedCosine<float, 2>(r, m, rng, cones[i % nc]));
    }
  std::array<Vec3D<float>, 2> box{Vec3D<float>{-1.f, -2.f, 0.f}, Vec3D<float>{3.f, 4.f, 5.f}};
  std::vector<Vec3D<float>> grid{Vec3D<float>{0.f, 0.f, 0.f}};
  for (float power : {1.f, 5.f}) {
    const std::array<int, 5> ts3{2, 0, 1, 1, -1}, ts2{1, 0, 2, 1, -1}; // POS_Z (3-D), POS_Y (2-D)
    SourceGrid<float, 3> s3(box, grid, power, ts3);
    SourceGrid<float, 2> s2(box, grid, power, ts2);
    for (int i = 0; i < n; ++i) {
      RNG rng(seed0 + (unsigned)i);
      put(s3.getOriginAndDirection(i, rng)[1]);
    }
    for (int i = 0; i < n; ++i) {
      RNG rng(seed0 + (unsigned)i);
      put(s2.getOriginAndDirection(i, rng)[1]);
    }
    if (s3.getSourceArea() != 24.f || s2.getSourceArea() != 4.f || s3.getNumPoints() != 1)
      return 5;
  }
  f = std::fopen(argv[2], "wb");
  if (!f || std::fwrite(out.data(), 4, out.size(), f) != out.size())
    return 6;
  std::fclose(f);
  std::puts("facade samples ok");
  return 0;
}

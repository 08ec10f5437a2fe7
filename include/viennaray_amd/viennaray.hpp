// viennaray.hpp — header-only C++ façade with the reference's class, method and
// enum names for the accelerated path, implemented on the C ABI of
// include/viennaray_amd.h (link with libviennaray_amd.so).
//
// It mirrors, for the path Trace<T,D>::apply() covers:
//   viennaray::Trace / TraceDisk / TraceTriangle   (rayTrace.hpp:15-180,
//       rayTraceDisk.hpp:13-224, rayTraceTriangle.hpp:13-154)
//   viennaray::DiffuseParticle / SpecularParticle  (rayParticle.hpp:126-204)
//   viennaray::TracingData                         (rayTracingData.hpp:16-219)
//   viennaray::TraceInfo, BoundaryCondition, TraceDirection, NormalizationType
//   viennaray::DiskMesh / TriangleMesh             (rayMesh.hpp:82-131)
//   rayInternal::readGridFromFile / readMeshFromFile / createPlaneGrid / writeVTK / writeVTP
//   LineMesh + convertLinesToTriangles (rayMesh.hpp)
// so a reference example builds by pointing its include path here (see
// examples/ and INTEGRATION.md).  The device computes in float; NumericType
// double is accepted and converted at the boundary, as the reference does when
// it fills Embree's float buffers (rayGeometryDisk.hpp:137-175).
//
// Particles and sources.  AbstractParticle / Particle<Derived> / Source carry the reference's
// virtual interfaces (rayParticle.hpp:21-122, raySource.hpp:10-19), so user classes compile
// unchanged.  What runs on the device is decided by AbstractParticle::deviceModel(): the
// built-in particles (and the plug-ins of viennaray_amd/csrc/vr_particles.hpp) describe
// themselves as a vr_particle POD; a user particle whose per-hit logic exists only as host
// virtuals cannot be called from a HIP kernel — apply() reports that instead of tracing
// something else, and the way in is a model in the device registry plus a deviceModel()
// override.  Sources: SourceGrid runs natively in the generator kernel; any other Source is
// evaluated on the host once per ray (getOriginAndDirection with a counting stand-in of the
// per-ray engine) and the rays are handed to the device, which continues each ray's random
// stream where the callback left it.
#pragma once

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <fstream>
#include <functional>
#include <iostream>
#include <limits>
#include <memory>
#include <random>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

// (the reference's headers pull in OpenMP: programs written against them call omp_set_num_threads() without
//  including <omp.h> themselves — tests/traceInterface/traceInterface.cpp:27, tests/rngSeed/rngSeed.cpp:10)
#if defined(_OPENMP) && __has_include(<omp.h>)
#include <omp.h>
#endif

#include "../viennaray_amd.h"

namespace viennacore {
template <class T, size_t D> using VectorType = std::array<T, D>;
template <class T> using Vec2D = std::array<T, 2>;
template <class T> using Vec3D = std::array<T, 3>;
using Vec3Df = Vec3D<float>;
using Vec3Dd = Vec3D<double>;

// The handful of ViennaCore vector helpers (vcVectorType.hpp) that ViennaRay programs use on
// their points and normals, so that such a program builds without ViennaCore.  Plain
// component loops; known answers: the reference's tests/utilFuncs (tests/aux/facade_units.cpp).
template <class T, size_t D> VectorType<T, D> operator+(const VectorType<T, D> &a, const VectorType<T, D> &b) {
  VectorType<T, D> r;
  for (size_t i = 0; i < D; ++i)
    r[i] = a[i] + b[i];
  return r;
}
template <class T, size_t D> VectorType<T, D> operator-(const VectorType<T, D> &a, const VectorType<T, D> &b) {
  VectorType<T, D> r;
  for (size_t i = 0; i < D; ++i)
    r[i] = a[i] - b[i];
  return r;
}
template <class T, size_t D> VectorType<T, D> operator*(const T &f, const VectorType<T, D> &a) {
  VectorType<T, D> r;
  for (size_t i = 0; i < D; ++i)
    r[i] = f * a[i];
  return r;
}
template <class T, size_t D> VectorType<T, D> operator*(const VectorType<T, D> &a, const T &f) { return f * a; }
template <class T, size_t D>
VectorType<T, D> Sum(const VectorType<T, D> &a, const VectorType<T, D> &b, const VectorType<T, D> &c) {
  return a + b + c;
}
template <class T, size_t D> T DotProduct(const VectorType<T, D> &a, const VectorType<T, D> &b) {
  T s = 0;
  for (size_t i = 0; i < D; ++i)
    s += a[i] * b[i];
  return s;
}
template <class T> Vec3D<T> CrossProduct(const Vec3D<T> &a, const Vec3D<T> &b) {
  return Vec3D<T>{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}
template <class T, size_t D> T Norm2(const VectorType<T, D> &a) { return DotProduct(a, a); }
template <class T, size_t D> T Norm(const VectorType<T, D> &a) { return std::sqrt(Norm2(a)); }
template <class T, size_t D> void Normalize(VectorType<T, D> &a) {
  const T n = Norm(a);
  if (n > T(0))
    for (size_t i = 0; i < D; ++i)
      a[i] /= n;
}
// (ViennaCore has both forms: in place for an lvalue, a normalised copy for a const / temporary argument —
//  tests/smoothing/smoothing.cpp:26 uses the second)
template <class T, size_t D> VectorType<T, D> Normalize(const VectorType<T, D> &a) {
  VectorType<T, D> r = a;
  Normalize(r);
  return r;
}
template <class T, size_t D> bool IsNormalized(const VectorType<T, D> &a) {
  return std::fabs(Norm(a) - T(1)) < T(1e-4);
}
template <class T, size_t D> T Distance(const VectorType<T, D> &a, const VectorType<T, D> &b) { return Norm(a - b); }
template <class T, size_t D> VectorType<T, D> Inv(const VectorType<T, D> &a) { return T(-1) * a; }
template <class T, size_t D>
VectorType<T, D> ScaleAdd(const VectorType<T, D> &a, const VectorType<T, D> &b, const T &f) { // a * f + b
  return f * a + b;
}
template <class T> Vec3D<T> ComputeNormal(const Vec3D<Vec3D<T>> &p) { return CrossProduct(p[1] - p[0], p[2] - p[0]); }

// vcLogger.hpp: the part ViennaRay programs touch (examples/triangle3D/triangle3D.cpp:14:
// Logger::setLogLevel(LogLevel::DEBUG)).  Messages at or below the current level go to std::cout / std::cerr;
// nothing here aborts.
enum class LogLevel : unsigned { ERROR = 0, WARNING = 1, INFO = 2, INTERMEDIATE = 3, TIMING = 4, DEBUG = 5 };
class Logger {
  std::string pending_;
  bool toError_ = false;
  static LogLevel &level() {
    static LogLevel l = LogLevel::INFO;
    return l;
  }
  Logger() = default;
  Logger &add(LogLevel l, const char *tag, const std::string &msg, bool err) {
    if (static_cast<unsigned>(level()) >= static_cast<unsigned>(l)) {
      pending_ += std::string("\n    ") + tag + msg + "\n";
      toError_ = toError_ || err;
    }
    return *this;
  }

public:
  Logger(const Logger &) = delete;
  Logger &operator=(const Logger &) = delete;
  static Logger &getInstance() {
    static Logger instance;
    return instance;
  }
  static void setLogLevel(LogLevel l) { level() = l; }
  static LogLevel getLogLevel() { return level(); }
  Logger &addDebug(const std::string &s) { return add(LogLevel::DEBUG, "DEBUG: ", s, false); }
  Logger &addTiming(const std::string &s, double seconds) {
    return add(LogLevel::TIMING, "", s + ": " + std::to_string(seconds) + " s", false);
  }
  Logger &addInfo(const std::string &s) { return add(LogLevel::INFO, "", s, false); }
  Logger &addWarning(const std::string &s) { return add(LogLevel::WARNING, "WARNING: ", s, false); }
  Logger &addError(const std::string &s, bool = true) { return add(LogLevel::ERROR, "ERROR: ", s, true); }
  void print(std::ostream &out = std::cout) {
    (toError_ ? std::cerr : out) << pending_;
    pending_.clear();
    toError_ = false;
  }
};

// vcRNG.hpp: viennacore::RNG is std::mt19937_64.  Here a thin wrapper with the same
// UniformRandomBitGenerator interface that also counts the outputs drawn: a host-side Source
// callback consumes part of a ray's random stream and the device continues after it.
struct RNG {
  using result_type = std::mt19937_64::result_type;
  std::mt19937_64 engine;
  uint64_t draws = 0;
  RNG() = default;
  explicit RNG(result_type seed) : engine(seed) {}
  void seed(result_type s) {
    engine.seed(s);
    draws = 0;
  }
  static constexpr result_type min() { return std::mt19937_64::min(); }
  static constexpr result_type max() { return std::mt19937_64::max(); }
  result_type operator()() {
    ++draws;
    return engine();
  }
};

// vcRNG.hpp tea<N>: the per-ray seed hash (rayTraceKernel.hpp:120)
template <unsigned N> inline unsigned tea(unsigned v0, unsigned v1) {
  unsigned s0 = 0;
  for (unsigned n = 0; n < N; ++n) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  return v0;
}

struct Timer {
  std::chrono::steady_clock::time_point t0;
  long long currentDuration = 0; // ns
  void start() { t0 = std::chrono::steady_clock::now(); }
  void finish() {
    currentDuration = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  }
};
} // namespace viennacore

namespace viennaray {
using namespace viennacore;

enum class BoundaryCondition : unsigned { REFLECTIVE_BOUNDARY = 0, PERIODIC_BOUNDARY = 1, IGNORE_BOUNDARY = 2 };
enum class TraceDirection : unsigned { POS_X = 0, NEG_X = 1, POS_Y = 2, NEG_Y = 3, POS_Z = 4, NEG_Z = 5 };
enum class NormalizationType : unsigned { SOURCE = 0, MAX = 1 };
enum class TracingDataMergeEnum : unsigned { SUM = 0, APPEND = 1, AVERAGE = 2 };

struct TraceInfo {
  size_t numRays = 0;
  size_t totalRaysTraced = 0;
  size_t nonGeometryHits = 0;
  size_t geometryHits = 0;
  size_t particleHits = 0;
  size_t boundaryHits = 0;
  size_t reflections = 0;
  double time = 0.0;
  bool warning = false;
  bool error = false;
};

// ---- meshes (rayMesh.hpp) -------------------------------------------------------
struct DiskMesh {
  DiskMesh() = default;
  DiskMesh(const std::vector<Vec3Df> &pts, const std::vector<Vec3Df> &nms, float delta)
      : nodes(pts), normals(nms), gridDelta(delta) {}
  std::vector<Vec3Df> nodes;
  std::vector<Vec3Df> normals;
  float gridDelta = 0.f;
};

struct TriangleMesh {
  TriangleMesh() = default;
  TriangleMesh(std::vector<Vec3Df> const &pts, std::vector<Vec3D<unsigned>> const &tris, float delta)
      : nodes(pts), triangles(tris), gridDelta(delta) {}
  std::vector<Vec3Df> nodes;
  std::vector<Vec3D<unsigned>> triangles;
  float gridDelta = 0.f;
};

// rayMesh.hpp:27-80: 2-D surface as line segments; zero-length segments are dropped
struct LineMesh {
  LineMesh() = default;
  LineMesh(const std::vector<Vec3Df> &pts, const std::vector<Vec2D<unsigned>> &lns, float delta)
      : nodes(pts), gridDelta(delta) {
    for (auto const &l : lns) {
      const auto &p0 = nodes[l[0]];
      const auto &p1 = nodes[l[1]];
      const float dx = p1[0] - p0[0], dy = p1[1] - p0[1], dz = p1[2] - p0[2];
      if (std::sqrt(dx * dx + dy * dy + dz * dz) > 1e-6f)
        lines.push_back(l);
    }
  }
  std::vector<Vec3Df> nodes;
  std::vector<Vec2D<unsigned>> lines;
  float gridDelta = 0.f;
};

// rayMesh.hpp:133-175: every line becomes a strip of two triangles of height gridDelta
// (z = +-gridDelta/2); node 2i is node i at +z, node 2i+1 at -z
inline TriangleMesh convertLinesToTriangles(const LineMesh &lineMesh) {
  TriangleMesh mesh;
  mesh.gridDelta = lineMesh.gridDelta;
  const float w2 = lineMesh.gridDelta * 0.5f;
  mesh.nodes.reserve(lineMesh.nodes.size() * 2);
  for (auto const &p : lineMesh.nodes) {
    mesh.nodes.push_back(Vec3Df{p[0], p[1], w2});
    mesh.nodes.push_back(Vec3Df{p[0], p[1], -w2});
  }
  mesh.triangles.reserve(lineMesh.lines.size() * 2);
  for (auto const &line : lineMesh.lines) {
    const unsigned p0 = line[0] * 2, p1 = line[1] * 2;
    mesh.triangles.push_back(Vec3D<unsigned>{p0, p1, p0 + 1});
    mesh.triangles.push_back(Vec3D<unsigned>{p0 + 1, p1, p1 + 1});
  }
  return mesh;
}

// ---- TracingData (rayTracingData.hpp:16-219) ---------------------------------------------
// Labelled per-primitive vectors and labelled scalars with a merge type each.  The device
// path fills vector 0 of Trace::getLocalData() (merge type SUM); everything else is a plain
// container for the caller, with the reference's method names and defaults.
template <typename NumericType> class TracingData {
  std::vector<NumericType> scalars_;
  std::vector<std::vector<NumericType>> vectors_;
  std::vector<std::string> scalarLabels_, vectorLabels_;
  std::vector<TracingDataMergeEnum> scalarMerge_, vectorMerge_;

  static int find(const std::vector<std::string> &labels, const std::string &label, const char *what) {
    for (int i = 0; i < (int)labels.size(); ++i)
      if (labels[i] == label)
        return i;
    std::cerr << "Can not find " << what << " data label in TracingData.\n";
    return -1;
  }

public:
  TracingData() = default;
  TracingData(const TracingData &) = default;
  TracingData &operator=(const TracingData &) = default;
  // a moved-from object is left EMPTY (the reference's test checks data() == nullptr)
  TracingData(TracingData &&o) noexcept { *this = std::move(o); }
  TracingData &operator=(TracingData &&o) noexcept {
    scalars_ = std::exchange(o.scalars_, {});
    vectors_ = std::exchange(o.vectors_, {});
    scalarLabels_ = std::exchange(o.scalarLabels_, {});
    vectorLabels_ = std::exchange(o.vectorLabels_, {});
    scalarMerge_ = std::exchange(o.scalarMerge_, {});
    vectorMerge_ = std::exchange(o.vectorMerge_, {});
    return *this;
  }

  void setNumberOfVectorData(int size) {
    vectors_.clear();
    vectors_.resize(size);
    vectorMerge_.resize(size, TracingDataMergeEnum::SUM);
    vectorLabels_.resize(size, "vectorData");
  }
  void setNumberOfScalarData(int size) {
    scalars_.clear();
    scalars_.resize(size);
    scalarMerge_.resize(size, TracingDataMergeEnum::SUM);
    scalarLabels_.resize(size, "scalarData");
  }
  void setScalarData(int num, NumericType value, std::string label = "scalarData") {
    scalars_[num] = value;
    scalarLabels_[num] = std::move(label);
  }
  void setVectorData(int num, std::vector<NumericType> &v, std::string label = "vectorData") {
    vectors_[num] = v;
    vectorLabels_[num] = std::move(label);
  }
  void setVectorData(int num, std::vector<NumericType> &&v, std::string label = "vectorData") {
    vectors_[num] = std::move(v);
    vectorLabels_[num] = std::move(label);
  }
  void setVectorData(int num, size_t size, NumericType value, std::string label = "vectorData") {
    vectors_[num].assign(size, value);
    vectorLabels_[num] = std::move(label);
  }
  void setVectorData(int num, NumericType value, std::string label = "vectorData") {
    vectors_[num].assign(vectors_[num].size(), value);
    vectorLabels_[num] = std::move(label);
  }
  void appendVectorData(int num, const std::vector<NumericType> &v) {
    vectors_[num].insert(vectors_[num].end(), v.begin(), v.end());
  }
  void resizeAllVectorData(size_t size, NumericType val = 0) {
    for (auto &v : vectors_) {
      v.clear();
      v.resize(size, val);
    }
  }
  void setVectorMergeType(const std::vector<TracingDataMergeEnum> &m) { vectorMerge_ = m; }
  void setVectorMergeType(int num, TracingDataMergeEnum m) { vectorMerge_[num] = m; }
  void setScalarMergeType(const std::vector<TracingDataMergeEnum> &m) { scalarMerge_ = m; }
  void setScalarMergeType(int num, TracingDataMergeEnum m) { scalarMerge_[num] = m; }

  [[nodiscard]] std::vector<NumericType> &getVectorData(int i) { return vectors_[i]; }
  [[nodiscard]] const std::vector<NumericType> &getVectorData(int i) const { return vectors_[i]; }
  [[nodiscard]] std::vector<NumericType> &getVectorData(const std::string &label) {
    return vectors_[getVectorDataIndex(label)];
  }
  [[nodiscard]] std::vector<std::vector<NumericType>> &getVectorData() { return vectors_; }
  [[nodiscard]] const std::vector<std::vector<NumericType>> &getVectorData() const { return vectors_; }
  [[nodiscard]] NumericType &getScalarData(int i) { return scalars_[i]; }
  [[nodiscard]] const NumericType &getScalarData(int i) const { return scalars_[i]; }
  [[nodiscard]] NumericType &getScalarData(const std::string &label) { return scalars_[getScalarDataIndex(label)]; }
  [[nodiscard]] std::vector<NumericType> &getScalarData() { return scalars_; }
  [[nodiscard]] const std::vector<NumericType> &getScalarData() const { return scalars_; }
  [[nodiscard]] std::string getVectorDataLabel(int i) const {
    return i < (int)vectorLabels_.size() ? vectorLabels_[i] : std::string();
  }
  [[nodiscard]] std::string getScalarDataLabel(int i) const {
    return i < (int)scalarLabels_.size() ? scalarLabels_[i] : std::string();
  }
  [[nodiscard]] int getVectorDataIndex(const std::string &label) const { return find(vectorLabels_, label, "vector"); }
  [[nodiscard]] int getScalarDataIndex(const std::string &label) const { return find(scalarLabels_, label, "scalar"); }
  [[nodiscard]] TracingDataMergeEnum getVectorMergeType(int num) const { return vectorMerge_[num]; }
  [[nodiscard]] TracingDataMergeEnum getScalarMergeType(int num) const { return scalarMerge_[num]; }
};

// ---- particles --------------------------------------------------------------------
// rayUtil.hpp:49-63.  The built-in particles log nothing; the type is here so programs that
// pass a DataLog around keep compiling.
template <class NumericType> struct DataLog {
  std::vector<std::vector<NumericType>> data;
  // element-wise sum of two logs (the shorter extent of each row counts)
  void merge(DataLog<NumericType> &pOther) {
    const std::size_t rows = std::min(data.size(), pOther.data.size());
    for (std::size_t r = 0; r < rows; ++r) {
      const std::size_t cols = std::min(data[r].size(), pOther.data[r].size());
      std::transform(data[r].begin(), data[r].begin() + cols, pOther.data[r].begin(), data[r].begin(), std::plus<NumericType>());
    }
  }
};

} // namespace viennaray

namespace rayInternal {
using namespace viennacore;
// rayUtil.hpp:266-283: Marsaglia's rejection method — a point of the unit disc (pairs from U(-1,1) until inside)
// lifted onto the sphere.  The reference's arithmetic: the squared radius and the lift factor in double, the
// coordinates narrowed back to NumericType.
template <typename NumericType> viennacore::Vec3D<NumericType> pickRandomPointOnUnitSphere(viennacore::RNG &rngState) {
  static thread_local std::uniform_real_distribution<NumericType> symmetric(NumericType(-1), NumericType(1));
  NumericType u = 0, v = 0;
  double radiusSq = 2.;
  while (radiusSq >= 1.) {
    u = symmetric(rngState);
    v = symmetric(rngState);
    radiusSq = u * u + v * v;
  }
  const double lift = 2. * std::sqrt(1. - radiusSq);
  u *= lift;
  v *= lift;
  const NumericType w = 1. - 2 * radiusSq;
  return viennacore::Vec3D<NumericType>{u, v, w};
}
// rayUtil.hpp:145-202: {rayDir, firstDir, secondDir, minMax, posNeg}
inline std::array<int, 5> getTraceSettings(unsigned sourceDir) {
  switch (sourceDir) {
  case 0: return {0, 1, 2, 1, -1}; // POS_X
  case 1: return {0, 1, 2, 0, 1};  // NEG_X
  case 2: return {1, 0, 2, 1, -1}; // POS_Y
  case 3: return {1, 0, 2, 0, 1};  // NEG_Y
  case 4: return {2, 0, 1, 1, -1}; // POS_Z
  default: return {2, 0, 1, 0, 1}; // NEG_Z
  }
}
template <int D> constexpr double DiskFactor = 0.5 * (D == 3 ? 1.7320508 : 1.41421356237) * (1 + 1e-5); // rayUtil.hpp:99-101

// rayUtil.hpp:104-143 (direction: viennaray::TraceDirection, passed as its underlying value 0..5)
template <typename NumericType, int D, class Direction>
void adjustBoundingBox(std::array<viennacore::Vec3D<NumericType>, 2> &bdBox, Direction const direction,
                       NumericType discRadius) {
  if constexpr (D == 2) {
    bdBox[0][2] -= discRadius;
    bdBox[1][2] += discRadius;
  }
  switch ((unsigned)direction) {
  case 0: bdBox[1][0] += 2 * discRadius; break;
  case 1: bdBox[0][0] -= 2 * discRadius; break;
  case 2: bdBox[1][1] += 2 * discRadius; break;
  case 3: bdBox[0][1] -= 2 * discRadius; break;
  case 4: bdBox[1][2] += 2 * discRadius; break;
  default: bdBox[0][2] -= 2 * discRadius; break;
  }
}
template <class Direction> std::array<int, 5> getTraceSettings(Direction d) { return getTraceSettings((unsigned)d); }

// rayUtil.hpp:564-611
// rayUtil.hpp:564-611: a regular lattice of ray origins on the source face of the bounding box, about `pNumPoints`
// of them, in the proportion of the face's sides; margins of 1e-4 keep the origins off the walls.  The lattice is
// walked by repeated addition like the reference's (the float sums ARE the coordinates: tests/createSourceGrid).
template <typename NumericType, int D>
[[nodiscard]] std::vector<viennacore::Vec3D<NumericType>>
createSourceGrid(const std::array<viennacore::Vec3D<NumericType>, 2> &pBdBox, const size_t pNumPoints,
                 const NumericType pGridDelta, const std::array<int, 5> &pTraceSettings) {
  const int axisRay = pTraceSettings[0], axisA = pTraceSettings[1], axisB = pTraceSettings[2];
  const int sourceSide = pTraceSettings[3];
  constexpr double margin = 1e-4;
  const auto sideA = pBdBox[1][axisA] - pBdBox[0][axisA];
  const auto sideB = pBdBox[1][axisB] - pBdBox[0][axisB];
  // lattice counts: first from the grid spacing (only their integer ratio survives), then from the requested total
  size_t countA = static_cast<size_t>(round(sideA / pGridDelta));
  size_t countB = static_cast<size_t>(round(sideB / pGridDelta));
  const unsigned long aspect = countA / countB;
  countA = static_cast<size_t>(std::sqrt(pNumPoints * aspect));
  countB = static_cast<size_t>(std::sqrt(pNumPoints / aspect));
  const auto stepA = (sideA - 2 * margin) / static_cast<NumericType>(countA - 1);
  const auto stepB = (sideB - 2 * margin) / static_cast<NumericType>(countB - 1);

  std::vector<viennacore::Vec3D<NumericType>> lattice;
  lattice.reserve(pNumPoints);
  viennacore::Vec3D<NumericType> origin;
  origin[axisRay] = pBdBox[sourceSide][axisRay];
  auto b = pBdBox[0][axisB] + margin;
  while (b <= pBdBox[1][axisB] - margin) {
    origin[axisB] = D == 2 ? decltype(b)(0) : b;
    auto a = pBdBox[0][axisA] + margin;
    while (a <= pBdBox[1][axisA] - margin) {
      origin[axisA] = a;
      lattice.push_back(origin);
      a += stepA;
    }
    b += stepB;
  }
  lattice.shrink_to_fit();
  return lattice;
}
} // namespace rayInternal

namespace viennaray {
using namespace viennacore;

// rayReflection.hpp:13-120 — the host-side reflection functions user particles call
template <typename NumericType, int D = 3>
[[nodiscard]] Vec3D<NumericType> ReflectionSpecular(const Vec3D<NumericType> &rayDir, const Vec3D<NumericType> &geomNormal) {
  auto dirOldInv = Inv(rayDir);
  return NumericType(2 * DotProduct(geomNormal, dirOldInv)) * geomNormal - dirOldInv;
}
// rayReflection.hpp:31-50: cosine-distributed about the normal = a uniform point of the unit sphere pushed along the
// normal and renormalised (2-D: the z component is dropped before the renormalisation)
template <typename NumericType, int D>
[[nodiscard]] Vec3D<NumericType> ReflectionDiffuse(const Vec3D<NumericType> &geomNormal, RNG &rngState) {
  Vec3D<NumericType> lobe = rayInternal::pickRandomPointOnUnitSphere<NumericType>(rngState);
  for (int axis = 0; axis < 2; ++axis)
    lobe[axis] += geomNormal[axis];
  lobe[2] = D == 3 ? NumericType(lobe[2] + geomNormal[2]) : NumericType(0);
  Normalize(lobe);
  return lobe;
}
// ReflectionConedCosine (rayReflection.hpp:52-120) assembled from two pieces of this library's own: the frame
// around the mirror direction and the accept-reject draw of the lobe's polar angle.  The arithmetic (types,
// operation order, engine outputs consumed) is the reference's — tests/aux/facade_units.cpp compares 10^5
// samples bit for bit with the oracle's restatement — the code is not.
namespace detail {
// two unit vectors completing `axis` (unit length) to a right-handed frame; the closed form has its pole at -z
template <typename NumericType> struct LobeFrame {
  Vec3D<NumericType> u, v;
  explicit LobeFrame(const Vec3D<NumericType> &axis) {
    const NumericType one(1);
    if (axis[2] < NumericType(-0.999999)) {
      u = Vec3D<NumericType>{NumericType(0), -one, NumericType(0)};
      v = Vec3D<NumericType>{-one, NumericType(0), NumericType(0)};
      return;
    }
    const NumericType k = one / (one + axis[2]);
    const NumericType mixed = -axis[0] * axis[1] * k;
    u = Vec3D<NumericType>{one - axis[0] * axis[0] * k, mixed, -axis[0]};
    v = Vec3D<NumericType>{mixed, one - axis[1] * axis[1] * k, -axis[1]};
  }
  // the direction with polar angle (sin, cos) = (sp, cp) and azimuth (sin, cos) = (sa, ca) about `axis`
  Vec3D<NumericType> at(const Vec3D<NumericType> &axis, NumericType sp, NumericType cp, NumericType sa, NumericType ca) const {
    Vec3D<NumericType> r;
    for (int k = 0; k < 3; ++k)
      r[k] = sp * (ca * u[k] + sa * v[k]) + cp * axis[k];
    return r;
  }
};
// polar angle of the cosine lobe squeezed into [0, cone]: proposals theta = cone * sqrt(1 - sqrt(U1)) are accepted
// when U2 * theta * sqrt(U1) <= cos(pi/2 * theta / cone) * sin(theta); two engine outputs per proposal
inline double lobePolarAngle(RNG &rng, double cone) {
  std::uniform_real_distribution<double> unit(0.0, 1.0);
  while (true) {
    const double rootU = std::sqrt(unit(rng));
    const double frac = std::sqrt(std::max(1.0 - rootU, 0.0));
    const double theta = cone * frac;
    const double bound = std::cos(M_PI_2 * frac) * std::sin(theta);
    if (unit(rng) * theta * rootU <= bound)
      return theta;
  }
}
} // namespace detail

template <typename NumericType, int D>
[[nodiscard]] Vec3D<NumericType> ReflectionConedCosine(const Vec3D<NumericType> &rayDir, const Vec3D<NumericType> &geomNormal,
                                                       RNG &rng, const NumericType maxConeAngle) {
  // the two ends of the family: a mirror, and the plain cosine lobe about the normal
  if (!(maxConeAngle > NumericType(0)))
    return ReflectionSpecular<NumericType>(rayDir, geomNormal);
  if (maxConeAngle >= M_PI_2)
    return ReflectionDiffuse<NumericType, D>(geomNormal, rng);
  Vec3D<NumericType> mirror = ReflectionSpecular<NumericType>(rayDir, geomNormal);
  Normalize(mirror);
  const detail::LobeFrame<NumericType> frame(mirror);
  const double theta = detail::lobePolarAngle(rng, maxConeAngle);
  const NumericType sinTheta = std::sin(theta), cosTheta = std::cos(theta);
  std::uniform_real_distribution<double> unit(0.0, 1.0);
  const double azimuth = 2.0 * M_PI * unit(rng);
  const NumericType sinAz = std::sin(azimuth), cosAz = std::cos(azimuth);
  Vec3D<NumericType> out = frame.at(mirror, sinTheta, cosTheta, sinAz, cosAz);
  // a sample below the surface is mirrored back above it
  if (const NumericType below = DotProduct(out, geomNormal); below <= NumericType(0))
    out = out - NumericType(NumericType(2) * below) * geomNormal;
  if constexpr (D == 2)
    out[2] = NumericType(0);
  Normalize(out);
  return out;
}

// raySource.hpp:10-19
template <typename NumericType> class Source {
public:
  virtual ~Source() = default;
  virtual std::array<Vec3D<NumericType>, 2> getOriginAndDirection(size_t idx, RNG &rngState) const = 0;
  [[nodiscard]] virtual size_t getNumPoints() const = 0;
  virtual NumericType getSourceArea() const = 0;
  virtual NumericType getInitialRayWeight(const size_t) const { return 1.; }
};

// SourceGrid (raySourceGrid.hpp:9-74): rays start at explicit grid points (origin = grid[idx mod n]) with a
// power-cosine direction about the tracing axis.  The tracer recognises the type and runs it in the generator
// kernel (vr_set_source_grid); this host version exists for user code that calls it directly and follows the
// reference's arithmetic (float cosf / sinf / sqrtf, two engine outputs per ray).
template <typename NumericType, int D> class SourceGrid : public Source<NumericType> {
  using boundingBoxType = std::array<Vec3D<NumericType>, 2>;

public:
  SourceGrid(const boundingBoxType &boundingBox, std::vector<Vec3D<NumericType>> &sourceGrid, NumericType cosinePower,
             const std::array<int, 5> &traceSettings)
      : box_(boundingBox), points_(sourceGrid), axis_{traceSettings[0], traceSettings[1], traceSettings[2]},
        sign_(static_cast<NumericType>(traceSettings[4])), exponent_(NumericType(2) / (cosinePower + NumericType(1))) {}

  std::array<Vec3D<NumericType>, 2> getOriginAndDirection(const size_t idx, RNG &rngState) const override {
    std::uniform_real_distribution<NumericType> unit;
    const NumericType azimuthDraw = unit(rngState);
    const NumericType polarDraw = unit(rngState);
    const NumericType cosSq = pow(polarDraw, exponent_); // cos^2 of the polar angle
    const float along = sqrtf(cosSq), across = sqrtf(1.f - cosSq);
    Vec3D<NumericType> d{NumericType(0), NumericType(0), NumericType(0)};
    d[axis_[0]] = sign_ * along;
    d[axis_[1]] = cosf(M_PI * 2.f * azimuthDraw) * across;
    if constexpr (D == 3)
      d[axis_[2]] = sinf(M_PI * 2.f * azimuthDraw) * across;
    Normalize(d);
    return {points_[idx % points_.size()], d};
  }
  [[nodiscard]] size_t getNumPoints() const override { return points_.size(); }
  // the face of the bounding box the rays start from
  NumericType getSourceArea() const override {
    NumericType area = box_[1][axis_[1]] - box_[0][axis_[1]];
    if constexpr (D == 3)
      area *= box_[1][axis_[2]] - box_[0][axis_[2]];
    return area;
  }
  [[nodiscard]] const std::vector<Vec3D<NumericType>> &getGrid() const { return points_; }

private:
  const boundingBoxType box_;
  const std::vector<Vec3D<NumericType>> &points_;
  const std::array<int, 3> axis_; // tracing axis, first and second transverse axis
  const NumericType sign_;        // rays travel towards -/+ the tracing axis
  const NumericType exponent_;
};

#define VIENNARAY_PARTICLE_STOP                                                                                         \
  std::pair<NumericType, Vec3D<NumericType>> { NumericType(1), Vec3D<NumericType>{} }

// rayParticle.hpp:21-81
template <typename NumericType> class AbstractParticle {
public:
  virtual ~AbstractParticle() = default;
  virtual std::unique_ptr<AbstractParticle> clone() const = 0;
  virtual void initNew(RNG &rngState) = 0;
  virtual Vec3D<NumericType> initNewWithDirection(RNG &rngState) = 0;
  virtual std::pair<NumericType, Vec3D<NumericType>>
  surfaceReflection(NumericType rayWeight, const Vec3D<NumericType> &rayDir, const Vec3D<NumericType> &geomNormal,
                    const unsigned int primId, const int materialId, const TracingData<NumericType> *globalData,
                    RNG &rngState) = 0;
  virtual void surfaceCollision(NumericType rayWeight, const Vec3D<NumericType> &rayDir,
                                const Vec3D<NumericType> &geomNormal, const unsigned int primID, const int materialId,
                                TracingData<NumericType> &localData, const TracingData<NumericType> *globalData,
                                RNG &rngState) = 0;
  virtual NumericType getSourceDistributionPower() const = 0;
  virtual NumericType getMeanFreePath() const = 0;
  [[nodiscard]] virtual std::vector<std::string> getLocalDataLabels() const = 0;
  virtual void logData(DataLog<NumericType> &log) = 0;

  /// NOT in the reference: how this particle runs on the device.  Fill `pod` with a model of the
  /// device registry (viennaray_amd/csrc/vr_particles.hpp) and return true.  The default (false)
  /// means the particle exists only as host virtuals, which a HIP kernel cannot call.
  virtual bool deviceModel(vr_particle &pod) const {
    (void)pod;
    return false;
  }
};

// rayParticle.hpp:83-122: CRTP base implementing clone() and the no-op defaults
template <typename Derived, typename NumericType> class Particle : public AbstractParticle<NumericType> {
public:
  std::unique_ptr<AbstractParticle<NumericType>> clone() const final {
    return std::make_unique<Derived>(static_cast<Derived const &>(*this));
  }
  void initNew(RNG &) override {}
  Vec3D<NumericType> initNewWithDirection(RNG &) override { return Vec3D<NumericType>{0, 0, 0}; }
  std::pair<NumericType, Vec3D<NumericType>> surfaceReflection(NumericType, const Vec3D<NumericType> &,
                                                               const Vec3D<NumericType> &, const unsigned int, const int,
                                                               const TracingData<NumericType> *, RNG &) override {
    return VIENNARAY_PARTICLE_STOP;
  }
  void surfaceCollision(NumericType, const Vec3D<NumericType> &, const Vec3D<NumericType> &, const unsigned int, const int,
                        TracingData<NumericType> &, const TracingData<NumericType> *, RNG &) override {}
  NumericType getSourceDistributionPower() const override { return 1.; }
  NumericType getMeanFreePath() const override { return -1.; }
  [[nodiscard]] std::vector<std::string> getLocalDataLabels() const override { return {}; }
  void logData(DataLog<NumericType> &) override {}

protected:
  Particle() = default;
  Particle(const Particle &) = default;
  Particle(Particle &&) = default;
};

// rayParticle.hpp:126-163
template <typename NumericType, int D>
class DiffuseParticle : public Particle<DiffuseParticle<NumericType, D>, NumericType> {
  const NumericType stickingProbability_;
  const std::string dataLabel_;

public:
  DiffuseParticle(NumericType stickingProbability, std::string dataLabel)
      : stickingProbability_(stickingProbability), dataLabel_(std::move(dataLabel)) {}
  std::pair<NumericType, Vec3D<NumericType>> surfaceReflection(NumericType, const Vec3D<NumericType> &,
                                                               const Vec3D<NumericType> &geomNormal, const unsigned int,
                                                               const int, const TracingData<NumericType> *,
                                                               RNG &rngState) final {
    return {stickingProbability_, ReflectionDiffuse<NumericType, D>(geomNormal, rngState)};
  }
  void surfaceCollision(NumericType rayWeight, const Vec3D<NumericType> &, const Vec3D<NumericType> &,
                        const unsigned int primID, const int, TracingData<NumericType> &localData,
                        const TracingData<NumericType> *, RNG &) final {
    localData.getVectorData(0)[primID] += rayWeight;
  }
  NumericType getSourceDistributionPower() const final { return 1.; }
  [[nodiscard]] std::vector<std::string> getLocalDataLabels() const final { return {dataLabel_}; }
  bool deviceModel(vr_particle &pod) const final {
    pod = vr_particle{VR_PARTICLE_DIFFUSE, (float)stickingProbability_, 1.f, 0, nullptr, nullptr, 0.f, -1.f};
    return true;
  }
};

// rayParticle.hpp:165-204
template <typename NumericType, int D>
class SpecularParticle : public Particle<SpecularParticle<NumericType, D>, NumericType> {
  const NumericType stickingProbability_;
  const NumericType sourcePower_;
  const std::string dataLabel_;

public:
  SpecularParticle(NumericType stickingProbability, NumericType sourcePower, std::string dataLabel)
      : stickingProbability_(stickingProbability), sourcePower_(sourcePower), dataLabel_(std::move(dataLabel)) {}
  std::pair<NumericType, Vec3D<NumericType>> surfaceReflection(NumericType, const Vec3D<NumericType> &rayDir,
                                                               const Vec3D<NumericType> &geomNormal, const unsigned int,
                                                               const int, const TracingData<NumericType> *, RNG &) final {
    return {stickingProbability_, ReflectionSpecular<NumericType, D>(rayDir, geomNormal)};
  }
  void surfaceCollision(NumericType rayWeight, const Vec3D<NumericType> &, const Vec3D<NumericType> &,
                        const unsigned int primID, const int, TracingData<NumericType> &localData,
                        const TracingData<NumericType> *, RNG &) final {
    localData.getVectorData(0)[primID] += rayWeight;
  }
  NumericType getSourceDistributionPower() const final { return sourcePower_; }
  [[nodiscard]] std::vector<std::string> getLocalDataLabels() const final { return {dataLabel_}; }
  bool deviceModel(vr_particle &pod) const final {
    pod = vr_particle{VR_PARTICLE_SPECULAR, (float)stickingProbability_, (float)sourcePower_, 0, nullptr, nullptr, 0.f, -1.f};
    return true;
  }
};

// ---- plug-in particles of the device registry (NOT in the reference's rayParticle.hpp; written
//      against AbstractParticle the way a ViennaPS model is, with a device model each) ------------
// sticking + ReflectionConedCosine(maxConeAngle), optional mean free path; collects like SpecularParticle
template <typename NumericType, int D>
class ConedCosineParticle : public Particle<ConedCosineParticle<NumericType, D>, NumericType> {
  const NumericType stickingProbability_, sourcePower_, coneAngle_, meanFreePath_;
  const std::string dataLabel_;

public:
  ConedCosineParticle(NumericType stickingProbability, NumericType sourcePower, NumericType maxConeAngle,
                      std::string dataLabel, NumericType meanFreePath = NumericType(-1))
      : stickingProbability_(stickingProbability), sourcePower_(sourcePower), coneAngle_(maxConeAngle),
        meanFreePath_(meanFreePath), dataLabel_(std::move(dataLabel)) {}
  std::pair<NumericType, Vec3D<NumericType>> surfaceReflection(NumericType, const Vec3D<NumericType> &rayDir,
                                                               const Vec3D<NumericType> &geomNormal, const unsigned int,
                                                               const int, const TracingData<NumericType> *,
                                                               RNG &rngState) final {
    return {stickingProbability_, ReflectionConedCosine<NumericType, D>(rayDir, geomNormal, rngState, coneAngle_)};
  }
  void surfaceCollision(NumericType rayWeight, const Vec3D<NumericType> &, const Vec3D<NumericType> &,
                        const unsigned int primID, const int, TracingData<NumericType> &localData,
                        const TracingData<NumericType> *, RNG &) final {
    localData.getVectorData(0)[primID] += rayWeight;
  }
  NumericType getSourceDistributionPower() const final { return sourcePower_; }
  NumericType getMeanFreePath() const final { return meanFreePath_; }
  [[nodiscard]] std::vector<std::string> getLocalDataLabels() const final { return {dataLabel_}; }
  bool deviceModel(vr_particle &pod) const final {
    pod = vr_particle{VR_PARTICLE_CONED_COSINE, (float)stickingProbability_, (float)sourcePower_, 0, nullptr, nullptr,
                      (float)coneAngle_, (float)meanFreePath_};
    return true;
  }
};

// a DiffuseParticle with two data labels: label 0 += w, label 1 += w * max(0, -d.n)
template <typename NumericType, int D>
class DiffuseCosineParticle : public Particle<DiffuseCosineParticle<NumericType, D>, NumericType> {
  const NumericType stickingProbability_;
  const std::string dataLabel_, cosineLabel_;

public:
  DiffuseCosineParticle(NumericType stickingProbability, std::string dataLabel, std::string cosineLabel)
      : stickingProbability_(stickingProbability), dataLabel_(std::move(dataLabel)), cosineLabel_(std::move(cosineLabel)) {}
  std::pair<NumericType, Vec3D<NumericType>> surfaceReflection(NumericType, const Vec3D<NumericType> &,
                                                               const Vec3D<NumericType> &geomNormal, const unsigned int,
                                                               const int, const TracingData<NumericType> *,
                                                               RNG &rngState) final {
    return {stickingProbability_, ReflectionDiffuse<NumericType, D>(geomNormal, rngState)};
  }
  void surfaceCollision(NumericType rayWeight, const Vec3D<NumericType> &rayDir, const Vec3D<NumericType> &geomNormal,
                        const unsigned int primID, const int, TracingData<NumericType> &localData,
                        const TracingData<NumericType> *, RNG &) final {
    localData.getVectorData(0)[primID] += rayWeight;
    localData.getVectorData(1)[primID] += rayWeight * std::max(-DotProduct(rayDir, geomNormal), NumericType(0));
  }
  NumericType getSourceDistributionPower() const final { return 1.; }
  [[nodiscard]] std::vector<std::string> getLocalDataLabels() const final { return {dataLabel_, cosineLabel_}; }
  bool deviceModel(vr_particle &pod) const final {
    pod = vr_particle{VR_PARTICLE_DIFFUSE_COSINE, (float)stickingProbability_, 1.f, 0, nullptr, nullptr, 0.f, -1.f};
    return true;
  }
};

// The pattern `globalData` exists for (ViennaPS surface models): a diffuse particle whose sticking falls with the coverage
// of the primitive it meets.  coverage = vector `coverageVector` of the TracingData given to Trace::setGlobalData.
template <typename NumericType, int D>
class CoverageStickingParticle : public Particle<CoverageStickingParticle<NumericType, D>, NumericType> {
  const NumericType stickingProbability_;
  const std::string dataLabel_;
  const int coverageVector_;

public:
  CoverageStickingParticle(NumericType stickingProbability, std::string dataLabel, int coverageVector = 0)
      : stickingProbability_(stickingProbability), dataLabel_(std::move(dataLabel)), coverageVector_(coverageVector) {}
  std::pair<NumericType, Vec3D<NumericType>> surfaceReflection(NumericType, const Vec3D<NumericType> &,
                                                               const Vec3D<NumericType> &geomNormal,
                                                               const unsigned int primID, const int,
                                                               const TracingData<NumericType> *globalData,
                                                               RNG &rngState) final {
    NumericType coverage = 0;
    if (globalData && coverageVector_ < (int)globalData->getVectorData().size() &&
        primID < globalData->getVectorData(coverageVector_).size())
      coverage = globalData->getVectorData(coverageVector_)[primID];
    return {stickingProbability_ * (NumericType(1) - coverage), ReflectionDiffuse<NumericType, D>(geomNormal, rngState)};
  }
  void surfaceCollision(NumericType rayWeight, const Vec3D<NumericType> &, const Vec3D<NumericType> &,
                        const unsigned int primID, const int, TracingData<NumericType> &localData,
                        const TracingData<NumericType> *, RNG &) final {
    localData.getVectorData(0)[primID] += rayWeight;
  }
  NumericType getSourceDistributionPower() const final { return 1.; }
  [[nodiscard]] std::vector<std::string> getLocalDataLabels() const final { return {dataLabel_}; }
  bool deviceModel(vr_particle &pod) const final {
    pod = vr_particle{VR_PARTICLE_COVERAGE_STICKING, (float)stickingProbability_, 1.f, 0, nullptr, nullptr, 0.f, -1.f, {}};
    pod.params[0] = (float)coverageVector_;
    return true;
  }
};

// A particle whose device model was registered at run time (Trace::registerParticleModel — the counterpart of the
// reference's GPU path naming user callables per particle, gpu/raygCallableConfig.hpp:7-18): `kind` is what the
// registration returned, `params` go to the model as ModelCtx::params.  The host virtuals keep the no-op defaults of
// Particle<>: the per-hit logic IS the device model.
template <typename NumericType, int D>
class UserModelParticle : public Particle<UserModelParticle<NumericType, D>, NumericType> {
  const int kind_;
  const NumericType stickingProbability_, sourcePower_;
  const std::vector<std::string> dataLabels_;
  std::array<float, 8> params_{};

public:
  UserModelParticle(int kind, NumericType stickingProbability, std::vector<std::string> dataLabels,
                    NumericType sourcePower = NumericType(1), const std::vector<float> &params = {})
      : kind_(kind), stickingProbability_(stickingProbability), sourcePower_(sourcePower), dataLabels_(std::move(dataLabels)) {
    for (size_t k = 0; k < params.size() && k < params_.size(); ++k)
      params_[k] = params[k];
  }
  NumericType getSourceDistributionPower() const final { return sourcePower_; }
  [[nodiscard]] std::vector<std::string> getLocalDataLabels() const final { return dataLabels_; }
  bool deviceModel(vr_particle &pod) const final {
    pod = vr_particle{kind_, (float)stickingProbability_, (float)sourcePower_, 0, nullptr, nullptr, 0.f, -1.f, {}};
    for (size_t k = 0; k < params_.size(); ++k)
      pod.params[k] = params_[k];
    return true;
  }
};

// ---- Trace<T,D> ----------------------------------------------------------------------
template <class NumericType, int D> class Trace {
public:
  Trace() {
    if (vr_create(&ctx_, 0) != VR_OK) {
      ctx_ = nullptr;
      RTInfo_.error = true;
      std::cerr << "viennaray_amd: no usable HIP device (there is no CPU fallback)\n";
    }
  }
  Trace(const Trace &) = delete;
  Trace &operator=(const Trace &) = delete;
  Trace(Trace &&) = delete;
  Trace &operator=(Trace &&) = delete;
  virtual ~Trace() { vr_destroy(ctx_); }

  /// Run the ray tracer
  virtual void apply() {
    if (!ctx_ || pParticle_ == nullptr) {
      RTInfo_.error = true;
      std::cerr << "No particle was specified in rayTrace. Aborting.\n";
      return;
    }
    if (!particleOnDevice_) {
      RTInfo_.error = true;
      std::cerr << "viennaray_amd: this particle type has no device model (AbstractParticle::deviceModel): its "
                   "surfaceCollision / surfaceReflection exist only as host virtuals, which a HIP kernel cannot call. "
                   "Add a model to viennaray_amd/csrc/vr_particles.hpp and override deviceModel().\n";
      return;
    }
    if (setterError_) { // a setter was refused (e.g. too many primitives): do not trace something else
      RTInfo_.error = true;
      std::cerr << vr_last_error(ctx_) << "\n";
      return;
    }
    if (pSource_ && !sourceOnDevice_ && !uploadHostSource())
      return;
    uploadGlobalData();
    const int rc = world_ > 1 ? vr_apply_sharded(ctx_, rank_, world_, reduce_, reduceUser_) : vr_apply(ctx_);
    vr_trace_info i{};
    vr_get_trace_info(ctx_, &i);
    RTInfo_.numRays = i.numRays;
    RTInfo_.totalRaysTraced = i.totalRaysTraced;
    RTInfo_.nonGeometryHits = i.nonGeometryHits;
    RTInfo_.geometryHits = i.geometryHits;
    RTInfo_.particleHits = i.particleHits;
    RTInfo_.boundaryHits = i.boundaryHits;
    RTInfo_.reflections = i.reflections;
    RTInfo_.time = i.time;
    // NumericType = double (rayTrace.hpp:15 allows it): positions, directions and weights are narrowed to float at the
    // C ABI and traced in float — the result is NumericType = float's.  Narrower arithmetic than asked for: flagged.
    RTInfo_.warning = i.warning != 0 || !std::is_same_v<NumericType, float>;
    if constexpr (!std::is_same_v<NumericType, float>) {
      static bool told = false;
      if (!told) {
        told = true;
        std::cerr << "viennaray_amd: NumericType is not float: geometry and results are narrowed to float at the device "
                     "boundary (TraceInfo.warning is set).\n";
      }
    }
    RTInfo_.error = i.error != 0 || rc != VR_OK;
    if (rc != VR_OK) {
      std::cerr << vr_last_error(ctx_) << "\n";
      return;
    }
    // rayTraceDisk.hpp:40-47: one vector per data label of the particle (of every particle of a list, in order)
    auto labels = pParticle_->getLocalDataLabels();
    for (const auto &extra : moreParticles_) {
      auto l = extra->getLocalDataLabels();
      labels.insert(labels.end(), l.begin(), l.end());
    }
    localData_.setNumberOfVectorData((int)labels.size());
    const uint32_t n = vr_num_primitives(ctx_);
    const uint32_t nd = vr_num_data(ctx_);
    for (int k = 0; k < (int)labels.size(); ++k) {
      localData_.setVectorData(k, n, NumericType(0), labels[k]);
      if ((uint32_t)k >= nd)
        continue;
      if constexpr (std::is_same_v<NumericType, float>) {
        vr_get_flux_data(ctx_, (uint32_t)k, localData_.getVectorData(k).data(), n);
      } else {
        std::vector<float> tmp(n);
        vr_get_flux_data(ctx_, (uint32_t)k, tmp.data(), n);
        for (uint32_t j = 0; j < n; ++j)
          localData_.getVectorData(k)[j] = (NumericType)tmp[j];
      }
    }
  }

  template <typename ParticleType,
            std::enable_if_t<std::is_base_of_v<AbstractParticle<NumericType>, ParticleType>, bool> = true>
  void setParticleType(std::unique_ptr<ParticleType> const &particle) {
    pParticle_ = particle->clone();
    moreParticles_.clear();
    vr_particle pod{};
    particleOnDevice_ = pParticle_->deviceModel(pod);
    if (ctx_ && particleOnDevice_)
      check(vr_set_particle(ctx_, &pod));
  }
  /// NOT in the reference's CPU Trace: compile a device particle model at run time (vr_register_particle_model: HIP source
  /// of `struct VrUserModel`, see viennaray_amd/csrc/vr_particles.hpp) and get the kind id for a UserModelParticle.
  /// Returns -1 (and sets TraceInfo.error) if the model does not compile.
  int registerParticleModel(const std::string &name, const std::string &source, int numData = 1, bool needsFull = false) {
    int32_t kind = -1;
    if (ctx_)
      check(vr_register_particle_model(ctx_, name.c_str(), source.c_str(), numData, needsFull ? VR_MODEL_NEEDS_FULL : 0, &kind));
    return kind;
  }
  /// NOT in the reference's CPU Trace (its gpu::Trace keeps a particle list, gpu/raygTrace.hpp:163-248): several
  /// particles traced in ONE apply() — the same seed for all, one generator pass per source distribution;
  /// getLocalData() holds particle 0's data labels, then particle 1's, ...; getRayTraceInfo() their summed counters.
  void setParticleTypes(const std::vector<std::unique_ptr<AbstractParticle<NumericType>>> &particles) {
    if (particles.empty())
      return;
    std::vector<vr_particle> pods(particles.size());
    moreParticles_.clear();
    particleOnDevice_ = true;
    for (size_t q = 0; q < particles.size(); ++q) {
      auto copy = particles[q]->clone();
      particleOnDevice_ = copy->deviceModel(pods[q]) && particleOnDevice_;
      if (q == 0)
        pParticle_ = std::move(copy);
      else
        moreParticles_.push_back(std::move(copy));
    }
    if (ctx_ && particleOnDevice_)
      check(vr_set_particles(ctx_, pods.data(), (uint32_t)pods.size()));
  }

  void setBoundaryConditions(BoundaryCondition boundaryConditions[D]) {
    int32_t b[3] = {0, 0, 0};
    for (int i = 0; i < D; ++i)
      b[i] = (int32_t)boundaryConditions[i];
    if (ctx_)
      check(vr_set_boundary_conditions(ctx_, b, D));
  }
  void setNumberOfRaysPerPoint(const size_t n) {
    numRaysPerPoint_ = n;
    numRaysFixed_ = 0;
    if (ctx_)
      check(vr_set_number_of_rays_per_point(ctx_, n));
  }
  void setNumberOfRaysFixed(const size_t n) {
    numRaysFixed_ = n;
    numRaysPerPoint_ = 0;
    if (ctx_)
      check(vr_set_number_of_rays_fixed(ctx_, n));
  }
  void setMaxReflections(const unsigned n) {
    if (ctx_)
      check(vr_set_max_reflections(ctx_, n));
  }
  void setMaxBoundaryHits(const unsigned n) {
    if (ctx_)
      check(vr_set_max_boundary_hits(ctx_, n));
  }
  void setSourceDirection(const TraceDirection direction) {
    if (ctx_)
      check(vr_set_source_direction(ctx_, (int)direction));
  }
  void setPrimaryDirection(const Vec3D<NumericType> primaryDirection) {
    const float d[3] = {(float)primaryDirection[0], (float)primaryDirection[1], (float)primaryDirection[2]};
    if (ctx_)
      check(vr_set_primary_direction(ctx_, d));
  }
  void setUseRandomSeeds(const bool useRand) {
    useRandomSeeds_ = useRand;
    if (ctx_)
      check(vr_set_use_random_seeds(ctx_, useRand ? 1 : 0));
  }
  void setRngSeed(const unsigned int seed) {
    rngSeed_ = seed;
    useRandomSeeds_ = false;
    if (ctx_)
      check(vr_set_rng_seed(ctx_, seed));
  }
  /// NOT in the reference: make apply() one rank of a multi-GPU trace (one process per GPU).  This rank
  /// traces its share of the global ray indices and `reduce` sums the int64 accumulators over the ranks
  /// (vr_rccl_allreduce of viennaray_amd_rccl.h = RCCL over xGMI); every rank ends with the full flux.
  void setDistributed(int rank, int world, vr_allreduce_fn reduce, void *user) {
    rank_ = rank;
    world_ = world;
    reduce_ = reduce;
    reduceUser_ = user;
  }
  /// VIENNARAY_USE_WDIST (a compile-time option of the reference, CMakeLists.txt:15) as a run-time switch
  void setUseWdist(const bool on) {
    if (ctx_)
      check(vr_set_use_wdist(ctx_, on ? 1 : 0));
  }
  // rayTrace.hpp:53-61.  SourceGrid goes to the generator kernel; any other Source is evaluated on the
  // host at apply() (uploadHostSource)
  void setSource(std::shared_ptr<Source<NumericType>> source) {
    pSource_ = std::move(source);
    sourceOnDevice_ = false;
    if (!ctx_ || !pSource_)
      return;
    if (auto *g = dynamic_cast<SourceGrid<NumericType, D> *>(pSource_.get())) {
      auto pts = flatten3(g->getGrid());
      check(vr_set_source_grid(ctx_, pts.data(), (uint32_t)g->getGrid().size()));
      check(vr_set_source_area(ctx_, (float)g->getSourceArea()));
      sourceOnDevice_ = true;
    }
  }
  void resetSource() {
    pSource_.reset();
    sourceOnDevice_ = false;
    if (ctx_)
      check(vr_set_source_grid(ctx_, nullptr, 0)); // also drops host rays
    if (ctx_)
      check(vr_set_source_area(ctx_, 0.f));
  }
  void enableProgressBar() {}
  void disableProgressBar() {}

  virtual void normalizeFlux(std::vector<NumericType> &flux, NormalizationType norm = NormalizationType::SOURCE) {
    inPlace(flux, [&](float *p, uint32_t n) { return vr_normalize_flux(ctx_, p, n, (int)norm); });
  }
  virtual void smoothFlux(std::vector<NumericType> &flux, int numNeighbors = 1) {
    inPlace(flux, [&](float *p, uint32_t n) { return vr_smooth_flux(ctx_, p, n, numNeighbors); });
  }

  [[nodiscard]] TracingData<NumericType> &getLocalData() { return localData_; }
  // rayTrace.hpp:137-145: global data is a borrowed pointer handed to user particles; apply() copies its vectors and
  // scalars to the device, where the registry's particle models read them (ModelCtx::global)
  [[nodiscard]] TracingData<NumericType> *getGlobalData() { return pGlobalData_; }
  void setGlobalData(TracingData<NumericType> &data) { pGlobalData_ = &data; }
  [[nodiscard]] TraceInfo getRayTraceInfo() const { return RTInfo_; }
  [[nodiscard]] DataLog<NumericType> &getDataLog() { return dataLog_; }
  /// the underlying C-ABI context (multi-GPU drivers use vr_set_ray_range etc.)
  [[nodiscard]] vr_context *getContext() { return ctx_; }

protected:
  void check(int rc) {
    if (rc != VR_OK) {
      setterError_ = true;
      RTInfo_.error = true;
      std::cerr << "viennaray_amd: " << vr_last_error(ctx_) << "\n";
    }
  }
  void geometryAccepted(int rc) { // a new geometry clears an earlier refusal
    setterError_ = false;
    check(rc);
  }
  // rayTrace.hpp:137-145: the borrowed global data may have changed since the last apply — its vectors and scalars
  // go to HBM again (they are what the device particle models read; a few MB at most)
  void uploadGlobalData() {
    if (!ctx_)
      return;
    check(vr_set_global_data(ctx_, 0, nullptr, 0));
    check(vr_set_global_scalars(ctx_, nullptr, 0));
    if (!pGlobalData_)
      return;
    const auto &vecs = pGlobalData_->getVectorData();
    for (size_t v = 0; v < vecs.size() && v < 16; ++v) {
      std::vector<float> tmp(vecs[v].begin(), vecs[v].end());
      if (!tmp.empty())
        check(vr_set_global_data(ctx_, (uint32_t)v, tmp.data(), (uint32_t)tmp.size()));
    }
    const auto &sc = pGlobalData_->getScalarData();
    std::vector<float> st(sc.begin(), sc.end());
    if (!st.empty())
      check(vr_set_global_scalars(ctx_, st.data(), (uint32_t)st.size()));
  }
  // A user Source is a host callback (raySource.hpp:10-19): evaluate it for every ray of the coming
  // apply() exactly as the reference's loop would (rayTraceKernel.hpp:118-140: engine seeded with
  // tea<3>(idx, seed), particle.initNew, source.getOriginAndDirection) and hand the rays over together
  // with the number of engine outputs each consumed.  Needs a fixed seed (setRngSeed).
  bool uploadHostSource() {
    if (useRandomSeeds_) {
      RTInfo_.error = true;
      std::cerr << "viennaray_amd: a host-callback Source needs a fixed seed (setRngSeed): the rays are "
                   "generated on the host before the launch.\n";
      return false;
    }
    uint32_t runNumber = 1;
    vr_get_run_number(ctx_, &runNumber);
    const unsigned seed = runNumber + rngSeed_; // rayTraceKernel.hpp:100
    const size_t perPoint = numRaysPerPoint_, fixed = numRaysFixed_;
    const size_t numRays = fixed ? fixed : pSource_->getNumPoints() * perPoint;
    if (numRays == 0 || numRays > 0xFFFFFFFFull) {
      RTInfo_.error = true;
      std::cerr << "viennaray_amd: host-callback Source: number of rays must be in [1, 2^32).\n";
      return false;
    }
    std::vector<float> org(numRays * 3), dir(numRays * 3), weights(numRays);
    std::vector<uint32_t> draws(numRays);
    bool unitWeights = true;
    auto particle = pParticle_->clone();
    for (size_t idx = 0; idx < numRays; ++idx) {
      weights[idx] = (float)pSource_->getInitialRayWeight(idx); // rayTraceKernel.hpp:124
      unitWeights = unitWeights && weights[idx] == 1.f;
      RNG rng(tea<3>((unsigned)idx, seed));
      particle->initNew(rng);
      auto pd = particle->initNewWithDirection(rng);
      auto od = pSource_->getOriginAndDirection(idx, rng);
      if (pd[0] != NumericType(0) || pd[1] != NumericType(0) || pd[2] != NumericType(0))
        od[1] = pd; // rayTraceKernel.hpp:138-140
      for (int k = 0; k < 3; ++k) {
        org[3 * idx + k] = (float)od[0][k];
        dir[3 * idx + k] = (float)od[1][k];
      }
      draws[idx] = (uint32_t)rng.draws;
    }
    int rc = vr_set_host_rays(ctx_, org.data(), dir.data(), draws.data(), numRays);
    if (rc == VR_OK && !unitWeights)
      rc = vr_set_host_ray_weights(ctx_, weights.data(), numRays);
    if (rc == VR_OK)
      rc = vr_set_source_area(ctx_, (float)pSource_->getSourceArea()); // rayTraceDisk.hpp:127: the source's own area
    if (rc != VR_OK) {
      RTInfo_.error = true;
      std::cerr << vr_last_error(ctx_) << "\n";
      return false;
    }
    return true;
  }
  template <class F> void inPlace(std::vector<NumericType> &flux, F f) {
    if (!ctx_)
      return;
    if constexpr (std::is_same_v<NumericType, float>) {
      if (f(flux.data(), (uint32_t)flux.size()) != VR_OK)
        std::cerr << vr_last_error(ctx_) << "\n";
    } else {
      std::vector<float> tmp(flux.begin(), flux.end());
      if (f(tmp.data(), (uint32_t)tmp.size()) != VR_OK)
        std::cerr << vr_last_error(ctx_) << "\n";
      std::copy(tmp.begin(), tmp.end(), flux.begin());
    }
  }
  template <class Vec> static std::vector<float> flatten3(const std::vector<Vec> &v) {
    std::vector<float> out(v.size() * 3, 0.f);
    for (size_t i = 0; i < v.size(); ++i)
      for (size_t k = 0; k < v[i].size() && k < 3; ++k)
        out[3 * i + k] = (float)v[i][k];
    return out;
  }

  vr_context *ctx_ = nullptr;
  std::unique_ptr<AbstractParticle<NumericType>> pParticle_ = nullptr;
  std::vector<std::unique_ptr<AbstractParticle<NumericType>>> moreParticles_; // setParticleTypes: particles 1 ..
  TracingData<NumericType> localData_;
  TracingData<NumericType> *pGlobalData_ = nullptr;
  DataLog<NumericType> dataLog_;
  std::shared_ptr<Source<NumericType>> pSource_;
  TraceInfo RTInfo_;
  bool particleOnDevice_ = false, sourceOnDevice_ = false, setterError_ = false;
  bool useRandomSeeds_ = true;
  unsigned rngSeed_ = 0;
  size_t numRaysPerPoint_ = 1000, numRaysFixed_ = 0;
  int rank_ = 0, world_ = 1;
  vr_allreduce_fn reduce_ = nullptr;
  void *reduceUser_ = nullptr;
};

template <class NumericType, int D> class TraceDisk final : public Trace<NumericType, D> {
public:
  template <size_t Dim>
  void setGeometry(std::vector<VectorType<NumericType, Dim>> const &points,
                   std::vector<VectorType<NumericType, Dim>> const &normals, const NumericType gridDelta) {
    setGeometry(points, normals, gridDelta, NumericType(0));
  }
  template <size_t Dim>
  void setGeometry(std::vector<VectorType<NumericType, Dim>> const &points,
                   std::vector<VectorType<NumericType, Dim>> const &normals, const NumericType gridDelta,
                   const NumericType diskRadii) {
    static_assert(!(D == 3 && Dim == 2), "Setting 2D geometry in 3D trace object");
    auto p = this->flatten3(points), n = this->flatten3(normals);
    if (this->ctx_)
      this->geometryAccepted(vr_set_disks(this->ctx_, p.data(), n.data(), (uint32_t)points.size(), (float)gridDelta,
                                          (float)diskRadii, D));
  }
  void setGeometry(const DiskMesh &mesh) {
    auto p = this->flatten3(mesh.nodes), n = this->flatten3(mesh.normals);
    if (this->ctx_)
      this->geometryAccepted(vr_set_disks(this->ctx_, p.data(), n.data(), (uint32_t)mesh.nodes.size(), mesh.gridDelta, 0.f, D));
  }
  template <typename T> void setMaterialIds(std::vector<T> const &materialIds) {
    std::vector<int32_t> ids(materialIds.begin(), materialIds.end());
    if (this->ctx_)
      this->check(vr_set_material_ids(this->ctx_, ids.data(), (uint32_t)ids.size()));
  }
};

template <class NumericType, int D> class TraceTriangle final : public Trace<NumericType, D> {
public:
  void setGeometry(std::vector<VectorType<NumericType, 3>> const &points,
                   std::vector<VectorType<unsigned, 3>> const &triangles, const NumericType gridDelta) {
    auto p = this->flatten3(points);
    std::vector<uint32_t> t(triangles.size() * 3);
    for (size_t i = 0; i < triangles.size(); ++i)
      for (int k = 0; k < 3; ++k)
        t[3 * i + k] = triangles[i][k];
    if (this->ctx_)
      this->geometryAccepted(vr_set_triangles(this->ctx_, p.data(), (uint32_t)points.size(), t.data(),
                                              (uint32_t)triangles.size(), (float)gridDelta, D));
  }
  void setGeometry(const TriangleMesh &mesh) {
    std::vector<VectorType<NumericType, 3>> pts(mesh.nodes.size());
    for (size_t i = 0; i < pts.size(); ++i)
      pts[i] = {(NumericType)mesh.nodes[i][0], (NumericType)mesh.nodes[i][1], (NumericType)mesh.nodes[i][2]};
    setGeometry(pts, mesh.triangles, (NumericType)mesh.gridDelta);
  }
  // rayTraceTriangle.hpp:76-81 (2-D only)
  void setGeometry(const LineMesh &mesh) {
    static_assert(D == 2 || D == 3, "dimension");
    setGeometry(convertLinesToTriangles(mesh));
  }
  template <typename T> void setMaterialIds(std::vector<T> const &materialIds) {
    std::vector<int32_t> ids(materialIds.begin(), materialIds.end());
    if (this->ctx_)
      this->check(vr_set_material_ids(this->ctx_, ids.data(), (uint32_t)ids.size()));
  }
};

} // namespace viennaray

namespace rayInternal {
using namespace viennaray;

// file formats: rayUtil.hpp:353-411
template <typename NumericType>
void readGridFromFile(const std::string &fileName, NumericType &gridDelta, std::vector<Vec3D<NumericType>> &points,
                      std::vector<Vec3D<NumericType>> &normals) {
  std::ifstream f(fileName);
  if (!f.is_open()) {
    std::cout << "Cannot read file " << fileName << std::endl;
    return;
  }
  size_t n;
  f >> n >> gridDelta;
  points.resize(n);
  normals.resize(n);
  for (auto &p : points)
    f >> p[0] >> p[1] >> p[2];
  for (auto &p : normals)
    f >> p[0] >> p[1] >> p[2];
}

template <typename NumericType, int D>
void readMeshFromFile(const std::string &fileName, NumericType &gridDelta, std::vector<Vec3D<NumericType>> &nodes,
                      std::vector<VectorType<unsigned, D>> &elements) {
  std::ifstream f(fileName);
  if (!f.is_open()) {
    std::cerr << "Failed to open mesh file: " << fileName << "\n";
    return;
  }
  std::string id;
  size_t nn = 0, ne = 0;
  f >> id >> gridDelta >> id >> nn >> id >> ne;
  nodes.resize(nn);
  for (auto &p : nodes)
    f >> id >> p[0] >> p[1] >> p[2];
  // like the reference, a file holding fewer elements than declared leaves the rest
  // value-initialised (lineMesh.dat: 130 declared, 129 present -> one zero-length line)
  elements.assign(ne, VectorType<unsigned, D>{});
  for (size_t i = 0; i < ne && (f >> id); ++i) {
    VectorType<unsigned, D> e{};
    for (int j = 0; j < D; ++j)
      f >> e[j];
    if (f)
      elements[i] = e;
  }
}

// rayUtil.hpp:340-351: a flat point cloud in the plane spanned by direction[0] x direction[1] at coordinate 0 of
// direction[2], from -extent to +extent in steps of gridDelta (coordinates by repeated addition, like the reference's
// loops: the float sums are the coordinates), normals along +direction[2]
template <typename NumericType>
void createPlaneGrid(const NumericType gridDelta, const NumericType extent, const std::array<int, 3> direction,
                     std::vector<Vec3D<NumericType>> &points, std::vector<Vec3D<NumericType>> &normals) {
  const int outer = direction[0], inner = direction[1], up = direction[2];
  Vec3D<NumericType> unitNormal{0, 0, 0};
  unitNormal[up] = 1;
  Vec3D<NumericType> cursor{-extent, -extent, -extent};
  cursor[up] = 0;
  points.clear();
  normals.clear();
  while (cursor[outer] <= extent) {
    cursor[inner] = -extent;
    while (cursor[inner] <= extent) {
      points.push_back(cursor);
      cursor[inner] += gridDelta;
    }
    cursor[outer] += gridDelta;
  }
  normals.assign(points.size(), unitNormal);
}

template <typename NumericType, int D = 3, typename FluxType = NumericType>
void writeVTK(const std::string &filename, const std::vector<Vec3D<NumericType>> &points,
              const std::vector<FluxType> &flux) {
  // byte for byte the reference's output (rayUtil.hpp:413-449): each coordinate is followed by a
  // blank, |flux| < 1e-6 prints as 0
  std::ofstream f(filename.c_str());
  f << "# vtk DataFile Version 2.0\n" << D << "D Surface\nASCII\nDATASET UNSTRUCTURED_GRID\n";
  f << "POINTS " << points.size() << " float\n";
  for (auto const &p : points) {
    for (int j = 0; j < 3; j++)
      f << static_cast<float>(p[j]) << " ";
    f << "\n";
  }
  f << "CELLS " << points.size() << " " << points.size() * 2 << "\n";
  for (size_t i = 0; i < points.size(); ++i)
    f << 1 << " " << i << "\n";
  f << "CELL_TYPES " << points.size() << "\n";
  for (size_t i = 0; i < points.size(); ++i)
    f << 1 << "\n";
  f << "CELL_DATA " << flux.size() << "\nSCALARS flux float\nLOOKUP_TABLE default\n";
  for (size_t j = 0; j < flux.size(); ++j)
    f << ((std::abs(flux[j]) < 1e-6) ? 0.0 : flux[j]) << "\n";
}

// rayUtil.hpp:451-555: VTK PolyData (lines for D == 2, polygons for D == 3); the flux goes to the
// points if its length is the point count, else to the cells if it is the element count
template <typename NumericType, int D = 3, typename ResultType = NumericType>
void writeVTP(const std::string &filename, const std::vector<Vec3D<NumericType>> &points,
              const std::vector<VectorType<unsigned, D>> &elements, const std::vector<ResultType> &flux) {
  std::ofstream f(filename.c_str());
  if (!f.is_open())
    return;
  const size_t nPoints = points.size(), nElements = elements.size();
  f << "<?xml version=\"1.0\"?>\n<VTKFile type=\"PolyData\" version=\"0.1\" byte_order=\"LittleEndian\">\n  <PolyData>\n";
  f << "    <Piece NumberOfPoints=\"" << nPoints << "\" NumberOfVerts=\"0\" NumberOfLines=\"" << (D == 2 ? nElements : 0)
    << "\" NumberOfStrips=\"0\" NumberOfPolys=\"" << (D == 2 ? 0 : nElements)
    << "\">\n      <Points>\n        <DataArray type=\"Float32\" NumberOfComponents=\"3\" format=\"ascii\">\n";
  for (auto const &p : points)
    f << static_cast<float>(p[0]) << " " << static_cast<float>(p[1]) << " " << static_cast<float>(p[2]) << "\n";
  f << "        </DataArray>\n      </Points>\n      " << (D == 2 ? "<Lines>" : "<Polys>")
    << "\n        <DataArray type=\"Int32\" Name=\"connectivity\" format=\"ascii\">\n";
  for (auto const &e : elements) {
    for (int j = 0; j < D; ++j)
      f << static_cast<int>(e[j]) << " ";
    f << "\n";
  }
  f << "        </DataArray>\n        <DataArray type=\"Int32\" Name=\"offsets\" format=\"ascii\">\n";
  int offset = 0;
  for (size_t i = 0; i < nElements; ++i) {
    offset += D;
    f << offset << "\n";
  }
  f << "        </DataArray>\n      " << (D == 2 ? "</Lines>" : "</Polys>") << "\n";
  if (flux.size() == nPoints || flux.size() == nElements) {
    const bool point = flux.size() == nPoints;
    f << "      " << (point ? "<PointData" : "<CellData") << " Scalars=\"flux\">\n"
      << "        <DataArray type=\"Float32\" Name=\"flux\" format=\"ascii\">\n";
    for (size_t i = 0; i < flux.size(); ++i)
      f << ((std::abs(flux[i]) < 1e-6) ? 0.0f : static_cast<float>(flux[i])) << "\n";
    f << "        </DataArray>\n      " << (point ? "</PointData>" : "</CellData>") << "\n";
  } else if (!flux.empty()) {
    std::cerr << "writeVTP: flux size does not match points or polys; skipping data\n";
  }
  f << "    </Piece>\n  </PolyData>\n</VTKFile>\n";
}
inline void writeVTP(TriangleMesh const &mesh, const std::string &filename, const std::vector<double> &flux) {
  writeVTP<float, 3>(filename, mesh.nodes, mesh.triangles, flux);
}
} // namespace rayInternal

// The reference's API walk (tests/traceInterface/traceInterface.cpp:8-70) on the drop-in façade,
// extended over the extension points of SURVEY 8f N2: a user Source (host callback), SourceGrid, a
// plug-in particle with two data labels, a coned-cosine particle, and a host-only user particle
// (which must be refused, not silently replaced).  Needs a GPU to run; it must always COMPILE.
#include <rayParticle.hpp>
#include <raySourceGrid.hpp>
#include <rayTraceDisk.hpp>

#include <cstdio>
#include <cstdlib>

using namespace viennaray;

#define VC_TEST_ASSERT(x)                                                                                              \
  do {                                                                                                                 \
    if (!(x)) {                                                                                                        \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #x);                                              \
      std::exit(1);                                                                                                    \
    }                                                                                                                  \
  } while (0)

// tests/traceInterface/traceInterface.cpp:7-22, verbatim in shape
template <typename NumericType, int D> class MySource : public Source<NumericType> {
public:
  MySource() {}
  std::array<Vec3D<NumericType>, 2> getOriginAndDirection(const size_t idx, RNG &rngState) const override {
    Vec3D<NumericType> origin{0., 0., 0.};
    Vec3D<NumericType> direction{0., 0., 1.};
    return {origin, direction};
  }
  size_t getNumPoints() const override { return 0; }
  NumericType getSourceArea() const override { return 1; }
};

// a source that shoots straight down from above a given point and draws one number per ray
template <typename NumericType> class BeamSource : public Source<NumericType> {
  size_t n_;

public:
  explicit BeamSource(size_t n) : n_(n) {}
  std::array<Vec3D<NumericType>, 2> getOriginAndDirection(const size_t idx, RNG &rngState) const override {
    std::uniform_real_distribution<NumericType> u(-1, 1);
    Vec3D<NumericType> origin{u(rngState), NumericType(0.3), NumericType(3)};
    Vec3D<NumericType> direction{0., 0., -1.};
    return {origin, direction};
  }
  size_t getNumPoints() const override { return n_; }
  NumericType getSourceArea() const override { return 1; }
};

// a user particle that exists only as host virtuals
template <typename NumericType> class HostOnlyParticle : public Particle<HostOnlyParticle<NumericType>, NumericType> {
public:
  std::pair<NumericType, Vec3D<NumericType>> surfaceReflection(NumericType, const Vec3D<NumericType> &rayDir,
                                                               const Vec3D<NumericType> &geomNormal, const unsigned int,
                                                               const int, const TracingData<NumericType> *,
                                                               RNG &rng) override {
    return {NumericType(0.5), ReflectionConedCosine<NumericType, 3>(rayDir, geomNormal, rng, NumericType(0.4))};
  }
  std::vector<std::string> getLocalDataLabels() const override { return {"mine"}; }
};

int main() {
  constexpr int D = 3;
  using NumericType = float;
  NumericType extent = 5;
  NumericType gridDelta = 0.5;
  std::vector<VectorType<NumericType, D>> points;
  std::vector<VectorType<NumericType, D>> normals;
  rayInternal::createPlaneGrid(gridDelta, extent, {0, 1, 2}, points, normals);
  std::vector<NumericType> matIds(points.size(), 0);

  BoundaryCondition boundaryConds[D];
  boundaryConds[0] = BoundaryCondition::REFLECTIVE_BOUNDARY;
  boundaryConds[1] = BoundaryCondition::REFLECTIVE_BOUNDARY;
  boundaryConds[2] = BoundaryCondition::REFLECTIVE_BOUNDARY;
  auto particle = std::make_unique<DiffuseParticle<NumericType, D>>(NumericType(1), "hitFlux");

  TraceDisk<NumericType, D> rayTracer;
  rayTracer.setParticleType(particle);
  rayTracer.setGeometry(points, normals, gridDelta);
  rayTracer.setBoundaryConditions(boundaryConds);
  rayTracer.setSourceDirection(TraceDirection::POS_Z);
  rayTracer.setNumberOfRaysPerPoint(10);
  rayTracer.setUseRandomSeeds(false);
  rayTracer.setMaterialIds(matIds);

  auto mySource = std::make_shared<MySource<NumericType, D>>();
  rayTracer.setSource(mySource);
  rayTracer.resetSource();
  rayTracer.apply();

  auto flux = rayTracer.getLocalData().getVectorData(0);
  VC_TEST_ASSERT(flux.size() == points.size());
  rayTracer.normalizeFlux(flux);
  rayTracer.smoothFlux(flux, 2);
  VC_TEST_ASSERT(flux.size() == points.size());
  auto info = rayTracer.getRayTraceInfo();
  VC_TEST_ASSERT(info.numRays == 4410); // traceInterface.cpp:67
  VC_TEST_ASSERT(!info.error);
  std::printf("traceInterface: numRays %zu geometryHits %zu\n", info.numRays, info.geometryHits);

  { // host-callback source: every ray comes straight down at y = 0.3 -> only the disks under that line
    rayTracer.setRngSeed(7);
    rayTracer.setSource(std::make_shared<BeamSource<NumericType>>(1000));
    rayTracer.setNumberOfRaysPerPoint(3);
    rayTracer.apply();
    auto i2 = rayTracer.getRayTraceInfo();
    VC_TEST_ASSERT(!i2.error && i2.numRays == 3000 && i2.geometryHits == 3000);
    auto f2 = rayTracer.getLocalData().getVectorData("hitFlux");
    double inside = 0, outside = 0;
    for (size_t k = 0; k < points.size(); ++k)
      (std::fabs(points[k][1] - 0.3f) < 0.5f && std::fabs(points[k][0]) < 1.5f ? inside : outside) += f2[k];
    VC_TEST_ASSERT(inside > 0 && outside == 0);
    std::printf("host source: %g credited under the beam line, %g elsewhere\n", inside, outside);
    rayTracer.resetSource();
  }
  { // SourceGrid through the reference's own helper chain
    std::array<Vec3D<NumericType>, 2> bdBox{Vec3D<NumericType>{-extent, -extent, 0}, Vec3D<NumericType>{extent, extent, 0}};
    auto ts = rayInternal::getTraceSettings(TraceDirection::POS_Z);
    rayInternal::adjustBoundingBox<NumericType, D>(bdBox, TraceDirection::POS_Z, gridDelta);
    auto grid = rayInternal::createSourceGrid<NumericType, D>(bdBox, points.size(), gridDelta, ts);
    auto src = std::make_shared<SourceGrid<NumericType, D>>(bdBox, grid, NumericType(1), ts);
    rayTracer.setSource(src);
    rayTracer.setNumberOfRaysPerPoint(10);
    rayTracer.apply();
    auto i3 = rayTracer.getRayTraceInfo();
    VC_TEST_ASSERT(!i3.error && i3.numRays == grid.size() * 10);
    std::printf("source grid: %zu points, numRays %zu\n", grid.size(), i3.numRays);
    rayTracer.resetSource();
  }
  { // plug-in particle with two data labels
    rayTracer.setParticleType(std::make_unique<DiffuseCosineParticle<NumericType, D>>(NumericType(0.5), "flux", "cosFlux"));
    rayTracer.apply();
    auto &ld = rayTracer.getLocalData();
    VC_TEST_ASSERT(ld.getVectorDataIndex("cosFlux") == 1);
    double a = 0, b = 0;
    for (auto v : ld.getVectorData("flux"))
      a += v;
    for (auto v : ld.getVectorData("cosFlux"))
      b += v;
    VC_TEST_ASSERT(a > 0 && b > 0 && b < a);
    std::printf("two labels: sum flux %g, sum cosFlux %g\n", a, b);
  }
  { // coned-cosine plug-in
    rayTracer.setParticleType(std::make_unique<ConedCosineParticle<NumericType, D>>(NumericType(0.3), NumericType(2),
                                                                                   NumericType(0.5), "flux"));
    rayTracer.apply();
    VC_TEST_ASSERT(!rayTracer.getRayTraceInfo().error && rayTracer.getRayTraceInfo().reflections > 0);
  }
  { // a host-only user particle is refused, loudly
    rayTracer.setParticleType(std::make_unique<HostOnlyParticle<NumericType>>());
    rayTracer.apply();
    VC_TEST_ASSERT(rayTracer.getRayTraceInfo().error);
  }
  std::printf("facade interface ok\n");
  return 0;
}

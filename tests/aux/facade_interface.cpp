// Run-time walk over the drop-in façade's public API (the method list of SURVEY.md 8(b)), then the extension points of
// a user Source (host callback), SourceGrid, a
// plug-in particle with two data labels, a coned-cosine particle, and a host-only user particle
// (which must be refused, not silently replaced).  Needs a GPU to run; it must always COMPILE.  (That the reference's
// own tests/traceInterface and examples compile against the façade unchanged is checked on their sources in place:
// tests/test_facade_units.py::test_reference_sources_compile_against_the_facade.)
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

// the smallest possible user Source: one fixed ray, no points (set and dropped again before the first apply)
template <typename T> struct PinnedRaySource final : Source<T> {
  std::array<Vec3D<T>, 2> getOriginAndDirection(const size_t, RNG &) const override {
    return {Vec3D<T>{T(0), T(0), T(0)}, Vec3D<T>{T(0), T(0), T(1)}};
  }
  size_t getNumPoints() const override { return 0; }
  T getSourceArea() const override { return T(1); }
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
  NumericType getSourceArea() const override { return NumericType(1); }
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

// Every public method of Trace / TraceDisk that SURVEY.md 8(b) lists, once, on a flat 21 x 21 cloud (the plane of the
// reference's tests/traceInterface: 441 points x 10 rays per point = 4410 rays, its one known answer).
static void api_walk(TraceDisk<float, 3> &tracer, const std::vector<Vec3D<float>> &cloud,
                     const std::vector<Vec3D<float>> &cloudNormals, float delta, TracingData<float> &global) {
  using T = float;
  // configuration first, geometry last: the setters are order independent
  tracer.setUseRandomSeeds(false);
  tracer.setNumberOfRaysPerPoint(10);
  tracer.setSourceDirection(TraceDirection::POS_Z);
  BoundaryCondition walls[3] = {BoundaryCondition::REFLECTIVE_BOUNDARY, BoundaryCondition::REFLECTIVE_BOUNDARY,
                                BoundaryCondition::REFLECTIVE_BOUNDARY};
  tracer.setBoundaryConditions(walls);
  tracer.setMaxReflections(1000);
  tracer.setMaxBoundaryHits(500);
  std::unique_ptr<DiffuseParticle<T, 3>> absorber = std::make_unique<DiffuseParticle<T, 3>>(T(1), "hitFlux");
  tracer.setParticleType(absorber);
  tracer.setGeometry(cloud, cloudNormals, delta);
  tracer.setMaterialIds(std::vector<T>(cloud.size(), T(0)));
  // a user source that is dropped again leaves the built-in random source in charge
  tracer.setSource(std::make_shared<PinnedRaySource<T>>());
  tracer.resetSource();
  tracer.apply();

  const TraceInfo first = tracer.getRayTraceInfo();
  VC_TEST_ASSERT(!first.error);
  VC_TEST_ASSERT(first.numRays == 4410);                            // tests/traceInterface/traceInterface.cpp:67
  VC_TEST_ASSERT(first.geometryHits + first.nonGeometryHits == 4410); // absorbing: one outcome per ray
  VC_TEST_ASSERT(first.totalRaysTraced == first.geometryHits + first.nonGeometryHits + first.boundaryHits);
  TracingData<T> &local = tracer.getLocalData();
  VC_TEST_ASSERT(local.getVectorDataIndex("hitFlux") == 0 && local.getVectorData("hitFlux").size() == cloud.size());
  std::vector<T> raw = local.getVectorData(0);
  double credited = 0;
  for (T v : raw)
    credited += v;
  VC_TEST_ASSERT(credited >= (double)first.geometryHits); // the closest disk and every overlapping neighbour
  std::vector<T> shaped = raw;
  tracer.normalizeFlux(shaped);                      // SOURCE
  tracer.smoothFlux(shaped, 2);                      // two-neighbourhood
  VC_TEST_ASSERT(shaped.size() == cloud.size());
  std::vector<T> byMax = raw;
  tracer.normalizeFlux(byMax, NormalizationType::MAX);
  T top = 0;
  for (size_t k = 0; k < byMax.size(); ++k)
    top = std::max(top, byMax[k]);
  VC_TEST_ASSERT(top >= T(1));                       // the largest raw value maps to >= 1 (area correction at the rim)
  // the same seed again: the same flux, bit for bit (tests/rngSeed/rngSeed.cpp:48-51) — apply() advanced runNumber
  tracer.setRngSeed(0);
  tracer.apply();
  const std::vector<T> second = tracer.getLocalData().getVectorData(0);
  tracer.setRngSeed(1); // seed + runNumber: (1, run 2) != (0, run 2)
  tracer.apply();
  const std::vector<T> third = tracer.getLocalData().getVectorData(0);
  VC_TEST_ASSERT(second != third);
  // borrowed global data comes back as it went in; the data log is reachable
  // (the tracer keeps the POINTER, rayTrace.hpp:141: the data must outlive the applies that follow — main owns it)
  global.setNumberOfVectorData(1);
  global.setVectorData(0, cloud.size(), T(0.25), "coverage");
  tracer.setGlobalData(global);
  VC_TEST_ASSERT(tracer.getGlobalData() == &global && tracer.getGlobalData()->getVectorData("coverage")[7] == T(0.25));
  (void)tracer.getDataLog();
  std::printf("traceInterface: numRays %zu geometryHits %zu\n", first.numRays, first.geometryHits);
}

int main() {
  constexpr int Dim = 3;
  using Real = float;
  const Real extent = 5, gridDelta = 0.5;
  std::vector<VectorType<Real, Dim>> points, normals;
  rayInternal::createPlaneGrid(gridDelta, extent, {0, 1, 2}, points, normals);
  VC_TEST_ASSERT(points.size() == 441);
  TracingData<Real> globalData; // (declared before the tracer that borrows it)
  TraceDisk<Real, Dim> tracer;
  api_walk(tracer, points, normals, gridDelta, globalData);

  { // host-callback source: every ray comes straight down at y = 0.3 -> only the disks under that line
    tracer.setRngSeed(7);
    tracer.setSource(std::make_shared<BeamSource<Real>>(1000));
    tracer.setNumberOfRaysPerPoint(3);
    tracer.apply();
    auto i2 = tracer.getRayTraceInfo();
    VC_TEST_ASSERT(!i2.error && i2.numRays == 3000 && i2.geometryHits == 3000);
    auto f2 = tracer.getLocalData().getVectorData("hitFlux");
    double inside = 0, outside = 0;
    for (size_t k = 0; k < points.size(); ++k)
      (std::fabs(points[k][1] - 0.3f) < 0.5f && std::fabs(points[k][0]) < 1.5f ? inside : outside) += f2[k];
    VC_TEST_ASSERT(inside > 0 && outside == 0);
    std::printf("host source: %g credited under the beam line, %g elsewhere\n", inside, outside);
    tracer.resetSource();
  }
  { // SourceGrid through the reference's own helper chain
    std::array<Vec3D<Real>, 2> bdBox{Vec3D<Real>{-extent, -extent, 0}, Vec3D<Real>{extent, extent, 0}};
    auto ts = rayInternal::getTraceSettings(TraceDirection::POS_Z);
    rayInternal::adjustBoundingBox<Real, Dim>(bdBox, TraceDirection::POS_Z, gridDelta);
    auto grid = rayInternal::createSourceGrid<Real, Dim>(bdBox, points.size(), gridDelta, ts);
    auto src = std::make_shared<SourceGrid<Real, Dim>>(bdBox, grid, Real(1), ts);
    tracer.setSource(src);
    tracer.setNumberOfRaysPerPoint(10);
    tracer.apply();
    auto i3 = tracer.getRayTraceInfo();
    VC_TEST_ASSERT(!i3.error && i3.numRays == grid.size() * 10);
    std::printf("source grid: %zu points, numRays %zu\n", grid.size(), i3.numRays);
    tracer.resetSource();
  }
  { // plug-in particle with two data labels
    tracer.setParticleType(std::make_unique<DiffuseCosineParticle<Real, Dim>>(Real(0.5), "flux", "cosFlux"));
    tracer.apply();
    auto &ld = tracer.getLocalData();
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
    tracer.setParticleType(std::make_unique<ConedCosineParticle<Real, Dim>>(Real(0.3), Real(2),
                                                                                   Real(0.5), "flux"));
    tracer.apply();
    VC_TEST_ASSERT(!tracer.getRayTraceInfo().error && tracer.getRayTraceInfo().reflections > 0);
  }
  { // global data on the device.  A floor with a wall on it, so that reflected rays meet the surface again: coverage 1
    // everywhere = sticking 0 = a ray keeps its whole weight for the next hit; coverage 0 = plain sticking 0.5
    std::vector<VectorType<Real, Dim>> cornerPts = points, cornerNrm = normals;
    for (Real y = -extent; y <= extent; y += gridDelta)
      for (Real z = gridDelta; z <= Real(3); z += gridDelta) {
        cornerPts.push_back({Real(-2), y, z});
        cornerNrm.push_back({Real(1), Real(0), Real(0)});
      }
    TraceDisk<Real, Dim> corner;
    TracingData<Real> coverage; // (outlives the applies below: the tracer keeps the pointer)
    coverage.setNumberOfVectorData(1);
    coverage.setVectorData(0, cornerPts.size(), Real(1), "coverage");
    corner.setGeometry(cornerPts, cornerNrm, gridDelta);
    BoundaryCondition mirrors[Dim] = {BoundaryCondition::REFLECTIVE_BOUNDARY, BoundaryCondition::REFLECTIVE_BOUNDARY,
                                    BoundaryCondition::REFLECTIVE_BOUNDARY};
    corner.setBoundaryConditions(mirrors);
    corner.setNumberOfRaysPerPoint(20);
    corner.setRngSeed(3);
    corner.setGlobalData(coverage);
    corner.setParticleType(std::make_unique<CoverageStickingParticle<Real, Dim>>(Real(0.5), "flux", 0));
    auto total = [&] {
      corner.setRngSeed(3); // (the seed is rngSeed + runNumber, and apply() advances runNumber)
      corner.apply();
      double s = 0;
      for (auto v : corner.getLocalData().getVectorData("flux"))
        s += v;
      return s;
    };
    const double covered = total();
    const auto infoCovered = corner.getRayTraceInfo();
    for (auto &v : coverage.getVectorData(0))
      v = Real(0);
    const double bare = total(); // (the borrowed data changed between the applies: it is uploaded again)
    VC_TEST_ASSERT(!infoCovered.error && !corner.getRayTraceInfo().error && infoCovered.reflections > infoCovered.numRays / 2);
    VC_TEST_ASSERT(covered > bare * 1.02);
    std::printf("coverage sticking: %g credited fully covered, %g bare\n", covered, bare);
  }
  { // a particle list in one apply: the labels of all particles, in order
    std::vector<std::unique_ptr<AbstractParticle<Real>>> list;
    list.push_back(std::make_unique<DiffuseParticle<Real, Dim>>(Real(0.2), "neutral"));
    list.push_back(std::make_unique<SpecularParticle<Real, Dim>>(Real(0.9), Real(20), "ion"));
    tracer.setParticleTypes(list);
    tracer.apply();
    auto &ld = tracer.getLocalData();
    VC_TEST_ASSERT(!tracer.getRayTraceInfo().error && ld.getVectorDataIndex("neutral") == 0 && ld.getVectorDataIndex("ion") == 1);
    double a = 0, b = 0;
    for (auto v : ld.getVectorData("neutral"))
      a += v;
    for (auto v : ld.getVectorData("ion"))
      b += v;
    VC_TEST_ASSERT(a > 0 && b > 0 && a != b);
    std::printf("particle list: sum neutral %g, sum ion %g\n", a, b);
  }
  { // a device model registered at run time: a diffuse particle that credits weight x height of the primitive it meets
    // (reads nothing global) — compiled by the library, then traced like a built-in one
    const char *src = "struct VrUserModel : ModelDiffuse {\n"
                      "  template <class Credit>\n"
                      "  __device__ static void collide(const ModelCtx &m, float w, const V3 &, const V3 &, unsigned, Credit &&credit) {\n"
                      "    credit(0, w * m.params[0]);\n"
                      "  }\n"
                      "};\n";
    const int kind = tracer.registerParticleModel("scaled", src, 1);
    VC_TEST_ASSERT(kind >= VR_PARTICLE_USER_BASE && !tracer.getRayTraceInfo().error);
    tracer.setNumberOfRaysPerPoint(200); // (the two applies below use different seeds: enough rays for a 1 % comparison)
    tracer.setParticleType(std::make_unique<UserModelParticle<Real, Dim>>(kind, Real(0.5), std::vector<std::string>{"scaled"},
                                                                                 Real(1), std::vector<float>{2.5f}));
    tracer.apply();
    const std::vector<Real> scaled = tracer.getLocalData().getVectorData("scaled");
    tracer.setParticleType(std::make_unique<DiffuseParticle<Real, Dim>>(Real(0.5), "plain"));
    tracer.apply();
    const auto &plain = tracer.getLocalData().getVectorData("plain");
    double a = 0, b = 0;
    for (size_t k = 0; k < plain.size(); ++k) {
      a += scaled[k];
      b += plain[k];
    }
    VC_TEST_ASSERT(b > 0 && std::fabs(a / b - 2.5) < 0.05);
    std::printf("run-time model: sum %g = %.3f x the plain particle's %g\n", a, a / b, b);
    tracer.setNumberOfRaysPerPoint(10);
  }
  { // a host-only user particle is refused, loudly
    tracer.setParticleType(std::make_unique<HostOnlyParticle<Real>>());
    tracer.apply();
    VC_TEST_ASSERT(tracer.getRayTraceInfo().error);
  }
  std::printf("facade interface ok\n");
  return 0;
}

// triangle3D — the reference's examples/triangle3D workload on the façade.
// usage: triangle3D <trenchMesh.dat> [raysPerPoint]
#include <rayTraceTriangle.hpp>
#include <rayUtil.hpp>
#include <vcTimer.hpp>

using namespace viennaray;

int main(int argc, char **argv) {
  constexpr int D = 3;
  using NumericType = float;
  const std::string file = argc > 1 ? argv[1] : "trenchMesh.dat";
  const size_t raysPerPoint = argc > 2 ? std::stoul(argv[2]) : 2000;

  std::vector<Vec3D<NumericType>> points;
  std::vector<Vec3D<unsigned>> triangles;
  NumericType gridDelta = 0;
  rayInternal::readMeshFromFile<NumericType, D>(file, gridDelta, points, triangles);
  if (triangles.empty())
    return 2;

  TriangleMesh mesh(points, triangles, gridDelta);
  TraceTriangle<NumericType, D> tracer;
  tracer.setGeometry(mesh);
  auto particle = std::make_unique<DiffuseParticle<NumericType, D>>(NumericType(0.1), "flux");
  tracer.setParticleType(particle);
  tracer.setNumberOfRaysPerPoint(raysPerPoint);
  tracer.setRngSeed(12345);

  Timer timer;
  timer.start();
  tracer.apply();
  timer.finish();
  auto info = tracer.getRayTraceInfo();
  if (info.error)
    return 1;
  std::cout << "Tracing time: " << timer.currentDuration / 1e9 << " s\n";
  std::cout << "rays " << info.numRays << " geometryHits " << info.geometryHits << "\n";
  auto &localData = tracer.getLocalData();
  tracer.normalizeFlux(localData.getVectorData(0), NormalizationType::SOURCE);
  double s = 0;
  for (auto v : localData.getVectorData(0))
    s += v;
  std::cout << "mean normalised flux " << s / localData.getVectorData(0).size() << "\n";
  return 0;
}

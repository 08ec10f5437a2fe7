// triangle2D — the reference's examples/triangle2D workload (line mesh -> triangle strips,
// D = 2, DiffuseParticle sticking 0.1, 5000 rays per element) written against the drop-in
// façade.   usage: triangle2D <lineMesh.dat> [raysPerPoint]
#include <rayTraceTriangle.hpp>
#include <rayUtil.hpp>
#include <vcTimer.hpp>

using namespace viennaray;

int main(int argc, char **argv) {
  constexpr int D = 2;
  using NumericType = float;
  const std::string file = argc > 1 ? argv[1] : "lineMesh.dat";
  const size_t raysPerPoint = argc > 2 ? std::stoul(argv[2]) : 5000;

  std::vector<Vec3D<NumericType>> points;
  std::vector<Vec2D<unsigned>> lines;
  NumericType gridDelta = 0;
  rayInternal::readMeshFromFile<NumericType, D>(file, gridDelta, points, lines);
  if (points.empty())
    return 2;

  LineMesh lineMesh(points, lines, gridDelta);
  TraceTriangle<NumericType, D> tracer;
  tracer.setGeometry(lineMesh);

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
  std::cout << "Tracing time: " << timer.currentDuration / 1e9 << " s (device " << info.time << " s)\n";
  std::cout << "lines " << lineMesh.lines.size() << " rays " << info.numRays << " traces " << info.totalRaysTraced
            << " geometryHits " << info.geometryHits << " reflections " << info.reflections << "\n";

  auto &localData = tracer.getLocalData();
  tracer.normalizeFlux(localData.getVectorData(0), NormalizationType::SOURCE);
  double s = 0;
  for (auto v : localData.getVectorData(0))
    s += v;
  std::cout << "mean normalised flux " << s / localData.getVectorData(0).size() << "\n";

  auto triMesh = convertLinesToTriangles(lineMesh);
  rayInternal::writeVTP<NumericType, 3>("lineGeometryOutput.vtp", triMesh.nodes, triMesh.triangles,
                                        localData.getVectorData(0));
  return 0;
}

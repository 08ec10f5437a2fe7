// disk2D — the reference's examples/disk2D workload (239-disk 2-D trench, DiffuseParticle
// sticking 0.1, periodic walls, POS_Y source, 2000 rays per point) written against the
// drop-in façade.   usage: disk2D <trenchGrid2D.dat> [raysPerPoint]
#include <rayParticle.hpp>
#include <rayTraceDisk.hpp>
#include <vcTimer.hpp>

using namespace viennaray;

int main(int argc, char **argv) {
  constexpr int D = 2;
  using NumericType = float;
  const std::string file = argc > 1 ? argv[1] : "trenchGrid2D.dat";
  const size_t raysPerPoint = argc > 2 ? std::stoul(argv[2]) : 2000;

  NumericType gridDelta = 0;
  std::vector<VectorType<NumericType, 3>> points, normals;
  rayInternal::readGridFromFile(file, gridDelta, points, normals);
  if (points.empty())
    return 2;

  BoundaryCondition boundaryConds[D];
  boundaryConds[0] = BoundaryCondition::PERIODIC_BOUNDARY; // x
  boundaryConds[1] = BoundaryCondition::PERIODIC_BOUNDARY; // y

  auto particle = std::make_unique<DiffuseParticle<NumericType, D>>(NumericType(0.1), "flux");

  TraceDisk<NumericType, D> rayTracer;
  rayTracer.setGeometry(points, normals, gridDelta);
  rayTracer.setBoundaryConditions(boundaryConds);
  rayTracer.setParticleType(particle);
  rayTracer.setSourceDirection(TraceDirection::POS_Y);
  rayTracer.setNumberOfRaysPerPoint(raysPerPoint);
  rayTracer.setRngSeed(12345);

  Timer timer;
  timer.start();
  rayTracer.apply();
  timer.finish();
  auto info = rayTracer.getRayTraceInfo();
  if (info.error)
    return 1;
  std::cout << "Tracing time: " << timer.currentDuration / 1e9 << " s (device " << info.time << " s)\n";
  std::cout << "rays " << info.numRays << " traces " << info.totalRaysTraced << " geometryHits " << info.geometryHits
            << " reflections " << info.reflections << "\n";

  auto &flux = rayTracer.getLocalData().getVectorData("flux");
  rayTracer.normalizeFlux(flux, NormalizationType::SOURCE);
  rayTracer.smoothFlux(flux, 1);
  double s = 0;
  for (auto v : flux)
    s += v;
  std::cout << "mean normalised flux " << s / flux.size() << "\n";
  rayInternal::writeVTK<NumericType, D>("trenchResult2D.vtk", points, flux);
  return 0;
}

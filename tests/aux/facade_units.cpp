// Known answers of the reference's tests/tracingData, tests/particle and tests/linesToTriangles,
// run against the drop-in façade (no device needed: nothing here instantiates Trace::apply).
#include <rayMesh.hpp>
#include <rayParticle.hpp>
#include <rayTracingData.hpp>
#include <rayUtil.hpp>

#include <cstdio>
#include <cstdlib>

using namespace viennaray;

#define CHECK(x)                                                                                                       \
  do {                                                                                                                 \
    if (!(x)) {                                                                                                        \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #x);                                              \
      std::exit(1);                                                                                                    \
    }                                                                                                                  \
  } while (0)

int main(int argc, char **argv) {
  { // tests/tracingData/tracingData.cpp
    TracingData<float> d;
    d.setNumberOfScalarData(1);
    d.setNumberOfVectorData(1);
    CHECK(d.getScalarDataLabel(0) == "scalarData");
    CHECK(d.getVectorDataLabel(0) == "vectorData");
    d.setVectorData(0, 1000, 0, "zeroData");
    CHECK(d.getVectorDataLabel(0) == "zeroData");
    CHECK(d.getVectorData("zeroData").size() == 1000);
    d.setScalarData(0, 1, "oneData");
    CHECK(d.getScalarDataLabel(0) == "oneData");
    CHECK(d.getScalarData("oneData") == 1);
    d.resizeAllVectorData(10, 0.5);
    int counter = 0;
    for (const auto v : d.getVectorData(0)) {
      CHECK(v == 0.5);
      counter++;
    }
    CHECK(counter == 10);
    TracingData<float> moved = std::move(d);
    CHECK(d.getScalarData().data() == nullptr);
    CHECK(d.getVectorData().data() == nullptr);
    CHECK(moved.getVectorData(0).size() == 10);
  }
  { // tests/particle/particle.cpp
    auto p = std::make_unique<DiffuseParticle<float, 3>>(1.f, "test");
    CHECK(p->getSourceDistributionPower() == 1.);
    CHECK(p->getLocalDataLabels().size() == 1 && p->getLocalDataLabels()[0] == "test");
    auto q = std::make_unique<SpecularParticle<float, 3>>(1.f, 50.f, "test");
    CHECK(q->getSourceDistributionPower() == 50.);
    CHECK(q->getLocalDataLabels().size() == 1 && q->getLocalDataLabels()[0] == "test");
    auto c = q->clone();
    CHECK(c->getSourceDistributionPower() == 50.);
  }
  if (argc > 1) { // tests/linesToTriangles + rayMesh.hpp:27-80,133-175 on lineMesh.dat
    std::vector<Vec3D<float>> points;
    std::vector<Vec2D<unsigned>> lines;
    float gridDelta = 0;
    rayInternal::readMeshFromFile<float, 2>(argv[1], gridDelta, points, lines);
    CHECK(points.size() == 131 && lines.size() == 130 && gridDelta == 0.2f);
    LineMesh lineMesh(points, lines, gridDelta);
    CHECK(lineMesh.lines.size() == 128); // two zero-length lines dropped
    auto tri = convertLinesToTriangles(lineMesh);
    CHECK(tri.nodes.size() == 262 && tri.triangles.size() == 256);
    CHECK(tri.nodes[0][2] == 0.1f && tri.nodes[1][2] == -0.1f);
    CHECK(tri.triangles[0][0] == 2 * lineMesh.lines[0][0] && tri.triangles[0][2] == tri.triangles[0][0] + 1);
  }
  std::puts("facade units ok");
  return 0;
}

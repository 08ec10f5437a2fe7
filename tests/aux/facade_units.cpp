// Known answers of the reference's tests/tracingData, tests/particle and tests/linesToTriangles,
// run against the drop-in façade (no device needed: nothing here instantiates Trace::apply).
#include <rayMesh.hpp>
#include <rayParticle.hpp>
#include <rayTracingData.hpp>
#include <rayUtil.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>

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
  { // tests/utilFuncs/utilFuncs.cpp:10-86 (the ViennaCore vector helpers ViennaRay programs use)
    const float eps = 1e-6f;
    auto close = [&](float a, double b) { return std::fabs((double)a - b) <= eps * std::max(1.0, std::fabs(b)); };
    Vec3D<float> v1{1.2f, 2.4f, 3.6f}, v2{2.8f, 3.6f, 4.4f}, v3{1.f, 1.f, 1.f};
    auto r = v1 + v2;
    CHECK(close(r[0], 4.) && close(r[1], 6.) && close(r[2], 8.));
    r = Sum(v1, v2, v3);
    CHECK(close(r[0], 5.) && close(r[1], 7.) && close(r[2], 9.));
    r = v1 - v3;
    CHECK(close(r[0], 0.2) && close(r[1], 1.4) && close(r[2], 2.6));
    Vec3D<float> a{1.f, 0.f, 1.f}, b{1.f, 0.f, 0.f};
    CHECK(close(DotProduct(a, b), 1.));
    auto cp = CrossProduct(a, b);
    CHECK(close(cp[0], 0.) && close(cp[1], 1.) && close(cp[2], 0.));
    Vec3D<float> n1{1.f, 1.f, 1.f};
    CHECK(std::fabs(Norm(n1) - 1.73205f) < 1e-5f);
    Normalize(n1);
    CHECK(close(Norm(n1), 1.) && IsNormalized(n1));
    Vec3D<float> d1{1.f, 1.f, 1.f}, d2{2.f, 1.f, 1.f};
    CHECK(close(Distance(d1, d2), 1.));
    d1 = 2.f * d1;
    CHECK(close(d1[0], 2.) && close(d1[1], 2.) && close(d1[2], 2.));
    auto inv = Inv(d1);
    CHECK(close(inv[0], -2.) && close(inv[1], -2.) && close(inv[2], -2.));
    Vec3D<Vec3D<float>> coords{Vec3D<float>{0.f, 0.f, 0.f}, Vec3D<float>{1.f, 0.f, 1.f}, Vec3D<float>{1.f, 0.f, 0.f}};
    auto nn = ComputeNormal(coords);
    CHECK(close(nn[0], 0.) && close(nn[1], 1.) && close(nn[2], 0.));
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
  { // writeVTK / writeVTP: byte for byte what rayUtil.hpp:413-555 writes for these inputs
    auto slurp = [](const std::string &fn) {
      std::ifstream f(fn);
      return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    };
    const std::string dir = argc > 2 ? argv[2] : "/tmp";
    std::vector<Vec3D<float>> pts = {{0.f, 0.5f, -1.f}, {1.25f, 2.f, 3.f}, {1e-3f, 0.f, 7.f}};
    std::vector<float> flux = {1.5f, 5e-7f, 2.f};
    rayInternal::writeVTK<float, 3>(dir + "/u.vtk", pts, flux);
    const std::string vtk = "# vtk DataFile Version 2.0\n3D Surface\nASCII\nDATASET UNSTRUCTURED_GRID\nPOINTS 3 float\n"
                            "0 0.5 -1 \n1.25 2 3 \n0.001 0 7 \nCELLS 3 6\n1 0\n1 1\n1 2\nCELL_TYPES 3\n1\n1\n1\n"
                            "CELL_DATA 3\nSCALARS flux float\nLOOKUP_TABLE default\n1.5\n0\n2\n";
    CHECK(slurp(dir + "/u.vtk") == vtk);
    std::vector<VectorType<unsigned, 3>> tris = {{0u, 1u, 2u}};
    std::vector<double> cellFlux = {0.25};
    rayInternal::writeVTP<float, 3>(dir + "/u.vtp", pts, tris, cellFlux);
    const std::string head = "<?xml version=\"1.0\"?>\n<VTKFile type=\"PolyData\" version=\"0.1\" byte_order=\"LittleEndian\">\n"
                             "  <PolyData>\n";
    const std::string body3 =
        "    <Piece NumberOfPoints=\"3\" NumberOfVerts=\"0\" NumberOfLines=\"0\" NumberOfStrips=\"0\" NumberOfPolys=\"1\">\n"
        "      <Points>\n        <DataArray type=\"Float32\" NumberOfComponents=\"3\" format=\"ascii\">\n"
        "0 0.5 -1\n1.25 2 3\n0.001 0 7\n        </DataArray>\n      </Points>\n      <Polys>\n"
        "        <DataArray type=\"Int32\" Name=\"connectivity\" format=\"ascii\">\n0 1 2 \n        </DataArray>\n"
        "        <DataArray type=\"Int32\" Name=\"offsets\" format=\"ascii\">\n3\n        </DataArray>\n      </Polys>\n"
        "      <CellData Scalars=\"flux\">\n        <DataArray type=\"Float32\" Name=\"flux\" format=\"ascii\">\n0.25\n"
        "        </DataArray>\n      </CellData>\n";
    const std::string tail = "    </Piece>\n  </PolyData>\n</VTKFile>\n";
    CHECK(slurp(dir + "/u.vtp") == head + body3 + tail);
    // D == 2: lines; a flux as long as the point list goes to the points (it takes precedence)
    std::vector<VectorType<unsigned, 2>> lines = {{0u, 1u}, {1u, 2u}};
    std::vector<float> pointFlux = {1.f, 1e-9f, 3.f};
    rayInternal::writeVTP<float, 2>(dir + "/u2.vtp", pts, lines, pointFlux);
    const std::string body2 =
        "    <Piece NumberOfPoints=\"3\" NumberOfVerts=\"0\" NumberOfLines=\"2\" NumberOfStrips=\"0\" NumberOfPolys=\"0\">\n"
        "      <Points>\n        <DataArray type=\"Float32\" NumberOfComponents=\"3\" format=\"ascii\">\n"
        "0 0.5 -1\n1.25 2 3\n0.001 0 7\n        </DataArray>\n      </Points>\n      <Lines>\n"
        "        <DataArray type=\"Int32\" Name=\"connectivity\" format=\"ascii\">\n0 1 \n1 2 \n        </DataArray>\n"
        "        <DataArray type=\"Int32\" Name=\"offsets\" format=\"ascii\">\n2\n4\n        </DataArray>\n      </Lines>\n"
        "      <PointData Scalars=\"flux\">\n        <DataArray type=\"Float32\" Name=\"flux\" format=\"ascii\">\n1\n0\n3\n"
        "        </DataArray>\n      </PointData>\n";
    CHECK(slurp(dir + "/u2.vtp") == head + body2 + tail);
  }
  std::puts("facade units ok");
  return 0;
}

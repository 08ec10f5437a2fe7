// vr_host.hpp — host-side setup of the flux tracer: bounding box, trace
// settings, boundary walls, disk neighbourhoods, areas and the (host) LBVH.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "vr_area.hpp"
#include "vr_types.hpp"

namespace vr {

struct HostGeometry {
  int D = 3;
  int geo = 0; // 0 disk, 1 triangle
  uint32_t numPrims = 0;
  float gridDelta = 0.f, diskRadius = 0.f;
  // disks: Embree-style buffers (rayGeometryDisk.hpp:363-375)
  std::vector<float> disk4;   // n x {x,y,z,r}
  std::vector<float> normal3; // n x 3 (disk normals / triangle unit normals)
  std::vector<float> points3; // caller's points (neighbourhood input)
  // triangles
  std::vector<float> verts;   // nv x 3
  std::vector<uint32_t> tris; // n x 3
  std::vector<float> triAreas;
  float minC[3] = {0, 0, 0}, maxC[3] = {0, 0, 0};
  std::vector<int32_t> materialIds;
  // neighbourhood CSR in ORIGINAL ids, lists ascending
  std::vector<uint32_t> nbOff, nbIds;
};

struct Bvh {
  std::vector<float> nodes;     // 8 floats per node
  std::vector<uint32_t> order;  // leaf position -> original primitive id
  uint32_t numNodes = 0, numLeaves = 0, maxDepth = 0;
};

// geometry ingestion (restates rayGeometryDisk.hpp:101-193, rayGeometryTriangle.hpp:14-88 + rayMesh.hpp:99-112)
void host_set_disks(HostGeometry &g, const float *pts, const float *nrm, uint32_t n, float gridDelta, float radius,
                    int D);
void host_set_triangles(HostGeometry &g, const float *verts, uint32_t nv, const uint32_t *tris, uint32_t nt,
                        float gridDelta, int D);
// Sort plane of the ray stream: rays are binned by where they cross one plane normal to
// the tracing axis, and a wavefront's rays are coherent where they HIT if that plane is
// where most first hits happen.  Estimated from the geometry alone: histogram of the
// primitives' coordinates on the axis, weighted by the area they show the source
// (r^2 |n_axis| for a disc, |Ng_axis| / 2 for a triangle); the weighted mean of the
// fullest of 256 slices.  (Only orders the work: no influence on any result.)
float host_sort_plane(const HostGeometry &g, int axis, float fallback, float *modeShare = nullptr);
// (*modeShare: the fullest slice's share of the total shown area; ~1 for a flat surface)
// rayPointNeighborhood.hpp:42-107 as a CSR (all pairs within `dist`)
void host_neighbors(int D, const float *pts3, uint32_t n, float dist, const float *minC, std::vector<uint32_t> &off,
                    std::vector<uint32_t> &ids);

// rayUtil.hpp:104-202
void host_adjust_bbox(float *lo, float *hi, int D, int direction, float pad);
std::array<int, 5> host_trace_settings(int direction);
// rayBoundary.hpp:164-245
void host_build_walls(const float *lo, const float *hi, int firstDir, int secondDir, Tri *wall);
// rayUtil.hpp:287-321
void host_orthonormal_basis(const float *v, float *basis9);
// rayGeometryDisk.hpp:266-354 (+ rayDiskBoundingBoxIntersector.hpp)
void host_disk_areas(const HostGeometry &g, const AreaParams &p, std::vector<float> &areas);

// LBVH over primitive boxes; fills bvh.nodes / bvh.order
void host_build_bvh(const HostGeometry &g, Bvh &bvh);
// leaf-ordered primitive records (vr_types.hpp)
void host_pack_prims(const HostGeometry &g, const Bvh &bvh, std::vector<float> &prims);

} // namespace vr

// vr_types.hpp — POD types shared by the host side and the HIP kernels.
#pragma once
#include <cstdint>

namespace vr {

// Node of the stackless (escape-link) BVH, 32 bytes = two float4:
//   q0 = {lo.x, lo.y, lo.z, bits(link)}   q1 = {hi.x, hi.y, hi.z, bits(escape)}
// link:   internal -> index of the left child (the right child is the left
//         child's escape); leaf -> VR_LEAF | count << 27 | firstPrim
// escape: node to continue with when this subtree is done / culled
//         (VR_END terminates the traversal)
constexpr uint32_t VR_LEAF = 0x80000000u;
constexpr uint32_t VR_END = 0xFFFFFFFFu;
constexpr uint32_t VR_LEAF_FIRST_MASK = (1u << 27) - 1;
constexpr int VR_LEAF_MAX = 4;

// Primitive records, stored in BVH-leaf (Morton) order so a leaf is one
// contiguous, coalescable run:
//   disk     (32 B): {cx, cy, cz, r} {nx, ny, nz, bits(origId)}
//   triangle (64 B): {v0.xyz, bits(origId)} {e1.xyz, nn.x} {e2.xyz, nn.y} {Ng.xyz, nn.z}
//     e1 = v0-v1, e2 = v2-v0, Ng = (v1-v0)x(v2-v0); nn = unit normal used for
//     the back-face test and the reflection (rayGeometryTriangle.hpp:198-203)

struct Tri {       // boundary wall triangle, Embree's precomputed form
  float v0[3], e1[3], e2[3], Ng[3];
};

// number of per-ray RNG outputs kept in the LDS tape (tier 1)
constexpr int VR_TAPE = 16;
constexpr int VR_BLOCK = 256;

struct TraceParams {
  const float *nodes;         // float4 pairs
  const float *prims;         // float4 records
  const uint32_t *nbOff;      // [numPrims+1], disk neighbourhood CSR (leaf order)
  const uint32_t *nbIds;      // leaf positions
  const float *primSticking;  // optional [numPrims] (leaf order) or nullptr
  unsigned long long *fluxAcc;   // [numPrims] leaf order, fixed point 2^-40
  unsigned long long *counters;  // [8]
  unsigned long long *workCounter;
  unsigned long long *rngScratch; // [gridWaves][312][64]
  uint64_t rayFirst, rayEnd;  // global ray index range of this launch
  uint32_t seed;
  uint32_t numPrims;
  uint32_t maxReflections, maxBoundaryHits;
  uint32_t chunk;             // rays per work-queue grab (multiple of 256)
  int32_t rayDir, firstDir, secondDir, minMax;
  float posNeg, ee, sticking;
  int32_t bc0, bc1;
  int32_t useBasis;
  float basis[9];             // [b][component]
  float bbLo[3], bbHi[3];     // adjusted bounding box
  Tri wall[8];
};

// counters[] slots
enum {
  C_TRACES = 0,
  C_NONGEO,
  C_GEO,
  C_PARTICLE,
  C_BOUNDARY,
  C_REFLECTIONS,
  C_TERMINATED,
  C_TIER2,      // diagnostic: rays that needed the full-state RNG
  C_COUNT
};

} // namespace vr

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
constexpr uint32_t VR_QEND = 0x7FFFFFFFu;
constexpr uint32_t VR_WIDE_PRIMS = 0x80000000u; // wide-tree node: its children are primitive boxes // "no escape" in a 16-byte node (bit 31 is the leaf flag)

// Primitive records, stored in BVH-leaf (Morton) order so a leaf is one
// contiguous, coalescable run:
//   disk     (32 B): {cx, cy, cz, r} {nx, ny, nz, bits(origId)}
//   triangle (64 B): {v0.xyz, bits(origId)} {e1.xyz, nn.x} {e2.xyz, nn.y} {Ng.xyz, nn.z}
//     e1 = v0-v1, e2 = v2-v0, Ng = (v1-v0)x(v2-v0); nn = unit normal used for
//     the back-face test and the reflection (rayGeometryTriangle.hpp:198-203)

struct Tri {       // boundary wall triangle, Embree's precomputed form
  float v0[3], e1[3], e2[3], Ng[3];
};

// Ray stream record (HBM-resident, sorted by source-plane cell before tracing):
//   32 B: A = {org.x, org.y, org.z, dir.x}   B = {dir.y, dir.z, bits(idx - batchFirst), bits(k)}
//   k = number of engine outputs the source sampling consumed
//   (the form the absorbing kernels read: nothing after the first hit is observable, no engine state needed)
// Particles that keep going after a hit need the engine's streaming cursors {s[k], s[k+156]} (vr_device.hpp, struct
// Rng) as well — in 32 bytes too (round 3; it had been 48, and a 48-byte record straddles two 64-byte write bursts
// every other time: 96 bytes of fabric writes per ray against 64):
//   A = {org[firstDir], org[secondDir], dir.x, dir.y}   B = {dir.z, bits(idx - batchFirst), s[k+156] lo, hi}
//   org[rayDir] is the source plane, k the generator's fixed draw count and s[k] = k chain steps from the seed: the
//   tracer rebuilds them.  The sources for which that does not hold (tilted primary direction: k varies; SourceGrid and
//   host rays: any origin) add 16 bytes per ray in a side array, TraceParams::recExtra.
constexpr unsigned VR_BIN_CAP = 128; // record slots per sort bin (4 KB of 32-byte records: the generator's scattered stores
                                    // and the bin cursors do better with bins a page apart than with 64 slots)
constexpr int VR_BLOCK = 256;
constexpr int VR_MAX_LABELS = 4;         // data labels one particle model may have
constexpr unsigned VR_QUEUES = 8;        // work queues of the trace kernel: one per XCD
constexpr unsigned VR_QUEUE_STRIDE = 16; // u64 words between two queue cursors
// stack of the ordered per-lane walk: the first entries of a lane live in LDS ([entry][lane], 12 in
// the kernels that walk a lot, 4 in the absorbing flat-scene kernel), deeper ones in a per-wave global slab
constexpr unsigned VR_STACK_GLOBAL = 48;
#ifndef VR_STACK_LDS_ENTRIES
#define VR_STACK_LDS_ENTRIES 12 // (a -DVR_STACK_LDS_ENTRIES=2 build sends almost every deferred child through the slab: test variant)
#endif
constexpr int VR_STACK_LDS = VR_STACK_LDS_ENTRIES;
// scenes of a few hundred primitives (2-D simulations) live in LDS as a whole — pair nodes, primitive records,
// neighbourhood, flux accumulators: trace_kernel MODE 4 stages up to this many bytes per block (dynamic LDS)
constexpr int VR_RELIEF_STEPS = 16; // tiles relief_clip walks under one ray before it falls back to the scene box's exit
constexpr unsigned VR_SMALL_LDS = 25600; // (up to 17.5 KB: five blocks per CU; up to this: four, still ahead of the HBM path)
constexpr int VR_SMALL_STACK = 6; // LDS stack entries of that kernel (its trees are shallow)

struct TraceParams {
  // geometry (device pointers)
  const float *nodes;         // float4 pairs (pre-order; packet traversal, scalar fetch)
  const uint32_t *qnodes;     // uint4 per node: 16-bit boxes + link (per-lane traversal)
  const uint32_t *pnodes;     // uint4 pairs per internal node: both children, box + link each (ordered per-lane traversal)
  uint32_t *walkStack;        // [waves][VR_STACK_GLOBAL][64]: stack entries beyond the LDS-resident ones
  uint32_t numNodes;
  float qbase[3], qscale[3];  // quantised coordinate = (x - qbase) * qscale
  const float *prims;         // float4 records
  // 64-ary box tree over the primitives in leaf order (packet query, vr_device.hpp: pq_hit_packet):
  // entries of two float4 {lo.xyz, bits(first child)} {hi.xyz, bits(child count | VR_WIDE_PRIMS)};
  // children of a VR_WIDE_PRIMS node are the primitives at leaf positions first .. first + count - 1
  const float *wide;
  uint32_t wideTopFirst, wideTopCount; // the root's children (count carries VR_WIDE_PRIMS if they are primitives)
  uint32_t widePrimBase;
  uint32_t pqMaxFrontier, pqMaxCand;   // give-up thresholds of the packet query
  float sceneLo[3], sceneHi[3];        // root box of the BVH (every padded primitive box)
  float pqPad;                         // outward padding of the packet's box (float rounding of the clip)
  float nbDist;                        // neighbourhood radius = 2 x disk radius (rayGeometryDisk.hpp:191-192)
  int32_t geoD;                        // dimension of the geometry (2: z does not enter the neighbour test's boxes)
  // MODE 4 (scene resident in LDS): byte offsets of {pair nodes, primitive records, nbOff, nbIds, flux
  // accumulators (numData planes of numPrims), per-primitive sticking} inside the block's dynamic LDS, the number
  // of neighbour ids and the size of the whole copy
  uint32_t smallOff[6], smallNb, smallBytes;
  const uint32_t *nbOff;      // [numPrims+1], disk neighbourhood CSR (leaf order)
  const uint32_t *nbIds;      // leaf positions
  const float *primSticking;  // optional [numPrims] (leaf order) or nullptr
  const float *wallTable;     // 8 x {v0, e1, e2, Ng} = 96 floats
  unsigned long long *fluxAcc;   // [numData][replicas][accStride] leaf order, fixed point 2^-40
  uint32_t accStride, accMask;   // replica r of the accumulators starts at fluxAcc + r * accStride; r = blockIdx & accMask
  uint32_t numData, planeStride; // data label l (TracingData vector l) lives at fluxAcc + l * planeStride
  // particle plug-ins (vr_particles.hpp): run-time kind of the extended kernel instantiation
  int32_t particleKind;
  float reserved0, meanFreePath;
  int32_t useWdist;              // VIENNARAY_USE_WDIST crediting (rayTraceKernel.hpp:258-296)
  // sources other than SourceRandom: SourceGrid origins (raySourceGrid.hpp), or rays a host-side
  // Source callback produced (origin, direction, engine outputs it consumed)
  const float *gridPoints;
  uint32_t gridCount;
  float eeGrid;
  const float *hostOrg, *hostDir;
  const uint32_t *hostDraws;
  unsigned long long *counters;  // [8]
  unsigned long long *workCounter; // numQueues span cursors, VR_QUEUE_STRIDE words apart (a 128-byte line each)
  unsigned long long *rngScratch; // [waves][312][64]
  // ray stream of the current batch: VR_BIN_CAP record slots per sort bin, then the
  // overflow region (rays whose bin was full), all in one array of 32-byte records
  float *slotRec;                 // [(numBins * binCap + ovCap)] records of 32 B (48 B with the RNG cursors)
  uint32_t *binCount;             // [numBins + 1]; [numBins] counts the overflow rays
                                  // (nullptr: diagnostics, record i goes to slot i)
  const unsigned long long *idxList; // diagnostics: explicit ray indices (or nullptr)
  uint64_t batchFirst;            // global ray index of the batch's ray 0
  uint32_t batchCount;            // rays in this batch
  uint32_t ovCap;                 // capacity of the overflow region
  uint32_t binCap;                // record slots per sort bin
  uint32_t numBins;
  uint32_t seed;
  uint32_t numPrims;
  uint32_t maxReflections, maxBoundaryHits;
  uint32_t chunk;                 // rays per work-queue grab of one wave
  int32_t rayDir, firstDir, secondDir, minMax;
  float posNeg, ee, sticking;
  int32_t bc0, bc1;
  int32_t useBasis;
  float basis[9];                 // [b][component]
  // adjusted bounding box, as scalars (no dynamic indexing of kernel arguments)
  float srcCoord;                 // origin[rayDir]
  float lo1, hi1, lo2, hi2;       // extents along firstDir / secondDir
  float wallLoR, wallHiR;         // wall extent along rayDir, widened by a safety margin
  // sort-key binning (far-plane crossing cell)
  float keyCoord;                 // sort plane on rayDir: where most first hits are expected (host_sort_plane)
  float invExt1, invExt2;         // 1 / (hi - lo) along firstDir / secondDir (0 if degenerate)
  int32_t binT1, binT2;           // cells per axis
  int32_t binTiles;               // 8x8-cell tiles per row (3-D)
  uint32_t packetBudget;          // node visits a packet traversal may spend before giving up
  uint32_t walkPark;              // per-lane walk: leaves are tested when this % of the lanes under way are parked
  uint32_t walkExit;              // the per-lane walk of a round ends when fewer lanes than this are still walking
  uint32_t packetRatio;           // ... and it gives up when union visits > ratio x mean per-ray path
  uint32_t debugFlags;            // VR_DEBUG_FLAGS (timing experiments; 0 in production)
  // ---- round 3 (appended: the kernels' scalar loads of the fields above keep their offsets and alignment — the
  //      absorbing kernels sit at a fragile optimum of the register allocator) ----
  uint32_t numQueues;              // 1, or 8: one queue of sort bins per XCD (vr_trace.hip, refill)
  const float *recExtra;           // [batchCount] x {origin[rayDir], bits(k), s[k] lo, hi} for the sources whose origin plane or
                                   // draw count varies (tilted, grid, host rays); nullptr for the plain SourceRandom generator
  const float *hostWeights;        // Source::getInitialRayWeight(idx) of a host-callback source (nullptr: 1, raySource.hpp:18)
  float particleParams[8];         // vr_particle::params: the model's own parameters (ModelCtx::params)
  // Trace::setGlobalData (rayTrace.hpp:137-145): read-only vectors indexed by the ORIGINAL primitive id, and scalars
  const float *globalVec;          // [numGlobalVec][globalStride] or nullptr
  const float *globalScalars;      // [numGlobalScalars] or nullptr
  uint32_t numGlobalVec, globalStride, numGlobalScalars;
  // ---- round 4 (appended): scenes that are flat WITH RELIEF (ReliefParams below).  The generator sorts a ray by the
  //      cell of its PREDICTED first hit (two look-ups in the coarse field) and files the rays whose stretch through
  //      the local slab is long — grazing rays — in a second, coarser set of bins ("loose"), traced by the kernels
  //      for structured scenes; the flat-scene kernels then see waves whose packet query stays small.
  const float *reliefCoarse;       // [rcNy][rcNx] x {mid height, largest fine-tile thickness} or nullptr (plain binning)
  float rcLo1, rcLo2, rcInvT;      // coarse tile (ix, iy) of a point: (coord - rcLo) * rcInvT
  int32_t rcNx, rcNy;
  float reliefTravel;              // a ray is loose when thickness x tan(theta) exceeds this
  uint32_t looseCntBase;           // binCount[looseCntBase + b]: cursor of loose bin b; [looseCntBase + looseNumBins]: their overflow
  uint32_t looseSlotBase;          // first record slot of the loose bins (their overflow region follows them)
  uint32_t looseNumBins;
  int32_t looseT1, looseT2, looseTiles;
  int32_t reliefLookups;           // coarse look-ups of the generator's hit prediction (1 or 2)
  // Spill queue of the general relief kernel (MODE 6): a ray that would go on into the NEXT round — its follow-up segment
  // was not finished inside the round of its packet query — leaves the kernel as a 64-byte full-state record
  //   {org.xyz, weight} {rayDirection.xyz, bits(seed)} {bits(k), bits(reflections), bits(boundaryHits | back << 31), 0 (~0: no ray)}
  //   {s[k] lo, hi, s[k+156] lo, hi}
  // in blocks of 64 records (VR_SPILL_BLOCK, vr_trace.hip), one wave per block: a wave reserves a block with ONE atomic
  // on spillCount and fills it over its next rounds; the unused end of a wave's last block is marked empty
  // and the launch over the loose bins (MODE 7 = MODE 0 + these records) traces it to its end: the tight kernel's waves
  // then hold fresh, sorted primary rays only, whose packet queries stay small.
  float *spillRec;                 // [batchCount + 64 per wave] x 16 floats, or nullptr
  uint32_t *spillCount;            // [1] records reserved: a multiple of 64
  float reliefTanMax;              // ... and when tan(theta) exceeds this: its stretch through the SCENE box would span more tiles
                                   // than relief_clip should walk (a plane with one bump: thin tiles everywhere, a thick scene box)
  float pqMargin;                  // flat-scene kernels: a packet query searches the 64-ary tree with its box enlarged by this
                                   // much, and the frontier it finds serves the following rounds whose boxes lie inside (0: off)
};

// Relief field over the source plane (vr_setup.hip: relief_field_kernel): per fine tile the [lo, hi] range — along the
// source axis, padded — of every primitive whose (padded) box meets the tile.  A ray can only meet geometry inside a
// tile while its own height is within that range: relief_clip (vr_device.hpp) walks the tiles under a ray and returns
// the stretch that covers all such tiles, which replaces the clip to the SCENE box in the packet query of the
// flat-scene kernels (trace_kernel MODE 5 / 6).  The coarse field (<= 256 x 256, L2 resident for the generator's random
// look-ups) holds per coarse tile the mid height of its geometry and the largest thickness of its fine tiles.
struct ReliefParams {
  const float *prims;
  uint32_t n;
  int32_t geo, ax, a1, a2;
  float lo1, lo2, invTile, tile, pad;
  int32_t nx, ny;           // fine tiles
  int32_t k, cnx, cny;      // a coarse tile = k x k fine tiles
  float travel;             // statistics: TraceParams::reliefTravel (a ray is loose when thickness x tan(theta) exceeds it)
  float emptyMid;           // mid height of a coarse tile nothing reaches into
  uint32_t *rawLo, *rawHi;  // nx * ny ordered-uint minima / maxima
  float *fine;              // nx * ny x {lo, hi}
  float *coarse;            // cnx * cny x {mid, max fine thickness}
  uint32_t *stats;          // [0] non-empty coarse tiles, [1] sum over them of 4096 x the share of a cosine source's rays
                            //     that would be loose there: T^2 / (T^2 + travel^2), T the tile's largest fine thickness
};

// Height field over the source plane (vr_setup.hip: height_field_kernel): per tile of side `tile` the highest point —
// along the source axis, towards the source — of any primitive that reaches into the tile or one of its eight
// neighbours, plus a rounding margin.  A ray that is above it from tnear on, and rises above the whole scene before it
// has travelled one tile sideways, cannot meet the geometry (trace_kernel, "segments that rise clear").
struct HeightFieldParams {
  const float *prims;   // primitive records (disks: 2 x float4, triangles: 4 x float4)
  uint32_t n;
  int32_t geo, ax, a1, a2; // source axis and the two axes of the plane (a2 unused where ny == 1)
  float sign;           // +1: the source lies on the max side of the source axis
  float lo1, lo2, invTile, pad;
  int32_t nx, ny;
  uint32_t *raw;        // nx * ny ordered-uint maxima (zeroed)
  float *field;         // nx * ny: the 3 x 3 dilation of raw
};

// device-side scene setup (vr_setup.hip)
// Morton grid of the LBVH: the cell is the scene box's proportions, but at most VR_MORTON_ANISO : 1 (see morton_kernel)
constexpr float VR_MORTON_ANISO = 2.0f;
constexpr unsigned VR_NB_KEEP = 24; // ids the one-pass neighbourhood query keeps per primitive (more: the two-pass path)
struct SetupParams {
  // inputs (device copies of the caller's arrays)
  const float *disk4;     // n x {x,y,z,r}
  const float *normal3;   // n x 3 (disk normals / triangle unit normals)
  const float *points3;   // n x 3 caller's points (neighbourhood)
  const float *verts;     // nv x 3
  const uint32_t *tris;   // n x 3
  uint32_t n;
  int32_t geo, D;
  float nbDist;
  uint32_t leafMax;       // subtrees of <= leafMax primitives become one leaf (<= 15)
  int32_t orderAxis;      // children are ordered along this axis ...
  float orderSign;        // ... larger sign * coordinate (= nearer the source plane) first
  int32_t strictFence;    // fit_kernel: agent-scope release/acquire fences around the arrival counter (fallback)
  // work buffers
  float *box, *sbox;      // 6 per primitive: original order / sorted order
  uint32_t *bounds;       // 6 ordered-uint scene bounds
  unsigned long long *keysA, *keysB;
  uint32_t *valsA, *valsB, *sortTable;
  uint32_t *rangeLo, *rangeHi, *childL, *childR, *parentInt, *parentLeaf, *arrive;
  float *nodeBox;         // 6 per internal node
  uint32_t *subSize;      // per internal node: emitted subtree size | right-child-first << 31
  // outputs
  float *nodes;           // (2n-1) x 8 floats, build numbering (siblings adjacent)
  float *nodesPre;        // the emitted nodes in pre-order (source of the 16-byte nodes)
  float *prims;
  uint32_t *leafOfOrig, *order;
  uint32_t *nbOff, *nbIds;
  uint32_t *nbTmp;        // pass 2 (count AND keep): VR_NB_KEEP ids per primitive; nbTmp[n * VR_NB_KEEP] = overflow flag
  float *wide;            // 64-ary box tree (see TraceParams::wide), (n + n/64 + ...) x 8 floats
  float mortonAniso;      // cells of the Morton grid: at most this much finer along an axis than along the longest one
};

// particle kinds of the device registry (include/viennaray_amd.h: VR_PARTICLE_*)
enum { P_DIFFUSE = 0, P_SPECULAR = 1, P_CONED_COSINE = 2, P_DIFFUSE_COSINE = 3, P_COVERAGE_STICKING = 4,
       P_EXT = 2 /* template id of the extended kernel */,
       P_EXT_FULL = 3 /* ... with the coned-cosine model, WDIST crediting and mean-free-path scattering compiled in
                         (rare options that cost every particle of the instantiation registers: 157 spilled VGPRs
                         with them, 27 without) */ };

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

// vr_setup.hip — device-side scene setup: LBVH build and disk-neighbourhood CSR.
//
// The reference builds its acceleration structure inside the timed region
// (rtcJoinCommitScene, rayTraceKernel.hpp:91) and its point
// neighbourhood at setGeometry (rayGeometryDisk.hpp:191-192, recursive host
// vectors).  Here both are HIP kernels:
//
//   prim_box_kernel     primitive AABBs (oriented-disc extents r*sqrt(1-n_k^2),
//                       triangle min/max) + scene bounds (ordered-int atomics)
//   morton_kernel       63-bit Morton code of each box centre
//   radix sort          LSD, 8-bit digits, one wavefront per 1024-key tile,
//                       ballot-based stable ranking (no LDS scatter buffers)
//   karras_kernel       binary radix tree over the sorted codes (Karras 2012)
//   fit_kernel          bottom-up AABB fit with arrival counters
//   fit_kernel          ... + emitted subtree sizes and the source-side-first child order
//   finalize_kernel     traversal nodes {lo,link}{hi,escape} in pre-order: ranges of
//                       <= leafMax primitives collapse into leaves
//   quantize_nodes      16-byte nodes (16-bit conservative boxes) for per-lane traversal
//   pack_*_kernel       primitive records in leaf order
//   nb_count / nb_fill  neighbourhood = stackless BVH range query around every
//                       disc centre (per-axis |d| <= dist and |d|^2 <= dist^2,
//                       rayPointNeighborhood.hpp:287-298), written as CSR of leaf
//                       positions
#include <hip/hip_runtime.h>

#include <cfloat>

#include "vr_area.hpp"
#include "vr_kernels.hpp"
#include "vr_types.hpp"

namespace vr {

typedef unsigned long long u64;

// ---------------------------------------------------------------------------
// float <-> order-preserving uint (for atomicMin/Max on floats)
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned f2ord(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(unsigned u) {
  unsigned v = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
#ifdef __HIP_DEVICE_COMPILE__
  return __uint_as_float(v);
#else
  float f;
  __builtin_memcpy(&f, &v, 4);
  return f;
#endif
}

// bounds[0..2] = min (ordered), bounds[3..5] = max (ordered)
__global__ void prim_box_kernel(SetupParams s) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (i < s.n) {
    if (s.geo == 0) {
      const float4 d = reinterpret_cast<const float4 *>(s.disk4)[i];
      float nx = s.normal3[3 * (size_t)i], ny = s.normal3[3 * (size_t)i + 1], nz = s.normal3[3 * (size_t)i + 2];
      const float nn = sqrtf((nx * nx + ny * ny) + nz * nz);
      if (nn > 0.f) {
        nx /= nn;
        ny /= nn;
        nz /= nn;
      }
      const float c[3] = {d.x, d.y, d.z}, nv[3] = {nx, ny, nz};
      for (int k = 0; k < 3; ++k) {
        const float h = d.w * sqrtf(fmaxf(0.f, 1.f - nv[k] * nv[k])) * 1.0001f;
        lo[k] = c[k] - h;
        hi[k] = c[k] + h;
      }
    } else {
      const unsigned a = s.tris[3 * (size_t)i], b = s.tris[3 * (size_t)i + 1], c = s.tris[3 * (size_t)i + 2];
      for (int k = 0; k < 3; ++k) {
        const float v0 = s.verts[3 * (size_t)a + k], v1 = s.verts[3 * (size_t)b + k], v2 = s.verts[3 * (size_t)c + k];
        lo[k] = fminf(v0, fminf(v1, v2));
        hi[k] = fmaxf(v0, fmaxf(v1, v2));
      }
    }
    float *b = s.box + 6 * (size_t)i;
    for (int k = 0; k < 3; ++k) {
      b[k] = lo[k];
      b[3 + k] = hi[k];
    }
  }
  // wave reduce, block reduce through LDS, then six atomics per BLOCK (one per wave made
  // 10^5 same-address atomics on a 10^6-disk scene: 1 ms)
  __shared__ float red[6][4];
  for (int k = 0; k < 3; ++k) {
    float a = lo[k], b = hi[k];
    for (int off = 32; off > 0; off >>= 1) {
      a = fminf(a, __shfl_down(a, off, 64));
      b = fmaxf(b, __shfl_down(b, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      red[k][threadIdx.x >> 6] = a;
      red[3 + k][threadIdx.x >> 6] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    const unsigned nw = (blockDim.x + 63) >> 6;
    float v = red[k][0];
    for (unsigned w = 1; w < nw; ++w)
      v = k < 3 ? fminf(v, red[k][w]) : fmaxf(v, red[k][w]);
    if (k < 3)
      atomicMin(&s.bounds[k], f2ord(v));
    else
      atomicMax(&s.bounds[k], f2ord(v));
  }
}

__device__ __forceinline__ u64 spread21(u64 v) {
  v &= 0x1FFFFFull;
  v = (v | v << 32) & 0x1F00000000FFFFull;
  v = (v | v << 16) & 0x1F0000FF0000FFull;
  v = (v | v << 8) & 0x100F00F00F00F00Full;
  v = (v | v << 4) & 0x10C30C30C30C30C3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}

// pads the boxes (more than the rounding of the slab test) and computes codes
__global__ void morton_kernel(SetupParams s) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n)
    return;
  float slo[3], shi[3], scale = 0.f;
  for (int k = 0; k < 3; ++k) {
    slo[k] = ord2f(s.bounds[k]);
    shi[k] = ord2f(s.bounds[3 + k]);
    scale = fmaxf(scale, fmaxf(fabsf(slo[k]), fabsf(shi[k])));
  }
  const float pad = 4e-6f * fmaxf(scale, 1e-3f);
  float *b = s.box + 6 * (size_t)i;
  u64 q[3];
  // The grid's cell has the proportions of the scene box — but only up to mortonAniso : 1.  Scaled freely axis by
  // axis, a thin sheet (1000 x 1000 cells wide, one cell of relief) spends every third bit of the code on its relief:
  // the tree cuts it into contour bands whose boxes overlap everywhere in plan (measured on a 10^6-disk rippled sheet:
  // 110 pair visits and 18 leaf tests per ray, 33 ms for 3e7 rays; 9.7 ms with bounded proportions).  Cubes throughout
  // cost the 60 x 60 x 30 trench 5 %: cells twice as fine along the short (source) axis serve it better.
  const float extMax = fmaxf(fmaxf(shi[0] - slo[0], shi[1] - slo[1]), shi[2] - slo[2]);
  for (int k = 0; k < 3; ++k) {
    const float ext = fmaxf(shi[k] - slo[k], extMax / s.mortonAniso);
    const float inv = ext > 0.f ? 2097151.0f / ext : 0.f;
    float c = (0.5f * (b[k] + b[3 + k]) - slo[k]) * inv;
    c = fminf(fmaxf(c, 0.f), 2097151.0f);
    q[k] = (u64)c;
    b[k] -= pad;
    b[3 + k] += pad;
  }
  s.keysA[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
  s.valsA[i] = i;
}

// ---------------------------------------------------------------------------
// LSD radix sort, one wavefront per tile of 1024 keys
// ---------------------------------------------------------------------------
constexpr unsigned SORT_TILE = 1024;

__global__ __launch_bounds__(64) void sort_count_kernel(const u64 *keys, unsigned n, unsigned shift, unsigned tiles,
                                                        unsigned *table) {
  __shared__ unsigned hist[256];
  const unsigned lane = threadIdx.x, tile = blockIdx.x;
  for (unsigned k = lane; k < 256; k += 64)
    hist[k] = 0;
  __syncthreads();
  for (unsigned it = 0; it < SORT_TILE / 64; ++it) {
    const unsigned idx = tile * SORT_TILE + it * 64 + lane;
    if (idx < n)
      atomicAdd(&hist[(unsigned)(keys[idx] >> shift) & 255u], 1u);
  }
  __syncthreads();
  for (unsigned k = lane; k < 256; k += 64)
    table[k * tiles + tile] = hist[k];
}

__global__ __launch_bounds__(64) void sort_scatter_kernel(const u64 *keysIn, const unsigned *valsIn, u64 *keysOut,
                                                          unsigned *valsOut, unsigned n, unsigned shift,
                                                          unsigned tiles, const unsigned *table) {
  __shared__ unsigned offs[256];
  const unsigned lane = threadIdx.x, tile = blockIdx.x;
  for (unsigned k = lane; k < 256; k += 64)
    offs[k] = table[k * tiles + tile];
  __syncthreads();
  const u64 ltMask = (1ull << lane) - 1ull;
  for (unsigned it = 0; it < SORT_TILE / 64; ++it) {
    const unsigned idx = tile * SORT_TILE + it * 64 + lane;
    const bool valid = idx < n;
    u64 key = 0;
    unsigned val = 0, d = 0;
    if (valid) {
      key = keysIn[idx];
      val = valsIn[idx];
      d = (unsigned)(key >> shift) & 255u;
    }
    // lanes holding the same digit: AND of 8 per-bit ballots
    u64 peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const u64 m = __ballot(valid && ((d >> b) & 1u));
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    if (valid) {
      const unsigned rank = __popcll(peers & ltMask);
      const unsigned base = offs[d];
      keysOut[base + rank] = key;
      valsOut[base + rank] = val;
      if (rank == 0)
        offs[d] = base + __popcll(peers); // in-order LDS: every peer has read `base`
    }
  }
}

// ---------------------------------------------------------------------------
// Karras binary radix tree
// ---------------------------------------------------------------------------
constexpr unsigned CHILD_LEAF = 0x80000000u; // child is the singleton leaf of that sorted position

__device__ __forceinline__ int delta(const u64 *code, int n, int i, int j) {
  if (j < 0 || j >= n)
    return -1;
  const u64 a = code[i], b = code[j];
  if (a == b)
    return 64 + __clz((unsigned)i ^ (unsigned)j);
  return __clzll((long long)(a ^ b));
}

__global__ void karras_kernel(SetupParams s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = (int)s.n;
  if (i >= n - 1)
    return;
  const u64 *code = s.keysA;
  const int d = (delta(code, n, i, i + 1) - delta(code, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(code, n, i, i - d);
  int lmax = 2;
  while (delta(code, n, i, i + lmax * d) > dmin)
    lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(code, n, i, i + (l + t) * d) > dmin)
      l += t;
  const int j = i + l * d;
  const int dnode = delta(code, n, i, j);
  int sp = 0;
  int t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(code, n, i, i + (sp + t) * d) > dnode)
      sp += t;
  } while (t > 1);
  const int gamma = i + sp * d + (d < 0 ? d : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const unsigned left = (lo == gamma) ? (CHILD_LEAF | (unsigned)gamma) : (unsigned)gamma;
  const unsigned right = (hi == gamma + 1) ? (CHILD_LEAF | (unsigned)(gamma + 1)) : (unsigned)(gamma + 1);
  s.rangeLo[i] = (unsigned)lo;
  s.rangeHi[i] = (unsigned)hi;
  s.childL[i] = left;
  s.childR[i] = right;
  // parent links; bit 31 of the stored parent marks "I am the right child"
  if (left & CHILD_LEAF)
    s.parentLeaf[left & ~CHILD_LEAF] = (unsigned)i;
  else
    s.parentInt[left] = (unsigned)i;
  if (right & CHILD_LEAF)
    s.parentLeaf[right & ~CHILD_LEAF] = (unsigned)i | 0x80000000u;
  else
    s.parentInt[right] = (unsigned)i | 0x80000000u;
}

// bottom-up AABB fit: thread = sorted position; the second arriver at a node fits it.
// The same pass computes what the pre-order layout needs: the number of traversal
// nodes each subtree EMITS (a range of <= leafMax primitives collapses into one leaf)
// and which child the traversal should visit first (bit 31): the one whose box centre
// lies closer to the source plane, so that primary rays meet their first hit early
// and the escape-link walk culls everything behind it.
__global__ void fit_kernel(SetupParams s) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= s.n || s.n < 2)
    return;
  unsigned p = s.parentLeaf[q] & 0x7FFFFFFFu;
  for (;;) {
    // Hand-over between the two arrivers of a node.  Everything a fitter publishes (box, subtree
    // size) is written with agent-scope atomic stores (`global_store ... sc1`: write-through, not
    // kept in this XCD's L2) and read with agent-scope atomic loads (`sc1`: served below the
    // reader's L1), which are coherent across the XCDs by themselves; what remains is ORDER: every
    // such store of this lane must have been acknowledged before the arrival counter is bumped.
    // That is the explicit `s_waitcnt vmcnt(0)` below (inline asm: no compiler pass can drop or
    // move it; tests/test_isa_contracts.py checks the emitted ISA).  It replaces an agent-scope
    // release fence, whose `buffer_wbl2` write-back of the whole L2 made this kernel 3.0 ms instead
    // of 0.2 ms on 10^6 disks and which these sc1 stores do not need.  s.strictFence selects the
    // textbook form (agent-scope release / acquire fences) instead: build_scene() re-runs the fit
    // that way should launch_bvh_check ever report a violation.
    if (s.strictFence)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned arrived = __hip_atomic_fetch_add(&s.arrive[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived == 0u)
      return; // first arriver: the sibling will come
    if (s.strictFence)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const unsigned L = s.childL[p], R = s.childR[p];
    const float *a = (L & CHILD_LEAF) ? s.sbox + 6 * (size_t)(L & ~CHILD_LEAF) : s.nodeBox + 6 * (size_t)L;
    const float *b = (R & CHILD_LEAF) ? s.sbox + 6 * (size_t)(R & ~CHILD_LEAF) : s.nodeBox + 6 * (size_t)R;
    float *o = s.nodeBox + 6 * (size_t)p;
    float ba[6], bb[6];
    for (int k = 0; k < 6; ++k) {
      ba[k] = __hip_atomic_load(&a[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bb[k] = __hip_atomic_load(&b[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int k = 0; k < 3; ++k) {
      __hip_atomic_store(&o[k], fminf(ba[k], bb[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&o[3 + k], fmaxf(ba[3 + k], bb[3 + k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned sl =
        (L & CHILD_LEAF) ? 1u : (__hip_atomic_load(&s.subSize[L], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x7FFFFFFFu);
    const unsigned sr =
        (R & CHILD_LEAF) ? 1u : (__hip_atomic_load(&s.subSize[R], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x7FFFFFFFu);
    const unsigned cnt = s.rangeHi[p] - s.rangeLo[p] + 1u;
    const unsigned size = cnt <= s.leafMax ? 1u : 1u + sl + sr;
    const int ax = s.orderAxis;
    const bool rightFirst = s.orderSign * ((bb[ax] + bb[3 + ax]) - (ba[ax] + ba[3 + ax])) > 0.f;
    __hip_atomic_store(&s.subSize[p], size | (rightFirst ? 0x80000000u : 0u), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    if (p == 0)
      return;
    p = s.parentInt[p] & 0x7FFFFFFFu;
  }
}

// Traversal nodes, written twice from the (child-ordered, leaf-collapsed) tree:
//  * s.nodes   : build numbering (internal i -> i, singleton leaf of sorted position q ->
//                n-1+q): the two children of a node are NEIGHBOURS in memory, which is what
//                the packet traversal's scalar fetches like (a missed first child is
//                followed by its sibling, same 64-byte line).  Explicit link + escape.
//  * s.nodesPre: PRE-ORDER: the first child of an internal node is the next node, the
//                escape of any node is the node after its subtree.  Source of the 16-byte
//                nodes of the per-lane walk, which keep ONE link word thanks to that.
// Thread t = node of the build numbering; a node below a collapsed range is not emitted.
// One walk to the root yields both the pre-order index (every ancestor contributes 1,
// plus the size of the sibling subtree where the path is the second child) and the
// escape in build numbering (sibling of the nearest ancestor-or-self that is a first
// child).
__device__ __forceinline__ unsigned sub_size(const SetupParams &s, unsigned child) {
  return (child & CHILD_LEAF) ? 1u : (s.subSize[child] & 0x7FFFFFFFu);
}
__device__ __forceinline__ unsigned node_of_child(const SetupParams &s, unsigned c) {
  return (c & CHILD_LEAF) ? (s.n - 1u) + (c & ~CHILD_LEAF) : c;
}

__global__ void finalize_kernel(SetupParams s) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned n = s.n;
  float4 *nodes = reinterpret_cast<float4 *>(s.nodes);
  float4 *nodesPre = reinterpret_cast<float4 *>(s.nodesPre);
  if (n == 1) {
    if (t == 0) {
      const float *b = s.sbox;
      const float4 a0 = make_float4(b[0], b[1], b[2], __uint_as_float(VR_LEAF | (1u << 27) | 0u));
      const float4 a1 = make_float4(b[3], b[4], b[5], __uint_as_float(VR_END));
      nodes[0] = nodesPre[0] = a0;
      nodes[1] = nodesPre[1] = a1;
      s.subSize[0] = 1u;
    }
    return;
  }
  if (t >= 2 * n - 1)
    return;
  const bool internal = t < n - 1;
  const unsigned q = internal ? 0u : t - (n - 1);
  unsigned pw = 0; // parent index | "I am the right child"
  bool haveParent = true;
  if (internal) {
    if (t == 0)
      haveParent = false;
    else
      pw = s.parentInt[t];
  } else {
    pw = s.parentLeaf[q];
  }
  if (haveParent) { // not emitted below a collapsed range (ancestors' ranges only grow)
    const unsigned pp = pw & 0x7FFFFFFFu;
    if (s.rangeHi[pp] - s.rangeLo[pp] + 1u <= s.leafMax)
      return;
  }
  unsigned pre = 0, escBuild = VR_END;
  bool escFound = false;
  while (haveParent) {
    const unsigned pp = pw & 0x7FFFFFFFu;
    const bool amRight = (pw & 0x80000000u) != 0u;
    const bool rightFirst = (s.subSize[pp] & 0x80000000u) != 0u;
    pre += 1u;
    if (amRight != rightFirst) { // second child: the first child's subtree comes before
      pre += sub_size(s, rightFirst ? s.childR[pp] : s.childL[pp]);
    } else if (!escFound) {      // first child: the walk continues with the sibling
      escBuild = node_of_child(s, rightFirst ? s.childL[pp] : s.childR[pp]);
      escFound = true;
    }
    if (pp == 0)
      break;
    pw = s.parentInt[pp];
  }
  const unsigned total = s.subSize[0] & 0x7FFFFFFFu;
  unsigned linkPre, linkBuild, mySize;
  const float *b;
  if (internal) {
    const unsigned lo = s.rangeLo[t], cnt = s.rangeHi[t] - lo + 1u;
    const unsigned w = s.subSize[t];
    mySize = w & 0x7FFFFFFFu;
    if (cnt <= s.leafMax) {
      linkPre = linkBuild = VR_LEAF | (cnt << 27) | lo;
    } else {
      linkPre = pre + 1u;
      linkBuild = node_of_child(s, (w & 0x80000000u) ? s.childR[t] : s.childL[t]);
    }
    b = s.nodeBox + 6 * (size_t)t;
  } else {
    mySize = 1u;
    linkPre = linkBuild = VR_LEAF | (1u << 27) | q;
    b = s.sbox + 6 * (size_t)q;
  }
  const unsigned escPre = pre + mySize >= total ? VR_END : pre + mySize;
  nodes[2 * (size_t)t] = make_float4(b[0], b[1], b[2], __uint_as_float(linkBuild));
  nodes[2 * (size_t)t + 1] = make_float4(b[3], b[4], b[5], __uint_as_float(escBuild));
  nodesPre[2 * (size_t)pre] = make_float4(b[0], b[1], b[2], __uint_as_float(linkPre));
  nodesPre[2 * (size_t)pre + 1] = make_float4(b[3], b[4], b[5], __uint_as_float(escPre));
}

// 16-byte form of the same nodes for the per-lane traversal: the box on a 16-bit grid of
// the scene (rounded outwards and widened by one cell, so the test stays conservative
// under the float rounding of the quantised-space slab test) + ONE link word:
//   internal: escape index (VR_QEND when none)        [first child = this + 1]
//   leaf    : VR_LEAF | cnt << 27 | first primitive   [escape      = this + 1]
__global__ void quantize_nodes_kernel(const float4 *nodes, unsigned numNodes, float bx, float by, float bz, float sx,
                                      float sy, float sz, uint4 *qnodes) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= numNodes)
    return;
  const float4 a = nodes[2 * (size_t)i], b = nodes[2 * (size_t)i + 1];
  const float base[3] = {bx, by, bz}, sc[3] = {sx, sy, sz};
  const float lo[3] = {a.x, a.y, a.z}, hi[3] = {b.x, b.y, b.z};
  unsigned ql[3], qh[3];
  for (int k = 0; k < 3; ++k) {
    const float l = floorf((lo[k] - base[k]) * sc[k]) - 1.f;
    const float h = ceilf((hi[k] - base[k]) * sc[k]) + 1.f;
    ql[k] = (unsigned)fminf(fmaxf(l, 0.f), 65535.f);
    qh[k] = (unsigned)fminf(fmaxf(h, 0.f), 65535.f);
  }
  const unsigned link = __float_as_uint(a.w), esc = __float_as_uint(b.w);
  const unsigned w = (link & VR_LEAF) ? link : (esc == VR_END ? VR_QEND : esc);
  qnodes[i] = make_uint4(ql[0] | (ql[1] << 16), ql[2] | (qh[0] << 16), qh[1] | (qh[2] << 16), w);
}

// PAIR nodes for the ordered per-lane walk (vr_device.hpp: pair_walk_lanes): entry i (i = pre-order index of
// an internal node) holds BOTH children — {box(c0), link(c0)} {box(c1), link(c1)}, 32 bytes in one cache
// line — so one visit decides both, descends into the nearer one and defers the other.
//   link: leaf -> its leaf word (VR_LEAF | cnt << 27 | first), internal -> its pre-order index
// c0 = i + 1, c1 = the node after c0's subtree.  A scene that is one leaf gets the pair {that leaf, nothing}.
__global__ void pair_nodes_kernel(const uint4 *qnodes, unsigned numNodes, uint4 *pnodes) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= numNodes)
    return;
  const uint4 me = qnodes[i];
  if (me.w & VR_LEAF) {
    if (i == 0u) { // the whole scene is one leaf
      pnodes[0] = me;
      pnodes[1] = make_uint4(0xFFFFFFFFu, 0x0000FFFFu, 0u, VR_LEAF); // lo = 65535 > hi = 0: never hit; empty leaf
    }
    return;
  }
  const unsigned c0 = i + 1u;
  const uint4 a = qnodes[c0];
  const unsigned c1 = (a.w & VR_LEAF) ? c0 + 1u : a.w;
  const uint4 b = qnodes[c1];
  pnodes[2 * (size_t)i] = make_uint4(a.x, a.y, a.z, (a.w & VR_LEAF) ? a.w : c0);
  pnodes[2 * (size_t)i + 1] = make_uint4(b.x, b.y, b.z, (b.w & VR_LEAF) ? b.w : c1);
}

// sorted boxes, leafOfOrig, primitive records in leaf order
__global__ void pack_kernel(SetupParams s) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= s.n)
    return;
  const unsigned o = s.valsA[q];
  s.leafOfOrig[o] = q;
  s.order[q] = o;
  for (int k = 0; k < 6; ++k)
    s.sbox[6 * (size_t)q + k] = s.box[6 * (size_t)o + k];
  float4 *pr = reinterpret_cast<float4 *>(s.prims);
  if (s.geo == 0) {
    pr[2 * (size_t)q] = reinterpret_cast<const float4 *>(s.disk4)[o];
    pr[2 * (size_t)q + 1] = make_float4(s.normal3[3 * (size_t)o], s.normal3[3 * (size_t)o + 1],
                                        s.normal3[3 * (size_t)o + 2], __uint_as_float(o));
  } else {
    const float *a = s.verts + 3 * (size_t)s.tris[3 * (size_t)o];
    const float *b = s.verts + 3 * (size_t)s.tris[3 * (size_t)o + 1];
    const float *c = s.verts + 3 * (size_t)s.tris[3 * (size_t)o + 2];
    const float e1[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    const float e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    // Ng = cross(e2, e1), separate multiplies and subtract (no contraction)
    const float Ng[3] = {e2[1] * e1[2] - e2[2] * e1[1], e2[2] * e1[0] - e2[0] * e1[2], e2[0] * e1[1] - e2[1] * e1[0]};
    const float *nn = s.normal3 + 3 * (size_t)o;
    pr[4 * (size_t)q] = make_float4(a[0], a[1], a[2], __uint_as_float(o));
    pr[4 * (size_t)q + 1] = make_float4(e1[0], e1[1], e1[2], nn[0]);
    pr[4 * (size_t)q + 2] = make_float4(e2[0], e2[1], e2[2], nn[1]);
    pr[4 * (size_t)q + 3] = make_float4(Ng[0], Ng[1], Ng[2], nn[2]);
  }
}

// ---------------------------------------------------------------------------
// neighbourhood by BVH range query (disks).  Pass 0 counts, pass 1 fills.
// The caller's points (not the float4 disc buffer) define the distance test,
// like the reference (rayGeometryDisk.hpp:191: `init<Dim>(points, ...)`).
// ---------------------------------------------------------------------------
template <int PASS> __global__ void nb_kernel(SetupParams s) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= s.n)
    return;
  const float4 *nodes = reinterpret_cast<const float4 *>(s.nodes);
  const unsigned me = s.order[q];
  const float px = s.points3[3 * (size_t)me], py = s.points3[3 * (size_t)me + 1], pz = s.points3[3 * (size_t)me + 2];
  const float dist = s.nbDist, dist2 = dist * dist;
  // query box: every disc whose centre is within `dist` has that centre inside it,
  // and a disc's box contains its centre
  const float qlo[3] = {px - dist, py - dist, s.D == 2 ? -FLT_MAX : pz - dist};
  const float qhi[3] = {px + dist, py + dist, s.D == 2 ? FLT_MAX : pz + dist};
  unsigned count = 0;
  const unsigned base = PASS == 1 ? s.nbOff[q] : 0u;
  unsigned node = 0;
  while (node != VR_END) {
    const float4 a = nodes[2 * (size_t)node], b = nodes[2 * (size_t)node + 1];
    const unsigned link = __float_as_uint(a.w), esc = __float_as_uint(b.w);
    const bool hit = a.x <= qhi[0] && b.x >= qlo[0] && a.y <= qhi[1] && b.y >= qlo[1] && a.z <= qhi[2] && b.z >= qlo[2];
    if (hit) {
      if (link & VR_LEAF) {
        const unsigned first = link & VR_LEAF_FIRST_MASK, cnt = (link >> 27) & 15u;
        for (unsigned k = 0; k < cnt; ++k) {
          const unsigned r = first + k;
          if (r == q)
            continue;
          const unsigned o = s.order[r];
          const float dx = px - s.points3[3 * (size_t)o], dy = py - s.points3[3 * (size_t)o + 1],
                      dz = pz - s.points3[3 * (size_t)o + 2];
          bool near = fabsf(dx) <= dist && fabsf(dy) <= dist && (s.D == 2 || fabsf(dz) <= dist);
          near = near && ((dx * dx + dy * dy) + dz * dz) <= dist2;
          if (near) {
            if (PASS == 1)
              s.nbIds[base + count] = r;
            else if (PASS == 2 && count < VR_NB_KEEP)
              s.nbTmp[(size_t)q * VR_NB_KEEP + count] = r;
            ++count;
          }
        }
        node = esc;
      } else {
        node = link;
      }
    } else {
      node = esc;
    }
  }
  if (PASS != 1)
    s.nbOff[q] = count;
  if (PASS == 2 && count > VR_NB_KEEP)
    s.nbTmp[(size_t)s.n * VR_NB_KEEP] = 1u; // (more neighbours than kept: the caller falls back to count + fill)
}

// pass 2's lists, packed behind the scanned offsets (the query itself — a range walk of the BVH per primitive, the most
// expensive kernel of a scene build — runs once instead of twice)
__global__ void nb_compact_kernel(SetupParams s) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= s.n)
    return;
  const unsigned b = s.nbOff[q], e = s.nbOff[q + 1];
  for (unsigned j = b; j < e; ++j)
    s.nbIds[j] = s.nbTmp[(size_t)q * VR_NB_KEEP + (j - b)];
}

// ---------------------------------------------------------------------------
hipError_t launch_setup_bvh(const SetupParams &sp, unsigned *scanTmp, hipStream_t st) {
  SetupParams s = sp;
  const unsigned n = s.n;
  if (n == 0)
    return hipSuccess;
  const unsigned g256 = (n + 255) / 256;
  // bounds init: min = ord(+inf side), max = ord(-inf side)
  const unsigned initB[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  hipError_t e = hipMemcpyAsync(s.bounds, initB, sizeof(initB), hipMemcpyHostToDevice, st);
  if (e != hipSuccess)
    return e;
  hipLaunchKernelGGL(prim_box_kernel, dim3(g256), dim3(256), 0, st, s);
  hipLaunchKernelGGL(morton_kernel, dim3(g256), dim3(256), 0, st, s);
  // radix sort (keysA, valsA) -> ping-pong with (keysB, valsB), 8 passes end in A
  const unsigned tiles = (n + SORT_TILE - 1) / SORT_TILE;
  u64 *kin = s.keysA, *kout = s.keysB;
  unsigned *vin = s.valsA, *vout = s.valsB;
  for (unsigned pass = 0; pass < 8; ++pass) {
    hipLaunchKernelGGL(sort_count_kernel, dim3(tiles), dim3(64), 0, st, kin, n, pass * 8, tiles, s.sortTable);
    e = launch_scan(s.sortTable, 256u * tiles, scanTmp, st);
    if (e != hipSuccess)
      return e;
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(tiles), dim3(64), 0, st, kin, vin, kout, vout, n, pass * 8, tiles,
                       s.sortTable);
    u64 *tk = kin;
    kin = kout;
    kout = tk;
    unsigned *tv = vin;
    vin = vout;
    vout = tv;
  }
  // (8 passes: result is back in A)
  hipLaunchKernelGGL(pack_kernel, dim3(g256), dim3(256), 0, st, s);
  if (n > 1)
    hipLaunchKernelGGL(karras_kernel, dim3((n - 1 + 255) / 256), dim3(256), 0, st, s);
  return launch_fit_bvh(s, st);
}

// bottom-up fit + traversal-node emission over the resident radix tree (also the re-run with
// s.strictFence after a failed launch_bvh_check)
hipError_t launch_fit_bvh(const SetupParams &s, hipStream_t st) {
  const unsigned n = s.n;
  if (n == 0)
    return hipSuccess;
  if (n > 1) {
    hipError_t e = hipMemsetAsync(s.arrive, 0, (size_t)(n - 1) * 4, st);
    if (e != hipSuccess)
      return e;
    hipLaunchKernelGGL(fit_kernel, dim3((n + 255) / 256), dim3(256), 0, st, s);
  }
  hipLaunchKernelGGL(finalize_kernel, dim3((2 * n - 1 + 255) / 256), dim3(256), 0, st, s);
  return hipGetLastError();
}

// validation of the bottom-up fit (the hand-over in fit_kernel is the one place where the
// build relies on cross-workgroup ordering): every internal node's box must be exactly the
// union of its children's, its emitted size consistent.  Counts violations.
__global__ void bvh_check_kernel(SetupParams s, unsigned *bad) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (s.n < 2 || p >= s.n - 1)
    return;
  const unsigned L = s.childL[p], R = s.childR[p];
  const float *a = (L & CHILD_LEAF) ? s.sbox + 6 * (size_t)(L & ~CHILD_LEAF) : s.nodeBox + 6 * (size_t)L;
  const float *b = (R & CHILD_LEAF) ? s.sbox + 6 * (size_t)(R & ~CHILD_LEAF) : s.nodeBox + 6 * (size_t)R;
  const float *o = s.nodeBox + 6 * (size_t)p;
  bool ok = true;
  for (int k = 0; k < 3; ++k)
    ok = ok && o[k] == fminf(a[k], b[k]) && o[3 + k] == fmaxf(a[3 + k], b[3 + k]);
  const unsigned sl = (L & CHILD_LEAF) ? 1u : (s.subSize[L] & 0x7FFFFFFFu);
  const unsigned sr = (R & CHILD_LEAF) ? 1u : (s.subSize[R] & 0x7FFFFFFFu);
  const unsigned cnt = s.rangeHi[p] - s.rangeLo[p] + 1u;
  ok = ok && (s.subSize[p] & 0x7FFFFFFFu) == (cnt <= s.leafMax ? 1u : 1u + sl + sr);
  if (!ok)
    atomicAdd(bad, 1u);
}

hipError_t launch_bvh_check(const SetupParams &s, unsigned *bad, hipStream_t st) {
  if (s.n < 2)
    return hipSuccess;
  hipLaunchKernelGGL(bvh_check_kernel, dim3((s.n - 1 + 255) / 256), dim3(256), 0, st, s, bad);
  return hipGetLastError();
}

// smoothFlux (rayTraceDisk.hpp:146-193) on the device neighbourhood: weighted average over the
// neighbours whose normal points the same way, weights = normal dot products.  The sum runs
// over the neighbours in ASCENDING ORIGINAL ID like the host path (float addition is ordered),
// so both give the same bits: each thread sorts its (short) list first.  A list longer than the
// local buffer raises *overflow and the host path takes over.
constexpr unsigned SMOOTH_MAX = 48;
__global__ void smooth_flux_kernel(const float *fluxIn, float *fluxOut, const float *normal3, const uint32_t *nbOff,
                                   const uint32_t *nbIds, const uint32_t *order, const uint32_t *leafOfOrig,
                                   unsigned n, unsigned *overflow) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const unsigned q = leafOfOrig[i];
  const unsigned b = nbOff[q], e = nbOff[q + 1];
  if (e - b > SMOOTH_MAX) {
    atomicAdd(overflow, 1u);
    fluxOut[i] = fluxIn[i];
    return;
  }
  unsigned ids[SMOOTH_MAX];
  unsigned cnt = 0;
  for (unsigned j = b; j < e; ++j) { // insertion sort by original id
    const unsigned o = order[nbIds[j]];
    unsigned k = cnt++;
    while (k > 0 && ids[k - 1] > o) {
      ids[k] = ids[k - 1];
      --k;
    }
    ids[k] = o;
  }
  const float nx = normal3[3 * (size_t)i], ny = normal3[3 * (size_t)i + 1], nz = normal3[3 * (size_t)i + 2];
  float vv = fluxIn[i], sum = 1.f;
  for (unsigned k = 0; k < cnt; ++k) {
    const unsigned o = ids[k];
    const float w = (nx * normal3[3 * (size_t)o] + ny * normal3[3 * (size_t)o + 1]) + nz * normal3[3 * (size_t)o + 2];
    if (w > 0.f) {
      vv += fluxIn[o] * w;
      sum += w;
    }
  }
  fluxOut[i] = vv / sum;
}

// smoothFlux(flux, k > 1) (rayTraceDisk.hpp:146-193: a PointNeighborhood of radius k * 2 r, built for the call): the
// neighbourhood is not stored — every thread runs the range query of nb_kernel with the wider radius over the resident
// BVH, keeps the ids it finds (ascending original id, like the host path: float addition is ordered) and averages.
constexpr unsigned SMOOTH_WIDE_MAX = 128;
__global__ void smooth_wide_kernel(const float *fluxIn, float *fluxOut, const float *normal3, SetupParams s, float dist,
                                   unsigned *overflow) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n)
    return;
  const float4 *nodes = reinterpret_cast<const float4 *>(s.nodes);
  const float px = s.points3[3 * (size_t)i], py = s.points3[3 * (size_t)i + 1], pz = s.points3[3 * (size_t)i + 2];
  const float dist2 = dist * dist;
  const float qlo[3] = {px - dist, py - dist, s.D == 2 ? -FLT_MAX : pz - dist};
  const float qhi[3] = {px + dist, py + dist, s.D == 2 ? FLT_MAX : pz + dist};
  unsigned ids[SMOOTH_WIDE_MAX];
  unsigned cnt = 0;
  bool over = false;
  unsigned node = 0;
  while (node != VR_END) {
    const float4 a = nodes[2 * (size_t)node], b = nodes[2 * (size_t)node + 1];
    const unsigned link = __float_as_uint(a.w), esc = __float_as_uint(b.w);
    // (the boxes are the discs' boxes: a disc's box contains its centre, and every centre within `dist` lies in the query box)
    const bool hit = a.x <= qhi[0] && b.x >= qlo[0] && a.y <= qhi[1] && b.y >= qlo[1] && a.z <= qhi[2] && b.z >= qlo[2];
    if (hit && (link & VR_LEAF)) {
      const unsigned first = link & VR_LEAF_FIRST_MASK, num = (link >> 27) & 15u;
      for (unsigned k = 0; k < num; ++k) {
        const unsigned o = s.order[first + k];
        if (o == i)
          continue;
        const float dx = px - s.points3[3 * (size_t)o], dy = py - s.points3[3 * (size_t)o + 1],
                    dz = pz - s.points3[3 * (size_t)o + 2];
        bool near = fabsf(dx) <= dist && fabsf(dy) <= dist && (s.D == 2 || fabsf(dz) <= dist);
        near = near && ((dx * dx + dy * dy) + dz * dz) <= dist2;
        if (!near)
          continue;
        if (cnt == SMOOTH_WIDE_MAX) {
          over = true;
          continue;
        }
        unsigned k2 = cnt++;
        while (k2 > 0 && ids[k2 - 1] > o) { // insertion sort by original id
          ids[k2] = ids[k2 - 1];
          --k2;
        }
        ids[k2] = o;
      }
      node = esc;
    } else {
      node = hit ? link : esc;
    }
  }
  if (over) {
    atomicAdd(overflow, 1u);
    fluxOut[i] = fluxIn[i];
    return;
  }
  const float nx = normal3[3 * (size_t)i], ny = normal3[3 * (size_t)i + 1], nz = normal3[3 * (size_t)i + 2];
  float vv = fluxIn[i], sum = 1.f;
  for (unsigned k = 0; k < cnt; ++k) {
    const unsigned o = ids[k];
    const float w = (nx * normal3[3 * (size_t)o] + ny * normal3[3 * (size_t)o + 1]) + nz * normal3[3 * (size_t)o + 2];
    if (w > 0.f) {
      vv += fluxIn[o] * w;
      sum += w;
    }
  }
  fluxOut[i] = vv / sum;
}

hipError_t launch_smooth_wide(const float *fluxIn, float *fluxOut, const float *normal3, const SetupParams &s, float dist,
                              unsigned *overflow, hipStream_t st) {
  if (s.n == 0)
    return hipSuccess;
  hipLaunchKernelGGL(smooth_wide_kernel, dim3((s.n + 127) / 128), dim3(128), 0, st, fluxIn, fluxOut, normal3, s, dist, overflow);
  return hipGetLastError();
}

hipError_t launch_smooth_flux(const float *fluxIn, float *fluxOut, const float *normal3, const uint32_t *nbOff,
                              const uint32_t *nbIds, const uint32_t *order, const uint32_t *leafOfOrig, unsigned n,
                              unsigned *overflow, hipStream_t st) {
  if (n == 0)
    return hipSuccess;
  hipLaunchKernelGGL(smooth_flux_kernel, dim3((n + 127) / 128), dim3(128), 0, st, fluxIn, fluxOut, normal3, nbOff, nbIds,
                     order, leafOfOrig, n, overflow);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Post-processing on the device: exposed disk areas
// (computeDiskAreas + DiskBoundingBoxXYIntersector, vr_area.hpp) and normalizeFlux
// (rayTraceDisk.hpp:103-142, rayTraceTriangle.hpp:92-130; the reference's own GPU path:
// gpu/kernels/normKernels.cu:58-74).  One thread per primitive, caller's order.
// ---------------------------------------------------------------------------
__global__ void disk_areas_kernel(const float *disk4, const float *normal3, unsigned n, AreaParams p, float *out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const float4 d = reinterpret_cast<const float4 *>(disk4)[i];
  const float disk[4] = {d.x, d.y, d.z, d.w};
  const float nrm[3] = {normal3[3 * (size_t)i], normal3[3 * (size_t)i + 1], normal3[3 * (size_t)i + 2]};
  out[i] = disk_exposed_area(p, disk, nrm);
}

hipError_t launch_disk_areas(const float *disk4, const float *normal3, unsigned n, const AreaParams &p, float *out,
                             hipStream_t st) {
  if (n == 0)
    return hipSuccess;
  hipLaunchKernelGGL(disk_areas_kernel, dim3((n + 127) / 128), dim3(128), 0, st, disk4, normal3, n, p, out);
  return hipGetLastError();
}

// raw flux as the reference's float vector: float(acc * 2^-40) (acc: int64 fixed point)
__global__ void flux_from_acc_kernel(const unsigned long long *acc, unsigned n, float *flux) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    flux[i] = (float)((double)acc[i] * 9.094947017729282379150390625e-13); // 2^-40
}

// std::max_element over the flux (ordered-uint atomicMax; NaNs never win a `<` in the reference either)
__global__ void flux_max_kernel(const float *flux, unsigned n, unsigned *maxOrd) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? flux[i] : -FLT_MAX;
  if (!(v == v))
    v = -FLT_MAX;
  for (int off = 32; off > 0; off >>= 1)
    v = fmaxf(v, __shfl_down(v, off, 64));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0)
    red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(maxOrd, f2ord(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

// SOURCE: flux *= normFactor / area (all float).  MAX, disks: flux *= (pi r^2 / area) / max in
// double like the reference's `totalDiskArea` (a double); triangles: flux /= max * area (float).
template <int NORM, int GEO>
__global__ void normalize_flux_kernel(float *flux, const float *area, unsigned n, float normFactor, double totalDiskArea,
                                      const unsigned *maxOrd) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  if (NORM == 0) {
    flux[i] *= normFactor / area[i];
  } else {
    const float maxv = ord2f(*maxOrd);
    if (GEO == 0)
      flux[i] = (float)((double)flux[i] * ((totalDiskArea / (double)area[i]) / (double)maxv));
    else
      flux[i] /= maxv * area[i];
  }
}

hipError_t launch_flux_from_acc(const unsigned long long *acc, unsigned n, float *flux, hipStream_t st) {
  if (n == 0)
    return hipSuccess;
  hipLaunchKernelGGL(flux_from_acc_kernel, dim3((n + 255) / 256), dim3(256), 0, st, acc, n, flux);
  return hipGetLastError();
}

hipError_t launch_normalize_flux(float *flux, const float *area, unsigned n, int geo, int normType, float normFactor,
                                 double totalDiskArea, unsigned *maxOrd, hipStream_t st) {
  if (n == 0)
    return hipSuccess;
  const dim3 g((n + 255) / 256), b(256);
  if (normType == 0) {
    hipLaunchKernelGGL((normalize_flux_kernel<0, 0>), g, b, 0, st, flux, area, n, normFactor, totalDiskArea, maxOrd);
  } else {
    hipError_t e = hipMemsetAsync(maxOrd, 0, 4, st); // ordered 0 = below every float
    if (e != hipSuccess)
      return e;
    hipLaunchKernelGGL(flux_max_kernel, g, b, 0, st, flux, n, maxOrd);
    if (geo == 0)
      hipLaunchKernelGGL((normalize_flux_kernel<1, 0>), g, b, 0, st, flux, area, n, normFactor, totalDiskArea, maxOrd);
    else
      hipLaunchKernelGGL((normalize_flux_kernel<1, 1>), g, b, 0, st, flux, area, n, normFactor, totalDiskArea, maxOrd);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// 64-ary box tree over the Morton-sorted primitives (packet query): a lowest-level node = union of
// the padded boxes of 64 consecutive primitives, a node of level l+1 = union of 64 consecutive nodes of level l.
// Implicit topology (children are contiguous), so there is nothing to sort or link; the
// Morton order makes 64 consecutive primitives a compact patch.  Stored top level first.
// ---------------------------------------------------------------------------
// lowest level: a node = 64 consecutive primitives (children implicit: leaf positions first .. first+cnt-1)
__global__ void wide_leafnodes_kernel(const float *sbox, unsigned n, float4 *wide, unsigned nodeBase,
                                      unsigned nodeCount) {
  const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nodeCount)
    return;
  const unsigned first = 64u * g;
  const unsigned cnt = min(64u, n - first);
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (unsigned k = 0; k < cnt; ++k) {
    const float *b = sbox + 6 * (size_t)(first + k);
    for (int c = 0; c < 3; ++c) {
      lo[c] = fminf(lo[c], b[c]);
      hi[c] = fmaxf(hi[c], b[3 + c]);
    }
  }
  wide[2 * (size_t)(nodeBase + g)] = make_float4(lo[0], lo[1], lo[2], __uint_as_float(first));
  wide[2 * (size_t)(nodeBase + g) + 1] = make_float4(hi[0], hi[1], hi[2], __uint_as_float(cnt | VR_WIDE_PRIMS));
}

__global__ void wide_level_kernel(float4 *wide, unsigned childBase, unsigned childCount, unsigned nodeBase,
                                  unsigned nodeCount) {
  const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nodeCount)
    return;
  const unsigned first = childBase + 64u * g;
  const unsigned cnt = min(64u, childCount - 64u * g);
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (unsigned k = 0; k < cnt; ++k) {
    const float4 a = wide[2 * (size_t)(first + k)], b = wide[2 * (size_t)(first + k) + 1];
    lo[0] = fminf(lo[0], a.x);
    lo[1] = fminf(lo[1], a.y);
    lo[2] = fminf(lo[2], a.z);
    hi[0] = fmaxf(hi[0], b.x);
    hi[1] = fmaxf(hi[1], b.y);
    hi[2] = fmaxf(hi[2], b.z);
  }
  wide[2 * (size_t)(nodeBase + g)] = make_float4(lo[0], lo[1], lo[2], __uint_as_float(first));
  wide[2 * (size_t)(nodeBase + g) + 1] = make_float4(hi[0], hi[1], hi[2], __uint_as_float(cnt));
}

// entries the tree of n primitives needs (levels above the primitives)
size_t wide_tree_entries(unsigned n) {
  size_t total = 0, c = n;
  do {
    c = (c + 63) / 64;
    total += c;
  } while (c > 64);
  return total + 1;
}

// the discs' {centre, radius} records from the caller's points (one radius for all: rayGeometryDisk.hpp:60-75; 2-D: the z
// column is ignored) — 16 bytes per disk that need not cross PCIe
__global__ void disk4_kernel(const float *points3, unsigned n, float radius, int D, float4 *disk4) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    disk4[i] = make_float4(points3[3 * (size_t)i], points3[3 * (size_t)i + 1], D == 2 ? 0.f : points3[3 * (size_t)i + 2], radius);
}
hipError_t launch_disk4(const float *points3, unsigned n, float radius, int D, float *disk4, hipStream_t st) {
  if (n)
    hipLaunchKernelGGL(disk4_kernel, dim3((n + 255) / 256), dim3(256), 0, st, points3, n, radius, D, reinterpret_cast<float4 *>(disk4));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// height field over the source plane (HeightFieldParams, vr_types.hpp)
// ---------------------------------------------------------------------------
__global__ void height_field_kernel(HeightFieldParams q) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= q.n)
    return;
  const float4 *pr = reinterpret_cast<const float4 *>(q.prims);
  float lo[3], hi[3];
  if (q.geo == 0) {
    const float4 c = pr[2 * (size_t)i], n = pr[2 * (size_t)i + 1];
    const float cc[3] = {c.x, c.y, c.z}, nn[3] = {n.x, n.y, n.z};
    for (int k = 0; k < 3; ++k) { // the disc's own extent along axis k (as the BVH's boxes: vr_setup.hip, box_kernel)
      const float e = c.w * sqrtf(fmaxf(0.f, 1.f - nn[k] * nn[k])) * 1.0001f;
      lo[k] = cc[k] - e;
      hi[k] = cc[k] + e;
    }
  } else {
    const float4 a = pr[4 * (size_t)i], e1 = pr[4 * (size_t)i + 1], e2 = pr[4 * (size_t)i + 2];
    const float v0[3] = {a.x, a.y, a.z}, v1[3] = {a.x - e1.x, a.y - e1.y, a.z - e1.z}, v2[3] = {a.x + e2.x, a.y + e2.y, a.z + e2.z};
    for (int k = 0; k < 3; ++k) {
      lo[k] = fminf(v0[k], fminf(v1[k], v2[k]));
      hi[k] = fmaxf(v0[k], fmaxf(v1[k], v2[k]));
    }
  }
  const float top = (q.sign > 0.f ? hi[q.ax] : -lo[q.ax]) + q.pad;
  const int ix0 = min(max((int)floorf((lo[q.a1] - q.pad - q.lo1) * q.invTile), 0), q.nx - 1);
  const int ix1 = min(max((int)floorf((hi[q.a1] + q.pad - q.lo1) * q.invTile), 0), q.nx - 1);
  int iy0 = 0, iy1 = 0;
  if (q.ny > 1) {
    iy0 = min(max((int)floorf((lo[q.a2] - q.pad - q.lo2) * q.invTile), 0), q.ny - 1);
    iy1 = min(max((int)floorf((hi[q.a2] + q.pad - q.lo2) * q.invTile), 0), q.ny - 1);
  }
  for (int iy = iy0; iy <= iy1; ++iy)
    for (int ix = ix0; ix <= ix1; ++ix)
      atomicMax(&q.raw[iy * q.nx + ix], f2ord(top));
}

__global__ void height_dilate_kernel(HeightFieldParams q) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= q.nx * q.ny)
    return;
  const int ix = t % q.nx, iy = t / q.nx;
  unsigned m = 0u; // (f2ord: 0 is below every float — a tile nothing reaches into)
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int x = ix + dx, y = iy + dy;
      if (x >= 0 && x < q.nx && y >= 0 && y < q.ny)
        m = max(m, q.raw[y * q.nx + x]);
    }
  q.field[t] = m ? ord2f(m) : -3.0e38f;
}

hipError_t launch_height_field(const HeightFieldParams &q, hipStream_t st) {
  hipError_t e = hipMemsetAsync(q.raw, 0, (size_t)q.nx * q.ny * 4, st);
  if (e != hipSuccess)
    return e;
  if (q.n)
    hipLaunchKernelGGL(height_field_kernel, dim3((q.n + 255) / 256), dim3(256), 0, st, q);
  hipLaunchKernelGGL(height_dilate_kernel, dim3((q.nx * q.ny + 255) / 256), dim3(256), 0, st, q);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// relief field over the source plane (ReliefParams, vr_types.hpp): fine tiles {lo, hi} for the tracer's per-ray clip
// (relief_clip), coarse tiles {mid height, largest fine thickness} for the generator's sort key
// ---------------------------------------------------------------------------
__device__ __forceinline__ void relief_prim_box(const ReliefParams &q, unsigned i, float (&lo)[3], float (&hi)[3]) {
  const float4 *pr = reinterpret_cast<const float4 *>(q.prims);
  if (q.geo == 0) {
    const float4 c = pr[2 * (size_t)i], n = pr[2 * (size_t)i + 1];
    const float cc[3] = {c.x, c.y, c.z}, nn[3] = {n.x, n.y, n.z};
    for (int k = 0; k < 3; ++k) { // the disc's own extent along axis k (as the BVH's boxes)
      const float e = c.w * sqrtf(fmaxf(0.f, 1.f - nn[k] * nn[k])) * 1.0001f;
      lo[k] = cc[k] - e;
      hi[k] = cc[k] + e;
    }
  } else {
    const float4 a = pr[4 * (size_t)i], e1 = pr[4 * (size_t)i + 1], e2 = pr[4 * (size_t)i + 2];
    const float v0[3] = {a.x, a.y, a.z}, v1[3] = {a.x - e1.x, a.y - e1.y, a.z - e1.z}, v2[3] = {a.x + e2.x, a.y + e2.y, a.z + e2.z};
    for (int k = 0; k < 3; ++k) {
      lo[k] = fminf(v0[k], fminf(v1[k], v2[k]));
      hi[k] = fmaxf(v0[k], fmaxf(v1[k], v2[k]));
    }
  }
}

__global__ void relief_field_kernel(ReliefParams q) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= q.n)
    return;
  float lo[3], hi[3];
  relief_prim_box(q, i, lo, hi);
  // (the pad — 1e-5 of the largest coordinate — is far above the rounding of a ray's position and of the tile walk: a
  //  point of a primitive that the walk files under the neighbouring tile is still inside that tile's range)
  const unsigned zl = f2ord(lo[q.ax] - q.pad), zh = f2ord(hi[q.ax] + q.pad);
  const int ix0 = min(max((int)floorf((lo[q.a1] - q.pad - q.lo1) * q.invTile), 0), q.nx - 1);
  const int ix1 = min(max((int)floorf((hi[q.a1] + q.pad - q.lo1) * q.invTile), 0), q.nx - 1);
  int iy0 = 0, iy1 = 0;
  if (q.ny > 1) {
    iy0 = min(max((int)floorf((lo[q.a2] - q.pad - q.lo2) * q.invTile), 0), q.ny - 1);
    iy1 = min(max((int)floorf((hi[q.a2] + q.pad - q.lo2) * q.invTile), 0), q.ny - 1);
  }
  for (int iy = iy0; iy <= iy1; ++iy)
    for (int ix = ix0; ix <= ix1; ++ix) {
      atomicMin(&q.rawLo[iy * q.nx + ix], zl);
      atomicMax(&q.rawHi[iy * q.nx + ix], zh);
    }
}

// one thread per COARSE tile: its k x k fine tiles -> {lo, hi} floats, the coarse entry and the statistics
__global__ void relief_finish_kernel(ReliefParams q) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= q.cnx * q.cny)
    return;
  const int cx = t % q.cnx, cy = t / q.cnx;
  float mlo = 3.0e38f, mhi = -3.0e38f, thick = 0.f;
  unsigned filled = 0;
  for (int dy = 0; dy < q.k; ++dy)
    for (int dx = 0; dx < q.k; ++dx) {
      const int ix = cx * q.k + dx, iy = cy * q.k + dy;
      if (ix >= q.nx || iy >= q.ny)
        continue;
      const unsigned ul = q.rawLo[iy * q.nx + ix], uh = q.rawHi[iy * q.nx + ix];
      float2 f = make_float2(3.0e38f, -3.0e38f); // (a tile nothing reaches into: no height overlaps it)
      if (uh != 0u) {
        f = make_float2(ord2f(ul), ord2f(uh));
        mlo = fminf(mlo, f.x);
        mhi = fmaxf(mhi, f.y);
        thick = fmaxf(thick, f.y - f.x);
        ++filled;
      }
      reinterpret_cast<float2 *>(q.fine)[iy * q.nx + ix] = f;
    }
  reinterpret_cast<float2 *>(q.coarse)[t] = filled ? make_float2(0.5f * (mlo + mhi), thick) : make_float2(q.emptyMid, 0.f);
  if (filled) {
    atomicAdd(&q.stats[0], 1u);
    atomicAdd(&q.stats[1], (unsigned)(4096.f * thick * thick / (thick * thick + q.travel * q.travel)));
  }
}

hipError_t launch_relief_field(const ReliefParams &q, hipStream_t st) {
  hipError_t e = hipMemsetAsync(q.rawLo, 0xFF, (size_t)q.nx * q.ny * 4, st);
  if (e == hipSuccess)
    e = hipMemsetAsync(q.rawHi, 0, (size_t)q.nx * q.ny * 4, st);
  if (e == hipSuccess)
    e = hipMemsetAsync(q.stats, 0, 2 * 4, st);
  if (e != hipSuccess)
    return e;
  if (q.n)
    hipLaunchKernelGGL(relief_field_kernel, dim3((q.n + 255) / 256), dim3(256), 0, st, q);
  hipLaunchKernelGGL(relief_finish_kernel, dim3((q.cnx * q.cny + 255) / 256), dim3(256), 0, st, q);
  return hipGetLastError();
}

// builds the tree from s.sbox (sorted, padded boxes); out3 = the root's {first child entry,
// child count | VR_WIDE_PRIMS if the root's children are the primitives themselves, 0}
hipError_t launch_wide_tree(const SetupParams &s, unsigned *out3, hipStream_t st) {
  const unsigned n = s.n;
  out3[0] = out3[1] = out3[2] = 0;
  if (n == 0)
    return hipSuccess;
  if (n <= 64) { // the root's children are the primitives
    out3[1] = n | VR_WIDE_PRIMS;
    return hipSuccess;
  }
  unsigned counts[8], nl = 0;
  counts[nl++] = (n + 63) / 64;
  while (counts[nl - 1] > 64) {
    counts[nl] = (counts[nl - 1] + 63) / 64;
    ++nl;
  }
  // memory order: top level first
  unsigned base[8];
  unsigned off = 0;
  for (int l = (int)nl - 1; l >= 0; --l) {
    base[l] = off;
    off += counts[l];
  }
  float4 *wide = reinterpret_cast<float4 *>(s.wide);
  hipLaunchKernelGGL(wide_leafnodes_kernel, dim3((counts[0] + 63) / 64), dim3(64), 0, st, s.sbox, n, wide, base[0],
                     counts[0]);
  for (unsigned l = 1; l < nl; ++l)
    hipLaunchKernelGGL(wide_level_kernel, dim3((counts[l] + 63) / 64), dim3(64), 0, st, wide, base[l - 1], counts[l - 1],
                       base[l], counts[l]);
  out3[0] = base[nl - 1];
  out3[1] = counts[nl - 1];
  return hipGetLastError();
}

hipError_t launch_quantize_nodes(const float *nodes, unsigned numNodes, const float *base3, const float *scale3,
                                 uint32_t *qnodes, uint32_t *pnodes, hipStream_t st) {
  if (numNodes == 0)
    return hipSuccess;
  hipLaunchKernelGGL(quantize_nodes_kernel, dim3((numNodes + 255) / 256), dim3(256), 0, st,
                     reinterpret_cast<const float4 *>(nodes), numNodes, base3[0], base3[1], base3[2], scale3[0],
                     scale3[1], scale3[2], reinterpret_cast<uint4 *>(qnodes));
  hipLaunchKernelGGL(pair_nodes_kernel, dim3((numNodes + 255) / 256), dim3(256), 0, st,
                     reinterpret_cast<const uint4 *>(qnodes), numNodes, reinterpret_cast<uint4 *>(pnodes));
  return hipGetLastError();
}

hipError_t launch_setup_neighbors(const SetupParams &s, int pass, hipStream_t st) {
  const unsigned g = (s.n + 255) / 256;
  if (s.n == 0)
    return hipSuccess;
  if (pass == 0)
    hipLaunchKernelGGL((nb_kernel<0>), dim3(g), dim3(256), 0, st, s);
  else if (pass == 1)
    hipLaunchKernelGGL((nb_kernel<1>), dim3(g), dim3(256), 0, st, s);
  else if (pass == 2) // count and keep (s.nbTmp; its overflow word zeroed by the caller)
    hipLaunchKernelGGL((nb_kernel<2>), dim3(g), dim3(256), 0, st, s);
  else                // pack pass 2's lists behind the scanned offsets
    hipLaunchKernelGGL(nb_compact_kernel, dim3(g), dim3(256), 0, st, s);
  return hipGetLastError();
}

} // namespace vr

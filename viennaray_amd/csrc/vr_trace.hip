// vr_trace.hip — the HIP kernels of the flux tracer (gfx950).
//
// The reference traces rays one by one in index order (rayTraceKernel.hpp:118).
// Ray i's whole random stream is a pure function of (i, seed), so any order
// gives the same flux; the GPU path is therefore organised as an HBM-resident
// RAY STREAM, processed in batches:
//
//   gen_kernel      one lane per ray index: per-ray mt19937_64 (lazy, streaming),
//                   power-cosine source sample -> 32-byte ray record (+ the 16-byte
//                   RNG cursors when the particle keeps going after a hit), written
//                   DIRECTLY into the sort bin of the cell where the ray crosses the
//                   sort plane (bin cursor = one atomic; no separate sort pass), so a
//                   wavefront's 64 rays end in the same neighbourhood: their BVH node
//                   / primitive fetches are wave-uniform (scalar loads) and their
//                   traversal loops stay converged.
//   trace_kernel    persistent wavefronts pull bins; a lane whose ray ends pulls
//                   the next one (wave-wide compaction by ballot + prefix
//                   popcount), so bounce chains of different length do not idle
//                   the wave.  Per segment: closest hit (wave-uniform packet query /
//                   packet traversal, per-lane ordered walk with its stack in LDS as
//                   the fallback; then the boundary walls), then the reference's
//                   state machine (rayTraceKernel.hpp:155-335).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "vr_device.hpp"
#include "vr_kernels.hpp"
#include "vr_particles.hpp"
#include "vr_trace_kernel.hpp"

namespace vr {

// ---------------------------------------------------------------------------
// source sampling (raySourceRandom.hpp:25-116)
// ---------------------------------------------------------------------------
// `draw()` returns the next raw 64-bit engine output
template <int D, class Draw>
__device__ __forceinline__ void source_sample(const TraceParams &p, Draw &&draw, V3 &org, V3 &dir) {
  // origin draws first (raySourceRandom.hpp:50-68)
  org = mk(0.f, 0.f, 0.f);
  const float r1 = canon_f32(draw());
  setc(org, p.rayDir, p.srcCoord);
  setc(org, p.firstDir, p.lo1 + (p.hi1 - p.lo1) * r1);
  if (D == 2) {
    setc(org, p.secondDir, 0.f);
  } else {
    const float r2 = canon_f32(draw());
    setc(org, p.secondDir, p.lo2 + (p.hi2 - p.lo2) * r2);
  }
  // then the direction draws (raySourceRandom.hpp:70-116)
  if (!p.useBasis) {
    const float d1 = canon_f32(draw());
    const float d2 = canon_f32(draw());
    float ct, st, cp, sp;
    cosine_sample(d1, d2, p.ee, ct, st, cp, sp);
    dir = mk(0.f, 0.f, 0.f);
    setc(dir, p.rayDir, p.posNeg * ct);
    setc(dir, p.firstDir, cp * st);
    setc(dir, p.secondDir, sp * st);
  } else {
    float dr;
    do {
      const float d1 = canon_f32(draw());
      const float d2 = canon_f32(draw());
      float ct, st, cp, sp;
      cosine_sample(d1, d2, p.ee, ct, st, cp, sp);
      const float a = ct, b = cp * st, c = sp * st;
      dir.x = (p.basis[0] * a + p.basis[3] * b) + p.basis[6] * c;
      dir.y = (p.basis[1] * a + p.basis[4] * b) + p.basis[7] * c;
      dir.z = (p.basis[2] * a + p.basis[5] * b) + p.basis[8] * c;
      dr = getc(dir, p.rayDir);
    } while ((p.posNeg < 0.f && dr > 0.f) || (p.posNeg > 0.f && dr < 0.f));
  }
}

__device__ __forceinline__ unsigned part1by1(unsigned v) {
  v &= 0x0000FFFFu;
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

// Sort key of a ray: the cell in which it crosses the FAR plane of the geometry's
// bounding box (the plane opposite the source), folded back into the domain the
// way the side walls would (periodic wrap / mirror).  For surface-like
// geometry that is where the ray ends up and where the BVH is deepest, so the 64
// rays of a wavefront walk (almost) the same nodes and leaves.  Cells are
// Morton-ordered so consecutive bins are spatial neighbours.  The key only
// orders the work; it has no influence on any result.
__device__ __forceinline__ float fold_unit(float u, int bc) {
  if (bc == 1) // periodic
    return u - floorf(u);
  if (bc == 0) { // reflective: mirror fold with period 2
    float v = u - 2.f * floorf(0.5f * u);
    return v > 1.f ? 2.f - v : v;
  }
  return u; // ignore: clamped below
}

template <int D> __device__ __forceinline__ unsigned bin_of(const TraceParams &p, const V3 &org, const V3 &dir) {
  // sort plane: the coordinate on the tracing axis where most first hits are expected
  const float keyCoord = p.keyCoord;
  const float dr = getc(dir, p.rayDir);
  float t = (keyCoord - p.srcCoord) / (fabsf(dr) > 1e-6f ? dr : copysignf(1e-6f, dr == 0.f ? -p.posNeg : dr));
  t = (p.debugFlags & 2u) ? 0.f : (t > 0.f ? t : 0.f); // flag 2: key on the origin instead
  const float u1 = fold_unit((getc(org, p.firstDir) + getc(dir, p.firstDir) * t - p.lo1) * p.invExt1, p.bc0);
  int c1 = (int)(u1 * (float)p.binT1);
  c1 = c1 < 0 ? 0 : (c1 >= p.binT1 ? p.binT1 - 1 : c1);
  if (D == 2)
    return (unsigned)c1;
  const float u2 = fold_unit((getc(org, p.secondDir) + getc(dir, p.secondDir) * t - p.lo2) * p.invExt2, p.bc1);
  int c2 = (int)(u2 * (float)p.binT2);
  c2 = c2 < 0 ? 0 : (c2 >= p.binT2 ? p.binT2 - 1 : c2);
  // 8x8 tiles in row-major order; inside a tile the COLUMNS run in alternating directions (boustrophedon: up
  // column 0, down column 1, ...), so consecutive bins are always adjacent cells — also from one tile to the next
  // in a row of tiles (a tile ends bottom right, its neighbour starts bottom left).  A round of the trace kernel
  // swallows two or three bins; with plain row-major cells one round in four straddled a row end: a packet box
  // eight cells wide.
  const unsigned tile = (unsigned)(c2 >> 3) * (unsigned)p.binTiles + (unsigned)(c1 >> 3);
  const unsigned row = (unsigned)c2 & 7u, col = (unsigned)c1 & 7u;
  return tile * 64u + (col << 3 | ((col & 1u) ? 7u - row : row));
}

// ---------------------------------------------------------------------------
// gen_kernel: ray index -> ray record
// ---------------------------------------------------------------------------
// Writes the ray record straight into its sort bin (no separate sort pass): the bin's
// cursor hands out one of p.binCap slots; a ray whose bin is full goes to the
// overflow region, which is traced after the bins.  Returns the record slot.
template <int D, bool KEEP>
__device__ __forceinline__ unsigned gen_store(const TraceParams &p, unsigned i, const V3 &o, const V3 &d, unsigned k,
                                              u64 lo, u64 hi) {
  unsigned slot = i;
  if (p.binCount && !(p.debugFlags & 64u)) { // flag 64: timing experiment, no binning
    const unsigned b = bin_of<D>(p, o, project_dir<D>(d));
    const unsigned pos = atomicAdd(&p.binCount[b], 1u);
    if (pos < p.binCap)
      slot = b * p.binCap + pos;
    else
      slot = p.numBins * p.binCap + atomicAdd(&p.binCount[p.numBins], 1u); // < ovCap by construction
  }
  // record = {A, B} (32 B) [+ the RNG cursors {s[k], s[k+156]} (16 B) when the particle keeps going]
  float4 *rec = reinterpret_cast<float4 *>(p.slotRec) + (size_t)(KEEP ? 3 : 2) * slot;
  rec[0] = make_float4(o.x, o.y, o.z, d.x);
  rec[1] = make_float4(d.y, d.z, __uint_as_float(i), __uint_as_float(k));
  if (KEEP)
    *reinterpret_cast<ulonglong2 *>(rec + 2) = make_ulonglong2(lo, hi);
  return slot;
}

// Fixed number of source draws (no tilted primary direction): the NS engine outputs
// the source sample needs are produced straight into registers by one 156+NS-step pass
// of the seeding recurrence, which also leaves the streaming cursors for the trace kernel.
// The bin cursor's returning atomic is the one long latency of a ray; it is issued as soon as the ray's bin is
// known and its answer is used one loop pass later, after the NEXT ray's seeding chain: the wave computes while
// its own atomic is under way instead of leaving that to the other waves of the SIMD.
template <int D, bool KEEP> __global__ __launch_bounds__(VR_BLOCK) void gen_kernel(const TraceParams p) {
  constexpr int NS = D == 3 ? 4 : 3;
  const bool binned = p.binCount && !(p.debugFlags & 64u); // flag 64: timing experiment, no binning
  bool havePrev = false;
  V3 po = mk(0, 0, 0), pd = mk(0, 0, 1);
  unsigned pi = 0, pbin = 0, ppos = 0;
  u64 plo = 0, phi = 0;
  for (unsigned i = blockIdx.x * VR_BLOCK + threadIdx.x;; i += gridDim.x * VR_BLOCK) {
    const bool cur = i < p.batchCount;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    u64 lo = 0, hi = 0;
    unsigned b = 0;
    if (cur) {
      const unsigned long long idx = p.idxList ? p.idxList[i] : p.batchFirst + i;
      u64 out[NS];
      mt_first_outputs<NS>(tea3((unsigned)idx, p.seed), out, lo, hi);
      int k = 0;
      source_sample<D>(p, [&]() { return out[k++]; }, o, d); // k is a compile-time sequence after unrolling
      if (binned)
        b = bin_of<D>(p, o, project_dir<D>(d));
    }
    if (havePrev) { // the previous ray of this lane: its slot has arrived
      unsigned slot = pi;
      if (binned) {
        if (ppos < p.binCap)
          slot = pbin * p.binCap + ppos;
        else
          slot = p.numBins * p.binCap + atomicAdd(&p.binCount[p.numBins], 1u); // < ovCap by construction
      }
      // record = {A, B} (32 B) [+ the RNG cursors {s[k], s[k+156]} (16 B) when the particle keeps going]
      float4 *rec = reinterpret_cast<float4 *>(p.slotRec) + (size_t)(KEEP ? 3 : 2) * slot;
      rec[0] = make_float4(po.x, po.y, po.z, pd.x);
      rec[1] = make_float4(pd.y, pd.z, __uint_as_float(pi), __uint_as_float((unsigned)NS));
      if (KEEP)
        *reinterpret_cast<ulonglong2 *>(rec + 2) = make_ulonglong2(plo, phi);
    }
    if (!cur)
      break;
    if (binned)
      ppos = atomicAdd(&p.binCount[b], 1u); // (answer used in the next pass)
    po = o;
    pd = d;
    pi = i;
    pbin = b;
    plo = lo;
    phi = hi;
    havePrev = true;
  }
}

// General generator (tilted primary direction: the rejection loop makes the number
// of draws data dependent): the streaming generator from draw 0 (+ tier 2).
template <int D, bool KEEP> __global__ __launch_bounds__(VR_BLOCK) void gen_basis_kernel(const TraceParams p) {
  const unsigned tid = threadIdx.x;
  const unsigned gwave = (blockIdx.x * VR_BLOCK + tid) >> 6; // physical wave of this (bounded) grid
  u64 *scratchLane = p.rngScratch + (size_t)gwave * (312u * 64u) + (tid & 63u);
  for (unsigned i = blockIdx.x * VR_BLOCK + tid; i < p.batchCount; i += gridDim.x * VR_BLOCK) {
    const unsigned long long idx = p.idxList ? p.idxList[i] : p.batchFirst + i;
    Rng rng;
    rng_init(rng, tea3((unsigned)idx, p.seed), scratchLane);
    unsigned t2 = 0;
    V3 o, d;
    source_sample<D>(p, [&]() { return rng_next(rng, t2); }, o, d);
    gen_store<D, KEEP>(p, i, o, d, rng.k, rng.lo, rng.hi); // (k >= 156: the trace kernel rebuilds tier 2 from the seed)
  }
}

// SourceGrid (raySourceGrid.hpp:25-66): origin = grid[idx % numPoints], direction from two draws
// (cosf / sinf / powf / sqrtf in float, then Normalize)
template <int D, bool KEEP> __global__ __launch_bounds__(VR_BLOCK) void gen_grid_kernel(const TraceParams p) {
  for (unsigned i = blockIdx.x * VR_BLOCK + threadIdx.x; i < p.batchCount; i += gridDim.x * VR_BLOCK) {
    const unsigned long long idx = p.idxList ? p.idxList[i] : p.batchFirst + i;
    u64 out[2], lo, hi;
    mt_first_outputs<2>(tea3((unsigned)idx, p.seed), out, lo, hi);
    const float r1 = canon_f32(out[0]), r2 = canon_f32(out[1]);
    const float *g = p.gridPoints + 3 * (size_t)(idx % p.gridCount);
    const V3 o = mk(g[0], g[1], g[2]);
    const float tt = glibc_powf(r2, p.eeGrid);
    const float ang = (float)(3.14159265358979323846 * 2.f * (double)r1);
    float sn, cs;
    glibc_sincosf(ang, sn, cs);
    V3 d = mk(0.f, 0.f, 0.f);
    setc(d, p.rayDir, p.posNeg * sqrtf(tt));
    setc(d, p.firstDir, cs * sqrtf(1.f - tt));
    setc(d, p.secondDir, D == 2 ? 0.f : sn * sqrtf(1.f - tt));
    vnormalize(d);
    gen_store<D, KEEP>(p, i, o, d, 2u, lo, hi);
  }
}

// Rays produced by a host-side Source callback (raySource.hpp:10-19): origin, direction and the
// number of engine outputs the callback consumed; the record's RNG cursors continue from there
template <int D, bool KEEP> __global__ __launch_bounds__(VR_BLOCK) void gen_host_kernel(const TraceParams p) {
  const unsigned tid = threadIdx.x;
  const unsigned gwave = (blockIdx.x * VR_BLOCK + tid) >> 6;
  u64 *scratchLane = p.rngScratch + (size_t)gwave * (312u * 64u) + (tid & 63u);
  for (unsigned i = blockIdx.x * VR_BLOCK + tid; i < p.batchCount; i += gridDim.x * VR_BLOCK) {
    const unsigned long long idx = p.idxList ? p.idxList[i] : p.batchFirst + i;
    const V3 o = mk(p.hostOrg[3 * idx], p.hostOrg[3 * idx + 1], p.hostOrg[3 * idx + 2]);
    const V3 d = mk(p.hostDir[3 * idx], p.hostDir[3 * idx + 1], p.hostDir[3 * idx + 2]);
    Rng rng;
    rng_init(rng, tea3((unsigned)idx, p.seed), scratchLane);
    if (KEEP) {
      unsigned t2 = 0;
      const unsigned k = p.hostDraws ? p.hostDraws[idx] : 0u;
      for (unsigned j = 0; j < k && j < 156u; ++j)
        (void)rng_next(rng, t2);
      rng.k = k; // (k >= 156: the trace kernel rebuilds the full state from the seed and skips k outputs)
    }
    gen_store<D, KEEP>(p, i, o, d, rng.k, rng.lo, rng.hi);
  }
}

// ---------------------------------------------------------------------------
// exclusive scan (in place), 2048 elements per block: radix-sort digit tables, neighbour offsets
// ---------------------------------------------------------------------------
constexpr unsigned SCAN_PER_THREAD = 8;
constexpr unsigned SCAN_PER_BLOCK = SCAN_PER_THREAD * VR_BLOCK;

__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned *sh, unsigned &total) {
  const unsigned tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  unsigned x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    unsigned y = __shfl_up(x, off, 64);
    if ((int)lane >= off)
      x += y;
  }
  if (lane == 63)
    sh[w] = x;
  __syncthreads();
  unsigned base = 0;
  for (unsigned k = 0; k < w; ++k)
    base += sh[k];
  total = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return base + x - v;
}

__global__ __launch_bounds__(VR_BLOCK) void scan_block_kernel(unsigned *data, unsigned n, unsigned *blockSums) {
  __shared__ unsigned sh[4];
  const unsigned base = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * SCAN_PER_THREAD;
  unsigned v[SCAN_PER_THREAD];
  unsigned sum = 0;
#pragma unroll
  for (unsigned k = 0; k < SCAN_PER_THREAD; ++k) {
    v[k] = base + k < n ? data[base + k] : 0u;
    sum += v[k];
  }
  unsigned total;
  unsigned ex = block_exclusive_scan(sum, sh, total);
#pragma unroll
  for (unsigned k = 0; k < SCAN_PER_THREAD; ++k) {
    if (base + k < n)
      data[base + k] = ex;
    ex += v[k];
  }
  if (threadIdx.x == 0 && blockSums)
    blockSums[blockIdx.x] = total;
}

__global__ __launch_bounds__(VR_BLOCK) void scan_add_kernel(unsigned *data, unsigned n, const unsigned *blockOffsets) {
  const unsigned off = blockOffsets[blockIdx.x];
  const unsigned base = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * SCAN_PER_THREAD;
#pragma unroll
  for (unsigned k = 0; k < SCAN_PER_THREAD; ++k)
    if (base + k < n)
      data[base + k] += off;
}


// ---------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------
hipError_t launch_gen(const TraceParams &p, int D, bool keepRng, unsigned maxBlocks, hipStream_t s) {
  unsigned grid = (p.batchCount + VR_BLOCK - 1) / VR_BLOCK;
  if (grid == 0)
    return hipSuccess;
  if (grid > maxBlocks)
    grid = maxBlocks; // grid-stride; bounds the tier-2 slabs to grid waves
  // source: SourceRandom (0: axis-aligned, 1: tilted primary direction), SourceGrid (2), host rays (3)
  const int src = p.hostOrg ? 3 : (p.gridPoints ? 2 : (p.useBasis ? 1 : 0));
  const int key = src * 4 + (D == 2 ? 0 : 2) + (keepRng ? 1 : 0);
#define VR_GEN(K, DD, KEEP)                                                                                           \
  case K: hipLaunchKernelGGL((gen_kernel<DD, KEEP>), dim3(grid), dim3(VR_BLOCK), 0, s, p); break;                     \
  case 4 + K: hipLaunchKernelGGL((gen_basis_kernel<DD, KEEP>), dim3(grid), dim3(VR_BLOCK), 0, s, p); break;            \
  case 8 + K: hipLaunchKernelGGL((gen_grid_kernel<DD, KEEP>), dim3(grid), dim3(VR_BLOCK), 0, s, p); break;             \
  case 12 + K: hipLaunchKernelGGL((gen_host_kernel<DD, KEEP>), dim3(grid), dim3(VR_BLOCK), 0, s, p); break;
  switch (key) {
    VR_GEN(0, 2, false)
    VR_GEN(1, 2, true)
    VR_GEN(2, 3, false)
    VR_GEN(3, 3, true)
  }
#undef VR_GEN
  return hipGetLastError();
}

hipError_t launch_scan(unsigned *data, unsigned n, unsigned *tmp /* >= 2 * ceil(n/2048) + 2 */, hipStream_t s) {
  const unsigned blocks = (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
  if (blocks <= 1) {
    hipLaunchKernelGGL(scan_block_kernel, dim3(1), dim3(VR_BLOCK), 0, s, data, n, (unsigned *)nullptr);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(scan_block_kernel, dim3(blocks), dim3(VR_BLOCK), 0, s, data, n, tmp);
  hipError_t e = launch_scan(tmp, blocks, tmp + blocks, s);
  if (e != hipSuccess)
    return e;
  hipLaunchKernelGGL(scan_add_kernel, dim3(blocks), dim3(VR_BLOCK), 0, s, data, n, tmp);
  return hipGetLastError();
}

template <int D, int GEO, int PARTICLE>
static hipError_t launch_trace_t(const TraceParams &p, int mode, unsigned grid, hipStream_t s) {
  if (mode == 1)
    hipLaunchKernelGGL((trace_kernel<D, GEO, 0, 1>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
  else if (mode == 2)
    hipLaunchKernelGGL((trace_kernel<D, GEO, 0, 2>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
  else if (mode == 3 && GEO == 0 && PARTICLE != P_EXT)
    hipLaunchKernelGGL((trace_kernel<D, 0, PARTICLE == P_EXT ? 0 : PARTICLE, 3>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
  else if (mode == 4)
    hipLaunchKernelGGL((trace_kernel<D, GEO, PARTICLE, 4>), dim3(grid), dim3(VR_BLOCK), p.smallBytes, s, p);
  else
    hipLaunchKernelGGL((trace_kernel<D, GEO, PARTICLE, 0>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
  return hipGetLastError();
}

// mode: 0 general, 1 absorbing + flat scene, 2 absorbing + structured scene
// particle: 0 DiffuseParticle, 1 SpecularParticle, 2 (P_EXT) extended kernel (always mode 0)
template <class F> static auto dispatch_variant(int D, int geo, int particle, F &&f) {
  const int key = (D == 2 ? 0 : 6) + (geo ? 3 : 0) + particle;
  switch (key) {
  case 0: return f(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  case 1: return f(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  case 2: return f(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
  case 3: return f(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  case 4: return f(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  case 5: return f(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
  case 6: return f(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  case 7: return f(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  case 8: return f(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
  case 9: return f(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  case 10: return f(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  default: return f(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
  }
}

hipError_t launch_trace(const TraceParams &p, int D, int geo, int particle, int mode, unsigned grid,
                        hipStream_t s) {
  if (mode == 1 || mode == 2)
    particle = 0; // the reflection model is unobservable: one instantiation serves all
  return dispatch_variant(D, geo, particle, [&](auto d, auto g, auto pt) {
    return launch_trace_t<decltype(d)::value, decltype(g)::value, decltype(pt)::value>(p, mode, grid, s);
  });
}

template <int D, int GEO, int PARTICLE> static int occ_t(int mode, unsigned smallBytes) {
  int nb = 0;
  hipError_t e;
  if (mode == 1)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, GEO, 0, 1>, VR_BLOCK, 0);
  else if (mode == 2)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, GEO, 0, 2>, VR_BLOCK, 0);
  else if (mode == 3 && GEO == 0 && PARTICLE != P_EXT)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, 0, PARTICLE == P_EXT ? 0 : PARTICLE, 3>, VR_BLOCK, 0);
  else if (mode == 4)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, GEO, PARTICLE, 4>, VR_BLOCK, smallBytes);
  else
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, GEO, PARTICLE, 0>, VR_BLOCK, 0);
  return e == hipSuccess ? nb : 2;
}

int trace_blocks_per_cu(int D, int geo, int particle, int mode, unsigned smallBytes) {
  if (mode == 1 || mode == 2)
    particle = 0;
  return dispatch_variant(D, geo, particle, [&](auto d, auto g, auto pt) {
    return occ_t<decltype(d)::value, decltype(g)::value, decltype(pt)::value>(mode, smallBytes);
  });
}

// ---- diagnostics -----------------------------------------------------------
template <int GEO>
__global__ void debug_intersect_kernel(const TraceParams p, const float *org, const float *dir, const float *tnear,
                                       unsigned n, int *geomID, unsigned *primID, float *t, int ordered,
                                       unsigned walkStackWaves) {
  __shared__ float wallS[96];
  __shared__ unsigned stackS[VR_STACK_LDS * VR_BLOCK]; // (64-thread blocks: lane columns 0..63 of the [entry][VR_BLOCK] layout)
  for (unsigned k = threadIdx.x; k < 96; k += blockDim.x)
    wallS[k] = p.wallTable[k];
  __syncthreads();
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  // (the walk votes wave-wide: every lane of the block takes part in the call)
  const unsigned j = i < n ? i : 0u;
  const V3 o = mk(org[3 * j], org[3 * j + 1], org[3 * j + 2]), d = mk(dir[3 * j], dir[3 * j + 1], dir[3 * j + 2]);
  HitRec h;
  hit_clear(h);
#ifdef VR_DIAG
  unsigned long long phaseDummy[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tLast = 0ull;
  unsigned long long *const phaseT = phaseDummy;
#endif
  if (ordered) {
    unsigned node = 0u, sp = 0u;
    VR_DIAG_DECL
    // (diagnostic launches are small: the block index serves as the wave index of the global slab; the host bounds it)
    pair_walk_lanes<GEO, VR_STACK_LDS>(p, reinterpret_cast<const uint4 *>(p.pnodes), reinterpret_cast<const float4 *>(p.prims),
                                       stackS + threadIdx.x, p.walkStack + (size_t)(blockIdx.x % walkStackWaves) * (VR_STACK_GLOBAL * 64u) + threadIdx.x,
                             i < n, o, d, tnear[j], h, node, sp, 1u VR_DIAG_PASS);
  } else {
    unsigned node = 0u;
    VR_DIAG_DECL
    bvh_walk_lanes<GEO>(p, i < n, o, d, tnear[j], h, node, 1u VR_DIAG_PASS);
  }
  hit_walls(p, wallS, o, d, tnear[j], h);
  if (i >= n)
    return;
  geomID[i] = h.geom;
  primID[i] = h.prim;
  t[i] = h.t;
}

hipError_t launch_debug_intersect(const TraceParams &p, int geo, const float *org, const float *dir,
                                  const float *tnear, unsigned n, int *geomID, unsigned *primID, float *t, int ordered,
                                  unsigned walkStackWaves, hipStream_t s) {
  const unsigned grid = (n + 63) / 64;
  if (geo == 0)
    hipLaunchKernelGGL((debug_intersect_kernel<0>), dim3(grid), dim3(64), 0, s, p, org, dir, tnear, n, geomID, primID, t,
                       ordered, walkStackWaves);
  else
    hipLaunchKernelGGL((debug_intersect_kernel<1>), dim3(grid), dim3(64), 0, s, p, org, dir, tnear, n, geomID, primID, t,
                       ordered, walkStackWaves);
  return hipGetLastError();
}

template <int D>
__global__ void debug_process_hit_kernel(const TraceParams p, const float *org, const float *dir, const float *tfar,
                                         const unsigned *prim, unsigned n, float *outOrg, float *outDir, int *outReflect) {
  __shared__ float wallS[96];
  for (unsigned k = threadIdx.x; k < 96; k += blockDim.x)
    wallS[k] = p.wallTable[k];
  __syncthreads();
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  V3 o = mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
  V3 rd = mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
  V3 d = project_dir<D>(rd);
  const V3 hp = mk(o.x + d.x * tfar[i], o.y + d.y * tfar[i], o.z + d.z * tfar[i]);
  bool active = true;
  process_boundary_hit<D>(p, wallS, prim[i], hp, o, rd, d, active);
  outOrg[3 * i] = o.x;
  outOrg[3 * i + 1] = o.y;
  outOrg[3 * i + 2] = o.z;
  outDir[3 * i] = d.x;
  outDir[3 * i + 1] = d.y;
  outDir[3 * i + 2] = d.z;
  outReflect[i] = active ? 1 : 0;
}

hipError_t launch_debug_process_hit(const TraceParams &p, int D, const float *org, const float *dir, const float *tfar,
                                    const unsigned *prim, unsigned n, float *outOrg, float *outDir, int *outReflect,
                                    hipStream_t s) {
  const unsigned grid = (n + 63) / 64;
  if (D == 2)
    hipLaunchKernelGGL((debug_process_hit_kernel<2>), dim3(grid), dim3(64), 0, s, p, org, dir, tfar, prim, n, outOrg, outDir, outReflect);
  else
    hipLaunchKernelGGL((debug_process_hit_kernel<3>), dim3(grid), dim3(64), 0, s, p, org, dir, tfar, prim, n, outOrg, outDir, outReflect);
  return hipGetLastError();
}

__global__ __launch_bounds__(VR_BLOCK) void debug_rng_kernel(unsigned seed32, unsigned count, u64 *scratch, u64 *out) {
  if (threadIdx.x != 0)
    return;
  Rng rng;
  rng_init(rng, seed32, scratch);
  unsigned t2 = 0;
  for (unsigned i = 0; i < count; ++i)
    out[i] = rng_next(rng, t2);
}

hipError_t launch_debug_rng(unsigned seed32, unsigned count, unsigned long long *scratch, unsigned long long *out,
                            hipStream_t s) {
  hipLaunchKernelGGL(debug_rng_kernel, dim3(1), dim3(VR_BLOCK), 0, s, seed32, count, scratch, out);
  return hipGetLastError();
}

// un-permute the leaf-ordered accumulators into the caller's primitive order
__global__ void gather_flux_kernel(const unsigned long long *acc, unsigned stride, unsigned replicas,
                                   const unsigned *leafOfOrig, unsigned n, unsigned long long *outAcc) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const unsigned q = leafOfOrig[i];
    unsigned long long s = 0; // (integer sum: replica order is irrelevant)
    for (unsigned r = 0; r < replicas; ++r)
      s += acc[(size_t)r * stride + q];
    outAcc[i] = s;
  }
}

hipError_t launch_gather_flux(const unsigned long long *acc, unsigned stride, unsigned replicas,
                              const unsigned *leafOfOrig, unsigned n, unsigned long long *outAcc, hipStream_t s) {
  hipLaunchKernelGGL(gather_flux_kernel, dim3((n + 255) / 256), dim3(256), 0, s, acc, stride, replicas, leafOfOrig, n,
                     outAcc);
  return hipGetLastError();
}

} // namespace vr

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

namespace vr {

// fixed-point weight: 2^40 per unit (order-independent integer accumulation)
__device__ __forceinline__ u64 weight_fx(float w) { return (u64)((double)w * 1099511627776.0 + 0.5); }

#ifndef VR_USER_MODULE // (a run-time compiled particle module holds trace kernels only, see the end of the file)
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

// Sort key on a scene that is flat WITH RELIEF (TraceParams, round 4): the cell of the ray's PREDICTED first hit — the
// crossing of the plane through the mid height of the coarse relief tile under the previous guess, two look-ups starting
// from the sort plane — so that the rays of a wave meet the surface, not some plane above or below it, in one
// neighbourhood whatever their angles.  A ray whose stretch through the local slab is long (thickness x tan(theta) >
// reliefTravel: a grazing ray) is filed in the coarser LOOSE bins instead (bit 31 of the result): one such ray in a wave
// stretches the packet query's box over dozens of cells.  Like bin_of this only orders the work.
template <int D> __device__ __forceinline__ unsigned bin_of_relief(const TraceParams &p, const V3 &org, const V3 &dir) {
  typedef float F2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(1))) F2 *GlobalF2;
  const GlobalF2 coarse = (GlobalF2)p.reliefCoarse;
  const float dr = getc(dir, p.rayDir);
  const float drs = fabsf(dr) > 1e-6f ? dr : copysignf(1e-6f, dr == 0.f ? -p.posNeg : dr);
  // (the key only orders the work: the approximate reciprocal and fused multiply-adds will do, and the look-ups take
  //  the unfolded position, clamped — a ray that crosses a side wall first is sorted a little less well)
  const float inv = __builtin_amdgcn_rcpf(drs);
  const float o1 = getc(org, p.firstDir), d1 = getc(dir, p.firstDir);
  const float o2 = D == 3 ? getc(org, p.secondDir) : 0.f, d2 = D == 3 ? getc(dir, p.secondDir) : 0.f;
  const float a1 = (o1 - p.rcLo1) * p.rcInvT, b1 = d1 * p.rcInvT, a2 = (o2 - p.rcLo2) * p.rcInvT, b2 = d2 * p.rcInvT;
  float t = fmaxf((p.keyCoord - p.srcCoord) * inv, 0.f), thick = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    if (it == p.reliefLookups)
      break;
    int cx = (int)__builtin_fmaf(b1, t, a1);
    cx = cx < 0 ? 0 : (cx >= p.rcNx ? p.rcNx - 1 : cx);
    int cy = 0;
    if (D == 3) {
      cy = (int)__builtin_fmaf(b2, t, a2);
      cy = cy < 0 ? 0 : (cy >= p.rcNy ? p.rcNy - 1 : cy);
    }
    const F2 f = coarse[cy * p.rcNx + cx];
    t = fmaxf((f.x - p.srcCoord) * inv, 0.f);
    thick = f.y;
  }
  const float u1 = fold_unit((__builtin_fmaf(d1, t, o1) - p.lo1) * p.invExt1, p.bc0);
  const float sin2 = fmaxf(0.f, 1.f - dr * dr);
  const bool loose = thick * thick * sin2 > p.reliefTravel * p.reliefTravel * (drs * drs) || sin2 > p.reliefTanMax * p.reliefTanMax * (drs * drs);
  const int T1 = loose ? p.looseT1 : p.binT1, T2 = loose ? p.looseT2 : p.binT2, tiles = loose ? p.looseTiles : p.binTiles;
  int c1 = (int)(u1 * (float)T1);
  c1 = c1 < 0 ? 0 : (c1 >= T1 ? T1 - 1 : c1);
  if (D == 2)
    return (unsigned)c1 | (loose ? 0x80000000u : 0u);
  const float u2 = fold_unit((__builtin_fmaf(d2, t, o2) - p.lo2) * p.invExt2, p.bc1);
  int c2 = (int)(u2 * (float)T2);
  c2 = c2 < 0 ? 0 : (c2 >= T2 ? T2 - 1 : c2);
  const unsigned tile = (unsigned)(c2 >> 3) * (unsigned)tiles + (unsigned)(c1 >> 3);
  const unsigned row = (unsigned)c2 & 7u, col = (unsigned)c1 & 7u;
  return (tile * 64u + (col << 3 | ((col & 1u) ? 7u - row : row))) | (loose ? 0x80000000u : 0u);
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
  float4 *rec = reinterpret_cast<float4 *>(p.slotRec) + (size_t)2 * slot;
  if (KEEP) {
    // the compact record (vr_types.hpp) + what the plain generator's rays do not need: origin[rayDir], k, s[k]
    rec[0] = make_float4(getc(o, p.firstDir), getc(o, p.secondDir), d.x, d.y);
    rec[1] = make_float4(d.z, __uint_as_float(i), __uint_as_float((unsigned)(hi & 0xFFFFFFFFull)), __uint_as_float((unsigned)(hi >> 32)));
    reinterpret_cast<float4 *>(const_cast<float *>(p.recExtra))[i] =
        make_float4(getc(o, p.rayDir), __uint_as_float(k), __uint_as_float((unsigned)(lo & 0xFFFFFFFFull)), __uint_as_float((unsigned)(lo >> 32)));
  } else {
    rec[0] = make_float4(o.x, o.y, o.z, d.x);
    rec[1] = make_float4(d.y, d.z, __uint_as_float(i), __uint_as_float(k));
  }
  return slot;
}

// Fixed number of source draws (no tilted primary direction): the NS engine outputs
// the source sample needs are produced straight into registers by one 156+NS-step pass
// of the seeding recurrence, which also leaves the streaming cursors for the trace kernel.
// The bin cursor's returning atomic is the one long latency of a ray; it is issued as soon as the ray's bin is
// known and its answer is used one loop pass later, after the NEXT ray's seeding chain: the wave computes while
// its own atomic is under way instead of leaving that to the other waves of the SIMD.
// RELIEF: the sort key of bin_of_relief and its second, loose set of bins
template <int D, bool KEEP, bool RELIEF> __global__ __launch_bounds__(VR_BLOCK) void gen_kernel(const TraceParams p) {
  constexpr int NS = D == 3 ? 4 : 3;
  const bool binned = p.binCount && !(p.debugFlags & 64u); // flag 64: timing experiment, no binning
  bool havePrev = false;
  V3 po = mk(0, 0, 0), pd = mk(0, 0, 1);
  unsigned pi = 0, pbin = 0, ppos = 0;
  u64 phi = 0;
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
        b = RELIEF ? bin_of_relief<D>(p, o, project_dir<D>(d)) : bin_of<D>(p, o, project_dir<D>(d));
    }
    if (havePrev) { // the previous ray of this lane: its slot has arrived
      unsigned slot = pi;
      if (binned) {
        if (RELIEF && (pbin >> 31)) { // a loose bin: its slots and its overflow region lie behind the tight bins'
          const unsigned lb = pbin & 0x7FFFFFFFu;
          if (ppos < p.binCap)
            slot = p.looseSlotBase + lb * p.binCap + ppos;
          else
            slot = p.looseSlotBase + p.looseNumBins * p.binCap + atomicAdd(&p.binCount[p.looseCntBase + p.looseNumBins], 1u);
        } else if (ppos < p.binCap)
          slot = pbin * p.binCap + ppos;
        else
          slot = p.numBins * p.binCap + atomicAdd(&p.binCount[p.numBins], 1u); // < ovCap by construction
      }
      // record = {A, B} (32 B) [+ the RNG cursors {s[k], s[k+156]} (16 B) when the particle keeps going]
      if (KEEP) {
        // compact: the origin's two free coordinates, the direction, the index and s[k+156]; the tracer knows the source
        // plane and the draw count and rebuilds s[k] from the seed (vr_types.hpp)
        float4 *rec = reinterpret_cast<float4 *>(p.slotRec) + (size_t)2 * slot;
        rec[0] = make_float4(getc(po, p.firstDir), getc(po, p.secondDir), pd.x, pd.y);
        rec[1] = make_float4(pd.z, __uint_as_float(pi), __uint_as_float((unsigned)(phi & 0xFFFFFFFFull)),
                             __uint_as_float((unsigned)(phi >> 32)));
      } else {
        float4 *rec = reinterpret_cast<float4 *>(p.slotRec) + (size_t)2 * slot;
        rec[0] = make_float4(po.x, po.y, po.z, pd.x);
        rec[1] = make_float4(pd.y, pd.z, __uint_as_float(pi), __uint_as_float((unsigned)NS));
      }
    }
    if (!cur)
      break;
    if (binned)
      ppos = atomicAdd(&p.binCount[(RELIEF && (b >> 31)) ? p.looseCntBase + (b & 0x7FFFFFFFu) : b], 1u); // (answer used in the next pass)
    po = o;
    pd = d;
    pi = i;
    pbin = b;
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

#endif // VR_USER_MODULE

__device__ __forceinline__ unsigned long long wave_sum(unsigned v) {
  unsigned long long s = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    s += __shfl_down(s, off, 64);
  return s;
}

// Credit `wfx` to accumulator `pos` for every lane with `cond`; lanes of the wave that
// credit the same accumulator with the same weight are merged into one atomic
// (sorted rays: a wavefront's hits fall on a handful of disks).
__device__ __forceinline__ void credit_aggregated(unsigned long long *acc, bool cond, unsigned pos, u64 wfx) {
  unsigned long long todo = ballot64(cond);
  const unsigned lane = threadIdx.x & 63u;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned P = __shfl(pos, leader, 64);
    const unsigned wlo = __shfl((unsigned)(wfx & 0xFFFFFFFFull), leader, 64);
    const unsigned whi = __shfl((unsigned)(wfx >> 32), leader, 64);
    const u64 W = ((u64)whi << 32) | wlo;
    const unsigned long long same = ballot64(cond && pos == P && wfx == W);
    if ((int)lane == leader)
      atomicAdd(&acc[P], W * (u64)__popcll(same));
    todo &= ~same;
  }
}

// Every lane of the wave adds `wfx` (0: nothing) to the SAME accumulator `pos`: one integer wave sum (exact, order
// independent) and one atomic.  Must be reached by the whole wave.
__device__ __forceinline__ void credit_wave_sum(unsigned long long *acc, unsigned pos, u64 wfx) {
  u64 s = wfx;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    s += (u64)__shfl_down((unsigned long long)s, off, 64);
  if ((threadIdx.x & 63u) == 0u && s)
    atomicAdd(&acc[pos], s);
}

__device__ __forceinline__ unsigned long long bcast64(unsigned long long v) {
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xFFFFFFFFull));
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// Boundary::processHit (rayBoundary.hpp:29-127): what a hit of wall triangle `prim` at hitPoint does to the
// ray.  Shared by trace_kernel and the debug entry point that checks the reference's boundaryHit known answers.
template <int D>
__device__ __forceinline__ void process_boundary_hit(const TraceParams &p, const float *__restrict__ wallS, unsigned prim,
                                                     const V3 &hitPoint, V3 &org, V3 &rayDirection, V3 &dir,
                                                     bool &active) {
  const float *w = wallS + 12 * prim;
  V3 ng = mk(w[9], w[10], w[11]);
  if (vdot(dir, ng) > 0.f) { // back side: pass through
    org = hitPoint;
    return;
  }
  int bc, axis;
  bool minWall;
  if (D == 2 || prim <= 3u) {
    bc = p.bc0;
    axis = p.firstDir;
    minWall = prim <= 1u;
  } else {
    bc = p.bc1;
    axis = p.secondDir;
    minWall = prim <= 5u;
  }
  if (bc == 0) { // REFLECTIVE, rayBoundary.hpp:261-271
    vnormalize(ng);
    rayDirection = reflect_specular(rayDirection, ng);
    dir = project_dir<D>(rayDirection);
    org = hitPoint;
  } else if (bc == 1) { // PERIODIC: wrap to the opposite face
    org = hitPoint;
    const bool first = (D == 2 || prim <= 3u);
    const float wrapTo = first ? (minWall ? p.hi1 : p.lo1) : (minWall ? p.hi2 : p.lo2);
    setc(org, axis, wrapTo);
  } else { // IGNORE
    active = false;
  }
}

// "Segments that rise clear" (trace_kernel): does the height field over the source plane (HeightFieldParams, the launch
// frame's VR_F_HF_*) say that a ray starting at `org` cannot meet the geometry?  It is above its tile's height — the highest
// point of anything in the tile or its eight neighbours — from tnear on, and rises above the whole scene before it has
// travelled a tile sideways.  One look-up, no loop.
template <int D>
__device__ __forceinline__ bool rises_clear(const float *__restrict__ wallS, const V3 &org, const V3 &dir, float tnear) {
  const int hnx = __float_as_int(wallS[VR_F_HF_NX]);
  if (hnx <= 0)
    return false;
  const int ax = __float_as_int(wallS[VR_F_RAYDIR]), a1 = __float_as_int(wallS[VR_F_FIRSTDIR]), a2 = __float_as_int(wallS[VR_F_SECONDDIR]);
  const float sgn = wallS[VR_F_HF_SIGN];
  const float up = sgn * getc(dir, ax);
  if (!(up > 0.f))
    return false;
  const float hz = sgn * getc(org, ax);
  const float invT = wallS[VR_F_HF_INVT];
  const int hny = __float_as_int(wallS[VR_F_HF_NY]);
  int ix = (int)floorf((getc(org, a1) - wallS[VR_F_HF_LO1]) * invT);
  ix = ix < 0 ? 0 : (ix >= hnx ? hnx - 1 : ix);
  int iy = 0;
  float d2 = 0.f;
  if (D == 3) {
    iy = (int)floorf((getc(org, a2) - wallS[VR_F_HF_LO2]) * invT);
    iy = iy < 0 ? 0 : (iy >= hny ? hny - 1 : iy);
    d2 = getc(dir, a2);
  }
  typedef const __attribute__((address_space(1))) float *GlobalF;
  const GlobalF field = reinterpret_cast<GlobalF>(((unsigned long long)__float_as_uint(wallS[VR_F_HF_PTR_HI]) << 32) |
                                                  __float_as_uint(wallS[VR_F_HF_PTR_LO]));
  const float height = field[iy * hnx + ix];
  const float d1 = getc(dir, a1);
  const float tTop = fmaxf(wallS[VR_F_HF_TOP] - hz, 0.f) / up; // where the ray passes the top of the scene box
  return hz + up * tnear > height && tTop * sqrtf(d1 * d1 + d2 * d2) <= 0.99f * wallS[VR_F_HF_TILE];
}

// ---------------------------------------------------------------------------
// trace_kernel
//   ABSORB: every hit absorbs the whole weight (sticking >= 1 everywhere), so
//   nothing after the first surface hit is observable and the reflection /
//   roulette code (and its RNG) is compiled out.
// ---------------------------------------------------------------------------
// (SGPR budget: 256-thread blocks per CU = min(8, 800 / (ceil(sgpr/16)*16 + 16)) on gfx950,
//  MI355X_MICROARCH.md; 80 keeps 8 blocks resident)
// MODE 0: general kernel.  MODE 1: absorbing, flat scene (packets carry the load).  MODE 2:
// absorbing, structured scene (most rounds end in per-lane walks): straggler carry-over on.
// MODE 3: general kernel for a flat scene: like 0, with the packet query's wave-uniform crediting.
// MODE 4: MODE 0 for scenes of a few hundred primitives (2-D simulations): pair nodes, primitive records,
// neighbourhood and flux accumulators are staged in LDS (VR_SMALL_LDS bytes per block) and every access of the
// round but the ray records stays there; no packets (a per-lane walk over LDS nodes is cheaper than their set-up).
#ifndef VR_GENERAL_WAVES
#define VR_GENERAL_WAVES 6 // waves per SIMD of the general kernel (MODE 0)
#endif
#ifndef VR_FLAT_ORDERED
#define VR_FLAT_ORDERED 0  // MODE 3 walks like MODE 1: the escape-link walk, no carry-over (1: the ordered pair walk).  It runs
                           // only on scenes whose box is thin (vr_api.cpp: flatScene), where a wave walks in the 5 % of its
                           // rounds whose query gives up; without the walk's 12 KB of LDS stack and ~10 VGPRs the kernel
                           // takes 6 waves per SIMD: C2 0.1 10.9 -> 10.0 ms.  (Forced onto a scene with relief
                           // — VR_GENERAL_FLAT=1 — it is 15 - 30 % slower than with the ordered walk at 5 waves.)
#endif
#ifndef VR_SMALL_WAVES
#define VR_SMALL_WAVES 5   // ... of the LDS-resident kernel (MODE 4; 6: C5 20.15 -> 21.9 ms)
#endif
#ifndef VR_PQ_CACHE
#define VR_PQ_CACHE 1      // flat-scene kernels: the packet query's frontier serves the neighbouring rounds (pq_hit_packet CACHE)
#endif
#ifndef VR_PQ_WALLS_FIRST
#define VR_PQ_WALLS_FIRST 1 // flat-scene kernels: a ray that meets a side wall before the scene box stays out of the packet query's box
#endif
#ifndef VR_PQ_CACHE_RELIEF
#define VR_PQ_CACHE_RELIEF 1 // ... in the relief kernels MODE 5 / 6 too
#endif
#ifndef VR_FLAT_WAVES
#define VR_FLAT_WAVES 6    // ... of the general flat-scene kernel (MODE 3)
#endif
// MODE 5 / 6: MODE 1 / 3 for a scene that is flat WITH RELIEF: the packet query clips its rays to the local relief
// (relief_clip, vr_device.hpp) instead of to the scene box; the generator has filed the grazing rays apart (TraceParams,
// round 4), and those are traced by a MODE 2 / 0 launch of their own.
#ifndef VR_RELIEF_WAVES
#define VR_RELIEF_WAVES 8  // waves per SIMD of the absorbing relief kernel (MODE 5)
#endif
// MODE 7: MODE 0 that also resumes the rays of MODE 6's spill queue (TraceParams::spillRec) behind its bins
constexpr unsigned VR_SPILL_BLOCK = 64u; // records of the spill queue a wave reserves at a time (TraceParams::spillRec)
// the unused records [used, VR_SPILL_BLOCK) of a wave's block marked empty (word 11 = ~0: no ray)
__device__ __forceinline__ void spill_pad(const TraceParams &p, unsigned base, unsigned used, unsigned lane) {
  if (used < VR_SPILL_BLOCK && lane >= used)
    reinterpret_cast<float4 *>(p.spillRec)[4 * (size_t)(base + lane) + 2] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
}
constexpr int vr_mode_waves(int m) {
  return m == 5 ? VR_RELIEF_WAVES : m == 1 ? 8 : (m == 2 ? 7 : ((m == 3 || m == 6) ? VR_FLAT_WAVES : (m == 4 ? VR_SMALL_WAVES : VR_GENERAL_WAVES)));
}
template <int D, int GEO, int PARTICLE, int MODE_>
__global__ __launch_bounds__(VR_BLOCK) __attribute__((amdgpu_num_sgpr(80)))
__attribute__((amdgpu_waves_per_eu(vr_mode_waves(MODE_), vr_mode_waves(MODE_)))) void
trace_kernel(const TraceParams p) {
  constexpr bool SMALL = MODE_ == 4;
  constexpr bool RELIEF = MODE_ == 5 || MODE_ == 6;
  constexpr bool RESUME = MODE_ == 7;
  constexpr int MODE = (SMALL || RESUME) ? 0 : (MODE_ == 5 ? 1 : (MODE_ == 6 ? 3 : MODE_));
  constexpr bool FRAME_LDS = MODE == 1; // (the wall / scene-box frame from LDS: hit_walls_lds, vr_device.hpp)
  constexpr bool FOLLOW = MODE == 3;    // (follow-up segments inside the round of a packet query: end of the round)
  constexpr bool ABSORB = MODE == 1 || MODE == 2;
  // PARTICLE 0 / 1: DiffuseParticle / SpecularParticle compiled in.  PARTICLE 2 (P_EXT): the
  // extended kernel — particle kind, data labels, WDIST crediting and mean-free-path scattering
  // decided at run time from TraceParams (vr_particles.hpp)
  constexpr bool EXT = PARTICLE >= P_EXT;           // (P_EXT, P_EXT_FULL)
  constexpr bool EXT_FULL = PARTICLE == P_EXT_FULL; // ... with the coned-cosine model, WDIST crediting and the mean free path
  // packet-query rounds credit disks wave-uniformly from the candidate list (pq_credit) instead of
  // walking the neighbour CSR per lane
  constexpr bool PQ_CREDIT = GEO == 0 && !EXT_FULL && (MODE == 1 || MODE == 3);
  PqCands cands;
  cands.local = 0ull;
  cands.count = 0u;
  cands.box = false;
  cands.mine = 0u;
  cands.rec = nullptr;
  // CARRY: lanes whose BVH walk is still under way when most of the wave is done keep
  // their cursor over the state-machine / refill phase (see the round structure below).
  // The absorbing kernel for flat scenes does without: its rounds are packets, and the extra
  // live registers would cost it the 8th wave per SIMD.
  constexpr bool CARRY = MODE != 1 && (MODE != 3 || VR_FLAT_ORDERED);
  __shared__ float wallS[VR_WALL_TABLE]; // (96 .. : the launch's scalar frame, vr_device.hpp)
  // per-lane event counters live in LDS (fire-and-forget ds_add), not in 8 VGPRs.  (Five of them counted per WAVE in
  // scalar registers — 5 KB of LDS less, room for a 7th block per CU — was built and measured in round 3: slower at
  // 7 waves per SIMD and at 6; removed.)
  __shared__ unsigned cntS[8 * VR_BLOCK];
  __shared__ unsigned pqS[(VR_BLOCK / 64) * 128]; // packet query: per-wave frontier lists
  constexpr bool PQ_CACHE = (MODE == 1 || MODE == 3) && (!RELIEF || VR_PQ_CACHE_RELIEF) && VR_PQ_CACHE != 0; // (pq_hit_packet CACHE: flat-scene kernels)
  __shared__ float pqBoxS[PQ_CACHE ? (VR_BLOCK / 64) * 6 * VR_PQ_KEEP : 1];         // ... the kept leaf nodes' boxes, VR_PQ_KEEP x 6 per wave
  __shared__ uint4 candS[PQ_CREDIT ? (VR_BLOCK / 64) * VR_PQ_RECORDS : 1]; // ... and candidate records (pq_credit)
  // ... and, where the credits of a round carry different weights (the general kernels), one int64 sum per candidate
  // and data label (two labels here; further ones are summed over the wave in registers)
  constexpr bool PQ_SUMS = PQ_CREDIT && !ABSORB;
  constexpr unsigned PQ_LAB = EXT ? 2u : 1u;
  __shared__ unsigned long long candAccS[PQ_SUMS ? (VR_BLOCK / 64) * VR_PQ_CANDS * PQ_LAB : 1];
  // per-lane stack of the ordered walk, [entry][lane]; the absorbing flat-scene kernel walks rarely and keeps its
  // 8 waves per SIMD with a short LDS part (deeper entries: global slab)
  constexpr bool ORDERED = MODE != 1 && (MODE != 3 || VR_FLAT_ORDERED); // (MODE 1 walks rarely: it keeps the escape-link walk, one register of state)
  constexpr int SD = SMALL ? VR_SMALL_STACK : VR_STACK_LDS;
  __shared__ unsigned stackS[ORDERED ? SD * VR_BLOCK : 1];
  // (MODE 4: the scene copy is the kernel's dynamic LDS — smallBytes of it, so a smaller scene leaves room for a
  //  fifth block per CU)
  extern __shared__ uint4 sceneS[];
  unsigned char *const sceneB = reinterpret_cast<unsigned char *>(sceneS);
  const unsigned tid = threadIdx.x;
  // (the wave's index as a SCALAR: the per-wave tables' addresses are then wave-uniform values the compiler keeps in
  //  SGPRs — as per-lane values one of them was spilled and came back from scratch three times a round, each reload
  //  waiting for every atomic and load the wave had in flight)
  const unsigned waveInBlock = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  cands.rec = (VR_LDS U4 *)(candS + (PQ_CREDIT ? waveInBlock * VR_PQ_RECORDS : 0u));
  const unsigned lane = tid & 63u;
  const unsigned gwave = (blockIdx.x * VR_BLOCK + tid) >> 6;
  if (tid < VR_WALL_TABLE)
    wallS[tid] = p.wallTable[tid];
  if (tid == VR_F_EXTRA_LO || tid == VR_F_EXTRA_HI) // (read per lane at a refill: as a kernel argument the pointer would be held in SGPRs throughout)
    wallS[tid] = __uint_as_float((unsigned)((unsigned long long)p.recExtra >> (tid == VR_F_EXTRA_LO ? 0 : 32)));
#pragma unroll
  for (int k = 0; k < 8; ++k)
    cntS[k * VR_BLOCK + tid] = 0u;
  if constexpr (PQ_CACHE) {
    if (tid < VR_BLOCK / 64)
      pqS[tid * 128 + 39] = 0u; // (no cached frontier yet: pq_hit_packet CACHE)
  }
  if (SMALL) {
    // stage the scene (the offsets are multiples of 16 bytes; vr_apply_prepare checked that it fits)
    const uint4 *gn = reinterpret_cast<const uint4 *>(p.pnodes);
    uint4 *ln = reinterpret_cast<uint4 *>(sceneB + p.smallOff[0]);
    for (unsigned k = tid; k < 2u * p.numNodes; k += VR_BLOCK)
      ln[k] = gn[k];
    const uint4 *gp = reinterpret_cast<const uint4 *>(p.prims);
    uint4 *lp = reinterpret_cast<uint4 *>(sceneB + p.smallOff[1]);
    for (unsigned k = tid; k < (GEO == 0 ? 2u : 4u) * p.numPrims; k += VR_BLOCK)
      lp[k] = gp[k];
    unsigned *lo = reinterpret_cast<unsigned *>(sceneB + p.smallOff[2]);
    for (unsigned k = tid; k <= p.numPrims; k += VR_BLOCK)
      lo[k] = p.nbOff[k];
    unsigned *li = reinterpret_cast<unsigned *>(sceneB + p.smallOff[3]);
    for (unsigned k = tid; k < p.smallNb; k += VR_BLOCK)
      li[k] = p.nbIds[k];
    unsigned long long *lf = reinterpret_cast<unsigned long long *>(sceneB + p.smallOff[4]);
    for (unsigned k = tid; k < p.numPrims * p.numData; k += VR_BLOCK) // (one plane per data label)
      lf[k] = 0ull;
    if (p.primSticking) {
      float *ls = reinterpret_cast<float *>(sceneB + p.smallOff[5]);
      for (unsigned k = tid; k < p.numPrims; k += VR_BLOCK)
        ls[k] = p.primSticking[k];
    }
  }
  __syncthreads();
  unsigned *const cnt = cntS + tid; // counter k of this lane: cnt[k * VR_BLOCK]
  enum { K_BOUNDARY = 0, K_REFL, K_TIER2, K_TRACES, K_NONGEO, K_GEO, K_TERM, K_PARTICLE };
#define VR_COUNT(k, v) atomicAdd(&cnt[(k)*VR_BLOCK], (unsigned)(v))

  // scene data: global memory, or (MODE 4) the block's LDS copies
  const float4 *__restrict__ prims = SMALL ? reinterpret_cast<const float4 *>(sceneB + p.smallOff[1])
                                           : reinterpret_cast<const float4 *>(p.prims);
  const uint4 *__restrict__ pnodes = SMALL ? reinterpret_cast<const uint4 *>(sceneB + p.smallOff[0])
                                           : reinterpret_cast<const uint4 *>(p.pnodes);
  const unsigned *__restrict__ nbOff = SMALL ? reinterpret_cast<const unsigned *>(sceneB + p.smallOff[2]) : p.nbOff;
  const unsigned *__restrict__ nbIds = SMALL ? reinterpret_cast<const unsigned *>(sceneB + p.smallOff[3]) : p.nbIds;
  const float *__restrict__ primSticking = SMALL ? reinterpret_cast<const float *>(sceneB + p.smallOff[5]) : p.primSticking;
  const float4 *__restrict__ rayAB = reinterpret_cast<const float4 *>(p.slotRec);
  unsigned long long *const fluxGlobal = p.fluxAcc + (size_t)(blockIdx.x & p.accMask) * p.accStride; // this block's replica
  unsigned long long *const fluxAcc = SMALL ? reinterpret_cast<unsigned long long *>(sceneB + p.smallOff[4]) : fluxGlobal;
  const float tnear = 1e-4f; // rayUtil.hpp:229-231

  // per-lane ray state
  bool active = false;
  // `dir` is what the intersector sees: the 2-D projection of rayDirection (rayUtil.hpp:204-227),
  // i.e. rayDirection itself in 3-D (then the same registers)
  V3 org = mk(0, 0, 0), rayDirection = mk(0, 0, 1), dir2 = mk(0, 0, 1);
  V3 &dir = D == 3 ? rayDirection : dir2;
  float rayWeight = 0.f;
  // The two per-ray counters are touched once per segment: in the general kernels they live in LDS, not in two of
  // the 80 VGPRs (left to the register allocator they went to scratch, and a scratch reload waits on the
  // vector-memory counter: for every load and atomic the wave has in flight).
  constexpr bool COLD_IN_LDS = !ABSORB && !SMALL;
  __shared__ unsigned coldS[COLD_IN_LDS ? 3 * VR_BLOCK : 1];
  unsigned numReflectionsR = 0, boundaryHitsR = 0;
  unsigned &numReflections = COLD_IN_LDS ? coldS[tid] : numReflectionsR;
  unsigned &boundaryHits = COLD_IN_LDS ? coldS[VR_BLOCK + tid] : boundaryHitsR;
  // Source::getInitialRayWeight(idx) (rayTraceKernel.hpp:124): 1 for every built-in source; a host-callback source may
  // hand over its own (p.hostWeights, wave-uniform test).  Read again only by the roulette's thresholds.
  unsigned initWeightR = 0x3F800000u;
  unsigned &initWeightBits = COLD_IN_LDS ? coldS[2 * VR_BLOCK + tid] : initWeightR;
  bool hitFromBack = false;
  bool start = false; // this lane begins a new trace segment in this round
  // (MODE 6) this wave's block of the spill queue: first record and records used (wave-uniform)
  unsigned spillBase = 0u, spillUsed = VR_SPILL_BLOCK;
  unsigned node = VR_END; // cursor of the lane's BVH walk (VR_END: none under way)
  unsigned sp = 0u;       // ... and the depth of its stack
  unsigned *const stackG = p.walkStack + (size_t)gwave * (VR_STACK_GLOBAL * 64u) + lane;
  HitRec h;               // closest hit so far of the lane's current segment
  h.t = 0.f;
  h.geom = -1;
  h.prim = 0u;
  h.pos = 0u;
  Rng rng;
  rng_resume(rng, 0u, 0u, 0ull, 0ull);
  rng.scratch = p.rngScratch + (size_t)gwave * (312u * 64u) + lane;
  // wave-uniform cursor over the sort bins: [curBin, spanEnd) is the span of (virtual)
  // bins this wave pulled from the queue; bins >= numBins are 64-ray chunks of the
  // overflow region
  typedef const unsigned __attribute__((address_space(4))) *ConstU32;
  ConstU32 binCount = (ConstU32)p.binCount;
  const unsigned ovCount = binCount[p.numBins] < p.ovCap ? binCount[p.numBins] : p.ovCap;
  const unsigned ovChunks = (ovCount + p.binCap - 1) / p.binCap;
  // (MODE 7: the spill queue's records, in chunks of a bin's capacity, are the virtual bins behind the overflow chunks)
  const unsigned spillN = (RESUME && p.spillRec) ? ((ConstU32)p.spillCount)[0] : 0u;
  // (the queue is made of 64-record blocks, each with a ray in its first record and possibly unused records at its end:
  //  a chunk of at least a block, block aligned, so that a refill which finds no ray has truly run out of work)
  const unsigned spillChunk = p.binCap < VR_SPILL_BLOCK ? VR_SPILL_BLOCK : (p.binCap / VR_SPILL_BLOCK) * VR_SPILL_BLOCK;
  const unsigned totalBins = p.numBins + ovChunks + (RESUME ? (spillN + spillChunk - 1) / spillChunk : 0u);
  unsigned curBin = 0, spanStart = 0, spanEnd = 0, curOff = 0, curCnt = 0, curBase = 0;
  unsigned spanCounts = 0; // lane i: ray count of bin spanStart + i
  // (only the general flat-scene kernel has the queues compiled in — it is the one they pay for, vr_api.cpp — the others
  //  keep the single queue's code: MODE 1 with the bookkeeping: L2 hit rate 74 -> 84 % but 6.60 -> 6.83 ms from the
  //  extra scalar spills of a kernel at 8 waves per SIMD)
  constexpr bool MULTIQ = MODE == 3;
  const unsigned numQueues = MULTIQ ? p.numQueues : 1u;
  const unsigned myQueue = MULTIQ ? (blockIdx.x & (numQueues - 1u)) : 0u; // (numQueues is 1 or 8)
  unsigned qTried = 0;                      // queues this wave has found empty (wave-uniform)
  unsigned packetSkip = 0, packetFails = 0; // wave-uniform back-off of packet attempts
  unsigned pqSkip = 0, pqFails = 0;         // ... and of packet-query attempts
  bool exhausted = false;
  VR_DIAG_DECL
#ifdef VR_DIAG
  __shared__ unsigned long long phaseS[(VR_BLOCK / 64) * 16];
  unsigned long long *const phaseT = phaseS + (tid >> 6) * 16;
  if (lane < 16)
    phaseT[lane] = 0ull;
  unsigned long long tLast = __builtin_amdgcn_s_memtime();
#endif

  for (;;) {
    // keep the compiler from hoisting the (loop-invariant) LDS wall table into
    // ~100 registers: occupancy matters more than 24 ds_reads per segment
    asm volatile("" ::: "memory");
    // ---- wave-wide compaction / restart: idle lanes pull the next sorted rays ----
    // Two steps: first every idle lane is ASSIGNED a record slot — a wave-uniform walk over the next
    // bins of the span, no memory but the (rare) grab of a new span — then all of them load at once.
    // (Loading bin by bin cost one full HBM round trip per bin: a round of the absorbing kernel
    //  swallows two or three bins.)
    {
      const unsigned long long idle = ballot64(!active);
      const unsigned need = (unsigned)__popcll(idle);
      const unsigned rank = (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
      unsigned slot = 0xFFFFFFFFu;
      unsigned assigned = 0;
      // (at most 12 bin changes per round — unless the wave has nothing at all to do: it owns its span, and
      //  leaving with bins of it unread would lose their rays)
      for (int adv = 0; assigned < need && (adv < 12 || (need == 64u && assigned == 0u));) {
        if (curOff >= curCnt) { // current bin used up: next bin of the span, or a new span
          ++adv;
          if (curBin + 1 >= spanEnd || spanEnd == 0) {
            if (exhausted)
              break;
            // One queue of bins PER XCD: queue q owns the q-th eighth of the (spatially ordered) bins, the waves of
            // an XCD (blockIdx & 7 labels the blocks that share one) drain their own queue first and then help
            // with the others.  The rounds that follow each other on an XCD are then neighbours in space: the
            // primitive records one round pulled into the XCD's L2 serve the next (one global queue dealt
            // neighbouring spans to all eight L2s: 37 line misses per C2 round beyond its ray records).
            if constexpr (!MULTIQ) {
              unsigned long long s = 0;
              if (lane == 0)
                s = atomicAdd(p.workCounter, (unsigned long long)p.chunk);
              s = bcast64(s);
              if (s >= totalBins) {
                exhausted = true;
                break;
              }
              curBin = spanStart = (unsigned)s;
              spanEnd = (unsigned)((s + p.chunk < totalBins) ? s + p.chunk : totalBins);
            } else {
              unsigned lo = 0, hi = 0;
              for (; qTried < numQueues; ++qTried) {
                const unsigned q = (myQueue + qTried) & (numQueues - 1u);
                const unsigned qLo = (unsigned)((unsigned long long)totalBins * q / numQueues);
                const unsigned qHi = (unsigned)((unsigned long long)totalBins * (q + 1u) / numQueues);
                unsigned long long s = 0;
                if (lane == 0)
                  s = atomicAdd(p.workCounter + (size_t)q * VR_QUEUE_STRIDE, (unsigned long long)p.chunk);
                s = bcast64(s);
                if (s < (unsigned long long)(qHi - qLo)) {
                  lo = qLo + (unsigned)s;
                  hi = (lo + p.chunk < qHi) ? lo + p.chunk : qHi;
                  break;
                }
              }
              if (lo == hi) { // every queue is empty
                exhausted = true;
                break;
              }
              curBin = spanStart = lo;
              spanEnd = hi;
            }
            // the span's bin counts in one coalesced load (lane i <- bin spanStart + i; chunk <= 64)
            const unsigned bi = spanStart + lane;
            spanCounts = (bi < spanEnd && bi < p.numBins) ? p.binCount[bi] : 0u;
          } else {
            ++curBin;
          }
          curOff = 0;
          if (curBin < p.numBins) {
            const unsigned c = __shfl(spanCounts, (int)(curBin - spanStart), 64);
            curCnt = c < p.binCap ? c : p.binCap;
            curBase = curBin * p.binCap;
          } else if (!RESUME || curBin < p.numBins + ovChunks) {
            const unsigned k = (curBin - p.numBins) * p.binCap;
            curCnt = ovCount - k < p.binCap ? ovCount - k : p.binCap;
            curBase = p.numBins * p.binCap + k;
          } else { // a chunk of the spill queue: bit 31 marks its record numbers
            const unsigned k = (curBin - p.numBins - ovChunks) * spillChunk;
            curCnt = spillN - k < spillChunk ? spillN - k : spillChunk;
            curBase = 0x80000000u | k;
          }
          curCnt = __builtin_amdgcn_readfirstlane(curCnt);
          continue;
        }
        const unsigned avail = curCnt - curOff;
        const unsigned take = avail < need - assigned ? avail : need - assigned;
        if (!active && rank >= assigned && rank < assigned + take)
          slot = curBase + curOff + (rank - assigned);
        curOff += take;
        assigned += take;
      }
      bool resumed = false;
      if (RESUME && slot != 0xFFFFFFFFu && (slot >> 31)) {
        // a ray of the spill queue: its whole state as the relief kernel left it
        const float4 *__restrict__ sr = reinterpret_cast<const float4 *>(p.spillRec) + 4 * (size_t)(slot & 0x7FFFFFFFu);
        const float4 r0 = sr[0], r1 = sr[1], r2 = sr[2], r3 = sr[3];
        const bool ray = __float_as_uint(r2.w) != 0xFFFFFFFFu; // (the unused end of a wave's block: spill_pad)
        org = mk(r0.x, r0.y, r0.z);
        rayWeight = r0.w;
        rayDirection = mk(r1.x, r1.y, r1.z);
        dir = project_dir<D>(rayDirection);
        rng_resume(rng, __float_as_uint(r1.w), __float_as_uint(r2.x),
                   ((u64)__float_as_uint(r3.y) << 32) | __float_as_uint(r3.x), ((u64)__float_as_uint(r3.w) << 32) | __float_as_uint(r3.z));
        numReflections = __float_as_uint(r2.y);
        boundaryHits = __float_as_uint(r2.z) & 0x7FFFFFFFu;
        hitFromBack = (__float_as_uint(r2.z) >> 31) != 0u;
        active = ray;
        start = ray;
        resumed = true;
      }
      if (slot != 0xFFFFFFFFu && !resumed) {
        DIAG(8);
        const unsigned j = slot;
        const float4 a = rayAB[2 * (size_t)j]; // (32-byte records in both forms, vr_types.hpp)
        const float4 b = rayAB[2 * (size_t)j + 1];
        if (!ABSORB) {
          // compact form.  (The source frame comes from LDS, per lane: as kernel arguments these loop-invariant scalars
          //  were hoisted and held across the whole kernel — scalar spills in every instantiation.)
          float srcPlane = wallS[VR_F_SRC_PLANE];
          const int rd = __float_as_int(wallS[VR_F_RAYDIR]), fd = __float_as_int(wallS[VR_F_FIRSTDIR]);
          const unsigned long long ex = ((unsigned long long)__float_as_uint(wallS[VR_F_EXTRA_HI]) << 32) | __float_as_uint(wallS[VR_F_EXTRA_LO]);
          const unsigned seed32 = tea3((unsigned)(p.batchFirst + __float_as_uint(b.y)), p.seed);
          constexpr unsigned NS = D == 3 ? 4u : 3u; // draws of the plain generator (gen_kernel)
          unsigned k = NS;
          u64 lo;
          if (ex) { // a source with its own origin plane / draw count: the side array has them and s[k]
            typedef float F4 __attribute__((ext_vector_type(4)));
            const F4 e = reinterpret_cast<const __attribute__((address_space(1))) F4 *>(ex)[__float_as_uint(b.y)]; // (global, not flat)
            srcPlane = e.x;
            k = __float_as_uint(e.y);
            lo = ((u64)__float_as_uint(e.w) << 32) | __float_as_uint(e.z);
          } else {
            lo = seed32; // s[NS]: NS steps of the seeding recurrence
#pragma unroll
            for (unsigned st = 1; st <= NS; ++st)
              lo = mt_step(lo, st);
          }
          org.x = rd == 0 ? srcPlane : (fd == 0 ? a.x : a.y);
          org.y = rd == 1 ? srcPlane : (fd == 1 ? a.x : a.y);
          org.z = rd == 2 ? srcPlane : (fd == 2 ? a.x : a.y);
          rayDirection = mk(a.z, a.w, b.x);
          rng_resume(rng, seed32, k, lo, ((u64)__float_as_uint(b.w) << 32) | __float_as_uint(b.z));
        } else {
          org = mk(a.x, a.y, a.z);
          rayDirection = mk(a.w, b.x, b.y);
        }
        dir = project_dir<D>(rayDirection); // what Embree sees (rayUtil.hpp:204-227)
        rayWeight = 1.f;                    // Source::getInitialRayWeight
        if (!ABSORB && p.hostWeights) {     // (a host-callback source with weights of its own never runs an absorbing kernel)
          rayWeight = p.hostWeights[p.batchFirst + __float_as_uint(b.y)];
          initWeightBits = __float_as_uint(rayWeight);
        }
        numReflections = 0;
        boundaryHits = 0;
        hitFromBack = false;
        active = true;
        start = true;
      }
    }
    if (!ballot64(active))
      break;
    TICK(0);

    // ---- closest hit of a trace segment (rtcIntersect1, rayTraceKernel.hpp:163-167) ----
    // A round: if the whole wave begins a segment together
    // (freshly sorted, coherent rays) it first tries the wave-uniform packet traversal with
    // a bounded number of node visits; otherwise, and when the packet gives up, every lane
    // walks its own path — but only until the number of lanes still walking drops below
    // p.walkExit: the lanes that are done run the state machine and start their next
    // segment (or pull a new ray) while the stragglers keep their cursor and closest hit
    // for the next round, so one long walk does not idle the other 63 lanes.
    if (active) {
      DIAG(0);
    }
    if (start) {
      DIAG(9);
    }
    if (!CARRY || start) { // (!CARRY: every active lane starts a segment in every round)
      hit_clear(h);
      node = 0u;
    }
    const unsigned long long carried = CARRY ? ballot64(active && !start) : 0ull;
    start = false;
    const bool usePacket = !SMALL &&
        !(p.debugFlags & 32u) && carried == 0ull && packetSkip == 0 && __popcll(ballot64(active)) >= 8;
    bool packetDone = false;
    bool pqCredit = false; // this round's surface hits are credited from the packet's candidate list
    if (!SMALL && usePacket && p.wide && !(p.debugFlags & 128u)) {
      // first choice: the box query (one wide-tree search for the whole wave)
      if (pqSkip == 0) {
        if (active) {
          DIAG(12);
        }
        // (the walls first for the flat-scene kernels' queries: see pq_hit_packet, tWall.  The conservative pre-tests of
        //  hit_walls let only the rays near a side wall through to the exact test)
        float tWall = 3.402823466e+38f;
        if constexpr ((MODE == 1 || MODE == 3) && VR_PQ_WALLS_FIRST) {
          if (active && !(p.debugFlags & 65536u)) { // (flag 65536: off, for comparison)
            HitRec hw;
            hit_clear(hw);
            if constexpr (FRAME_LDS)
              hit_walls_lds(p, wallS, org, dir, tnear, hw);
            else
              hit_walls(p, wallS, org, dir, tnear, hw);
            tWall = hw.geom == 0 ? hw.t : tWall;
          }
        }
        packetDone = pq_hit_packet<GEO, PQ_CREDIT, FRAME_LDS, FOLLOW, RELIEF, PQ_CACHE>(p, active, org, dir, tnear, h, (volatile VR_LDS unsigned *)(pqS + waveInBlock * 128u), cands, wallS, (volatile VR_LDS float *)(pqBoxS + (PQ_CACHE ? waveInBlock * (6u * VR_PQ_KEEP) : 0u)), tWall VR_DIAG_PASS);
        pqCredit = PQ_CREDIT && packetDone;
        pqFails = packetDone ? 0u : (pqFails < 6u ? pqFails + 1u : 6u);
        pqSkip = packetDone ? 0u : (1u << pqFails) - 1u;
        if (packetDone) {
          node = VR_END;
          if (active) {
            DIAG(13);
          }
        }
      } else {
        --pqSkip;
      }
    }
    if (!SMALL && usePacket && !packetDone) {
      packetDone = bvh_hit_packet<GEO>(p, active, org, dir, tnear, h, p.packetBudget, p.packetRatio VR_DIAG_PASS);
      // a wave whose rays have scattered stops paying for hopeless packets for a while
      packetFails = packetDone ? 0u : (packetFails < 6u ? packetFails + 1u : 6u);
      packetSkip = packetDone ? 0u : (1u << packetFails) - 1u;
      if (packetDone)
        node = VR_END;
    } else if (packetSkip) {
      --packetSkip;
    }
    TICK(1);
    if (!packetDone) {
      const unsigned walking = (unsigned)__popcll(ballot64(active && (ORDERED ? node != VR_END : node < p.numNodes)));
      const unsigned minLanes = (!CARRY || exhausted || walking <= p.walkExit) ? 1u : p.walkExit;
      if (ORDERED)
        pair_walk_lanes<GEO, SD, MODE != 2>(p, pnodes, prims, stackS + tid, stackG, active, org, dir, tnear, h, node, sp, minLanes VR_DIAG_PASS);
      else
        bvh_walk_lanes<GEO>(p, active, org, dir, tnear, h, node, minLanes VR_DIAG_PASS);
    }
    const bool fin = active && (ORDERED ? node == VR_END : node >= p.numNodes); // this lane's geometry walk is complete
#ifdef VR_SELFCHECK
    { // -DVR_SELFCHECK build: every finished segment again with the escape-link walk; disagreements are
      // counted in counters[48], the first one is kept in counters[50..]
      HitRec hb;
      hit_clear(hb);
      unsigned nb = fin ? 0u : VR_END;
      bvh_walk_lanes<GEO>(p, fin, org, dir, tnear, hb, nb, 1u VR_DIAG_PASS);
      // (the CLOSEST HIT is what is compared — geometry and walls: a packet query leaves a ray that meets a side wall before
      //  it can enter the scene box without a geometry hit, and the wall wins either way)
      HitRec hc = h;
      if (fin) {
        hit_walls(p, wallS, org, dir, tnear, hb);
        hit_walls(p, wallS, org, dir, tnear, hc);
      }
      if (fin && (hb.geom != hc.geom || hb.t != hc.t || (hb.geom == 1 && hb.pos != hc.pos) || (hb.geom == 0 && hb.prim != hc.prim))) {
        if (atomicAdd(&p.counters[48], 1ull) == 0ull) {
          const float v[8] = {org.x, org.y, org.z, dir.x, dir.y, dir.z, hc.t, hb.t};
          for (int k = 0; k < 8; ++k)
            p.counters[50 + k] = (unsigned long long)__float_as_uint(v[k]);
          p.counters[58] = ((unsigned long long)hc.pos << 32) | hb.pos;
          p.counters[59] = ((unsigned long long)(unsigned)hc.geom << 32) | (unsigned)hb.geom;
        }
      }
    }
#endif
    TICK(3);
    if (fin) { // boundary walls, where one can come before the hit
      if constexpr (FRAME_LDS)
        hit_walls_lds(p, wallS, org, dir, tnear, h);
      else
        hit_walls(p, wallS, org, dir, tnear, h);
    }
    TICK(4);
    // Merge same-disk credits of the wave into one atomic when that is likely to pay: rays of a
    // packet, or — sampled on one lane's target — when a good share of the wave's hits fall on
    // the same primitive (sorted rays on a coarse scene: one vector atomic with 64 lanes on ONE
    // address is serialised lane by lane in the L2 atomic unit).
    bool aggregate = packetDone || (p.debugFlags & 32768u) != 0u; // (flag 32768: always, a measurement)
    {
      const bool cand = fin && h.geom == 1;
      const unsigned long long cm = ballot64(cand);
      if (!aggregate && cm) {
        const unsigned sample = (unsigned)__shfl((int)h.pos, __ffsll((long long)cm) - 1, 64);
        const unsigned same = (unsigned)__popcll(ballot64(cand && h.pos == sample));
        aggregate = 4u * same >= (unsigned)__popcll(cm) && same >= 4u;
      }
    }

    bool creditLane = false;
    u64 creditW = 0;
    float creditWf = 0.f; // (registry particles: the weight as the model's collide sees it ...
    V3 creditDir = mk(0.f, 0.f, 0.f); //  ... and the INCOMING direction: the state machine replaces it by the reflected one)
    SUB_MARK(12); // (since the walls: the aggregation vote)
    if (fin) {
      DIAG(5);
      // ---- the reference's state machine for this segment (rayTraceKernel.hpp:169-335) ----
      VR_COUNT(K_TRACES, 1);
      if (h.geom < 0) { // miss, :172-176
        VR_COUNT(K_NONGEO, 1);
        active = false;
      } else {
        const V3 hitPoint = mk(org.x + dir.x * h.t, org.y + dir.y * h.t, org.z + dir.z * h.t);
        bool scattered = false;
        if (EXT && EXT_FULL && p.meanFreePath > 0.f) {
          // mean-free-path scatter (rayTraceKernel.hpp:179-203), quirk Q1 kept: tested after the
          // closest hit was found, and the origin moves by dir * rnd (the uniform number itself)
          const float rnd = canon_f32(rng_next(rng, cnt[K_TIER2 * VR_BLOCK]));
          const float scatterProbability = (float)(1. - (double)glibc_expf(-h.t / p.meanFreePath));
          if (rnd < scatterProbability) {
            org = mk(org.x + dir.x * rnd, org.y + dir.y * rnd, org.z + dir.z * rnd);
            rayDirection = pick_random_point_on_unit_sphere(rng, cnt[K_TIER2 * VR_BLOCK]);
            dir = project_dir<D>(rayDirection);
            VR_COUNT(K_PARTICLE, 1);
            scattered = true;
          }
        }
        if (scattered) {
          // (reflect = true; continue)
        } else if (h.geom == 0) { // boundary, :206-214 + rayBoundary.hpp:29-127
          SUB_START
          if (++boundaryHits > p.maxBoundaryHits) {
            VR_COUNT(K_TERM, 1);
            active = false;
          } else {
            process_boundary_hit<D>(p, wallS, h.prim, hitPoint, org, rayDirection, dir, active);
          }
          SUB_STOP(11);
        } else {
          // geometry hit
          V3 geomNormal;
          if (GEO == 0) {
            if (PQ_CREDIT && pqCredit) { // (found by the packet query: the candidate's record in LDS, not a dependent global load)
              const U4 nr = cands.rec[VR_PQ_NRM + cands.mine];
              geomNormal = mk(__uint_as_float(nr.x), __uint_as_float(nr.y), __uint_as_float(nr.z));
            } else {
              const float4 n4 = prims[2 * h.pos + 1];
              geomNormal = mk(n4.x, n4.y, n4.z);
            }
          } else {
            geomNormal = mk(prims[4 * h.pos + 1].w, prims[4 * h.pos + 2].w, prims[4 * h.pos + 3].w);
          }
          const bool backfaceHit = vdot(rayDirection, geomNormal) > 0.f; // :224
          SUB_MARK(10);
          if (backfaceHit) {
            if (GEO == 0 && !hitFromBack) { // first back hit of a disk: let through, :235-240
              hitFromBack = true;
              org = hitPoint;
            } else { // :229-233, :243-248
              VR_COUNT(K_TERM, 1);
              active = false;
            }
          } else {
            VR_COUNT(K_GEO, 1);
            DIAG(11);
            const u64 wfx = weight_fx(rayWeight);
            if (PQ_CREDIT && pqCredit) {
              creditLane = true; // credited after the state machine, for the whole wave at once (pq_credit)
              creditW = wfx;
              creditWf = rayWeight;
              if (EXT)
                creditDir = rayDirection;
            } else if (!EXT) {
              // surfaceCollision, rayParticle.hpp:148-156.  Without aggregation the credits of the neighbour
              // disks are first collected (three in registers; further ones, rare, go out at once) and then issued
              // together with the closest disk's: on gfx9 a load that follows an atomic waits for that atomic too
              // (one in-order counter), so an atomic inside the neighbour loop exposed its full L2 round trip to
              // the next neighbour's loads, iteration after iteration.
              unsigned cq0 = 0xFFFFFFFFu, cq1 = 0xFFFFFFFFu, cq2 = 0xFFFFFFFFu;
              if (aggregate && !(p.debugFlags & 1u))
                credit_aggregated(fluxAcc, true, h.pos, wfx);
              if (GEO == 0 && !(p.debugFlags & 4u)) {
                // every overlapping neighbour disk is credited the full weight (:271-300)
                SUB_START
                const unsigned nb = nbOff[h.pos], ne = nbOff[h.pos + 1];
                // One dependent access per neighbour instead of three: the next id is fetched while this
                // neighbour is tested, and both record words are requested together (left to itself the compiler
                // sinks the centre's load behind the normal's sign test).  Throughput of full launches does not
                // notice; a launch of 10^6 rays is as long as its longest bounce chain, and this loop was
                // half of a round's chain of memory latencies.
                unsigned qNext = nb < ne ? nbIds[nb] : 0u;
                for (unsigned j = nb; j < ne; ++j) {
                  DIAG(6);
                  const unsigned q = qNext;
                  qNext = nbIds[j + 1 < ne ? j + 1 : j];
                  const float4 c4 = prims[2 * q];
                  const float4 n4 = prims[2 * q + 1];
                  asm volatile("" ::"v"(c4.x), "v"(n4.x)); // (both in flight before the test branches)
                  const bool hitN = local_disc_hit(org, dir, c4, mk(n4.x, n4.y, n4.z)) && !(p.debugFlags & 1u);
                  if (aggregate) {
                    credit_aggregated(fluxAcc, hitN, q, wfx);
                  } else if (hitN) {
                    if (cq2 != 0xFFFFFFFFu)
                      atomicAdd(&fluxAcc[q], wfx);
                    else if (cq1 != 0xFFFFFFFFu)
                      cq2 = q;
                    else if (cq0 != 0xFFFFFFFFu)
                      cq1 = q;
                    else
                      cq0 = q;
                  }
                }
                SUB_STOP(8);
              }
              if (!aggregate && !(p.debugFlags & 1u)) {
                atomicAdd(&fluxAcc[h.pos], wfx);
                if (cq0 != 0xFFFFFFFFu)
                  atomicAdd(&fluxAcc[cq0], wfx);
                if (cq1 != 0xFFFFFFFFu)
                  atomicAdd(&fluxAcc[cq1], wfx);
                if (cq2 != 0xFFFFFFFFu)
                  atomicAdd(&fluxAcc[cq2], wfx);
              }
            } else {
              // plug-in particles: Particles::collide decides what each credited primitive's data
              // labels receive; with WDIST the weight is shared by inverse impact distance
              // (rayTraceKernel.hpp:258-296: w / d_i / sum(1/d) * numDisksHit, closest disk first)
              const int kind = p.particleKind;
              const ModelCtx mctx = model_ctx(p);
              // (a coarse scene under sorted rays: a good share of the wave credits ONE disk — merged per distinct
              //  weight like the built-in particles' credits, or the 64 lanes queue up on one address in L2)
              auto creditTo = [&](unsigned q, float w, const V3 &nq, unsigned origId) {
                Particles::collide<EXT_FULL>(kind, mctx, w, rayDirection, nq, origId, [&](int label, float v) {
                  unsigned long long *plane = fluxAcc + (size_t)label * (SMALL ? p.numPrims : p.planeStride);
                  if (aggregate && !SMALL) // (LDS accumulators take 64 adds on one address in their stride)
                    credit_aggregated(plane, true, q, weight_fx(v));
                  else
                    atomicAdd(&plane[q], weight_fx(v));
                });
              };
              if (GEO == 0) {
                const unsigned nb = nbOff[h.pos], ne = nbOff[h.pos + 1];
                float invSum = 0.f, dClosest = 0.f;
                unsigned numHit = 1;
                if (EXT_FULL && p.useWdist) {
                  const float4 cp = prims[2 * h.pos];
                  const V3 dv = mk(hitPoint.x - cp.x, hitPoint.y - cp.y, hitPoint.z - cp.z);
                  dClosest = sqrtf(vdot(dv, dv)) + 1e-6f;
                  invSum = 0.f + 1.f / dClosest;
                  for (unsigned j = nb; j < ne; ++j) {
                    const unsigned q = nbIds[j];
                    const float4 n4 = prims[2 * q + 1];
                    float dist;
                    if (local_disc_hit_dist(org, dir, prims[2 * q], mk(n4.x, n4.y, n4.z), dist)) {
                      invSum += 1.f / (dist + 1e-6f);
                      ++numHit;
                    }
                  }
                }
                creditTo(h.pos, (EXT_FULL && p.useWdist) ? rayWeight / dClosest / invSum * (float)numHit : rayWeight, geomNormal, h.prim);
                // (as in the built-in particles' loop: the next id is fetched while this neighbour is tested, and both
                //  record words are requested together — one dependent access per neighbour instead of three)
                unsigned qNext = nb < ne ? nbIds[nb] : 0u;
                for (unsigned j = nb; j < ne; ++j) {
                  const unsigned q = qNext;
                  qNext = nbIds[j + 1 < ne ? j + 1 : j];
                  const float4 c4 = prims[2 * q];
                  const float4 n4 = prims[2 * q + 1];
                  asm volatile("" ::"v"(c4.x), "v"(n4.x));
                  const V3 nq = mk(n4.x, n4.y, n4.z);
                  float dist;
                  if (local_disc_hit_dist(org, dir, c4, nq, dist))
                    creditTo(q, (EXT_FULL && p.useWdist) ? rayWeight / (dist + 1e-6f) / invSum * (float)numHit : rayWeight, nq,
                             __float_as_uint(n4.w));
                }
              } else {
                creditTo(h.pos, rayWeight, geomNormal, h.prim);
              }
            }
            if (ABSORB) {
              // sticking >= 1: weight drops to <= 0 (:316-319); the reflection draws
              // the reference makes before that test (Q2) are not observable.
              active = false;
            } else {
              // (not `p.primSticking ? primSticking[h.pos] : p.sticking`: the compiler makes that ONE load from a selected
              //  address — a generic pointer, i.e. a flat_load per reflection that waits on both memory counters)
              float sticking = p.sticking;
              asm volatile("" : "+s"(sticking)); // (a value in a register, not a second address to choose from)
              if (p.primSticking)
                sticking = primSticking[h.pos];
              if (EXT) // (a registry model may make it depend on the primitive and the caller's global data)
                sticking = Particles::sticking<EXT_FULL>(p.particleKind, model_ctx(p), h.prim, sticking);
              const float wAfter = rayWeight - rayWeight * sticking;
              if (wAfter <= 0.f) {
                active = false; // as above: the pending draws die with the ray
              } else {
                // surfaceReflection, rayParticle.hpp:137-146 / 178-187
                SUB_START
                V3 newDir;
                if (PARTICLE == 0)
                  newDir = reflection_diffuse<D>(geomNormal, rng, cnt[K_TIER2 * VR_BLOCK]);
                else if (PARTICLE == 1)
                  newDir = reflect_specular(rayDirection, geomNormal);
                else
                  newDir = Particles::reflect<D, EXT_FULL>(p.particleKind, model_ctx(p), rayDirection, geomNormal, rng, cnt[K_TIER2 * VR_BLOCK]);
                rayWeight = wAfter;
                if (++numReflections > p.maxReflections) { // :320-324
                  VR_COUNT(K_TERM, 1);
                  active = false;
                } else {
                  // rejectionControl, :435-460
                  const float initWeight = p.hostWeights ? __uint_as_float(initWeightBits) : 1.f;
                  const float lowerThreshold = (float)(0.1 * (double)initWeight);
                  const float renewWeight = (float)(0.3 * (double)initWeight);
                  bool reflect = true;
                  if (!(rayWeight >= lowerThreshold)) {
                    DIAG(10);
                    const double killProbability = 1.0 - (double)(rayWeight / renewWeight);
                    if (canon_f64(rng_next(rng, cnt[K_TIER2 * VR_BLOCK])) < killProbability)
                      reflect = false;
                    else
                      rayWeight = renewWeight;
                  }
                  if (!reflect) {
                    active = false;
                  } else {
                    rayDirection = newDir;
                    org = hitPoint;
                    dir = project_dir<D>(rayDirection);
                  }
                }
                SUB_STOP(9);
              }
            }
          }
        }
      }
      SUB_MARK(13); // (since the walls: everything but the per-ray end counters)
      if (!active) {
        VR_COUNT(K_BOUNDARY, boundaryHits);
        VR_COUNT(K_REFL, numReflections);
      }
      start = active; // still alive: the next segment begins in the next round
    }
    TICK(5);
    if (PQ_CREDIT && pqCredit && !(p.debugFlags & 1u)) {
      // ---- surfaceCollision for the round's surface hits, candidate by candidate (wave-uniform):
      // a lane credits candidate q if q is its closest disk, or q is a neighbour of that disk
      // (centres within nbDist: the relation the CSR was built from, rayPointNeighborhood.hpp:287-298,
      // evaluated on the same floats) and its ray passes the neighbour test on q.  All lanes
      // crediting q add to ONE address: a single atomic (absorbing: count x unit weight).
      if (ballot64(creditLane)) {
        // centre of this lane's closest disk
        const U4 own = cands.rec[creditLane ? cands.mine : 0u];
        const float px = __uint_as_float(own.y), py = __uint_as_float(own.z), pz = __uint_as_float(own.w);
        const float dist = p.nbDist, dist2 = dist * dist;
        // General kernels: the lanes crediting candidate c add their fixed-point weights to the wave's LDS sum of c
        // (ds_add_u64: exact, any order) and afterwards lane c sends candidate c's total to HBM — ONE wave instruction
        // of global atomics per round and label instead of one atomic per candidate and distinct weight.
        // (from the per-lane wave index: with a wave-uniform ADDRESS the compiler turns these LDS atomics into a reduction
        //  over the wave plus one atomic — more work than the two or three lanes that credit a candidate; C2 0.1 +19 %)
        unsigned long long *const candAcc = candAccS + (PQ_SUMS ? (tid >> 6) * (VR_PQ_CANDS * PQ_LAB) : 0u);
        if (PQ_SUMS) {
          for (unsigned k = lane; k < cands.count * PQ_LAB; k += 64u)
            candAcc[k] = 0ull;
          __builtin_amdgcn_wave_barrier();
        }
        for (unsigned c = 0; c < cands.count; ++c) {
          DIAG(6);
          const U4 cr = cands.rec[c];
          const unsigned q = (unsigned)__builtin_amdgcn_readfirstlane((int)cr.x);
          const float dx = px - __uint_as_float(cr.y), dy = py - __uint_as_float(cr.z), dz = pz - __uint_as_float(cr.w);
          bool near = fabsf(dx) <= dist && fabsf(dy) <= dist && (p.geoD == 2 || fabsf(dz) <= dist);
          near = near && ((dx * dx + dy * dy) + dz * dz) <= dist2;
          const bool sel = creditLane && (h.pos == q || (near && ((cands.local >> c) & 1ull)));
          if (EXT) {
            // registry particles: the model's collide runs per lane with candidate q's own normal and id; its credits
            // are RECORDED per lane (a model may credit under any condition of its own) and then added label by label
            if (ballot64(sel)) {
              const float4 n4 = prims[2 * (size_t)q + 1];
              float val[VR_MAX_LABELS];
#pragma unroll
              for (int l = 0; l < VR_MAX_LABELS; ++l)
                val[l] = 0.f;
              if (sel)
                Particles::collide<EXT_FULL>(p.particleKind, model_ctx(p), creditWf, creditDir, mk(n4.x, n4.y, n4.z),
                                             __float_as_uint(n4.w), [&](int label, float v) {
#pragma unroll
                                               for (int l = 0; l < VR_MAX_LABELS; ++l)
                                                 val[l] = l == label ? val[l] + v : val[l];
                                             });
#pragma unroll
              for (int l = 0; l < VR_MAX_LABELS; ++l) {
                if ((unsigned)l >= p.numData)
                  break;
                if ((unsigned)l < PQ_LAB) {
                  if (sel)
                    atomicAdd(&candAcc[c * PQ_LAB + (unsigned)l], weight_fx(val[l]));
                } else {
                  credit_wave_sum(fluxAcc + (size_t)l * p.planeStride, q, sel ? weight_fx(val[l]) : 0ull);
                }
              }
            }
          } else if (ABSORB) {
            const unsigned long long m = ballot64(sel);
            if (m && lane == (unsigned)(__ffsll((long long)m) - 1))
              atomicAdd(&fluxAcc[q], (u64)__popcll(m) * 1099511627776ull); // unit weights: count x 2^40
          } else {
            if (sel)
              atomicAdd(&candAcc[c], creditW);
          }
        }
        if (PQ_SUMS) {
          __builtin_amdgcn_wave_barrier();
          if (lane < cands.count) {
            const unsigned q = cands.rec[lane].x;
#pragma unroll
            for (unsigned l = 0; l < PQ_LAB; ++l) {
              const unsigned long long v = *(volatile VR_LDS unsigned long long *)&candAcc[lane * PQ_LAB + l];
              if (v && l < p.numData)
                atomicAdd(&fluxAcc[(size_t)l * p.planeStride + q], v);
            }
          }
        }
      }
    }
    if constexpr (FOLLOW) {
      // ---- follow-up segments.  A ray that goes on after this round's event (reflected off the surface, let through
      // a back face, turned round by a side wall) would search the geometry again in the next round — on a flat scene
      // only to leave it at once, and the wave would pay a second packet query for it.  Where the new segment's stretch
      // inside the scene box lies within the box Q of this round's query, the query's candidates are every primitive it
      // can meet (a disk it hits holds a point of that stretch, hence meets Q): they are tested here, and a segment that
      // meets none of them is finished in this round — its wall test and the miss / boundary branches of the state
      // machine (rayTraceKernel.hpp:169-214).  A segment that does meet one is left to the next round as before.
      // Same arithmetic, same closest-hit rule: nothing changes in the results (the parity tests run both ways,
      // VR_DEBUG_FLAGS=256 switches this off).
      if (pqCredit && cands.box && !(p.debugFlags & 256u)) {
        const bool again = fin && active;
        bool inside = false, reaches = false;
        if (again) {
          const U4 ql = cands.rec[VR_PQ_BOX], qh = cands.rec[VR_PQ_BOX + 1];
          const V3 inv = safe_inverse(dir);
          const float tx0 = (p.sceneLo[0] - org.x) * inv.x, tx1 = (p.sceneHi[0] - org.x) * inv.x;
          const float ty0 = (p.sceneLo[1] - org.y) * inv.y, ty1 = (p.sceneHi[1] - org.y) * inv.y;
          const float tz0 = (p.sceneLo[2] - org.z) * inv.z, tz1 = (p.sceneHi[2] - org.z) * inv.z;
          const float tEnter = fmaxf(fminf(tx0, tx1), fminf(ty0, ty1));
          const float tIn = fmaxf(tEnter, fmaxf(fminf(tz0, tz1), tnear));
          const float tQ = fmaxf(tEnter, fmaxf(fminf(tz0, tz1), 0.f));
          const float tOut = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
          reaches = tIn <= tOut; // (as pq_hit_packet's `valid`: otherwise no part of the segment is inside the scene box)
          float tBeg = tQ, tEnd = tOut;
          bool capped = false;
          if (RELIEF && !(p.debugFlags & 512u)) {
            // (a reflected ray mostly rises clear of everything near by: the height field's one look-up says so without a
            //  walk.  Otherwise, as the query's own rays: the stretch through the local relief — none: the ray cannot meet
            //  the geometry — by a SHORT walk: a grazing ray that is not through after six tiles is left to the next round,
            //  i.e. to the spill queue; the lanes of a wave walk together, and one such ray kept all of them waiting)
            if (reaches && rises_clear<D>(wallS, org, dir, tnear)) {
              reaches = false;
            } else {
              float tA, tB;
              capped = relief_clip<6>(wallS, reaches, org, dir, tQ, tOut, tA, tB);
              reaches = reaches && tA <= tB;
              tBeg = tA;
              tEnd = tB;
            }
          }
          const float ax = org.x + dir.x * tBeg, ay = org.y + dir.y * tBeg, az = org.z + dir.z * tBeg;
          const float bx = org.x + dir.x * tEnd, by = org.y + dir.y * tEnd, bz = org.z + dir.z * tEnd;
          // (inside Q proper: the padding absorbs the rounding of the clip, as it does for the query's own rays)
          const float pad = p.pqPad;
          const float lx = __uint_as_float(ql.x) + pad, ly = __uint_as_float(ql.y) + pad, lz = __uint_as_float(ql.z) + pad;
          const float hx = __uint_as_float(qh.x) - pad, hy = __uint_as_float(qh.y) - pad, hz = __uint_as_float(qh.z) - pad;
          inside = !capped && (!reaches || (fminf(ax, bx) >= lx && fmaxf(ax, bx) <= hx && fminf(ay, by) >= ly && fmaxf(ay, by) <= hy &&
                                            fminf(az, bz) >= lz && fmaxf(az, bz) <= hz));
          if (RELIEF && (p.debugFlags & 2048u) && !inside) { // EXPERIMENT (wrong results): long continuing rays vanish
            const float ex = bx - ax, ey = by - ay, ez = bz - az;
            if ((ex * ex + ey * ey) + ez * ez > p.reliefTravel * p.reliefTravel)
              active = false;
          }
          if (RELIEF && (p.debugFlags & 4096u) && !inside) // EXPERIMENT (wrong results): every continuing ray not finished here vanishes
            active = false;
        }
        if (ballot64(inside)) {
          bool meets = false;
          if (ballot64(inside && reaches)) {
            for (unsigned c = 0; c < cands.count; ++c) {
              const U4 cr = cands.rec[c], nr = cands.rec[VR_PQ_NRM + c]; // (LDS broadcasts)
              const float4 c4 = make_float4(__uint_as_float(cr.y), __uint_as_float(cr.z), __uint_as_float(cr.w), __uint_as_float(nr.w));
              float t;
              meets = meets || hit_disc(org, dir, tnear, c4, mk(__uint_as_float(nr.x), __uint_as_float(nr.y), __uint_as_float(nr.z)), t);
            }
          }
          if (inside && !(reaches && meets)) {
            HitRec h2;
            hit_clear(h2);
            hit_walls(p, wallS, org, dir, tnear, h2);
            VR_COUNT(K_TRACES, 1);
            if (h2.geom < 0) { // miss, :172-176
              VR_COUNT(K_NONGEO, 1);
              active = false;
            } else { // boundary, :206-214
              const V3 hitPoint = mk(org.x + dir.x * h2.t, org.y + dir.y * h2.t, org.z + dir.z * h2.t);
              if (++boundaryHits > p.maxBoundaryHits) {
                VR_COUNT(K_TERM, 1);
                active = false;
              } else {
                process_boundary_hit<D>(p, wallS, h2.prim, hitPoint, org, rayDirection, dir, active);
              }
            }
            if (!active) {
              VR_COUNT(K_BOUNDARY, boundaryHits);
              VR_COUNT(K_REFL, numReflections);
            }
            start = active;
          }
        }
      }
    }
    if constexpr (FOLLOW && RELIEF) {
      // ---- spill: a ray that would go on into the next round leaves as a full-state record (TraceParams::spillRec);
      // the launch over the loose bins resumes it.  This kernel's waves then hold fresh, sorted rays only.
      if (p.spillRec && !(p.debugFlags & 8192u)) { // (flag 8192: no spilling, for comparison)
        const bool sp = active && start;
        const unsigned long long sm = ballot64(sp);
        if (sm) {
          // (the queue in BLOCKS of 64 records, each filled by one wave: the records of a block are rays of one
          //  neighbourhood — this wave's consecutive rounds — and the resuming kernel takes a block per round; filed in
          //  order of arrival, 8 rays of a round side by side, its waves held rays of eight places.  Only a wave's last
          //  block has unused records: spill_pad at the end of the kernel)
          const unsigned n = (unsigned)__popcll(sm), room = VR_SPILL_BLOCK - spillUsed;
          unsigned nextBase = 0u;
          if (n > room) { // (the block is filled up, the rest of the round's rays open the next one)
            if (lane == 0u)
              nextBase = atomicAdd(p.spillCount, VR_SPILL_BLOCK);
            nextBase = (unsigned)__builtin_amdgcn_readfirstlane((int)nextBase);
          }
          if (sp) {
            const unsigned rank = (unsigned)__popcll(sm & ((1ull << lane) - 1ull));
            float4 *sr = reinterpret_cast<float4 *>(p.spillRec) + 4 * (size_t)(rank < room ? spillBase + spillUsed + rank : nextBase + (rank - room));
            sr[0] = make_float4(org.x, org.y, org.z, rayWeight);
            sr[1] = make_float4(rayDirection.x, rayDirection.y, rayDirection.z, __uint_as_float(rng.seed)); // (the engine's seed: tea3(idx, seed))
            sr[2] = make_float4(__uint_as_float(rng.k), __uint_as_float(numReflections),
                                __uint_as_float(boundaryHits | (hitFromBack ? 0x80000000u : 0u)), 0.f);
            sr[3] = make_float4(__uint_as_float((unsigned)(rng.lo & 0xFFFFFFFFull)), __uint_as_float((unsigned)(rng.lo >> 32)),
                                __uint_as_float((unsigned)(rng.hi & 0xFFFFFFFFull)), __uint_as_float((unsigned)(rng.hi >> 32)));
            active = false;
            start = false;
          }
          spillBase = n > room ? nextBase : spillBase;
          spillUsed = n > room ? n - room : spillUsed + n;
        }
      }
    }
    if constexpr (!ABSORB && !FOLLOW) {
      // ---- segments that rise clear (the general kernels without the packet query's candidate list).  A ray that goes
      // on after this round's event — reflected off the TOP surface of a structure: a fifth of all segments of a trench
      // — used to keep its lane for one more round: a walk of two or three steps while the other lanes walk thirty.
      // The height field over the source plane (HeightFieldParams: per tile the highest point of anything in the tile
      // or its eight neighbours, plus a rounding margin) decides it here: a ray that is above its tile's height from
      // tnear on, and rises above the whole scene before it has travelled a tile sideways, cannot meet the geometry —
      // a primitive it met would hold a point of the ray, hence reach up to the ray's height within those nine tiles.
      // Such a segment is finished in this round: wall test, then the miss / boundary branches of the state machine
      // (rayTraceKernel.hpp:169-214), and its lane pulls a new ray in the next round.  (Not with a mean free path: that
      // scatter is drawn before the boundary branch.  VR_DEBUG_FLAGS=256 switches this off; the tests run both ways.)
      if (!(p.debugFlags & 256u) && !(EXT && EXT_FULL && p.meanFreePath > 0.f)) {
        const bool clear = fin && active && rises_clear<D>(wallS, org, dir, tnear);
        if (clear) {
          HitRec h2;
          hit_clear(h2);
          hit_walls(p, wallS, org, dir, tnear, h2);
          VR_COUNT(K_TRACES, 1);
          if (h2.geom < 0) { // miss, :172-176
            VR_COUNT(K_NONGEO, 1);
            active = false;
          } else { // boundary, :206-214
            const V3 hitPoint = mk(org.x + dir.x * h2.t, org.y + dir.y * h2.t, org.z + dir.z * h2.t);
            if (++boundaryHits > p.maxBoundaryHits) {
              VR_COUNT(K_TERM, 1);
              active = false;
            } else {
              process_boundary_hit<D>(p, wallS, h2.prim, hitPoint, org, rayDirection, dir, active);
            }
          }
          if (!active) {
            VR_COUNT(K_BOUNDARY, boundaryHits);
            VR_COUNT(K_REFL, numReflections);
          }
          start = active;
        }
      }
    }
    TICK(6);
  }
  if constexpr (FOLLOW && RELIEF) {
    if (p.spillRec)
      spill_pad(p, spillBase, spillUsed, lane); // (the unused end of this wave's last block: no rays)
  }

  if (SMALL) {
    // every wave of the block has left the loop: the block's LDS accumulators go to its replica in HBM
    __syncthreads();
    for (unsigned l = 0; l < p.numData; ++l)
      for (unsigned k = tid; k < p.numPrims; k += VR_BLOCK)
        if (fluxAcc[(size_t)l * p.numPrims + k])
          atomicAdd(&fluxGlobal[(size_t)l * p.planeStride + k], fluxAcc[(size_t)l * p.numPrims + k]);
  }
#ifdef VR_DIAG
  TICK(7);
  if (lane < 16 && phaseT[lane])
    atomicAdd(&p.counters[64 + lane], phaseT[lane]);
  for (int k = 0; k < 16; ++k) {
    const unsigned long long sw = wave_sum(diagW[k]), sl = wave_sum(diagL[k]);
    if (lane == 0 && sl) {
      atomicAdd(&p.counters[16 + 2 * k], sw);
      atomicAdd(&p.counters[16 + 2 * k + 1], sl);
    }
  }
#endif
  // (slot order of vr_types.hpp: traces, nongeo, geo, particle, boundary, reflections, terminated, tier2)
  auto total = [&](int k) -> unsigned { return cnt[k * VR_BLOCK]; }; // this lane's share of counter k
  const unsigned vals[8] = {total(K_TRACES), total(K_NONGEO), total(K_GEO),  total(K_PARTICLE),
                            total(K_BOUNDARY), total(K_REFL), total(K_TERM), total(K_TIER2)};
#undef VR_COUNT
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned long long s = wave_sum(vals[i]);
    if (lane == 0 && s)
      atomicAdd(&p.counters[i], s);
  }
}

#ifndef VR_USER_MODULE
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
  if (src == 0 && p.reliefCoarse && p.binCount) { // the plain generator on a scene with relief: predicted-hit key, loose bins
    if (D == 2 && !keepRng)
      hipLaunchKernelGGL((gen_kernel<2, false, true>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
    else if (D == 2)
      hipLaunchKernelGGL((gen_kernel<2, true, true>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
    else if (!keepRng)
      hipLaunchKernelGGL((gen_kernel<3, false, true>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
    else
      hipLaunchKernelGGL((gen_kernel<3, true, true>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
    return hipGetLastError();
  }
#define VR_GEN(K, DD, KEEP)                                                                                           \
  case K: hipLaunchKernelGGL((gen_kernel<DD, KEEP, false>), dim3(grid), dim3(VR_BLOCK), 0, s, p); break;               \
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
  else if (mode == 3 && GEO == 0 && PARTICLE <= P_EXT)
    hipLaunchKernelGGL((trace_kernel<D, 0, ((PARTICLE > P_EXT) ? 0 : PARTICLE), 3>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
  else if (mode == 4)
    hipLaunchKernelGGL((trace_kernel<D, GEO, PARTICLE, 4>), dim3(grid), dim3(VR_BLOCK), p.smallBytes, s, p);
  else if (mode == 5)
    hipLaunchKernelGGL((trace_kernel<D, GEO, 0, 5>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
  else if (mode == 6 && GEO == 0 && PARTICLE <= P_EXT)
    hipLaunchKernelGGL((trace_kernel<D, 0, ((PARTICLE > P_EXT) ? 0 : PARTICLE), 6>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
  else if (mode == 7 && GEO == 0 && PARTICLE <= P_EXT)
    hipLaunchKernelGGL((trace_kernel<D, 0, ((PARTICLE > P_EXT) ? 0 : PARTICLE), 7>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
  else
    hipLaunchKernelGGL((trace_kernel<D, GEO, PARTICLE, 0>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
  return hipGetLastError();
}

// mode: 0 general, 1 absorbing + flat scene, 2 absorbing + structured scene
// particle: 0 DiffuseParticle, 1 SpecularParticle, 2 (P_EXT) extended kernel, 3 (P_EXT_FULL) ... with the coned-cosine
// model, WDIST crediting and mean-free-path scattering (always mode 0 or 4)
template <class F> static auto dispatch_variant(int D, int geo, int particle, F &&f) {
  const int key = (D == 2 ? 0 : 8) + (geo ? 4 : 0) + particle;
#define VR_VARIANT(K, DD, GG, PP)                                                                                      \
  case K: return f(std::integral_constant<int, DD>{}, std::integral_constant<int, GG>{}, std::integral_constant<int, PP>{});
  switch (key) {
    VR_VARIANT(0, 2, 0, 0) VR_VARIANT(1, 2, 0, 1) VR_VARIANT(2, 2, 0, 2) VR_VARIANT(3, 2, 0, 3)
    VR_VARIANT(4, 2, 1, 0) VR_VARIANT(5, 2, 1, 1) VR_VARIANT(6, 2, 1, 2) VR_VARIANT(7, 2, 1, 3)
    VR_VARIANT(8, 3, 0, 0) VR_VARIANT(9, 3, 0, 1) VR_VARIANT(10, 3, 0, 2) VR_VARIANT(11, 3, 0, 3)
    VR_VARIANT(12, 3, 1, 0) VR_VARIANT(13, 3, 1, 1) VR_VARIANT(14, 3, 1, 2)
  default: return f(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{});
  }
#undef VR_VARIANT
}

hipError_t launch_trace(const TraceParams &p, int D, int geo, int particle, int mode, unsigned grid,
                        hipStream_t s) {
  if (mode == 1 || mode == 2 || mode == 5)
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
  else if (mode == 3 && GEO == 0 && PARTICLE <= P_EXT)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, 0, ((PARTICLE > P_EXT) ? 0 : PARTICLE), 3>, VR_BLOCK, 0);
  else if (mode == 4)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, GEO, PARTICLE, 4>, VR_BLOCK, smallBytes);
  else if (mode == 5)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, GEO, 0, 5>, VR_BLOCK, 0);
  else if (mode == 6 && GEO == 0 && PARTICLE <= P_EXT)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, 0, ((PARTICLE > P_EXT) ? 0 : PARTICLE), 6>, VR_BLOCK, 0);
  else if (mode == 7 && GEO == 0 && PARTICLE <= P_EXT)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, 0, ((PARTICLE > P_EXT) ? 0 : PARTICLE), 7>, VR_BLOCK, 0);
  else
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, GEO, PARTICLE, 0>, VR_BLOCK, 0);
  return e == hipSuccess ? nb : 2;
}

int trace_blocks_per_cu(int D, int geo, int particle, int mode, unsigned smallBytes) {
  if (mode == 1 || mode == 2 || mode == 5)
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
  __shared__ float wallS[VR_WALL_TABLE];
  __shared__ unsigned stackS[VR_STACK_LDS * VR_BLOCK]; // (64-thread blocks: lane columns 0..63 of the [entry][VR_BLOCK] layout)
  for (unsigned k = threadIdx.x; k < VR_WALL_TABLE; k += blockDim.x)
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
  __shared__ float wallS[VR_WALL_TABLE];
  for (unsigned k = threadIdx.x; k < VR_WALL_TABLE; k += blockDim.x)
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

// un-permute the leaf-ordered accumulators into the caller's primitive order.
// Overflow is DETECTED, never silent: the accumulators are 64-bit fixed point (2^-40 per unit), summed over the replicas
// here and — as SIGNED int64 — over the ranks of a multi-GPU apply afterwards.  A primitive's sum must therefore stay
// below 2^(63 - headroomBits) (headroomBits = ceil(log2(ranks))): a replica with its top bit set, a carry out of the
// replica sum or a sum at or beyond that bound raises *overflowFlag, and vr_apply_finish fails the apply.
__global__ void gather_flux_kernel(const unsigned long long *acc, unsigned stride, unsigned replicas,
                                   const unsigned *leafOfOrig, unsigned n, unsigned long long *outAcc, unsigned headroomBits,
                                   unsigned long long *overflowFlag) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const unsigned q = leafOfOrig[i];
    unsigned long long s = 0; // (integer sum: replica order is irrelevant)
    bool bad = false;
    for (unsigned r = 0; r < replicas; ++r) {
      const unsigned long long v = acc[(size_t)r * stride + q];
      bad = bad || (v >> 63) != 0ull;
      s += v;
      bad = bad || s < v; // carry out of 64 bits
    }
    bad = bad || (s >> (63u - headroomBits)) != 0ull;
    outAcc[i] = s;
    if (bad)
      *overflowFlag = 1ull;
  }
}

hipError_t launch_gather_flux(const unsigned long long *acc, unsigned stride, unsigned replicas,
                              const unsigned *leafOfOrig, unsigned n, unsigned long long *outAcc, unsigned headroomBits,
                              unsigned long long *overflowFlag, hipStream_t s) {
  hipLaunchKernelGGL(gather_flux_kernel, dim3((n + 255) / 256), dim3(256), 0, s, acc, stride, replicas, leafOfOrig, n,
                     outAcc, headroomBits, overflowFlag);
  return hipGetLastError();
}

#else // VR_USER_MODULE
// ---------------------------------------------------------------------------
// A particle model registered at RUN TIME (vr_register_particle_model, include/viennaray_amd.h): the library writes a
// translation unit that defines VR_USER_MODEL_FILE (the caller's model source: `struct VrUserModel`, appended to the
// registry in vr_particles.hpp) and includes this file; `hipcc --genco` turns it into a code object holding the
// extended trace kernels with that model compiled in.  The host finds them by their mangled names.
// ---------------------------------------------------------------------------
static_assert(VrUserModel::kNumData >= 1 && VrUserModel::kNumData <= VR_MAX_LABELS, "a model has 1 .. VR_MAX_LABELS data labels");
static_assert(VrUserModel::kNumData == VR_USER_NUM_DATA, "kNumData differs from the count given at registration");
constexpr int VR_USER_P = VrUserModel::kNeedsFull ? P_EXT_FULL : P_EXT;
#define VR_INST(DD, GG, MM) template __global__ void trace_kernel<DD, GG, VR_USER_P, MM>(const TraceParams);
VR_INST(2, 0, 0) VR_INST(2, 0, 4) VR_INST(2, 1, 0) VR_INST(2, 1, 4)
VR_INST(3, 0, 0) VR_INST(3, 0, 4) VR_INST(3, 1, 0) VR_INST(3, 1, 4)
#undef VR_INST
template __global__ void trace_kernel<2, 0, P_EXT, VrUserModel::kNeedsFull ? 0 : 3>(const TraceParams);
template __global__ void trace_kernel<3, 0, P_EXT, VrUserModel::kNeedsFull ? 0 : 3>(const TraceParams);
#endif // VR_USER_MODULE

} // namespace vr

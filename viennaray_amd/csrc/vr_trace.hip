// vr_trace.hip — the flux-tracing megakernel and its diagnostic siblings.
//
// One lane = one primary ray followed to termination; the state machine is the
// reference's TraceKernel::apply() ray loop (include/viennaray/rayTraceKernel.hpp:118-338)
// with Embree's rtcIntersect1 replaced by closest_hit() (vr_device.hpp).
#include <hip/hip_runtime.h>

#include "vr_device.hpp"
#include "vr_kernels.hpp"

namespace vr {

// fixed-point weight: 2^40 per unit (order-independent integer accumulation)
__device__ __forceinline__ u64 weight_fx(float w) { return (u64)((double)w * 1099511627776.0 + 0.5); }

template <int D>
__device__ __forceinline__ void source_sample(const TraceParams &p, Rng &rng, unsigned &t2, V3 &org, V3 &dir) {
  // raySourceRandom.hpp:50-68 — origin draws first
  org = mk(0.f, 0.f, 0.f);
  const float r1 = canon_f32(rng_next(rng, t2));
  setc(org, p.rayDir, p.minMax ? p.bbHi[p.rayDir] : p.bbLo[p.rayDir]);
  {
    const float lo = p.bbLo[p.firstDir], hi = p.bbHi[p.firstDir];
    setc(org, p.firstDir, lo + (hi - lo) * r1);
  }
  if (D == 2) {
    setc(org, p.secondDir, 0.f);
  } else {
    const float r2 = canon_f32(rng_next(rng, t2));
    const float lo = p.bbLo[p.secondDir], hi = p.bbHi[p.secondDir];
    setc(org, p.secondDir, lo + (hi - lo) * r2);
  }
  // raySourceRandom.hpp:70-116 — then the direction draws
  if (!p.useBasis) {
    const float d1 = canon_f32(rng_next(rng, t2));
    const float d2 = canon_f32(rng_next(rng, t2));
    float ct, st, cp, sp;
    cosine_sample(d1, d2, p.ee, ct, st, cp, sp);
    dir = mk(0.f, 0.f, 0.f);
    setc(dir, p.rayDir, p.posNeg * ct);
    setc(dir, p.firstDir, cp * st);
    setc(dir, p.secondDir, sp * st);
  } else {
    float dr;
    do {
      const float d1 = canon_f32(rng_next(rng, t2));
      const float d2 = canon_f32(rng_next(rng, t2));
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

// rayUtil.hpp:266-283 + rayReflection.hpp:31-50
template <int D> __device__ __forceinline__ V3 reflect_diffuse(const V3 &n, Rng &rng, unsigned &t2) {
  float x, y;
  double x2py2;
  do {
    x = canon_f32(rng_next(rng, t2)) * 2.0f + -1.0f;
    y = canon_f32(rng_next(rng, t2)) * 2.0f + -1.0f;
    x2py2 = (double)(x * x + y * y);
  } while (x2py2 >= 1.);
  const double tmp = 2. * sqrt(1. - x2py2);
  x = (float)((double)x * tmp);
  y = (float)((double)y * tmp);
  const float z = (float)(1. - 2 * x2py2);
  V3 r = mk(x + n.x, y + n.y, D == 3 ? z + n.z : 0.f);
  vnormalize(r);
  return r;
}

struct LaneCounters {
  unsigned traces, nongeo, geo, boundary, reflections, terminated, tier2;
};

// One primary ray, start to finish.
template <int D, int GEO, int PARTICLE>
__device__ __forceinline__ void trace_ray(const TraceParams &p, unsigned long long idx, u64 *tapeLane,
                                          u64 *scratchLane, LaneCounters &cnt) {
  Rng rng;
  rng_init(rng, tea3((unsigned)idx, p.seed), tapeLane, scratchLane);

  const float initialRayWeight = 1.f;
  float rayWeight = initialRayWeight;
  unsigned numReflections = 0, boundaryHits = 0;
  V3 org, rayDirection;
  source_sample<D>(p, rng, cnt.tier2, org, rayDirection);
  V3 dir = project_dir<D>(rayDirection); // what Embree sees (rayUtil.hpp:204-227)
  const float tnear = 1e-4f;             // rayUtil.hpp:229-231

  const float4 *__restrict__ prims = reinterpret_cast<const float4 *>(p.prims);
  bool hitFromBack = false;
  bool reflect;
  do {
    reflect = false;
    HitRec h;
    closest_hit<GEO>(p, org, dir, tnear, h);
    ++cnt.traces;
    if (h.geom < 0) { // rayTraceKernel.hpp:172-176
      ++cnt.nongeo;
      break;
    }
    const V3 hitPoint = mk(org.x + dir.x * h.t, org.y + dir.y * h.t, org.z + dir.z * h.t);

    if (h.geom == 0) { // boundary, rayTraceKernel.hpp:206-214 + rayBoundary.hpp:29-127
      if (++boundaryHits > p.maxBoundaryHits) {
        ++cnt.terminated;
        break;
      }
      const Tri &w = p.wall[h.prim];
      V3 ng = mk(w.Ng[0], w.Ng[1], w.Ng[2]);
      reflect = true;
      if (vdot(dir, ng) > 0.f) { // back side: pass through
        org = hitPoint;
        continue;
      }
      int bc, axis;
      bool minWall;
      if (D == 2 || h.prim <= 3u) {
        bc = p.bc0;
        axis = p.firstDir;
        minWall = h.prim <= 1u;
      } else {
        bc = p.bc1;
        axis = p.secondDir;
        minWall = h.prim <= 5u;
      }
      if (bc == 0) { // REFLECTIVE: rayBoundary.hpp:261-271
        vnormalize(ng);
        rayDirection = reflect_specular(rayDirection, ng);
        dir = project_dir<D>(rayDirection);
        org = hitPoint;
      } else if (bc == 1) { // PERIODIC
        org = hitPoint;
        setc(org, axis, minWall ? p.bbHi[axis] : p.bbLo[axis]);
      } else { // IGNORE
        reflect = false;
      }
      continue;
    }

    // geometry hit
    V3 geomNormal;
    if (GEO == 0) {
      const float4 n4 = prims[2 * h.pos + 1];
      geomNormal = mk(n4.x, n4.y, n4.z);
    } else {
      geomNormal = mk(prims[4 * h.pos + 1].w, prims[4 * h.pos + 2].w, prims[4 * h.pos + 3].w);
    }
    const bool backfaceHit = vdot(rayDirection, geomNormal) > 0.f; // rayTraceKernel.hpp:224
    if (GEO == 0) {
      if (backfaceHit) {
        if (hitFromBack) {
          ++cnt.terminated;
          break;
        }
        hitFromBack = true;
        reflect = true;
        org = hitPoint;
        continue;
      }
    } else if (backfaceHit) {
      ++cnt.terminated;
      break;
    }

    ++cnt.geo;
    const u64 wfx = weight_fx(rayWeight);
    atomicAdd(&p.fluxAcc[h.pos], wfx); // surfaceCollision, rayParticle.hpp:148-156
    if (GEO == 0) {
      // every overlapping neighbour disk is credited the full weight
      // (rayTraceKernel.hpp:271-300)
      const unsigned b = p.nbOff[h.pos], e = p.nbOff[h.pos + 1];
      for (unsigned j = b; j < e; ++j) {
        const unsigned q = p.nbIds[j];
        const float4 c4 = prims[2 * q];
        const float4 n4 = prims[2 * q + 1];
        if (local_disc_hit(org, dir, c4, mk(n4.x, n4.y, n4.z)))
          atomicAdd(&p.fluxAcc[q], wfx);
      }
    }

    // surfaceReflection is evaluated (and draws) even when sticking == 1 (Q2)
    V3 newDir;
    if (PARTICLE == 0)
      newDir = reflect_diffuse<D>(geomNormal, rng, cnt.tier2);
    else
      newDir = reflect_specular(rayDirection, geomNormal);
    const float sticking = p.primSticking ? p.primSticking[h.pos] : p.sticking;

    rayWeight -= rayWeight * sticking; // rayTraceKernel.hpp:316-319
    if (rayWeight <= 0.f)
      break;
    if (++numReflections > p.maxReflections) {
      ++cnt.terminated;
      break;
    }
    // rejectionControl, rayTraceKernel.hpp:435-460
    {
      const float lowerThreshold = (float)(0.1 * (double)initialRayWeight);
      const float renewWeight = (float)(0.3 * (double)initialRayWeight);
      if (rayWeight >= lowerThreshold) {
        reflect = true;
      } else {
        const double killProbability = 1.0 - (double)(rayWeight / renewWeight);
        if (canon_f64(rng_next(rng, cnt.tier2)) < killProbability) {
          reflect = false;
        } else {
          rayWeight = renewWeight;
          reflect = true;
        }
      }
    }
    if (!reflect)
      break;
    rayDirection = newDir;
    org = hitPoint;
    dir = project_dir<D>(rayDirection);
  } while (reflect);
  cnt.boundary += boundaryHits;
  cnt.reflections += numReflections;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned v) {
  unsigned long long s = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    s += __shfl_down(s, off, 64);
  return s;
}

template <int D, int GEO, int PARTICLE>
__global__ __launch_bounds__(VR_BLOCK) void trace_kernel(const TraceParams p) {
  __shared__ u64 tape[VR_TAPE * VR_BLOCK];
  __shared__ unsigned long long chunkBase;
  const unsigned tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const unsigned gwave = (blockIdx.x * VR_BLOCK + tid) >> 6;
  u64 *tapeLane = tape + tid;
  u64 *scratchLane = p.rngScratch + (size_t)gwave * (312u * 64u) + lane;
  LaneCounters cnt = {0, 0, 0, 0, 0, 0, 0};

  const unsigned long long total = p.rayEnd - p.rayFirst;
  for (;;) {
    if (tid == 0)
      chunkBase = atomicAdd(p.workCounter, (unsigned long long)p.chunk);
    __syncthreads();
    const unsigned long long base = chunkBase;
    __syncthreads();
    if (base >= total)
      break;
    const unsigned long long end = (base + p.chunk < total) ? base + p.chunk : total;
    for (unsigned long long r = base + tid; r < end; r += VR_BLOCK)
      trace_ray<D, GEO, PARTICLE>(p, p.rayFirst + r, tapeLane, scratchLane, cnt);
  }

  const unsigned vals[8] = {cnt.traces, cnt.nongeo, cnt.geo, 0u, cnt.boundary, cnt.reflections, cnt.terminated, cnt.tier2};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned long long s = wave_sum(vals[i]);
    if (lane == 0 && s)
      atomicAdd(&p.counters[i], s);
  }
}

// ---------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------
template <int D, int GEO, int PARTICLE>
static hipError_t launch_t(const TraceParams &p, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL((trace_kernel<D, GEO, PARTICLE>), dim3(grid), dim3(VR_BLOCK), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_trace(const TraceParams &p, int D, int geo, int particle, unsigned grid, hipStream_t s) {
  const int key = (D == 2 ? 0 : 4) | (geo ? 2 : 0) | (particle ? 1 : 0);
  switch (key) {
  case 0: return launch_t<2, 0, 0>(p, grid, s);
  case 1: return launch_t<2, 0, 1>(p, grid, s);
  case 2: return launch_t<2, 1, 0>(p, grid, s);
  case 3: return launch_t<2, 1, 1>(p, grid, s);
  case 4: return launch_t<3, 0, 0>(p, grid, s);
  case 5: return launch_t<3, 0, 1>(p, grid, s);
  case 6: return launch_t<3, 1, 0>(p, grid, s);
  default: return launch_t<3, 1, 1>(p, grid, s);
  }
}

// ---- diagnostics -----------------------------------------------------------
template <int GEO>
__global__ void debug_intersect_kernel(const TraceParams p, const float *org, const float *dir,
                                       const float *tnear, unsigned n, int *geomID, unsigned *primID, float *t) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  HitRec h;
  closest_hit<GEO>(p, mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]), mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]),
                   tnear[i], h);
  geomID[i] = h.geom;
  primID[i] = h.prim;
  t[i] = h.t;
}

hipError_t launch_debug_intersect(const TraceParams &p, int geo, const float *org, const float *dir,
                                  const float *tnear, unsigned n, int *geomID, unsigned *primID, float *t,
                                  hipStream_t s) {
  const unsigned grid = (n + 63) / 64;
  if (geo == 0)
    hipLaunchKernelGGL((debug_intersect_kernel<0>), dim3(grid), dim3(64), 0, s, p, org, dir, tnear, n, geomID, primID, t);
  else
    hipLaunchKernelGGL((debug_intersect_kernel<1>), dim3(grid), dim3(64), 0, s, p, org, dir, tnear, n, geomID, primID, t);
  return hipGetLastError();
}

template <int D>
__global__ __launch_bounds__(VR_BLOCK) void debug_source_kernel(const TraceParams p, const unsigned long long *idx,
                                                                 unsigned n, float *org, float *dir) {
  __shared__ u64 tape[VR_TAPE * VR_BLOCK];
  const unsigned tid = threadIdx.x;
  const unsigned i = blockIdx.x * VR_BLOCK + tid;
  if (i >= n)
    return;
  const unsigned gwave = i >> 6;
  Rng rng;
  rng_init(rng, tea3((unsigned)idx[i], p.seed), tape + tid, p.rngScratch + (size_t)gwave * (312u * 64u) + (tid & 63u));
  unsigned t2 = 0;
  V3 o, d;
  source_sample<D>(p, rng, t2, o, d);
  org[3 * i] = o.x;
  org[3 * i + 1] = o.y;
  org[3 * i + 2] = o.z;
  dir[3 * i] = d.x;
  dir[3 * i + 1] = d.y;
  dir[3 * i + 2] = d.z;
}

hipError_t launch_debug_source(const TraceParams &p, int D, const unsigned long long *idx, unsigned n, float *org,
                               float *dir, hipStream_t s) {
  const unsigned grid = (n + VR_BLOCK - 1) / VR_BLOCK;
  if (D == 2)
    hipLaunchKernelGGL((debug_source_kernel<2>), dim3(grid), dim3(VR_BLOCK), 0, s, p, idx, n, org, dir);
  else
    hipLaunchKernelGGL((debug_source_kernel<3>), dim3(grid), dim3(VR_BLOCK), 0, s, p, idx, n, org, dir);
  return hipGetLastError();
}

__global__ __launch_bounds__(VR_BLOCK) void debug_rng_kernel(unsigned seed32, unsigned count, u64 *scratch, u64 *out) {
  __shared__ u64 tape[VR_TAPE * VR_BLOCK];
  const unsigned tid = threadIdx.x;
  if (tid != 0)
    return;
  Rng rng;
  rng_init(rng, seed32, tape + tid, scratch);
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
__global__ void gather_flux_kernel(const unsigned long long *acc, const unsigned *leafOfOrig, unsigned n,
                                   unsigned long long *outAcc) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    outAcc[i] = acc[leafOfOrig[i]];
}

hipError_t launch_gather_flux(const unsigned long long *acc, const unsigned *leafOfOrig, unsigned n,
                              unsigned long long *outAcc, hipStream_t s) {
  hipLaunchKernelGGL(gather_flux_kernel, dim3((n + 255) / 256), dim3(256), 0, s, acc, leafOfOrig, n, outAcc);
  return hipGetLastError();
}

} // namespace vr

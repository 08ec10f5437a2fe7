// vr_trace_kernel.hpp — trace_kernel, the persistent-wave tracer (see vr_trace.hip for the pipeline it belongs to),
// and the device helpers only it uses.  A header so that the two translation units that instantiate it
// (vr_trace.hip: the whole-ray kernels; vr_trace_wf.hip: the staged kernels of the generation-by-generation path)
// compile side by side.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "vr_device.hpp"
#include "vr_kernels.hpp"
#include "vr_particles.hpp"

namespace vr {

// fixed-point weight: 2^40 per unit (order-independent integer accumulation)
__device__ __forceinline__ u64 weight_fx(float w) { return (u64)((double)w * 1099511627776.0 + 0.5); }

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
// non-temporal 16-byte accesses (queue and hit records stream through once: they should not push the scene out of L2)
constexpr unsigned VR_EMIT_BLOCK = 2048;
typedef float vr_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4 *q) {
  const vr_f4 v = __builtin_nontemporal_load(reinterpret_cast<const vr_f4 *>(q));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float4 v, float4 *q) {
  vr_f4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<vr_f4 *>(q));
}

// STAGE (the generation-by-generation path, MODE 0 only; vr_trace_wf.hip):
//   0 whole rays: a ray stays in its lane from the source to its end (the kernels of all modes)
//   1 FIRST: sorted primaries from the sort bins as in 0, but a ray that survives its segment is appended to the
//     next generation's QUEUE (64-byte full-state record) instead of continuing in its lane
//   2 INTERSECT: rays of a queue, closest hit only (walk + walls) -> one 16-byte hit record per ray; no state
//     machine, no RNG: few registers, many waves, and a finished lane costs one store and one load
//   3 SHADE: queue + hit records -> the state machine on full wavefronts; survivors go to the next queue
//   4 RESUME: like 0, but the rays come from a queue (the last generations run to their end in their lanes)
#ifndef VR_WF_ISECT_WAVES
#define VR_WF_ISECT_WAVES 6
#endif
#ifndef VR_WF_SHADE_WAVES
#define VR_WF_SHADE_WAVES 6
#endif
constexpr int trace_waves(int mode, int stage) {
  return stage == 2 ? VR_WF_ISECT_WAVES : (stage == 3 ? VR_WF_SHADE_WAVES : (mode == 1 ? 8 : (mode == 2 ? 7 : (mode == 3 ? 5 : (mode == 4 ? 5 : 6)))));
}
template <int D, int GEO, int PARTICLE, int MODE_, int STAGE = 0>
__global__ __launch_bounds__(VR_BLOCK) __attribute__((amdgpu_num_sgpr(80)))
__attribute__((amdgpu_waves_per_eu(trace_waves(MODE_, STAGE), trace_waves(MODE_, STAGE)))) void
trace_kernel(const TraceParams p) {
  constexpr bool QUEUE_IN = STAGE >= 2;           // rays come from a queue of full-state records
  constexpr bool EMIT = STAGE == 1 || STAGE == 3; // a ray alive after its segment goes to the next queue
  constexpr bool ISECT = STAGE == 2, SHADE = STAGE == 3;
  static_assert(STAGE == 0 || MODE_ == 0, "staged kernels exist for the general kernel only");
  constexpr bool SMALL = MODE_ == 4;
  constexpr int MODE = SMALL ? 0 : MODE_;
  constexpr bool ABSORB = MODE == 1 || MODE == 2;
  // PARTICLE 0 / 1: DiffuseParticle / SpecularParticle compiled in.  PARTICLE 2 (P_EXT): the
  // extended kernel — particle kind, data labels, WDIST crediting and mean-free-path scattering
  // decided at run time from TraceParams (vr_particles.hpp)
  constexpr bool EXT = PARTICLE == P_EXT;
  // packet-query rounds credit disks wave-uniformly from the candidate list (pq_credit) instead of
  // walking the neighbour CSR per lane
  constexpr bool PQ_CREDIT = GEO == 0 && !EXT && (MODE == 1 || MODE == 3);
  PqCands cands;
  cands.local = 0ull;
  cands.count = 0u;
  cands.rec = nullptr;
  // CARRY: lanes whose BVH walk is still under way when most of the wave is done keep
  // their cursor over the state-machine / refill phase (see the round structure below).
  // The absorbing kernel for flat scenes does without: its rounds are packets, and the extra
  // live registers would cost it the 8th wave per SIMD.
  constexpr bool CARRY = MODE != 1;
  __shared__ float wallS[96];
  // per-lane event counters live in LDS (fire-and-forget ds_add), not in 7 VGPRs
  __shared__ unsigned cntS[ISECT ? 1 : 8 * VR_BLOCK];
  __shared__ unsigned pqS[(VR_BLOCK / 64) * 128]; // packet query: per-wave frontier lists
  __shared__ uint4 candS[PQ_CREDIT ? (VR_BLOCK / 64) * VR_PQ_CANDS : 1]; // ... and candidate records (pq_credit)
  // per-lane stack of the ordered walk, [entry][lane]; the absorbing flat-scene kernel walks rarely and keeps its
  // 8 waves per SIMD with a short LDS part (deeper entries: global slab)
  constexpr bool ORDERED = MODE != 1; // (MODE 1 walks rarely: it keeps the escape-link walk, one register of state)
  constexpr int SD = SMALL ? VR_SMALL_STACK : VR_STACK_LDS;
  __shared__ unsigned stackS[ORDERED && !SHADE ? SD * VR_BLOCK : 1];
  // (MODE 4: the scene copy is the kernel's dynamic LDS — smallBytes of it, so a smaller scene leaves room for a
  //  fifth block per CU)
  extern __shared__ uint4 sceneS[];
  unsigned char *const sceneB = reinterpret_cast<unsigned char *>(sceneS);
  const unsigned tid = threadIdx.x;
  cands.rec = candS + (PQ_CREDIT ? (tid >> 6) * VR_PQ_CANDS : 0u);
  const unsigned lane = tid & 63u;
  const unsigned gwave = (blockIdx.x * VR_BLOCK + tid) >> 6;
  if (tid < 96)
    wallS[tid] = p.wallTable[tid];
  if (!ISECT) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      cntS[k * VR_BLOCK + tid] = 0u;
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
  enum { K_TRACES = 0, K_NONGEO, K_GEO, K_BOUNDARY, K_REFL, K_TERM, K_TIER2, K_PARTICLE };
#define VR_COUNT(k, v) atomicAdd(&cnt[(k) * VR_BLOCK], (unsigned)(v))

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
  unsigned numReflections = 0, boundaryHits = 0;
  bool hitFromBack = false;
  bool start = false; // this lane begins a new trace segment in this round
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
  // (QUEUE_IN: the rays are records 0 .. qN - 1 of a queue; "bin" b is its b-th run of binCap records)
  unsigned qN = 0, qi = 0; // qi: queue index of the lane's ray (where INTERSECT stores its hit)
  if (QUEUE_IN) {
    qN = *(ConstU32)p.qInCount;
    qN = qN < p.qCap ? qN : p.qCap;
  }
  const unsigned numBinsEff = QUEUE_IN ? (qN + p.binCap - 1) / p.binCap : p.numBins;
  unsigned ovCount = 0;
  if (!QUEUE_IN)
    ovCount = binCount[p.numBins] < p.ovCap ? binCount[p.numBins] : p.ovCap;
  const unsigned totalBins = numBinsEff + (ovCount + p.binCap - 1) / p.binCap;
  // bins per grab of the work queue: the host's choice — but a queue's length is only known here, and a short one
  // must not go to a few waves only
  unsigned chunkEff = p.chunk;
  if (QUEUE_IN) {
    const unsigned share = numBinsEff / (gridDim.x * (VR_BLOCK / 64) * 2u);
    chunkEff = share < chunkEff ? (share ? share : 1u) : chunkEff;
  }
  unsigned curBin = 0, spanStart = 0, spanEnd = 0, curOff = 0, curCnt = 0, curBase = 0;
  unsigned spanCounts = 0; // lane i: ray count of bin spanStart + i
  unsigned packetSkip = 0, packetFails = 0; // wave-uniform back-off of packet attempts
  unsigned pqSkip = 0, pqFails = 0;         // ... and of packet-query attempts
  bool exhausted = false;
  unsigned emitBase = 0, emitUsed = VR_EMIT_BLOCK; // EMIT: this wave's block of the output queue (wave-uniform)
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
            unsigned long long s = 0;
            if (lane == 0)
              s = atomicAdd(p.workCounter, (unsigned long long)chunkEff);
            s = bcast64(s);
            if (s >= totalBins) {
              exhausted = true;
              break;
            }
            curBin = spanStart = (unsigned)s;
            spanEnd = (unsigned)((s + chunkEff < totalBins) ? s + chunkEff : totalBins);
            // the span's bin counts in one coalesced load (lane i <- bin spanStart + i; chunk <= 64)
            const unsigned bi = spanStart + lane;
            if (QUEUE_IN)
              spanCounts = bi < spanEnd ? (qN - bi * p.binCap < p.binCap ? qN - bi * p.binCap : p.binCap) : 0u;
            else
              spanCounts = (bi < spanEnd && bi < p.numBins) ? p.binCount[bi] : 0u;
          } else {
            ++curBin;
          }
          curOff = 0;
          if (curBin < numBinsEff) {
            const unsigned c = __shfl(spanCounts, (int)(curBin - spanStart), 64);
            curCnt = c < p.binCap ? c : p.binCap;
            curBase = curBin * p.binCap;
          } else {
            const unsigned k = (curBin - numBinsEff) * p.binCap;
            curCnt = ovCount - k < p.binCap ? ovCount - k : p.binCap;
            curBase = numBinsEff * p.binCap + k;
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
      if (slot != 0xFFFFFFFFu) {
        DIAG(8);
        const unsigned j = slot;
        constexpr unsigned REC = QUEUE_IN ? 4 : (ABSORB ? 2 : 3); // float4 per record
        const float4 *__restrict__ recs = QUEUE_IN ? reinterpret_cast<const float4 *>(p.qIn) : rayAB;
        // (queue records stream through once: non-temporal, so they do not push the scene out of L2)
        const float4 a = QUEUE_IN ? nt_load4(recs + REC * (size_t)j) : recs[REC * (size_t)j];
        const float4 b = QUEUE_IN ? nt_load4(recs + REC * (size_t)j + 1) : recs[REC * (size_t)j + 1];
        org = mk(a.x, a.y, a.z);
        rayDirection = mk(a.w, b.x, b.y);
        const bool dead = QUEUE_IN && a.w == 0.f && b.x == 0.f && b.y == 0.f; // (an unused slot of a wave's block)
        dir = project_dir<D>(rayDirection); // what Embree sees (rayUtil.hpp:204-227)
        rayWeight = 1.f;                    // Source::getInitialRayWeight
        numReflections = 0;
        boundaryHits = 0;
        hitFromBack = false;
        active = !dead;
        start = !dead;
        if (QUEUE_IN) {
          // full-state record of a ray under way (queue_store below)
          qi = j;
          if (!ISECT) {
            const float4 curF = nt_load4(recs + REC * (size_t)j + 2);
            const float4 e = nt_load4(recs + REC * (size_t)j + 3);
            ulonglong2 cur;
            cur.x = ((u64)__float_as_uint(curF.y) << 32) | __float_as_uint(curF.x);
            cur.y = ((u64)__float_as_uint(curF.w) << 32) | __float_as_uint(curF.z);
            rayWeight = b.z;
            numReflections = __float_as_uint(b.w);
            boundaryHits = __float_as_uint(e.z) & 0x7FFFFFFFu;
            hitFromBack = (__float_as_uint(e.z) >> 31) != 0u;
            rng_resume(rng, __float_as_uint(e.x), __float_as_uint(e.y), cur.x, cur.y);
          }
          if (SHADE) {
            const float4 hf = nt_load4(reinterpret_cast<const float4 *>(p.hitBuf) + j);
            const uint4 hh = make_uint4(__float_as_uint(hf.x), __float_as_uint(hf.y), __float_as_uint(hf.z), __float_as_uint(hf.w));
            h.t = __uint_as_float(hh.x);
            h.geom = (int)hh.y;
            h.prim = hh.z;
            h.pos = hh.w;
          }
        } else if (!ABSORB) {
          const unsigned idxOff = __float_as_uint(b.z);
          const ulonglong2 cur = *reinterpret_cast<const ulonglong2 *>(rayAB + REC * (size_t)j + 2);
          rng_resume(rng, tea3((unsigned)(p.batchFirst + idxOff), p.seed), __float_as_uint(b.w), cur.x, cur.y);
        }
      }
    }
    if (!ballot64(active)) {
      // (a queue has dead records — the unused ends of the emitting waves' blocks: a round that drew nothing but
      //  those is not the end of the queue)
      if (!QUEUE_IN || exhausted)
        break;
      continue;
    }
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
    if (!SHADE && (!CARRY || start)) { // (!CARRY: every active lane starts a segment in every round)
      hit_clear(h);
      node = 0u;
    }
    const unsigned long long carried = CARRY ? ballot64(active && !start) : 0ull;
    start = false;
    const bool usePacket = !SMALL && !QUEUE_IN &&
        !(p.debugFlags & 32u) && carried == 0ull && packetSkip == 0 && __popcll(ballot64(active)) >= 8;
    bool packetDone = false;
    bool pqCredit = false; // this round's surface hits are credited from the packet's candidate list
    if (!SMALL && usePacket && p.wide && !(p.debugFlags & 128u)) {
      // first choice: the box query (one wide-tree search for the whole wave)
      if (pqSkip == 0) {
        if (active) {
          DIAG(12);
        }
        packetDone = pq_hit_packet<GEO, PQ_CREDIT>(p, active, org, dir, tnear, h, pqS + (tid >> 6) * 128u, cands VR_DIAG_PASS);
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
    if (!packetDone && !SHADE) {
      const unsigned walking = (unsigned)__popcll(ballot64(active && (ORDERED ? node != VR_END : node < p.numNodes)));
      const unsigned minLanes = (!CARRY || exhausted || walking <= p.walkExit) ? 1u : p.walkExit;
      if (ORDERED)
        pair_walk_lanes<GEO, SD, MODE != 2 && !(ISECT && VR_WF_ISECT_WAVES >= 7)>(p, pnodes, prims, stackS + tid, stackG, active, org, dir, tnear, h, node, sp, minLanes VR_DIAG_PASS);
      else
        bvh_walk_lanes<GEO>(p, active, org, dir, tnear, h, node, minLanes VR_DIAG_PASS);
    }
    const bool fin = active && (SHADE || (ORDERED ? node == VR_END : node >= p.numNodes)); // this lane's geometry walk is complete
#ifdef VR_SELFCHECK
    { // -DVR_SELFCHECK build: every finished segment again with the escape-link walk; disagreements are
      // counted in counters[48], the first one is kept in counters[50..]
      HitRec hb;
      hit_clear(hb);
      unsigned nb = fin ? 0u : VR_END;
      bvh_walk_lanes<GEO>(p, fin, org, dir, tnear, hb, nb, 1u VR_DIAG_PASS);
      if (fin && (hb.geom != h.geom || (hb.geom == 1 && (hb.t != h.t || hb.pos != h.pos)))) {
        if (atomicAdd(&p.counters[48], 1ull) == 0ull) {
          const float v[8] = {org.x, org.y, org.z, dir.x, dir.y, dir.z, h.t, hb.t};
          for (int k = 0; k < 8; ++k)
            p.counters[50 + k] = (unsigned long long)__float_as_uint(v[k]);
          p.counters[58] = ((unsigned long long)h.pos << 32) | hb.pos;
          p.counters[59] = ((unsigned long long)(unsigned)h.geom << 32) | (unsigned)hb.geom;
        }
      }
    }
#endif
    TICK(3);
    if (fin && !SHADE)
      hit_walls(p, wallS, org, dir, tnear, h); // boundary walls, where one can come before the hit
    TICK(4);
    if (ISECT) {
      // the closest hit is all this stage computes: one 16-byte record per ray, and the lane is free
      if (fin) {
        nt_store4(make_float4(h.t, __uint_as_float((unsigned)h.geom), __uint_as_float(h.prim), __uint_as_float(h.pos)), reinterpret_cast<float4 *>(p.hitBuf) + qi);
        active = false;
      }
      continue;
    }
    // Merge same-disk credits of the wave into one atomic when that is likely to pay: rays of a
    // packet, or — sampled on one lane's target — when a good share of the wave's hits fall on
    // the same primitive (sorted rays on a coarse scene: one vector atomic with 64 lanes on ONE
    // address is serialised lane by lane in the L2 atomic unit).
    bool aggregate = packetDone;
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
        if (EXT && p.meanFreePath > 0.f) {
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
            const float4 n4 = prims[2 * h.pos + 1];
            geomNormal = mk(n4.x, n4.y, n4.z);
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
              auto creditTo = [&](unsigned q, float w, const V3 &nq) {
                Particles::collide(kind, w, rayDirection, nq, [&](int label, float v) {
                  atomicAdd(&fluxAcc[(size_t)label * (SMALL ? p.numPrims : p.planeStride) + q], weight_fx(v));
                });
              };
              if (GEO == 0) {
                const unsigned nb = nbOff[h.pos], ne = nbOff[h.pos + 1];
                float invSum = 0.f, dClosest = 0.f;
                unsigned numHit = 1;
                if (p.useWdist) {
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
                creditTo(h.pos, p.useWdist ? rayWeight / dClosest / invSum * (float)numHit : rayWeight, geomNormal);
                for (unsigned j = nb; j < ne; ++j) {
                  const unsigned q = nbIds[j];
                  const float4 n4 = prims[2 * q + 1];
                  const V3 nq = mk(n4.x, n4.y, n4.z);
                  float dist;
                  if (local_disc_hit_dist(org, dir, prims[2 * q], nq, dist))
                    creditTo(q, p.useWdist ? rayWeight / (dist + 1e-6f) / invSum * (float)numHit : rayWeight, nq);
                }
              } else {
                creditTo(h.pos, rayWeight, geomNormal);
              }
            }
            if (ABSORB) {
              // sticking >= 1: weight drops to <= 0 (:316-319); the reflection draws
              // the reference makes before that test (Q2) are not observable.
              active = false;
            } else {
              const float sticking = p.primSticking ? primSticking[h.pos] : p.sticking;
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
                  newDir = Particles::reflect<D>(p.particleKind, p, rayDirection, geomNormal, rng, cnt[K_TIER2 * VR_BLOCK]);
                rayWeight = wAfter;
                if (++numReflections > p.maxReflections) { // :320-324
                  VR_COUNT(K_TERM, 1);
                  active = false;
                } else {
                  // rejectionControl, :435-460
                  const float lowerThreshold = (float)(0.1 * 1.0);
                  const float renewWeight = (float)(0.3 * 1.0);
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
    if (EMIT) {
      // a ray that goes on leaves this kernel: its full state is appended to the next generation's queue
      // (one counter bump per wave and round)
      const bool out = fin && active;
      const unsigned long long om = ballot64(out);
      if (om) {
        // queue space is reserved per wave in blocks of VR_EMIT_BLOCK records (one bump of the shared counter per
        // block, not per round: 10^6 bumps of one address cost more than the stage's work); what a wave leaves
        // unused of its blocks stays zero-filled and reads as a dead record
        const unsigned need = (unsigned)__popcll(om);
        if (emitUsed + need > VR_EMIT_BLOCK) {
          unsigned nb = 0;
          if (lane == 0)
            nb = atomicAdd(p.qOutCount, VR_EMIT_BLOCK);
          emitBase = (unsigned)__builtin_amdgcn_readfirstlane((int)nb);
          emitUsed = 0;
        }
        const unsigned k = emitBase + emitUsed + (unsigned)__popcll(om & ((1ull << lane) - 1ull));
        emitUsed += need;
        if (out) {
          if (k < p.qCap) {
            float4 *q = reinterpret_cast<float4 *>(p.qOut) + 4 * (size_t)k;
            nt_store4(make_float4(org.x, org.y, org.z, rayDirection.x), q);
            nt_store4(make_float4(rayDirection.y, rayDirection.z, rayWeight, __uint_as_float(numReflections)), q + 1);
            nt_store4(make_float4(__uint_as_float((unsigned)rng.lo), __uint_as_float((unsigned)(rng.lo >> 32)),
                                  __uint_as_float((unsigned)rng.hi), __uint_as_float((unsigned)(rng.hi >> 32))), q + 2);
            // (a tier-2 engine state lives in this lane's slab: the next owner rebuilds it from {seed, k})
            nt_store4(make_float4(__uint_as_float(rng.seed), __uint_as_float(rng.k),
                                  __uint_as_float(boundaryHits | (hitFromBack ? 0x80000000u : 0u)), 0.f), q + 3);
          } else {
            p.counters[61] = 1ull; // queue overflow: reported by vr_apply_finish, never silent
          }
          active = false;
          start = false;
        }
      }
    }
    TICK(5);
    if (PQ_CREDIT && pqCredit && !(p.debugFlags & 1u)) {
      // ---- surfaceCollision for the round's surface hits, candidate by candidate (wave-uniform):
      // a lane credits candidate q if q is its closest disk, or q is a neighbour of that disk
      // (centres within nbDist: the relation the CSR was built from, rayPointNeighborhood.hpp:287-298,
      // evaluated on the same floats) and its ray passes the neighbour test on q.  All lanes
      // crediting q add to ONE address: a single atomic (absorbing: count x unit weight).
      if (ballot64(creditLane)) {
        float px = 0.f, py = 0.f, pz = 0.f; // centre of this lane's closest disk
        for (unsigned c = 0; c < cands.count; ++c) {
          const uint4 cr = cands.rec[c]; // (same address in every lane: an LDS broadcast)
          const bool mine = h.pos == cr.x;
          px = mine ? __uint_as_float(cr.y) : px;
          py = mine ? __uint_as_float(cr.z) : py;
          pz = mine ? __uint_as_float(cr.w) : pz;
        }
        const float dist = p.nbDist, dist2 = dist * dist;
        for (unsigned c = 0; c < cands.count; ++c) {
          DIAG(6);
          const uint4 cr = cands.rec[c];
          const unsigned q = (unsigned)__builtin_amdgcn_readfirstlane((int)cr.x);
          const float dx = px - __uint_as_float(cr.y), dy = py - __uint_as_float(cr.z), dz = pz - __uint_as_float(cr.w);
          bool near = fabsf(dx) <= dist && fabsf(dy) <= dist && (p.geoD == 2 || fabsf(dz) <= dist);
          near = near && ((dx * dx + dy * dy) + dz * dz) <= dist2;
          const bool sel = creditLane && (h.pos == q || (near && ((cands.local >> c) & 1ull)));
          if (ABSORB) {
            const unsigned long long m = ballot64(sel);
            if (m && lane == (unsigned)(__ffsll((long long)m) - 1))
              atomicAdd(&fluxAcc[q], (u64)__popcll(m) * 1099511627776ull); // unit weights: count x 2^40
          } else {
            credit_aggregated(fluxAcc, sel, q, creditW);
          }
        }
      }
    }
    TICK(6);
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
  if (ISECT)
    return;
  const unsigned vals[8] = {cnt[K_TRACES * VR_BLOCK], cnt[K_NONGEO * VR_BLOCK], cnt[K_GEO * VR_BLOCK], cnt[K_PARTICLE * VR_BLOCK],
                            cnt[K_BOUNDARY * VR_BLOCK], cnt[K_REFL * VR_BLOCK], cnt[K_TERM * VR_BLOCK],
                            cnt[K_TIER2 * VR_BLOCK]};
#undef VR_COUNT
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned long long s = wave_sum(vals[i]);
    if (lane == 0 && s)
      atomicAdd(&p.counters[i], s);
  }
}

} // namespace vr

// vr_device.hpp — device-side building blocks of the flux tracer (gfx950).
//
// Everything here is compiled with -ffp-contract=off: the order of float
// operations is part of the parity contract with the reference semantics
// (see DESIGN.md §Numerics); fused multiply-adds are written explicitly where
// they are wanted (BVH slab test only).
#pragma once
#include <hip/hip_runtime.h>

#include "vr_libm.hpp"
#include "vr_types.hpp"

namespace vr {

struct V3 {
  float x, y, z;
};

__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
// (the three components as VALUES before the choice: written as a choice between v.x, v.y and v.z the compiler selects an
//  ADDRESS and loads through it — a vector whose component is picked by a run-time axis (rises_clear, relief_clip) then
//  lives in scratch memory for the whole kernel: the general kernels kept the ray direction there, three scratch loads at
//  every use.  Round 4, found by reading the ISA for scratch_ after the vector L1 turned out to be the bound: C4 -7 %,
//  C2 sticking 0.1 -5 %, the headline -3 %, C5 -3 %, trench3D +- 0)
__device__ __forceinline__ float getc(const V3 &v, int a) {
  const float x = v.x, y = v.y, z = v.z;
  return a == 0 ? x : (a == 1 ? y : z);
}
__device__ __forceinline__ void setc(V3 &v, int a, float f) {
  v.x = a == 0 ? f : v.x;
  v.y = a == 1 ? f : v.y;
  v.z = a == 2 ? f : v.z;
}
// ViennaCore DotProduct: sequential accumulation starting from 0
__device__ __forceinline__ float vdot(const V3 &a, const V3 &b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// Embree SSE2 dot: x + (y + z)
__device__ __forceinline__ float edot(const V3 &a, const V3 &b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ V3 ecross(const V3 &a, const V3 &b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 vsub(const V3 &a, const V3 &b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ void vnormalize(V3 &a) {
  float n = sqrtf(vdot(a, a));
  if (n <= 0.f)
    return;
  a.x /= n;
  a.y /= n;
  a.z /= n;
}

// ---------------------------------------------------------------------------
// Per-ray RNG: mt19937_64 seeded with tea<3>(idx, seed)
// (reference: rayTraceKernel.hpp:100,120-121; engine = std::mt19937_64).
//
// A fresh 312-word engine per ray is what makes the reference's CPU loop
// expensive; on the GPU the stream is evaluated lazily from its recurrence
//   s[0] = seed, s[j] = 6364136223846793005 (s[j-1] ^ s[j-1]>>62) + j  (j < 312)
//   s[n+312] = s[n+156] ^ tw(s[n], s[n+1]),  output k = temper(s[k+312])
// Tier 1: outputs 0..155 depend on seeding words only (s[k], s[k+1], s[k+156]); the
//   generator reaches s[156] once per ray (the unavoidable 156 sequential steps) and
//   from there every draw advances two cursors by one step each (struct Rng below).
// Tier 2: a ray that draws more than 156 numbers builds the whole 312-word state in a
//   per-lane global scratch slab ([word][lane] so a wave's accesses coalesce) and
//   continues with the textbook block twist.
// ---------------------------------------------------------------------------
typedef unsigned long long u64;

// wave-wide vote as a lane mask.  (HIP's __ballot compares a materialised 0/1 value:
// v_cndmask + v_cmp per call; the builtin hands the condition mask through.)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// six comparisons of values already in registers as ONE straight line (a chain of && compiles to nested exec-mask branches)
__device__ __forceinline__ bool all6(bool a, bool b, bool c, bool d, bool e, bool f) {
  return ((int)a & (int)b & (int)c & (int)d & (int)e & (int)f) != 0;
}

// -DVR_DIAG: lane-occupancy counters.  DIAG(k) inside any (divergent) region counts one
// wave-level execution and the lanes that took part; summed into counters[16 + 2k, +1].
#ifdef VR_DIAG
#define VR_DIAG_DECL unsigned diagW[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, diagL[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define DIAG(k)                                                                                                        \
  do {                                                                                                                 \
    const unsigned long long m_ = ballot64(1);                                                                         \
    if ((int)(threadIdx.x & 63u) == __ffsll((long long)m_) - 1)                                                        \
      ++diagW[k];                                                                                                      \
    ++diagL[k];                                                                                                        \
  } while (0)
// TICK(k): wave time (s_memtime, core clock) since the previous TICK goes to phase k; phaseT = this wave's
// row of an LDS table, summed into counters[64 + k] at the end of the kernel
#define TICK(k)                                                                                                        \
  do {                                                                                                                 \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                      \
    if ((threadIdx.x & 63u) == 0u)                                                                                     \
      phaseT[k] += now_ - tLast;                                                                                       \
    tLast = now_;                                                                                                      \
  } while (0)
// SUB_START / SUB_STOP(k): a stretch inside divergent code (first active lane books it); part of the enclosing phase
#define SUB_START const unsigned long long sub0_ = __builtin_amdgcn_s_memtime();
#define SUB_MARK(k)                                                                                                    \
  do {                                                                                                                 \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                      \
    const unsigned long long m_ = ballot64(1);                                                                         \
    if ((int)(threadIdx.x & 63u) == __ffsll((long long)m_) - 1)                                                        \
      phaseT[k] += now_ - tLast;                                                                                       \
  } while (0)
#define SUB_STOP(k)                                                                                                    \
  do {                                                                                                                 \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                      \
    const unsigned long long m_ = ballot64(1);                                                                         \
    if ((int)(threadIdx.x & 63u) == __ffsll((long long)m_) - 1)                                                        \
      phaseT[k] += now_ - sub0_;                                                                                       \
  } while (0)
#define VR_DIAG_ARGS , unsigned (&diagW)[16], unsigned (&diagL)[16], unsigned long long *phaseT, unsigned long long &tLast
#define VR_DIAG_PASS , diagW, diagL, phaseT, tLast
#else
#define VR_DIAG_DECL
#define DIAG(k)
#define TICK(k)
#define SUB_START
#define SUB_MARK(k)
#define SUB_STOP(k)
#define VR_DIAG_ARGS
#define VR_DIAG_PASS
#endif

__device__ __forceinline__ unsigned tea3(unsigned v0, unsigned v1) {
  unsigned s0 = 0;
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  return v0;
}

__device__ __forceinline__ u64 mt_step(u64 p, unsigned j) { return 6364136223846793005ull * (p ^ (p >> 62)) + j; }
// The same step for the generator's seeding chain (step index a literal after unrolling), written over 32-bit
// halves with the multiplier's halves held in VGPRs: v_mad_u64_u32 can then take the step index as its scalar
// 64-bit addend (gfx9 VOP3: one scalar source per instruction) — 6 VALU per step (3 of them multiplies)
// instead of the 7 + a 64-bit add the compiler makes of the one-line form.
// (The high word by two more v_mad_u64_u32 chained onto the first — 5 VALU + a register move — issues in 20.1
//  SIMD-cycles per step against 24.4 in isolation (tools/mt_step_variants.py, vr_bench.hip kind 6), but the chip
//  then holds 2.0 - 2.27 GHz instead of 2.39, and the generator, which already runs at 1.9 GHz, got SLOWER: 4.66 ->
//  4.8 - 4.95 ms.  The generator is bound by power, not by issue slots; v_mul_lo_u32 is the cheaper multiply.)
struct MtMul {
  unsigned al, ah;
};
__device__ __forceinline__ MtMul mt_mul_init() {
  MtMul m{0x4C957F2Du, 0x5851F42Du}; // 6364136223846793005 = 0x5851F42D4C957F2D
  asm volatile("" : "+v"(m.al), "+v"(m.ah));
  return m;
}
__device__ __forceinline__ u64 mt_step_v(const MtMul &m, u64 p, unsigned j) {
  const unsigned xh = (unsigned)(p >> 32);
  const unsigned t = (unsigned)p ^ (xh >> 30); // low word of p ^ (p >> 62); the high word is unchanged
  u64 r; // = t * al + j, the index in a scalar register pair (written out: LLVM's constant hoisting otherwise
         //   rebuilds the indices from a base held in VGPRs, at two more VALU per step)
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(t), "v"(m.al), "s"((u64)j) : "vcc");
  const unsigned hi = (unsigned)(r >> 32) + t * m.ah + xh * m.al;
  return ((u64)hi << 32) | (unsigned)r;
}
__device__ __forceinline__ u64 mt_twist(u64 a, u64 b, u64 c) {
  u64 y = (a & 0xFFFFFFFF80000000ull) | (b & 0x7FFFFFFFull);
  return c ^ (y >> 1) ^ ((b & 1ull) ? 0xB5026F5AA96619E9ull : 0ull);
}
__device__ __forceinline__ u64 mt_temper(u64 v) {
  v ^= (v >> 29) & 0x5555555555555555ull;
  v ^= (v << 17) & 0x71D67FFFEDA60000ull;
  v ^= (v << 37) & 0xFFF7EEE000000000ull;
  v ^= (v >> 43);
  return v;
}

// Streaming form of the first 156 outputs: output k = temper(s[k+156] ^ tw(s[k], s[k+1]))
// depends on three words of the SEEDING recurrence only, and both cursors advance by one
// mt_step per draw.  A ray therefore carries {k, lo = s[k], hi = s[k+156]} (16 B + a
// counter) and every draw costs two 64-bit multiply-adds — no tape, no LDS, no state
// array — until draw 156, where the full 312-word state is needed (tier 2).
struct Rng {
  unsigned seed;   // engine seed (32 bit)
  unsigned k;      // index of the next output
  unsigned pos;    // tier 2: position in the 312-word block, 0xFFFFFFFF = not built
  u64 lo, hi;      // s[k], s[k+156] while k < 156
  u64 *scratch;    // global, this lane's column: scratch[word * 64]
};

__device__ __forceinline__ void rng_resume(Rng &r, unsigned seed, unsigned k, u64 lo, u64 hi) {
  r.seed = seed;
  r.k = k;
  r.pos = 0xFFFFFFFFu;
  r.lo = lo;
  r.hi = hi;
}

// cold start: 156 steps of the recurrence to reach s[156]
__device__ __forceinline__ void rng_init(Rng &r, unsigned seed, u64 *scratchLane) {
  const MtMul m = mt_mul_init();
  u64 x = seed;
#pragma unroll 39
  for (int j = 1; j <= 156; ++j)
    x = mt_step_v(m, x, j);
  rng_resume(r, seed, 0, (u64)seed, x);
  r.scratch = scratchLane;
}

// First K outputs of the engine, all in registers (static indexing): what the
// generator needs when the number of draws is known at compile time.  Also returns the
// streaming cursors {s[K], s[K+156]} for the draws that follow.
template <int K> __device__ __forceinline__ void mt_first_outputs(unsigned seed, u64 (&out)[K], u64 &lo, u64 &hi) {
  const MtMul m = mt_mul_init();
  u64 w[K + 1];
  u64 x = seed;
  w[0] = x;
#pragma unroll
  for (int j = 1; j <= K; ++j) {
    x = mt_step_v(m, x, j);
    w[j] = x;
  }
#pragma unroll 38 // (152 steps = 4 x 38: the step needs its index as a literal; deeper unrolling loses in the I-cache)
  for (int j = K + 1; j < 156; ++j)
    x = mt_step_v(m, x, j);
#pragma unroll
  for (int i = 0; i < K; ++i) {
    x = mt_step_v(m, x, 156 + i);
    out[i] = mt_temper(mt_twist(w[i], w[i + 1], x));
  }
  lo = w[K];
  hi = mt_step_v(m, x, 156 + K);
}

// (The two out-of-line tier-2 routines take the slab and the seed BY VALUE: handed a reference to the engine, the
//  whole struct had to live in scratch memory — every draw of the streaming tier stored its counter there and the
//  reflection loop waited on the vector-memory counter once per iteration.)
typedef __attribute__((address_space(1))) u64 GlobalU64; // (the tier-2 state is global memory: not through flat instructions)
__device__ __noinline__ void rng_tier2_build(GlobalU64 *s, unsigned seed) {
  u64 x = seed;
  s[0] = x;
  for (int j = 1; j < 312; ++j) {
    x = mt_step(x, j);
    s[j * 64] = x;
  }
}

__device__ __noinline__ void rng_tier2_twist(GlobalU64 *s) {
  u64 cur = s[0];
  for (int i = 0; i < 156; ++i) {
    u64 nxt = s[(i + 1) * 64];
    s[i * 64] = mt_twist(cur, nxt, s[(i + 156) * 64]);
    cur = nxt;
  }
  for (int i = 156; i < 311; ++i) {
    u64 nxt = s[(i + 1) * 64];
    s[i * 64] = mt_twist(cur, nxt, s[(i - 156) * 64]);
    cur = nxt;
  }
  s[311 * 64] = mt_twist(cur, s[0], s[155 * 64]);
}

__device__ __forceinline__ u64 rng_next(Rng &r, unsigned &tier2Count) {
  if (r.k < 156u) { // (tier 2 is only ever built at k >= 156)
    const u64 lo1 = mt_step(r.lo, r.k + 1u);
    const u64 v = mt_temper(mt_twist(r.lo, lo1, r.hi));
    r.lo = lo1;
    r.hi = mt_step(r.hi, r.k + 157u); // (meaningless after k = 155; never read again)
    ++r.k;
    return v;
  }
  if (r.pos == 0xFFFFFFFFu) {
    rng_tier2_build((GlobalU64 *)r.scratch, r.seed);
    rng_tier2_twist((GlobalU64 *)r.scratch);
    unsigned skip = r.k; // outputs already consumed by the streaming tier
    while (skip >= 312u) {
      rng_tier2_twist((GlobalU64 *)r.scratch);
      skip -= 312u;
    }
    r.pos = skip;
    ++tier2Count;
  }
  if (r.pos >= 312u) {
    rng_tier2_twist((GlobalU64 *)r.scratch);
    r.pos = 0;
  }
  u64 v = mt_temper(((GlobalU64 *)r.scratch)[r.pos * 64]);
  ++r.pos;
  ++r.k;
  return v;
}

// libstdc++ std::uniform_real_distribution<float> on a 64-bit engine:
// generate_canonical = float(u64) / 2^64, clamped below 1 (bits/random.tcc)
__device__ __forceinline__ float canon_f32(u64 v) {
  float f = (float)v * 5.42101086242752217e-20f; // 2^-64, exact scaling
  return f >= 1.0f ? 0.99999994f : f;
}
__device__ __forceinline__ double canon_f64(u64 v) {
  double f = (double)v * 5.42101086242752217e-20; // 2^-64
  return f >= 1.0 ? 0.99999999999999989 : f;
}

// ---------------------------------------------------------------------------
// Primitive tests — Embree 4.3.3 semantics restated
// ---------------------------------------------------------------------------
// oriented disc: plane hit, tnear <= t, dist^2 < r^2
__device__ __forceinline__ bool hit_disc(const V3 &o, const V3 &d, float tnear, const float4 &c4, const V3 &n,
                                         float &tOut) {
  const float divisor = edot(d, n);
  if (divisor == 0.f)
    return false;
  const V3 co = V3{c4.x - o.x, c4.y - o.y, c4.z - o.z};
  const float t = edot(co, n) / divisor;
  if (!(tnear <= t && t <= 3.402823466e+38f))
    return false;
  const V3 p = V3{o.x + d.x * t - c4.x, o.y + d.y * t - c4.y, o.z + d.z * t - c4.z};
  const float dist2 = edot(p, p);
  if (!(dist2 < c4.w * c4.w))
    return false;
  tOut = t;
  return true;
}

__device__ __forceinline__ float xorsign(float v, float s) {
  return __uint_as_float(__float_as_uint(v) ^ (__float_as_uint(s) & 0x80000000u));
}

// Moeller-Trumbore, Embree form: den != 0, U,V >= 0, U+V <= |den|, |den| tnear < T
__device__ __forceinline__ bool hit_tri(const V3 &o, const V3 &d, float tnear, const V3 &v0, const V3 &e1,
                                        const V3 &e2, const V3 &Ng, float &tOut) {
  const V3 C = vsub(v0, o);
  const V3 R = ecross(C, d);
  const float den = edot(Ng, d);
  const float absDen = fabsf(den);
  const float U = xorsign(edot(R, e2), den);
  const float V = xorsign(edot(R, e1), den);
  if (!(den != 0.f && U >= 0.f && V >= 0.f && U + V <= absDen))
    return false;
  const float T = xorsign(edot(Ng, C), den);
  if (!(absDen * tnear < T && T <= absDen * 3.402823466e+38f))
    return false;
  tOut = T / absDen;
  return true;
}

struct HitRec {
  float t;
  int geom;       // -1 miss, 0 boundary, 1 geometry
  unsigned prim;  // wall id, or ORIGINAL primitive id
  unsigned pos;   // leaf position of the geometry primitive
};

// Exact pre-test for an axis-aligned wall at coordinate W on axis a: the
// Moeller-Trumbore depth test can only pass when (W - o_a) and d_a have the
// same strict sign (the other two components of the wall's Ng are exactly 0),
// so walls failing it are skipped without changing any result.
__device__ __forceinline__ bool wall_reachable(float W, float oa, float da) {
  const float c = W - oa;
  return (c > 0.f && da > 0.f) || (c < 0.f && da < 0.f);
}

__device__ __forceinline__ void hit_clear(HitRec &h) {
  h.t = 3.402823466e+38f;
  h.geom = -1;
  h.prim = 0xFFFFFFFFu;
  h.pos = 0;
}

// boundary: 8 wall triangles {v0,e1,e2,Ng} in LDS; pairs (0,1) (2,3) lie on the
// firstDir min/max planes, (4,5) (6,7) on the secondDir min/max planes.
// Called AFTER the geometry walk with the geometry's closest hit in `h`: most rays meet
// the surface long before they could reach a wall plane, and for those the (long)
// triangle tests are skipped.  A wall wins against geometry at equal t (closest-hit rule:
// boundary first); among walls the lower id wins.
// The launch's scalar frame, staged behind the walls in the kernels' LDS table (vr_api.cpp fills it at every prepare).
// Loop-invariant kernel arguments are hoisted and held in SGPRs throughout; the spilled ones come back by v_readlane
// at every use.  The compact ray records' decode (vr_trace.hip) reads its four scalars from here in every kernel; the
// absorbing flat-scene kernel (8 waves per SIMD, 78 spilled SGPRs) also its wall and scene-box frame: the *_lds
// variants below (C2 trace 6.5 -> 6.3 ms; the general kernels measured 3 % SLOWER with them and keep the arguments).
constexpr int VR_WALL_TABLE = 144; // floats
enum { VR_F_SRC_PLANE = 96, VR_F_RAYDIR = 97, VR_F_FIRSTDIR = 98, VR_F_SECONDDIR = 99, VR_F_EXTRA_LO = 100, VR_F_EXTRA_HI = 101,
       VR_F_LO1 = 102 /* lo1, hi1, lo2, hi2 */, VR_F_WALL_LO_R = 106, VR_F_WALL_HI_R = 107, VR_F_SCENE_LO = 108, VR_F_SCENE_HI = 111,
       VR_F_PQ_PAD = 114, VR_F_BC0 = 115, VR_F_BC1 = 116, VR_F_NB_DIST = 117,
       // the height field over the source plane (HeightFieldParams; NX = 0: none)
       VR_F_HF_LO1 = 118, VR_F_HF_LO2 = 119, VR_F_HF_INVT = 120, VR_F_HF_TILE = 121, VR_F_HF_TOP = 122, VR_F_HF_SIGN = 123,
       VR_F_HF_NX = 124, VR_F_HF_NY = 125, VR_F_HF_PTR_LO = 126, VR_F_HF_PTR_HI = 127,
       // the relief field's fine tiles (ReliefParams; NX = 0: none): relief_clip below
       VR_F_RF_LO1 = 128, VR_F_RF_LO2 = 129, VR_F_RF_INVT = 130, VR_F_RF_TILE = 131, VR_F_RF_NX = 132, VR_F_RF_NY = 133,
       VR_F_RF_PTR_LO = 134, VR_F_RF_PTR_HI = 135 };

__device__ __forceinline__ void hit_walls(const TraceParams &p, const float *__restrict__ wallS, const V3 &o,
                                          const V3 &d, float tnear, HitRec &h) {
  if (p.debugFlags & 8u)
    return;
  HitRec hw;
  hit_clear(hw);
  const float tLimit = h.t * 1.001f; // (plane t below is approximate: generous margin; inf stays inf)
  const float o1 = getc(o, p.firstDir), d1 = getc(d, p.firstDir);
  const float o2 = getc(o, p.secondDir), d2 = getc(d, p.secondDir);
#pragma unroll
  for (int pair = 0; pair < 4; ++pair) {
    const float *w0 = wallS + 24 * pair;
    // the wall planes are the adjusted bbox faces: scalars, no LDS read needed to cull
    const float W = pair == 0 ? p.lo1 : (pair == 1 ? p.hi1 : (pair == 2 ? p.lo2 : p.hi2));
    const float oa = pair < 2 ? o1 : o2, da = pair < 2 ? d1 : d2;
    if (!wall_reachable(W, oa, da))
      continue;
    {
      // conservative pre-tests (the approximate reciprocal is fine, the margins are far
      // above its rounding): the plane must come before the geometry hit, and where the
      // ray meets it, it must lie inside the wall's extent along the tracing axis (the
      // walls span the whole adjusted bbox there)
      const float tw = (W - oa) * __builtin_amdgcn_rcpf(da);
      if (tw > tLimit)
        continue;
      const float cr = getc(o, p.rayDir) + getc(d, p.rayDir) * tw;
      if (cr < p.wallLoR || cr > p.wallHiR)
        continue;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float *w = w0 + 12 * j;
      float t;
      if (hit_tri(o, d, tnear, mk(w[0], w[1], w[2]), mk(w[3], w[4], w[5]), mk(w[6], w[7], w[8]),
                  mk(w[9], w[10], w[11]), t)) {
        if (t < hw.t) { // ascending wall id: ties keep the lower id
          hw.t = t;
          hw.geom = 0;
          hw.prim = (unsigned)(2 * pair + j);
        }
      }
    }
  }
  if (hw.geom == 0 && hw.t <= h.t)
    h = hw;
}

// ... the same with the frame read from LDS (see above)
__device__ __forceinline__ void hit_walls_lds(const TraceParams &p, const float *__restrict__ wallS, const V3 &o,
                                          const V3 &d, float tnear, HitRec &h) {
  if (p.debugFlags & 8u)
    return;
  HitRec hw;
  hit_clear(hw);
  const float tLimit = h.t * 1.001f; // (plane t below is approximate: generous margin; inf stays inf)
  const int aRay = __float_as_int(wallS[VR_F_RAYDIR]), aFirst = __float_as_int(wallS[VR_F_FIRSTDIR]), aSecond = __float_as_int(wallS[VR_F_SECONDDIR]);
  const float o1 = getc(o, aFirst), d1 = getc(d, aFirst);
  const float o2 = getc(o, aSecond), d2 = getc(d, aSecond);
#pragma unroll
  for (int pair = 0; pair < 4; ++pair) {
    const float *w0 = wallS + 24 * pair;
    // the wall planes are the adjusted bbox faces: scalars, no LDS read needed to cull
    const float W = wallS[VR_F_LO1 + pair]; // lo1, hi1, lo2, hi2
    const float oa = pair < 2 ? o1 : o2, da = pair < 2 ? d1 : d2;
    if (!wall_reachable(W, oa, da))
      continue;
    {
      // conservative pre-tests (the approximate reciprocal is fine, the margins are far
      // above its rounding): the plane must come before the geometry hit, and where the
      // ray meets it, it must lie inside the wall's extent along the tracing axis (the
      // walls span the whole adjusted bbox there)
      const float tw = (W - oa) * __builtin_amdgcn_rcpf(da);
      if (tw > tLimit)
        continue;
      const float cr = getc(o, aRay) + getc(d, aRay) * tw;
      if (cr < wallS[VR_F_WALL_LO_R] || cr > wallS[VR_F_WALL_HI_R])
        continue;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float *w = w0 + 12 * j;
      float t;
      if (hit_tri(o, d, tnear, mk(w[0], w[1], w[2]), mk(w[3], w[4], w[5]), mk(w[6], w[7], w[8]),
                  mk(w[9], w[10], w[11]), t)) {
        if (t < hw.t) { // ascending wall id: ties keep the lower id
          hw.t = t;
          hw.geom = 0;
          hw.prim = (unsigned)(2 * pair + j);
        }
      }
    }
  }
  if (hw.geom == 0 && hw.t <= h.t)
    h = hw;
}

// ---------------------------------------------------------------------------
// A ray's stretch through the LOCAL relief (ReliefParams, vr_types.hpp) instead of through the scene box.  The tiles of
// the relief field under the ray are walked from t0 to t1 (its stretch inside the scene box); inside a tile the ray can
// only meet a primitive while its height lies within the tile's [lo, hi] — every primitive that reaches into the tile
// lies in that range — so [tA, tB], the hull of those sub-stretches over all tiles walked, holds every point at which
// the ray meets ANY primitive (closest hit and neighbour crossings alike).  tA > tB: the ray meets nothing.
// Rounding: the walk may file a sliver of the ray under the neighbouring tile; the field's pad (1e-5 of the largest
// coordinate, three orders above the rounding of a position) makes that tile's range hold the primitive as well.
// After VR_RELIEF_STEPS tiles (a grazing ray that the generator did not file apart) the rest of the stretch is kept whole.
// ---------------------------------------------------------------------------
// Returns true for a lane whose walk was cut short after STEPS tiles (the rest of its stretch is then kept whole).
template <int STEPS = VR_RELIEF_STEPS>
__device__ __forceinline__ bool relief_clip(const float *__restrict__ wallS, bool on, const V3 &o, const V3 &d, float t0, float t1,
                                            float &tA, float &tB) {
  typedef float F2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(1))) F2 *GlobalF2;
  const GlobalF2 field = reinterpret_cast<GlobalF2>(((unsigned long long)__float_as_uint(wallS[VR_F_RF_PTR_HI]) << 32) |
                                                    __float_as_uint(wallS[VR_F_RF_PTR_LO]));
  const int ax = __float_as_int(wallS[VR_F_RAYDIR]), a1 = __float_as_int(wallS[VR_F_FIRSTDIR]), a2 = __float_as_int(wallS[VR_F_SECONDDIR]);
  const int nx = __float_as_int(wallS[VR_F_RF_NX]), ny = __float_as_int(wallS[VR_F_RF_NY]);
  const float T = wallS[VR_F_RF_TILE], invT = wallS[VR_F_RF_INVT], lo1 = wallS[VR_F_RF_LO1], lo2 = wallS[VR_F_RF_LO2];
  const float oz = getc(o, ax), dz = getc(d, ax), o1 = getc(o, a1), d1 = getc(d, a1);
  const float o2 = ny > 1 ? getc(o, a2) : 0.f, d2 = ny > 1 ? getc(d, a2) : 0.f;
  const float big = 3.0e38f;
  tA = big;
  tB = -big;
  int ix = (int)floorf((o1 + d1 * t0 - lo1) * invT), iy = ny > 1 ? (int)floorf((o2 + d2 * t0 - lo2) * invT) : 0;
  ix = ix < 0 ? 0 : (ix >= nx ? nx - 1 : ix);
  iy = iy < 0 ? 0 : (iy >= ny ? ny - 1 : iy);
  const int sx = d1 > 0.f ? 1 : -1, sy = d2 > 0.f ? 1 : -1;
  const float inv1 = d1 != 0.f ? 1.0f / d1 : 0.f, inv2 = d2 != 0.f ? 1.0f / d2 : 0.f;
  float tx = d1 != 0.f ? (lo1 + (float)(ix + (d1 > 0.f ? 1 : 0)) * T - o1) * inv1 : big;
  float ty = d2 != 0.f ? (lo2 + (float)(iy + (d2 > 0.f ? 1 : 0)) * T - o2) * inv2 : big;
  const float dtx = T * fabsf(inv1), dty = T * fabsf(inv2);
  const float invz = dz != 0.f ? 1.0f / dz : 0.f;
  // (Skipping the air above the relief first — down to the height field's 3 x 3-tile maximum, one look-up — was built and
  //  measured in round 4: +- 0 on a plane with a bump and on the rippled sheet, the walk being at most six tiles anyway;
  //  started BELOW a tile's top it also made the query's box too low for the follow-up segments.  Removed.)
  float tc = t0;
  bool go = on && t0 <= t1, cut = false;
  for (int s = 0; ballot64(go); ++s) {
    if (go) {
      const float tn = fminf(fminf(tx, ty), t1);
      // (staging the wave's window of tiles in LDS first — one load per lane, then LDS reads in the walk — was built and
      //  measured in round 4: +- 0, the tiles are L1 hits; the lists it used now hold the frontier cache)
      const F2 f = field[iy * nx + ix];
      const float z0 = oz + dz * tc, z1 = oz + dz * tn;
      if (fmaxf(z0, z1) >= f.x && fminf(z0, z1) <= f.y) {
        float ta = tc, tb = tn;
        if (dz != 0.f) {
          const float q0 = (f.x - oz) * invz, q1 = (f.y - oz) * invz;
          ta = fmaxf(ta, fminf(q0, q1));
          tb = fminf(tb, fmaxf(q0, q1));
        }
        tA = fminf(tA, ta);
        tB = fmaxf(tB, tb);
      }
      if (!(tn < t1)) {
        go = false;
      } else if (s == STEPS - 1) { // (too many tiles: the rest of the stretch as it is)
        tA = fminf(tA, tn);
        tB = t1;
        go = false;
        cut = true;
      } else {
        if (tx <= ty) {
          ix += sx;
          tx += dtx;
        } else {
          iy += sy;
          ty += dty;
        }
        go = (unsigned)ix < (unsigned)nx && (unsigned)iy < (unsigned)ny; // (beyond the field: beyond the scene box)
        tc = tn;
      }
    }
  }
  return cut;
}

__device__ __forceinline__ V3 safe_inverse(const V3 &d) {
  const float dx = fabsf(d.x) < 1e-30f ? copysignf(1e-30f, d.x) : d.x;
  const float dy = fabsf(d.y) < 1e-30f ? copysignf(1e-30f, d.y) : d.y;
  const float dz = fabsf(d.z) < 1e-30f ? copysignf(1e-30f, d.z) : d.z;
  return V3{1.0f / dx, 1.0f / dy, 1.0f / dz};
}

// closest-hit rule: min t; ties -> boundary first, then lower original id
__device__ __forceinline__ bool hit_update(HitRec &h, bool ok, float t, unsigned orig, unsigned q) {
  const bool take = ok && (t < h.t || (t == h.t && h.geom == 1 && orig < h.prim));
  if (take) {
    h.t = t;
    h.geom = 1;
    h.prim = orig;
    h.pos = q;
  }
  return take;
}

// Conservative pre-test of a disc from its first record word {c, r} alone: a ray can only pass
// hit_disc / local_disc_hit if its LINE comes within r of the centre and the disc is not entirely
// behind the origin.  The per-lane paths are bound by the vector memory pipeline (one address per
// clock per CU for scattered gathers), so a disc's second word (normal, id) is only
// fetched by the lanes that pass.  Margins cover the rounding of the cancellation in q (a few ulp
// of |c - o|^2); a pass decides nothing, the exact test follows.
__device__ __forceinline__ bool disc_may_hit(const V3 &o, const V3 &d, float invDD, const float4 &c4) {
  const V3 oc = V3{c4.x - o.x, c4.y - o.y, c4.z - o.z};
  const float oc2 = vdot(oc, oc), b = vdot(oc, d), r2 = c4.w * c4.w;
  const float q = oc2 - b * b * invDD; // squared distance of the centre from the line
  return q <= r2 * 1.001f + 2e-6f * oc2 && !(b < 0.f && oc2 > r2 * 1.01f + 1e-12f);
}

// geometry, per-lane, escape links: every lane walks its own path over the pre-order nodes.  Kept for the
// absorbing flat-scene kernel (MODE 1), whose rounds are packets and which wants the single register of walk
// state, as the reference walk of the -DVR_SELFCHECK build and of vr_debug_intersect (VR_DEBUG_WALK=0); the
// kernels that walk a lot use pair_walk_lanes below.  Divergent lanes
// make every node fetch 64 separate requests, so this walk reads the 16-byte nodes:
// one dwordx4 per visit.  The slab test runs in the quantised frame (ray transformed
// once per call; per-axis scaling leaves t unchanged); boxes were rounded outwards
// by more than the rounding of this test, and a box only ever culls.
// `node` is the lane's cursor (resumable): the loop runs while at least `minLanes`
// lanes of the wave are still walking, then returns with the stragglers' cursors and
// closest hits intact.
template <int GEO>
__device__ __forceinline__ void bvh_walk_lanes(const TraceParams &p, bool part, const V3 &o, const V3 &d, float tnear,
                                               HitRec &h, unsigned &node, unsigned minLanes VR_DIAG_ARGS) {
  const uint4 *__restrict__ qnodes = reinterpret_cast<const uint4 *>(p.qnodes);
  const float4 *__restrict__ prims = reinterpret_cast<const float4 *>(p.prims);
  const V3 inv = safe_inverse(V3{d.x * p.qscale[0], d.y * p.qscale[1], d.z * p.qscale[2]});
  const V3 oi = V3{(o.x - p.qbase[0]) * p.qscale[0] * inv.x, (o.y - p.qbase[1]) * p.qscale[1] * inv.y,
                   (o.z - p.qbase[2]) * p.qscale[2] * inv.z};
  if (!part)
    node = VR_END;
  const float invDD = 1.0f / vdot(d, d);
  // Two alternating phases ("while-while"): a lane SEARCHES for leaves whose box it hits.
  // The first such leaf is only remembered (`pend`) and the lane searches on
  // (speculatively: it may visit nodes the pending leaf's hit would have culled); at a
  // second leaf it parks on that node.  When a given share of the lanes under way are
  // parked, or nobody searches any more, the pending leaves' primitives are tested
  // together.  Testing a leaf the moment one lane reaches it would run the (long)
  // primitive test with a handful of lanes on almost every step.
  unsigned pend = 0u; // pending leaf link (VR_LEAF bit set) or 0
  // The loop bodies are PREDICATED, not branched: a lane that is not searching (parked, done, or
  // never took part) runs the step on node 0 and discards it.  An exec-masked step costs the same
  // issue slots as a full one, and the mask juggling of a divergent `if` was a third of the
  // instructions of this loop; two steps run between the wave-level votes for the same reason.
  for (;;) {
    bool parked = false;
    for (;;) {
      const bool search0 = node < p.numNodes && !parked;
      const unsigned long long sm = ballot64(search0);
      if (!sm)
        break;
      const unsigned long long km = ballot64(parked);
      if (100u * (unsigned)__popcll(km) >= p.walkPark * (unsigned)__popcll(km | sm))
        break;
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const bool search = node < p.numNodes && !parked;
#ifdef VR_DIAG
        if (search) {
          DIAG(1);
        }
#endif
        const uint4 nd = qnodes[search ? node : 0u];
        const float lx = (float)(nd.x & 0xFFFFu), ly = (float)(nd.x >> 16), lz = (float)(nd.y & 0xFFFFu);
        const float hx = (float)(nd.y >> 16), hy = (float)(nd.z & 0xFFFFu), hz = (float)(nd.z >> 16);
        const float tx0 = __builtin_fmaf(lx, inv.x, -oi.x), tx1 = __builtin_fmaf(hx, inv.x, -oi.x);
        const float ty0 = __builtin_fmaf(ly, inv.y, -oi.y), ty1 = __builtin_fmaf(hy, inv.y, -oi.y);
        const float tz0 = __builtin_fmaf(lz, inv.z, -oi.z), tz1 = __builtin_fmaf(hz, inv.z, -oi.z);
        const float tEntry = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tnear));
        const float tExit = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
        const unsigned link = nd.w;
        const bool leaf = (link & VR_LEAF) != 0u;
        const bool hitBox = tEntry <= tExit && tEntry <= h.t;
        const bool take = search && hitBox && leaf;
        const bool park = take && pend != 0u; // second leaf: wait here (the node is visited again afterwards)
        pend = (take && pend == 0u) ? link : pend;
        parked = parked || park;
        // pre-order: first child of an internal node / escape of a leaf = next node
        const unsigned nxt = (hitBox || leaf) ? node + 1u : link;
        node = (search && !park) ? nxt : node;
      }
    }
    if (ballot64(pend != 0u)) {
      const unsigned first = pend & VR_LEAF_FIRST_MASK;
      const unsigned cnt = pend ? (pend >> 27) & 15u : 0u;
      for (unsigned i = 0; ballot64(i < cnt); ++i) {
        const bool on = i < cnt;
#ifdef VR_DIAG
        if (on) {
          DIAG(2);
        }
#endif
        const unsigned q = on ? first + i : 0u;
        float t;
        if (GEO == 0) {
          const float4 c4 = prims[2 * q];
          if (on && disc_may_hit(o, d, invDD, c4)) { // (exec-masked: lanes that fail fetch nothing more)
            const float4 n4 = prims[2 * q + 1];
            const bool ok = hit_disc(o, d, tnear, c4, mk(n4.x, n4.y, n4.z), t);
            hit_update(h, ok, t, __float_as_uint(n4.w), q);
          }
        } else {
          const float4 a = prims[4 * q], b = prims[4 * q + 1], c = prims[4 * q + 2], e = prims[4 * q + 3];
          const bool ok =
              hit_tri(o, d, tnear, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), mk(e.x, e.y, e.z), t);
          hit_update(h, on && ok, t, __float_as_uint(a.w), q);
        }
      }
      pend = 0u;
    }
    if ((unsigned)__popcll(ballot64(node < p.numNodes)) < minLanes)
      break;
  }
}

// ---------------------------------------------------------------------------
// geometry, per-lane, ORDERED: the walk of bounced rays.  bvh_walk_lanes follows the escape links of a
// pre-order layout: one node (one 16-byte gather, one dependent cache access) per box test, and the
// order of the children is fixed at build time (source side first) — right for primary rays, wrong for
// half of the bounced ones, which then cannot be culled by a close hit.  Here a visit reads a PAIR node
// (both children of an internal node, 32 bytes of one line), tests both boxes, descends into the NEARER
// child and defers the other on a per-lane stack whose first SD entries live in LDS ([entry][lane]: conflict
// free) and the (rare) rest in a per-wave global slab.  Half the dependent accesses per box test, and the
// near-first order finds the closest hit early, so far subtrees fail `tEntry <= h.t` when they are popped.
// The closest-hit rule makes the result independent of the order: bit-identical to bvh_walk_lanes.
//   `node`: cursor — 0 = fresh (root pair), VR_END = finished, else a pair index or a leaf word.
//   `sp`:   stack depth of the lane (the caller keeps it with `node` between rounds).
// ---------------------------------------------------------------------------
template <int SD>
__device__ __forceinline__ void walk_push(unsigned *stackS, unsigned *stackG, unsigned long long *errFlag, unsigned &sp,
                                          unsigned v) {
  if (sp < (unsigned)SD)
    stackS[sp * VR_BLOCK] = v;
  else if (sp < (unsigned)SD + VR_STACK_GLOBAL)
    stackG[(sp - (unsigned)SD) * 64u] = v;
  else
    *errFlag = 1ull; // deeper than any tree the builder emits: reported by vr_apply_finish, never silent
  ++sp;
}
template <int SD> __device__ __forceinline__ unsigned walk_pop(const unsigned *stackS, const unsigned *stackG, unsigned &sp) {
  --sp;
  if (sp < (unsigned)SD)
    return stackS[sp * VR_BLOCK];
  return sp < (unsigned)SD + VR_STACK_GLOBAL ? stackG[(sp - (unsigned)SD) * 64u] : VR_END;
}

// stackS: this lane's column of the LDS stack (entry e at stackS[e * VR_BLOCK]); stackG: this lane's column of
// the wave's global slab (entry e at stackG[e * 64])
// pnodes / prims: the pair nodes and primitive records — global memory, or the LDS copies of MODE 4
// the pending leaves of a wave: every lane tests the primitives of ITS leaf word `pend` (0: none)
template <int GEO, bool LEAF2>
__device__ __forceinline__ void walk_leaf_tests(const float4 *__restrict__ prims, const V3 &o, const V3 &d, float invDD,
                                                float tnear, HitRec &h, unsigned &pend VR_DIAG_ARGS) {
    if (ballot64(pend != 0u)) {
      const unsigned first = pend & VR_LEAF_FIRST_MASK;
      const unsigned cnt = pend ? (pend >> 27) & 15u : 0u;
      for (unsigned i = 0; ballot64(i < cnt); i += ((GEO == 0 && LEAF2) ? 2u : 1u)) {
        const bool on = i < cnt;
#ifdef VR_DIAG
        if (on) {
          DIAG(2);
        }
#endif
        const unsigned q = on ? first + i : 0u;
        float t;
        if (GEO == 0) {
          // two disks per pass, all four record words requested together
          const bool on2 = LEAF2 && i + 1u < cnt;
          const unsigned q2 = on2 ? q + 1u : q;
          const float4 c4 = prims[2 * q];
          const float4 n4 = prims[2 * q + 1];
          const float4 c5 = prims[2 * q2];
          const float4 n5 = prims[2 * q2 + 1];
          asm volatile("" ::"v"(c4.x), "v"(n4.x), "v"(c5.x), "v"(n5.x));
          if (on && disc_may_hit(o, d, invDD, c4)) {
            const bool ok = hit_disc(o, d, tnear, c4, mk(n4.x, n4.y, n4.z), t);
            hit_update(h, ok, t, __float_as_uint(n4.w), q);
          }
          if (on2 && disc_may_hit(o, d, invDD, c5)) {
            const bool ok = hit_disc(o, d, tnear, c5, mk(n5.x, n5.y, n5.z), t);
            hit_update(h, ok, t, __float_as_uint(n5.w), q2);
          }
        } else {
          const float4 a = prims[4 * q], b = prims[4 * q + 1], c = prims[4 * q + 2], e = prims[4 * q + 3];
          const bool ok =
              hit_tri(o, d, tnear, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), mk(e.x, e.y, e.z), t);
          hit_update(h, on && ok, t, __float_as_uint(a.w), q);
        }
      }
      pend = 0u;
    }
}

// LEAF2: the leaf test takes two disks per pass, all four record words in flight together (8 more live registers:
// not for the 72-VGPR absorbing kernel)
template <int GEO, int SD, bool LEAF2 = true>
__device__ __forceinline__ void pair_walk_lanes(const TraceParams &p, const uint4 *__restrict__ pnodes,
                                                const float4 *__restrict__ prims, unsigned *stackS, unsigned *stackG,
                                                bool part, const V3 &o, const V3 &d, float tnear, HitRec &h,
                                                unsigned &node, unsigned &sp, unsigned minLanes VR_DIAG_ARGS) {
  const V3 inv = safe_inverse(V3{d.x * p.qscale[0], d.y * p.qscale[1], d.z * p.qscale[2]});
  const V3 oi = V3{(o.x - p.qbase[0]) * p.qscale[0] * inv.x, (o.y - p.qbase[1]) * p.qscale[1] * inv.y,
                   (o.z - p.qbase[2]) * p.qscale[2] * inv.z};
  if (!part)
    node = VR_END;
  sp = node == 0u ? 0u : sp;
  const float invDD = 1.0f / vdot(d, d);
  unsigned long long *const errFlag = p.counters + 60;
  unsigned pend = 0u; // pending leaf word or 0
  for (;;) {
    bool parked = false; // at a second leaf while the first is still pending
    for (;;) {
      const unsigned long long sm = ballot64(node != VR_END && !parked);
      if (!sm)
        break;
      const unsigned long long km = ballot64(parked);
      if (100u * (unsigned)__popcll(km) >= p.walkPark * (unsigned)__popcll(km | sm))
        break;
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const bool live = node != VR_END && !parked;
        const bool atLeaf = live && (node & VR_LEAF) != 0u; // (a deferred child that is a leaf, popped)
        const bool visit = live && !atLeaf;
#ifdef VR_DIAG
        if (visit) {
          DIAG(1);
        }
        if (node != VR_END) { // (lanes whose walk is not finished: searching, at a leaf, or parked)
          DIAG(7);
        }
#endif
        const size_t idx = visit ? (size_t)node : 0u;
        const uint4 a = pnodes[2 * idx], b = pnodes[2 * idx + 1];
        bool hit0, hit1;
        float e0, e1;
        {
          const float lx = (float)(a.x & 0xFFFFu), ly = (float)(a.x >> 16), lz = (float)(a.y & 0xFFFFu);
          const float hx = (float)(a.y >> 16), hy = (float)(a.z & 0xFFFFu), hz = (float)(a.z >> 16);
          const float tx0 = __builtin_fmaf(lx, inv.x, -oi.x), tx1 = __builtin_fmaf(hx, inv.x, -oi.x);
          const float ty0 = __builtin_fmaf(ly, inv.y, -oi.y), ty1 = __builtin_fmaf(hy, inv.y, -oi.y);
          const float tz0 = __builtin_fmaf(lz, inv.z, -oi.z), tz1 = __builtin_fmaf(hz, inv.z, -oi.z);
          e0 = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tnear));
          const float x0 = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
          hit0 = e0 <= x0 && e0 <= h.t;
        }
        {
          const float lx = (float)(b.x & 0xFFFFu), ly = (float)(b.x >> 16), lz = (float)(b.y & 0xFFFFu);
          const float hx = (float)(b.y >> 16), hy = (float)(b.z & 0xFFFFu), hz = (float)(b.z >> 16);
          const float tx0 = __builtin_fmaf(lx, inv.x, -oi.x), tx1 = __builtin_fmaf(hx, inv.x, -oi.x);
          const float ty0 = __builtin_fmaf(ly, inv.y, -oi.y), ty1 = __builtin_fmaf(hy, inv.y, -oi.y);
          const float tz0 = __builtin_fmaf(lz, inv.z, -oi.z), tz1 = __builtin_fmaf(hz, inv.z, -oi.z);
          e1 = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tnear));
          const float x1 = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
          hit1 = e1 <= x1 && e1 <= h.t;
        }
        const bool both = hit0 && hit1, any = hit0 || hit1;
#ifdef VR_DIAG
        if (visit && !any) {
          DIAG(14);
        }
        if (visit && both) {
          DIAG(15);
        }
#endif
        const bool second = hit1 && (!hit0 || e1 < e0); // child 1 is the nearer one
        const unsigned nearL = second ? b.w : a.w, farL = second ? a.w : b.w;
        // a leaf goes into the pending slot when that is free; with the slot taken the lane parks on it
        const bool nearLeaf = (nearL & VR_LEAF) != 0u;
        const bool takeNear = visit && any && nearLeaf && pend == 0u;
        const bool takeSelf = atLeaf && pend == 0u;
        const bool park = atLeaf && pend != 0u;
        pend = takeNear ? nearL : (takeSelf ? node : pend);
        // where next: the nearer child, or — nothing (more) to enter here — the far child / the stack
        const bool descend = visit && any && !takeNear;
        const bool toFar = takeNear && both;
        const bool pop = (visit && !any) || (takeNear && !both) || takeSelf;
        if (descend && both)
          walk_push<SD>(stackS, stackG, errFlag, sp, farL);
        unsigned nxt = descend ? nearL : (toFar ? farL : node);
        if (pop)
          nxt = sp ? walk_pop<SD>(stackS, stackG, sp) : VR_END;
        node = nxt;
        parked = parked || park;
      }
    }
    TICK(2);
    walk_leaf_tests<GEO, LEAF2>(prims, o, d, invDD, tnear, h, pend VR_DIAG_PASS);
    TICK(3);
    if ((unsigned)__popcll(ballot64(node != VR_END)) < minLanes)
      break;
  }
}

// ---------------------------------------------------------------------------
// geometry, wave-uniform ("packet"): the 64 rays of a wavefront that were sorted
// into the same far-plane cell walk the UNION of their paths in lock step.  The
// node index is a scalar, node and primitive records come through the scalar
// cache (s_load, constant address space) instead of 64 per-lane vector loads,
// and there is no per-lane exec masking inside the traversal.  A lane that
// misses a box still runs the subtree's primitive tests; those are exact, so the
// selected hit is unchanged (the box test only ever culls).
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef const vf4 __attribute__((address_space(4))) *ConstF4;

// `budget` bounds the number of node visits: a packet whose rays turn out to be
// incoherent (union of paths much larger than one path) gives up and returns false;
// the hits found so far are real hits and stay in `h`, the caller finishes with the
// per-lane traversal.  Besides the hard budget the walk keeps a running efficiency
// figure in scalar registers: `wants` = sum over visited nodes of the lanes whose own
// box test passed.  wants / lanes is the mean length of ONE ray's path; when the union
// walked so far exceeds `ratio` times that (plus a start-up allowance) the per-lane
// traversal is cheaper and the packet gives up (small scenes never reach the hard
// budget, so this is what protects them).
template <int GEO>
__device__ __forceinline__ bool bvh_hit_packet(const TraceParams &p, bool part, const V3 &o, const V3 &d, float tnear,
                                               HitRec &h, unsigned budget, unsigned ratio VR_DIAG_ARGS) {
  if (p.numPrims == 0)
    return true;
  const unsigned long long partMask = ballot64(part);
  const unsigned lanes = (unsigned)__popcll(partMask);
  unsigned wants = 16u * lanes; // start-up allowance: 16 visits
  unsigned visits = 0;
  ConstF4 nodes = (ConstF4)(p.nodes);
  ConstF4 prims = (ConstF4)(p.prims);
  const V3 inv = safe_inverse(d);
  const V3 oi = V3{o.x * inv.x, o.y * inv.y, o.z * inv.z};
  unsigned node = 0; // wave-uniform
  while (node != VR_END) {
    if (visits == budget || visits * lanes > ratio * wants)
      return false;
    ++visits;
    if (part) {
      DIAG(3);
    }
    const vf4 q0 = nodes[2 * node];
    const vf4 q1 = nodes[2 * node + 1];
    const float tx0 = __builtin_fmaf(q0.x, inv.x, -oi.x), tx1 = __builtin_fmaf(q1.x, inv.x, -oi.x);
    const float ty0 = __builtin_fmaf(q0.y, inv.y, -oi.y), ty1 = __builtin_fmaf(q1.y, inv.y, -oi.y);
    const float tz0 = __builtin_fmaf(q0.z, inv.z, -oi.z), tz1 = __builtin_fmaf(q1.z, inv.z, -oi.z);
    const float tEntry = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tnear));
    const float tExit = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
    const unsigned link = __builtin_amdgcn_readfirstlane(__float_as_uint(q0.w));
    const unsigned esc = __builtin_amdgcn_readfirstlane(__float_as_uint(q1.w));
    // (tEntry <= min(tExit, h.t), as two compares: a min with the loop-carried h.t costs a
    //  canonicalising v_max besides the v_min.
    //  Votes are taken per comparison and combined as scalar masks: a vote on a combined
    //  condition is lowered through a materialised 0/1 value, two more VALU per visit.)
    const unsigned long long want = partMask & ballot64(tEntry <= tExit) & ballot64(tEntry <= h.t);
    wants += (unsigned)__popcll(want);
    if (want) {
      if (link & VR_LEAF) {
        const unsigned first = link & VR_LEAF_FIRST_MASK;
        const unsigned cnt = (link >> 27) & 15u;
        for (unsigned i = 0; i < cnt; ++i) {
          if (part) {
            DIAG(4);
          }
          const unsigned q = first + i;
          float t;
          if (GEO == 0) {
            const vf4 c4v = prims[2 * q];
            const vf4 n4 = prims[2 * q + 1];
            const bool ok = hit_disc(o, d, tnear, make_float4(c4v.x, c4v.y, c4v.z, c4v.w), mk(n4.x, n4.y, n4.z), t);
            hit_update(h, part && ok, t, __float_as_uint(n4.w), q);
          } else {
            const vf4 a = prims[4 * q], b = prims[4 * q + 1], c = prims[4 * q + 2], e = prims[4 * q + 3];
            const bool ok =
                hit_tri(o, d, tnear, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), mk(e.x, e.y, e.z), t);
            hit_update(h, part && ok, t, __float_as_uint(a.w), q);
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
  return true;
}

// ---------------------------------------------------------------------------
// geometry, wave-uniform by BOX ("packet query"): for rays that are close together where it
// matters.  Every participating ray is clipped to the scene's box; the box Q around all those
// clipped segments (six wave-wide min/max reductions) contains every point at which any of the
// rays can meet a primitive.  The 64-ary box tree (TraceParams::wide) is then searched for Q
// breadth first with the LANES AS CHILDREN: one coalesced load and one box-box test per
// visited node for 64 children at once, 3-4 levels for 10^6 primitives — instead of ~50
// scalar node visits with a 64-ray slab test each.  The primitives whose own box meets Q
// are tested exactly against every ray (records through the scalar cache, as in
// bvh_hit_packet); the closest-hit rule makes the result independent of how candidates were
// found.  Pays when Q is small: sorted primary rays on any scene whose relief is small against
// its extent, and ALL segments of a flat scene (a bounced ray leaves the thin scene box at
// once, whatever its direction).  Gives up (false, nothing touched) when a level's frontier or
// the candidate list outgrows its limit; the caller then walks as before.
// ---------------------------------------------------------------------------
// Six wave-wide reductions at once (three minima, three maxima), result in every lane:
// row_shr 1/2/4/8 + row_bcast 15/31 with the DPP modifier fused into v_min / v_max.  The six
// chains are interleaved, so the two wait states a DPP read of a just-written VGPR needs are
// always filled.  Lanes whose DPP source is out of range keep their value (no bound_ctrl).
__device__ __forceinline__ void wave_minmax6(float &a, float &b, float &c, float &d, float &e, float &f) {
#define VR_DPP6(ctrl)                                                                                                  \
  "v_min_f32_dpp %0, %0, %0 " ctrl "\n v_min_f32_dpp %1, %1, %1 " ctrl "\n v_min_f32_dpp %2, %2, %2 " ctrl "\n"        \
  "v_max_f32_dpp %3, %3, %3 " ctrl "\n v_max_f32_dpp %4, %4, %4 " ctrl "\n v_max_f32_dpp %5, %5, %5 " ctrl "\n"
  // (s_nop 4: the compiler does not know that the block begins with DPP reads — a VGPR written by the VALU instruction
  //  just before it needs two wait states before a DPP instruction may read it, a VALU write of EXEC (v_cmpx, v_readlane
  //  into exec) five, and nothing inserts them for inline assembly.  Found when a second call site, scheduled
  //  differently, returned minima of stale registers now and then; five wait states cover both hazards of the gfx9 table)
  asm volatile("s_nop 4\n" VR_DPP6("row_shr:1 row_mask:0xf bank_mask:0xf") VR_DPP6("row_shr:2 row_mask:0xf bank_mask:0xf")
                   VR_DPP6("row_shr:4 row_mask:0xf bank_mask:0xf") VR_DPP6("row_shr:8 row_mask:0xf bank_mask:0xf")
                       VR_DPP6("row_bcast:15 row_mask:0xa bank_mask:0xf") VR_DPP6("row_bcast:31 row_mask:0xc bank_mask:0xf")
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
#undef VR_DPP6
  a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
  b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
  c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
  d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63));
  e = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 63));
  f = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), 63));
}

__device__ __forceinline__ bool local_disc_hit(const V3 &ro, const V3 &rd, const float4 &c4, const V3 &n);

__device__ __forceinline__ float lane_bcast(float v, int srcLane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srcLane));
}

// What a completed packet query leaves behind for the crediting of disks (pq_credit, vr_trace.hip):
// candidate c's leaf position and centre are record c of a small per-wave LDS table (written by the
// lane that holds the candidate's record, read back by all lanes at once: one ds_write_b128 / one
// broadcast ds_read_b128 per candidate instead of four v_writelane / v_readlane sequences), and bit c of each lane's
// `local` says whether that lane's ray passes the neighbour test on candidate c
// (checkLocalIntersection, rayTraceKernel.hpp:462-507).  Every disk a ray can be credited to is
// among the candidates: the test only passes where the ray crosses the disk, the disk lies in the
// scene box, and Q covers every participating ray from its origin to where it leaves that box.
// Pointers into LDS carry their address space in the type where they are volatile or travel through a struct: address-
// space inference leaves such accesses on GENERIC pointers, i.e. flat_load / flat_store (the slow path to LDS, waiting on
// both memory counters) — the packet query's frontier lists were read and written that way until round 3.
#define VR_LDS __attribute__((address_space(3)))
#ifndef VR_PQ_KEEP
#define VR_PQ_KEEP 12 // leaf nodes a packet query's frontier cache keeps per wave (<= 16: the compacted list starts at lst[16])
#endif
typedef unsigned U4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ U4 mk_u4(unsigned x, unsigned y, unsigned z, unsigned w) {
  U4 v;
  v.x = x, v.y = y, v.z = z, v.w = w;
  return v;
}
struct PqCands {
  VR_LDS U4 *rec; // this wave's candidate records in LDS: {leaf position, centre.xyz as bits}, VR_PQ_CANDS entries
  unsigned long long local;
  unsigned count; // wave-uniform
  bool box;       // KEEPQ: records VR_PQ_BOX, + 1 hold the query's box (false: no ray reached the scene, nothing stored)
  unsigned mine;  // per lane: the candidate that is this lane's closest hit so far (valid where the hit came from the query)
};
constexpr unsigned VR_PQ_CANDS = 52; // >= 2 * pqMaxCand + 1 (pqMaxCand <= 24, vr_api.cpp) + the two records below
// KEEPQ: records 50 / 51 keep the query's (padded) box {lo.xyz, -}{hi.xyz, -} for the round's follow-up segments
// (trace_kernel, "follow-up segments"); PqCands::box says whether there is one (stored as an "empty box", two constant
// 16-byte tuples were hoisted out of the round loop, spilled, and reloaded from scratch in every round)
constexpr unsigned VR_PQ_BOX = 50;
// records VR_PQ_NRM + c: candidate c's {normal.xyz, radius} — with record c everything the state machine (normal of the
// closest disk), the crediting and the follow-up segments need of a candidate, read from LDS instead of from its
// primitive record in global memory
constexpr unsigned VR_PQ_NRM = 52;
constexpr unsigned VR_PQ_RECORDS = 2 * 52;

// lst: 128 dwords of LDS private to this wave
// FRAME_LDS: the scene box and the padding come from the LDS frame `wallS` (see hit_walls_lds)
// RELIEF: the rays are clipped to the local relief (relief_clip) instead of to the scene box
// CACHE (the flat-scene kernels): consecutive rounds of a wave are neighbours in space — its bins follow a boustrophedon
// path — and the descent of the 64-ary tree is latency: three dependent node loads before the first primitive record
// (27 % of the flat kernel's wave time, -DVR_DIAG sub-phase timers).  A query therefore searches with its box ENLARGED by
// pqMargin and leaves its last-level frontier (the leaf nodes meeting the enlarged box S: complete for every box inside
// S, at most 12) with S in the wave's LDS lists (lst[0..], lst[64..]: the entries; lst[32..37]: S; lst[38]: their number;
// lst[39]: valid; cboxes: the leaf nodes' own boxes); a later round whose box lies inside S filters the kept nodes by ITS
// box — from LDS, no node load — and goes straight to the primitive records of those that meet it: the very nodes an
// exact descent would have found.  The candidates are filtered with the round's own box as before: bit-identical.
template <int GEO, bool CREDIT, bool FRAME_LDS = false, bool KEEPQ = false, bool RELIEF = false, bool CACHE = false>
__device__ __forceinline__ bool pq_hit_packet(const TraceParams &p, bool part, const V3 &o, const V3 &d, float tnear,
                                              HitRec &h, volatile VR_LDS unsigned *lst, PqCands &cd,
                                              const float *__restrict__ wallS, volatile VR_LDS float *cboxes, float tWall VR_DIAG_ARGS) {
  const unsigned lane = threadIdx.x & 63u;
  // the ray's stretch inside the scene box
  const V3 inv = safe_inverse(d);
  const float tx0 = ((FRAME_LDS ? wallS[VR_F_SCENE_LO] : p.sceneLo[0]) - o.x) * inv.x, tx1 = ((FRAME_LDS ? wallS[VR_F_SCENE_HI] : p.sceneHi[0]) - o.x) * inv.x;
  const float ty0 = ((FRAME_LDS ? wallS[VR_F_SCENE_LO + 1] : p.sceneLo[1]) - o.y) * inv.y, ty1 = ((FRAME_LDS ? wallS[VR_F_SCENE_HI + 1] : p.sceneHi[1]) - o.y) * inv.y;
  const float tz0 = ((FRAME_LDS ? wallS[VR_F_SCENE_LO + 2] : p.sceneLo[2]) - o.z) * inv.z, tz1 = ((FRAME_LDS ? wallS[VR_F_SCENE_HI + 2] : p.sceneHi[2]) - o.z) * inv.z;
  const float tIn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tnear));
  const float tOut = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
  // tWall: the exact wall test's hit of this ray (or infinity).  A ray that meets a side wall BEFORE it can enter the scene
  // box cannot meet the geometry first (a hit has t >= tIn up to a few ulp: the margin): it stays out of the query's box.
  // Such rays were binned — folded by the boundary condition — with the rays on the far side of the domain, and one of them
  // stretched its wave's box across the whole scene: every packet query of a flat plane that gave up was one of these.
  bool valid = part && tIn <= tOut && !(tWall < tIn * 0.99999f);
  cd.count = 0;
  cd.local = 0ull;
  const float big = 3.0e38f;
  cd.box = false;
  if (!ballot64(valid))
    return true; // nobody reaches the scene box: every ray misses the geometry
  // (Q starts at the ray's origin where that lies inside the box, not at tnear: the neighbour test
  //  accepts any t > 0)
  float tQ = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), 0.f));
  float tEnd = tOut;
  if (RELIEF && !(p.debugFlags & 512u)) { // (flag 512: the scene box's clip, for comparison)
    float tA, tB;
    relief_clip(wallS, valid, o, d, tQ, tOut, tA, tB);
    valid = valid && tA <= tB;
    if (!ballot64(valid))
      return true; // no ray's height meets the relief under it
    tQ = tA;
    tEnd = tB;
  }
  const float ax = o.x + d.x * tQ, ay = o.y + d.y * tQ, az = o.z + d.z * tQ;
  const float bx = o.x + d.x * tEnd, by = o.y + d.y * tEnd, bz = o.z + d.z * tEnd;
  float qlx = valid ? fminf(ax, bx) : big, qly = valid ? fminf(ay, by) : big, qlz = valid ? fminf(az, bz) : big;
  float qhx = valid ? fmaxf(ax, bx) : -big, qhy = valid ? fmaxf(ay, by) : -big, qhz = valid ? fmaxf(az, bz) : -big;
  wave_minmax6(qlx, qly, qlz, qhx, qhy, qhz);
  const float pad = FRAME_LDS ? wallS[VR_F_PQ_PAD] : p.pqPad;
  qlx -= pad;
  qly -= pad;
  qlz -= pad;
  qhx += pad;
  qhy += pad;
  qhz += pad;
  if (KEEPQ) {
    cd.box = true;
    if (lane == 0u) {
      cd.rec[VR_PQ_BOX] = mk_u4(__float_as_uint(qlx), __float_as_uint(qly), __float_as_uint(qlz), 0u);
      cd.rec[VR_PQ_BOX + 1] = mk_u4(__float_as_uint(qhx), __float_as_uint(qhy), __float_as_uint(qhz), 0u);
    }
  }
#ifdef VR_DIAG
  unsigned long long pqT0 = __builtin_amdgcn_s_memtime();
#define VR_PQ_MARK(k)                                                                                                  \
  do {                                                                                                                 \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                      \
    if ((threadIdx.x & 63u) == 0u)                                                                                     \
      phaseT[k] += now_ - pqT0;                                                                                        \
    pqT0 = now_;                                                                                                       \
  } while (0)
#else
#define VR_PQ_MARK(k)
#endif
  // breadth-first search of the 64-ary tree: a frontier entry = {first child, child count | prims flag}
  const float4 *__restrict__ wide = reinterpret_cast<const float4 *>(p.wide);
  const float4 *__restrict__ prims = reinterpret_cast<const float4 *>(p.prims);
  unsigned fFirst = p.wideTopFirst, fCnt = p.wideTopCount;
  unsigned nF = 1;
  const unsigned long long ltMask = (1ull << lane) - 1ull;
  if constexpr (!CACHE) {
  while (!((unsigned)__builtin_amdgcn_readlane((int)fCnt, 0) & VR_WIDE_PRIMS)) {
    unsigned nNext = 0;
    for (unsigned j = 0; j < nF; ++j) {
      const unsigned first = (unsigned)__builtin_amdgcn_readlane((int)fFirst, (int)j);
      const unsigned cnt = (unsigned)__builtin_amdgcn_readlane((int)fCnt, (int)j);
      bool hit = false;
      unsigned cf = 0, cc = 0;
      if (lane < cnt) {
        DIAG(3);
        const float4 a = wide[2 * (size_t)(first + lane)], b = wide[2 * (size_t)(first + lane) + 1];
        hit = all6(a.x <= qhx, b.x >= qlx, a.y <= qhy, b.y >= qly, a.z <= qhz, b.z >= qlz);
        cf = __float_as_uint(a.w);
        cc = __float_as_uint(b.w);
      }
      const unsigned long long m = ballot64(hit);
      const unsigned pos = nNext + (unsigned)__popcll(m & ltMask);
      if (hit && pos < 64u) {
        lst[pos] = cf;
        lst[64u + pos] = cc;
      }
      nNext += (unsigned)__popcll(m);
    }
    if (nNext > p.pqMaxFrontier)
      return false; // (nothing touched yet)
    if (nNext == 0u)
      return true;
    fFirst = lst[lane];
    fCnt = lst[64u + lane];
    nF = nNext;
  }
  } else { // CACHE: the same search with an enlarged box, its frontier kept for the neighbouring rounds
    constexpr unsigned KEEP = VR_PQ_KEEP; // cached leaf nodes (entries lst[0 ..], lst[64 ..]; their boxes: cboxes, 6 floats each)
    const bool caching = p.pqMargin > 0.f && !(fCnt & VR_WIDE_PRIMS);
    bool haveList = false;
    if (caching && lst[39] != 0u) {
      // is the frontier the last search left complete for this round's box?
      const bool inside = qlx >= __uint_as_float(lst[32]) && qly >= __uint_as_float(lst[33]) && qlz >= __uint_as_float(lst[34]) &&
                          qhx <= __uint_as_float(lst[35]) && qhy <= __uint_as_float(lst[36]) && qhz <= __uint_as_float(lst[37]);
      haveList = __builtin_amdgcn_readfirstlane((int)inside) != 0; // (wave-uniform: every operand is)
    }
    if (!haveList) {
      // the search box: the round's own, or — caching — enlarged, so that its frontier serves the neighbouring rounds too
      const float mg = caching ? p.pqMargin : 0.f;
      const float elx = qlx - mg, ely = qly - mg, elz = qlz - mg, ehx = qhx + mg, ehy = qhy + mg, ehz = qhz + mg;
      if (caching && lane == 0u)
        lst[39] = 0u; // (the lists are about to be overwritten; a search that gives up leaves no cache)
      while (!((unsigned)__builtin_amdgcn_readlane((int)fCnt, 0) & VR_WIDE_PRIMS)) {
        unsigned nNext = 0;
        for (unsigned j = 0; j < nF; ++j) {
          const unsigned first = (unsigned)__builtin_amdgcn_readlane((int)fFirst, (int)j);
          const unsigned cnt = (unsigned)__builtin_amdgcn_readlane((int)fCnt, (int)j);
          bool hit = false;
          unsigned cf = 0, cc = 0;
          float4 a = make_float4(0, 0, 0, 0), b = a;
          if (lane < cnt) {
            DIAG(3);
            a = wide[2 * (size_t)(first + lane)];
            b = wide[2 * (size_t)(first + lane) + 1];
            hit = all6(a.x <= ehx, b.x >= elx, a.y <= ehy, b.y >= ely, a.z <= ehz, b.z >= elz);
            cf = __float_as_uint(a.w);
            cc = __float_as_uint(b.w);
          }
          const unsigned long long m = ballot64(hit);
          const unsigned pos = nNext + (unsigned)__popcll(m & ltMask);
          if (hit && pos < 64u) {
            lst[pos] = cf;
            lst[64u + pos] = cc;
            if (caching && pos < KEEP) { // (kept only where this turns out to be the last level)
              cboxes[6u * pos] = a.x, cboxes[6u * pos + 1u] = a.y, cboxes[6u * pos + 2u] = a.z;
              cboxes[6u * pos + 3u] = b.x, cboxes[6u * pos + 4u] = b.y, cboxes[6u * pos + 5u] = b.z;
            }
          }
          nNext += (unsigned)__popcll(m);
        }
        if (nNext > p.pqMaxFrontier) {
          DIAG(14); // (diag: gave up on the frontier)
          return false; // (nothing touched yet)
        }
        if (nNext == 0u && !caching)
          return true;
        fFirst = lst[lane];
        fCnt = lst[64u + lane];
        nF = nNext;
        if (nNext == 0u)
          break; // (nothing meets the enlarged box: an empty frontier, complete for it)
      }
      if (caching) {
        if (nF <= KEEP) {
          if (lane == 0u) {
            lst[32] = __float_as_uint(elx), lst[33] = __float_as_uint(ely), lst[34] = __float_as_uint(elz);
            lst[35] = __float_as_uint(ehx), lst[36] = __float_as_uint(ehy), lst[37] = __float_as_uint(ehz);
            lst[38] = nF;
            lst[39] = 1u;
          }
          haveList = true;
        }
      }
    }
    if (haveList) {
      // the kept leaf nodes that meet THIS round's box (their boxes from LDS: no node load, no dependent trip)
      const unsigned nC = (unsigned)__builtin_amdgcn_readfirstlane((int)lst[38]);
      bool meet = false;
      unsigned f0 = 0, c0 = 0;
      if (lane < nC) {
        // (the six values FIRST, then one combined test: as a chain of && over volatile reads the ISA was six nested
        //  branches, each re-loading the spilled LDS address from scratch behind an s_waitcnt vmcnt(0) — six serial trips
        //  to memory per round of the headline kernel)
        const volatile VR_LDS float *bx = cboxes + 6u * lane;
        const float b0 = bx[0], b1 = bx[1], b2 = bx[2], b3 = bx[3], b4 = bx[4], b5 = bx[5];
        meet = (b0 <= qhx) & (b3 >= qlx) & (b1 <= qhy) & (b4 >= qly) & (b2 <= qhz) & (b5 >= qlz);
        f0 = lst[lane];
        c0 = lst[64u + lane];
      }
      const unsigned long long m = ballot64(meet);
      nF = (unsigned)__popcll(m);
      if (nF == 0u)
        return true;
      if (meet) {
        const unsigned pos = (unsigned)__popcll(m & ltMask);
        lst[16u + pos] = f0;
        lst[80u + pos] = c0;
      }
      fFirst = lst[16u + lane];
      fCnt = lst[80u + lane];
    }
  }
  // last level: the frontier's children are primitives.  Lanes load the RECORDS of a node's
  // (<= 64) primitives, keep those whose bounds meet Q, and every kept record is broadcast from
  // its lane to the whole wave for the exact test: no separate box level, no dependent record fetch.
  // how many candidates are there?  (boxes only, before any exact test: a packet over a relief as
  // deep as it is wide meets dozens of primitives, and for that the slab-test packet or the
  // per-lane walk is cheaper — give up with nothing touched)
  if (nF >= 3u) { // (one or two leaf nodes: the usual case of a compact packet, not worth the extra pass)
    unsigned total = 0;
    for (unsigned j = 0; j < nF; ++j) {
      const unsigned first = (unsigned)__builtin_amdgcn_readlane((int)fFirst, (int)j);
      const unsigned cnt = (unsigned)__builtin_amdgcn_readlane((int)fCnt, (int)j) & 0x7FFFFFFFu;
      bool cand = false;
      if (lane < cnt) {
        const unsigned q = first + lane;
        if (GEO == 0) {
          const float4 r0 = prims[2 * (size_t)q];
          cand = all6(r0.x - r0.w <= qhx, r0.x + r0.w >= qlx, r0.y - r0.w <= qhy, r0.y + r0.w >= qly, r0.z - r0.w <= qhz, r0.z + r0.w >= qlz);
        } else {
          const float4 r0 = prims[4 * (size_t)q], r1 = prims[4 * (size_t)q + 1], r2 = prims[4 * (size_t)q + 2];
          const float v1x = r0.x - r1.x, v1y = r0.y - r1.y, v1z = r0.z - r1.z;
          const float v2x = r0.x + r2.x, v2y = r0.y + r2.y, v2z = r0.z + r2.z;
          cand = fminf(r0.x, fminf(v1x, v2x)) <= qhx && fmaxf(r0.x, fmaxf(v1x, v2x)) >= qlx &&
                 fminf(r0.y, fminf(v1y, v2y)) <= qhy && fmaxf(r0.y, fmaxf(v1y, v2y)) >= qly &&
                 fminf(r0.z, fminf(v1z, v2z)) <= qhz && fmaxf(r0.z, fmaxf(v1z, v2z)) >= qlz;
        }
      }
      total += (unsigned)__popcll(ballot64(cand));
    }
    if (total > p.pqMaxCand) {
      DIAG(15); // (diag: gave up on the candidate count)
      return false;
    }
  }
  unsigned tests = 0;
  VR_PQ_MARK(14); // (diag: the descent)
  for (unsigned j = 0; j < nF; ++j) {
    const unsigned first = (unsigned)__builtin_amdgcn_readlane((int)fFirst, (int)j);
    const unsigned cnt = (unsigned)__builtin_amdgcn_readlane((int)fCnt, (int)j) & 0x7FFFFFFFu;
    const unsigned q = first + lane; // leaf position (primitive entries of the tree are implicit)
    bool cand = false;
    float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
    if (lane < cnt) {
      DIAG(3);
      if (GEO == 0) {
        r0 = prims[2 * (size_t)q];
        r1 = prims[2 * (size_t)q + 1];
        // a disc lies inside the ball of its radius
        cand = all6(r0.x - r0.w <= qhx, r0.x + r0.w >= qlx, r0.y - r0.w <= qhy, r0.y + r0.w >= qly, r0.z - r0.w <= qhz, r0.z + r0.w >= qlz);
      } else {
        r0 = prims[4 * (size_t)q];
        r1 = prims[4 * (size_t)q + 1];
        r2 = prims[4 * (size_t)q + 2];
        r3 = prims[4 * (size_t)q + 3];
        // vertices v0, v1 = v0 - e1, v2 = v0 + e2 (rounding of the two sums is far inside the pad of Q)
        const float v1x = r0.x - r1.x, v1y = r0.y - r1.y, v1z = r0.z - r1.z;
        const float v2x = r0.x + r2.x, v2y = r0.y + r2.y, v2z = r0.z + r2.z;
        cand = fminf(r0.x, fminf(v1x, v2x)) <= qhx && fmaxf(r0.x, fmaxf(v1x, v2x)) >= qlx &&
               fminf(r0.y, fminf(v1y, v2y)) <= qhy && fmaxf(r0.y, fmaxf(v1y, v2y)) >= qly &&
               fminf(r0.z, fminf(v1z, v2z)) <= qhz && fmaxf(r0.z, fmaxf(v1z, v2z)) >= qlz;
      }
    }
    unsigned long long m = ballot64(cand);
    VR_PQ_MARK(9); // (diag: the candidates' record loads + box tests)
    while (m) {
      const int k = __ffsll((long long)m) - 1;
      m &= m - 1ull;
      if (part) {
        DIAG(4);
      }
      const unsigned qq = first + (unsigned)k;
      float t;
      if (GEO == 0) {
        const float4 c4 = make_float4(lane_bcast(r0.x, k), lane_bcast(r0.y, k), lane_bcast(r0.z, k), lane_bcast(r0.w, k));
        const V3 n = mk(lane_bcast(r1.x, k), lane_bcast(r1.y, k), lane_bcast(r1.z, k));
        const unsigned orig = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(r1.w), k);
        const bool ok = hit_disc(o, d, tnear, c4, n, t);
        const bool took = hit_update(h, part && ok, t, orig, qq);
        if (CREDIT) {
          const int c = (int)tests;
          cd.mine = took ? (unsigned)c : cd.mine;
          if ((int)lane == k) { // (the lane that loaded the record files it: its r0 / r1 are the broadcast c4 / n)
            cd.rec[c] = mk_u4(qq, __float_as_uint(r0.x), __float_as_uint(r0.y), __float_as_uint(r0.z));
            cd.rec[VR_PQ_NRM + c] = mk_u4(__float_as_uint(r1.x), __float_as_uint(r1.y), __float_as_uint(r1.z), __float_as_uint(r0.w));
          }
          // (a wave-wide early out between the cheap sign tests and the division / distance part of
          //  these two tests was measured: the extra votes and branches cost more than they save)
          if (part && local_disc_hit(o, d, c4, n))
            cd.local |= 1ull << c;
          cd.count = tests + 1u;
        }
      } else {
        const V3 v0 = mk(lane_bcast(r0.x, k), lane_bcast(r0.y, k), lane_bcast(r0.z, k));
        const V3 e1 = mk(lane_bcast(r1.x, k), lane_bcast(r1.y, k), lane_bcast(r1.z, k));
        const V3 e2 = mk(lane_bcast(r2.x, k), lane_bcast(r2.y, k), lane_bcast(r2.z, k));
        const V3 Ng = mk(lane_bcast(r3.x, k), lane_bcast(r3.y, k), lane_bcast(r3.z, k));
        const unsigned orig = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(r0.w), k);
        const bool ok = hit_tri(o, d, tnear, v0, e1, e2, Ng, t);
        hit_update(h, part && ok, t, orig, qq);
      }
      if (++tests > 2u * p.pqMaxCand) {
        DIAG(10); // (diag: gave up in the exact tests)
        return false; // (only reachable with <= 2 leaf nodes: the hits found so far are real, the walk goes on from them)
      }
    }
    VR_PQ_MARK(15); // (diag: the exact tests)
  }
  return true;
}

// rayTraceKernel.hpp:462-507 (neighbour disk test)
__device__ __forceinline__ bool local_disc_hit(const V3 &ro, const V3 &rd, const float4 &c4, const V3 &n) {
  const float prod = vdot(n, rd);
  if (prod > 0.f)
    return false;
  if (fabsf(prod) < 1e-6f)
    return false;
  const V3 c = V3{c4.x, c4.y, c4.z};
  const float ddneg = vdot(c, n);
  const float tt = (ddneg - vdot(n, ro)) / prod;
  if (tt <= 0.f)
    return false;
  V3 hp = V3{rd.x * tt + ro.x, rd.y * tt + ro.y, rd.z * tt + ro.z};
  hp.x = hp.x - c.x;
  hp.y = hp.y - c.y;
  hp.z = hp.z - c.z;
  const float dist = sqrtf(vdot(hp, hp));
  return c4.w > dist;
}

// the same test, also returning the distance between the point of impact and the disc centre
// (VIENNARAY_USE_WDIST, rayTraceKernel.hpp:258-296); false leaves dist undefined
__device__ __forceinline__ bool local_disc_hit_dist(const V3 &ro, const V3 &rd, const float4 &c4, const V3 &n,
                                                    float &dist) {
  const float prod = vdot(n, rd);
  if (prod > 0.f)
    return false;
  if (fabsf(prod) < 1e-6f)
    return false;
  const V3 c = V3{c4.x, c4.y, c4.z};
  const float ddneg = vdot(c, n);
  const float tt = (ddneg - vdot(n, ro)) / prod;
  if (tt <= 0.f)
    return false;
  V3 hp = V3{rd.x * tt + ro.x, rd.y * tt + ro.y, rd.z * tt + ro.z};
  hp.x = hp.x - c.x;
  hp.y = hp.y - c.y;
  hp.z = hp.z - c.z;
  dist = sqrtf(vdot(hp, hp));
  return c4.w > dist;
}

// rayReflection.hpp:13-29
__device__ __forceinline__ V3 reflect_specular(const V3 &dir, const V3 &n) {
  const V3 inv = V3{-dir.x, -dir.y, -dir.z};
  const float f = 2 * vdot(n, inv);
  return V3{f * n.x - inv.x, f * n.y - inv.y, f * n.z - inv.z};
}

// rayUtil.hpp:204-215 (the D==2 projection of fillRayDirection)
template <int D> __device__ __forceinline__ V3 project_dir(V3 d) {
  if (D == 2) {
    if (d.z != 0.f) {
      d.z = 0.f;
      vnormalize(d);
    }
  }
  return d;
}

// raySourceRandom.hpp:70-116: one power-cosine sample in the local frame, with glibc's
// float sincosf / powf reproduced bit for bit (vr_libm.hpp), so the device traces the
// very rays the reference's CPU loop traces.
__device__ __forceinline__ void cosine_sample(float r1, float r2, float ee, float &cosTheta, float &sinTheta,
                                              float &cosPhi, float &sinPhi) {
  const float ang = (float)(3.14159265358979323846 * 2. * (double)r1); // rayUtil.hpp:247-256
  glibc_sincosf(ang, sinPhi, cosPhi);
  cosTheta = glibc_powf(r2, ee);                                      // std::pow(float, float)
  sinTheta = (float)sqrt(1. - (double)(cosTheta * cosTheta));
}

} // namespace vr

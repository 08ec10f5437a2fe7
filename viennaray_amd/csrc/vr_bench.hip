// vr_bench.hip — instruction-issue ceilings of the device, measured (vr_debug_issue_rate).
//
// The flux tracer is not bandwidth bound (sorted rays fetch nodes through the
// scalar cache; 6-11 % of the HBM peak); what bounds it is how fast a CU issues wave
// instructions.  These kernels measure that ceiling for the instruction mixes the two hot
// kernels are made of, at a chosen number of resident waves per SIMD, together with the
// clock the chip sustains while doing so:
//
//   KIND 0  f32 VALU, independent chains : v_fma_f32 / v_min_f32 / v_max_f32 (the slab test's mix)
//   KIND 1  f32 VALU, ONE dependent chain per wave (what a single traversal step looks like)
//   KIND 2  64-bit integer multiply-add chain = mt_step of the generator (mt19937_64 seeding)
//   KIND 3  SALU : s_add_u32 / s_and_b32 / s_lshl_b32 / s_xor_b32 on independent registers
//   KIND 4  packet-traversal mix: 24 VALU + 16 SALU interleaved, SALU consuming a VALU compare
//           (v_cmp -> s_and_b64), i.e. the vote pattern of bvh_hit_packet
//   KIND 5  24 VALU + 16 SALU interleaved, the two streams independent (can they co-issue?)
//
// Every wave runs `iters` passes over an unrolled body with a known instruction count, so
// rate = waves x iters x body / time (HIP events); rocprofv3's SQ_INSTS_VALU / SQ_INSTS_SALU
// of the same launch cross-check the count (tools/issue_ceiling.py).  The clock is read
// in-kernel: delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).
#include <hip/hip_runtime.h>

#include "vr_device.hpp"
#include "vr_kernels.hpp"

namespace vr {

struct IssueOut {
  u64 cycles;   // s_memtime delta of this wave
  u64 realtime; // s_memrealtime delta (100 MHz)
  u64 sink;     // keeps the results alive
};

#define VR_REP4(X) X X X X
#define VR_REP8(X) X X X X X X X X

template <int KIND> __global__ __launch_bounds__(256) void issue_kernel(unsigned iters, IssueOut *out, float seedf) {
  const unsigned tid = threadIdx.x;
  const unsigned gwave = (blockIdx.x * 256u + tid) >> 6;
  float a0 = seedf + tid, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f,
        a7 = a0 + 7.f;
  const float m = 0.999f, c = 0.5f;
  u64 x = 5489ull + tid;
  unsigned s0 = blockIdx.x + 1u, s1 = 3u, s2 = 5u, s3 = 7u;
  s0 = __builtin_amdgcn_readfirstlane(s0);
  __builtin_amdgcn_s_waitcnt(0);
  const u64 t0 = __builtin_amdgcn_s_memtime();
  const u64 r0 = __builtin_amdgcn_s_memrealtime();
  for (unsigned it = 0; it < iters; ++it) {
    if (KIND == 0) {
      // 32 VALU per pass: 8 independent chains x {fma, min, max, fma}
      asm volatile(VR_REP4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_min_f32 %2, %2, %9\n v_max_f32 %3, %3, %8\n"
                           "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_min_f32 %6, %6, %9\n v_max_f32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(m), "v"(c));
    } else if (KIND == 1) {
      // 32 VALU per pass, every one depending on the previous
      asm volatile(VR_REP8("v_fma_f32 %0, %0, %1, %2\n v_min_f32 %0, %0, %2\n v_fma_f32 %0, %0, %1, %2\n v_max_f32 %0, %0, %1\n")
                   : "+v"(a0)
                   : "v"(m), "v"(c));
    } else if (KIND == 2) {
      // 32 steps of the mt19937_64 seeding recurrence, the very step of gen_kernel (mt_step_v: 6 VALU, 3 of
      // them 32-bit multiplies)
      const MtMul mm = mt_mul_init();
#pragma unroll
      for (unsigned j = 1; j <= 32; ++j)
        x = mt_step_v(mm, x, j);
    } else if (KIND == 6) {
      // the step with its high word by two more v_mad_u64_u32 chained onto the first (5 VALU + a move): faster in
      // isolation, slower in the generator (vr_device.hpp, mt_step_v)
      unsigned al = 0x4C957F2Du, ah = 0x5851F42Du, zero = 0u;
      asm volatile("" : "+v"(al), "+v"(ah), "+v"(zero));
#pragma unroll
      for (unsigned j = 1; j <= 32; ++j) {
        const unsigned xh = (unsigned)(x >> 32);
        const unsigned t = (unsigned)x ^ (xh >> 30);
        u64 r, r2, r3;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(t), "v"(al), "s"((u64)j) : "vcc");
        const u64 carry = ((u64)zero << 32) | (unsigned)(r >> 32);
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r2) : "v"(t), "v"(ah), "v"(carry) : "vcc");
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r3) : "v"(xh), "v"(al), "v"(r2) : "vcc");
        x = ((u64)(unsigned)r3 << 32) | (unsigned)r;
      }
    } else if (KIND == 3) {
      // 32 SALU per pass on four independent registers
      asm volatile(VR_REP8("s_add_u32 %0, %0, %1\n s_and_b32 %1, %1, %2\n s_lshl_b32 %2, %2, 1\n s_xor_b32 %3, %3, %0\n")
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                   :
                   : "scc");
    } else if (KIND == 5) {
      // 24 VALU + 16 SALU per pass, the two streams independent of each other
      asm volatile(VR_REP8("v_fma_f32 %0, %0, %6, %7\n v_min_f32 %1, %1, %7\n s_add_u32 %4, %4, 1\n"
                           "v_fma_f32 %2, %2, %6, %7\n s_xor_b32 %5, %5, %4\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1)
                   : "v"(m), "v"(c)
                   : "scc");
    } else {
      // 24 VALU + 16 SALU per pass: slab-test arithmetic, a vote, scalar bookkeeping
      asm volatile(VR_REP8("v_fma_f32 %0, %0, %4, %5\n v_min_f32 %1, %1, %0\n s_add_u32 %2, %2, 1\n"
                           "v_cmp_le_f32 vcc, %0, %1\n s_and_b64 vcc, vcc, exec\n")
                   : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1)
                   : "v"(m), "v"(c)
                   : "vcc", "scc");
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  const u64 t1 = __builtin_amdgcn_s_memtime();
  const u64 r1 = __builtin_amdgcn_s_memrealtime();
  if ((tid & 63u) == 0) {
    out[gwave].cycles = t1 - t0;
    out[gwave].realtime = r1 - r0;
  }
  // results stay observable: lanes store a value nobody reads back
  const float fs = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (fs == 12345.678f || x == 42ull || (s0 ^ s1 ^ s2 ^ s3) == 0xDEADBEEFu)
    out[gwave].sink = x + (u64)s0;
}

// counted per pass of the body: 32 VALU (kinds 0, 1), 32 mt_step (kind 2: one 64-bit
// multiply-add each), 32 SALU (kind 3), 24 VALU + 16 SALU (kind 4)
hipError_t launch_issue_kernel(int kind, unsigned blocks, unsigned iters, void *out, hipStream_t s) {
  IssueOut *o = reinterpret_cast<IssueOut *>(out);
  switch (kind) {
  case 0: hipLaunchKernelGGL((issue_kernel<0>), dim3(blocks), dim3(256), 0, s, iters, o, 1.0f); break;
  case 1: hipLaunchKernelGGL((issue_kernel<1>), dim3(blocks), dim3(256), 0, s, iters, o, 1.0f); break;
  case 2: hipLaunchKernelGGL((issue_kernel<2>), dim3(blocks), dim3(256), 0, s, iters, o, 1.0f); break;
  case 3: hipLaunchKernelGGL((issue_kernel<3>), dim3(blocks), dim3(256), 0, s, iters, o, 1.0f); break;
  case 5: hipLaunchKernelGGL((issue_kernel<5>), dim3(blocks), dim3(256), 0, s, iters, o, 1.0f); break;
  case 6: hipLaunchKernelGGL((issue_kernel<6>), dim3(blocks), dim3(256), 0, s, iters, o, 1.0f); break;
  default: hipLaunchKernelGGL((issue_kernel<4>), dim3(blocks), dim3(256), 0, s, iters, o, 1.0f); break;
  }
  return hipGetLastError();
}

} // namespace vr

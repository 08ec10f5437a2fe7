// gather_rate.hip — how many scattered per-lane loads a CU's vector memory pipeline serves per clock (MI355X).
// The per-lane BVH walks of the general trace kernels issue ~70 such loads per trace segment (pair nodes, primitive
// records, neighbour records: 16 bytes each, a different cache line per lane); this measures the ceiling they run under.
//   build: hipcc --offload-arch=gfx950 -O3 tools/gather_rate.hip -o tools/gather_rate      run: tools/gather_rate
// Every lane walks its own LCG sequence over a table of `bytes` (L1-resident, L2-resident, beyond), LOADS independent
// loads of WIDTH dwords in flight per pass; rate = lanes x loads / (time x CUs x clock).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                                       \
  do {                                                                                                                 \
    hipError_t e_ = (x);                                                                                               \
    if (e_ != hipSuccess) {                                                                                            \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                     \
      std::exit(1);                                                                                                    \
    }                                                                                                                  \
  } while (0)

// PATTERN 0: every lane its own random 32-byte slot; 1: the two 16-byte halves of one slot (a pair node);
// 2: all lanes of a wave the same slot (a coherent wave); 3: lanes in groups of 4 share a slot;
// 4: 16 + 8 bytes of one 24-byte slot, five slots per 128-byte line (a 24-byte pair node)
template <int WIDTH, int PATTERN>
__global__ __launch_bounds__(256) void gather_kernel(const uint4 *__restrict__ table, unsigned slots, unsigned iters,
                                                      unsigned long long *out) {
  const unsigned tid = blockIdx.x * 256u + threadIdx.x;
  unsigned s = tid * 2654435761u + 12345u;
  if (PATTERN == 2)
    s = (tid >> 6) * 2654435761u + 12345u;
  if (PATTERN == 3)
    s = (tid >> 2) * 2654435761u + 12345u;
  unsigned acc = 0, slot = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (unsigned it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (PATTERN != 4 && (PATTERN != 1 || !(k & 1))) { // (PATTERN 1: the second load is the other half of the same slot)
        s = s * 1664525u + 1013904223u;
        slot = (s >> 8) & (slots - 1u); // (slots is a power of two)
      }
      const uint4 *p = table + 2u * slot + (PATTERN == 1 ? (k & 1) : 0);
      if (PATTERN == 4) {
        if (!(k & 1)) {
          s = s * 1664525u + 1013904223u;
          slot = (s >> 8) & (slots - 1u);
        }
        const unsigned line = slot / 5u, in = slot - line * 5u; // (slots counts 32-byte units: lines = slots / 4 >= slots / 5)
        const char *q = reinterpret_cast<const char *>(table) + (size_t)(line & (slots / 4u - 1u)) * 128u + in * 24u;
        if (k & 1) {
          const uint2 v = *reinterpret_cast<const uint2 *>(q + 16);
          acc ^= v.x ^ v.y;
        } else {
          const uint4 v = *reinterpret_cast<const uint4 *>(q);
          acc ^= v.x ^ v.w;
        }
        continue;
      }
      if (WIDTH == 4) {
        const uint4 v = *p;
        acc ^= v.x ^ v.w;
      } else if (WIDTH == 2) {
        const uint2 v = *reinterpret_cast<const uint2 *>(p);
        acc ^= v.x ^ v.y;
      } else {
        acc ^= *reinterpret_cast<const unsigned *>(p);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (acc == 0x12345678u || (threadIdx.x & 63u) == 0) { // (shader clocks and 100 MHz ticks of this wave: the clock the chip held)
    out[2 * (tid >> 6)] = (t1 - t0) + (acc == 0x12345678u ? 1 : 0);
    out[2 * (tid >> 6) + 1] = r1 - r0;
  }
}

template <int WIDTH, int PATTERN> static void run(const char *name, size_t bytes, int blocksPerCU, int cus) {
  const unsigned slots = (unsigned)(bytes / 32);
  uint4 *table;
  CHECK(hipMalloc(&table, bytes));
  CHECK(hipMemset(table, 1, bytes));
  const unsigned blocks = (unsigned)(cus * blocksPerCU), iters = 2000;
  unsigned long long *out;
  CHECK(hipMalloc(&out, (size_t)blocks * 4 * 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((gather_kernel<WIDTH, PATTERN>), dim3(blocks), dim3(256), 0, 0, table, slots, 50u, out); // warm
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((gather_kernel<WIDTH, PATTERN>), dim3(blocks), dim3(256), 0, 0, table, slots, iters, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> host((size_t)blocks * 8);
  CHECK(hipMemcpy(host.data(), out, host.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, real = 0;
  for (size_t w = 0; w < (size_t)blocks * 4; ++w) {
    cyc += (double)host[2 * w];
    real += (double)host[2 * w + 1];
  }
  const double ghz = real > 0 ? cyc / real * 0.1 : 0.0; // (s_memrealtime ticks at 100 MHz)
  const double loads = (double)blocks * 256.0 * iters * 8.0;
  std::printf("%-44s table %8zu KB  %d waves/SIMD: %7.3f ms  %6.1f G lane-loads/s  %.3f lane-loads per CU-clock (measured %.2f GHz)  %.0f GB/s\n",
              name, bytes >> 10, blocksPerCU, ms, loads / ms / 1e6, loads / (ms * 1e-3) / cus / (ghz * 1e9), ghz,
              loads * WIDTH * 4 / ms / 1e6);
  CHECK(hipFree(table));
  CHECK(hipFree(out));
}

int main() {
  hipDeviceProp_t pr;
  CHECK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  std::printf("%s: %d CUs\n", pr.name, cus);
  for (int w : {2, 6}) {
    run<4, 0>("16 B per lane, every lane its own line", 16 << 10, w, cus);
    run<4, 0>("16 B per lane, every lane its own line", 256 << 10, w, cus);
    run<4, 0>("16 B per lane, every lane its own line", 2 << 20, w, cus);
    run<4, 0>("16 B per lane, every lane its own line", 64 << 20, w, cus);
    run<4, 1>("2 x 16 B of one 32-byte slot (a pair node)", 16 << 10, w, cus);
    run<4, 1>("2 x 16 B of one 32-byte slot (a pair node)", 256 << 10, w, cus);
    run<4, 4>("16 + 8 B of one 24-byte slot, 5 per line", 16 << 10, w, cus);
    run<4, 4>("16 + 8 B of one 24-byte slot, 5 per line", 256 << 10, w, cus);
    run<2, 0>("8 B per lane, own line", 16 << 10, w, cus);
    run<1, 0>("4 B per lane, own line", 16 << 10, w, cus);
    run<4, 2>("16 B, all lanes of a wave one address", 16 << 10, w, cus);
    run<4, 3>("16 B, groups of 4 lanes one address", 16 << 10, w, cus);
  }
  return 0;
}

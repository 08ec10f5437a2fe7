// vr_grid.hip — device build of the CELL GRID (vr_grid.hpp): per-cell lists of primitive records
// and the occupancy word of every 4x4x4-cell brick.
//
//   grid_count_kernel   one thread per primitive: every cell its padded box overlaps gets +1
//   (scan)              cell counts -> first record of each cell
//   grid_fill_kernel    one thread per primitive: a copy of its record into each of those cells
//   grid_finish_kernel  one thread per brick: header words (first << 6 | count) of its 64 cells,
//                       the brick's occupancy word, the largest list
#include <hip/hip_runtime.h>

#include "vr_kernels.hpp"
#include "vr_types.hpp"

namespace vr {

__device__ __forceinline__ void grid_cell_range(const GridParams &g, const float *box, int (&lo)[3], int (&hi)[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int a = (int)floorf((box[k] - g.pad - g.lo[k]) * g.invH);
    const int b = (int)floorf((box[3 + k] + g.pad - g.lo[k]) * g.invH);
    lo[k] = min(max(a, 0), (int)g.dim[k] - 1);
    hi[k] = min(max(b, 0), (int)g.dim[k] - 1);
  }
}

__global__ void grid_count_kernel(GridParams g) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= g.n)
    return;
  int lo[3], hi[3];
  grid_cell_range(g, g.sbox + 6 * (size_t)q, lo, hi);
  for (int z = lo[2]; z <= hi[2]; ++z)
    for (int y = lo[1]; y <= hi[1]; ++y)
      for (int x = lo[0]; x <= hi[0]; ++x)
        atomicAdd(&g.cellStart[((size_t)z * g.dim[1] + y) * g.dim[0] + x], 1u);
}

__global__ void grid_fill_kernel(GridParams g) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= g.n)
    return;
  int lo[3], hi[3];
  grid_cell_range(g, g.sbox + 6 * (size_t)q, lo, hi);
  const float4 *pr = reinterpret_cast<const float4 *>(g.prims);
  float4 *out = reinterpret_cast<float4 *>(g.cellRecs);
  float4 r0, r1, r2 = make_float4(0, 0, 0, 0), r3 = r2;
  if (g.geo == 0) {
    r0 = pr[2 * (size_t)q];
    r1 = pr[2 * (size_t)q + 1];
    r1.w = __uint_as_float(q);
  } else {
    r0 = pr[4 * (size_t)q];
    r1 = pr[4 * (size_t)q + 1];
    r2 = pr[4 * (size_t)q + 2];
    r3 = pr[4 * (size_t)q + 3];
    r0.w = __uint_as_float(q);
  }
  for (int z = lo[2]; z <= hi[2]; ++z)
    for (int y = lo[1]; y <= hi[1]; ++y)
      for (int x = lo[0]; x <= hi[0]; ++x) {
        const size_t cell = ((size_t)z * g.dim[1] + y) * g.dim[0] + x;
        const size_t slot = (size_t)g.cellStart[cell] + atomicAdd(&g.cellFill[cell], 1u);
        if (g.geo == 0) {
          out[2 * slot] = r0;
          out[2 * slot + 1] = r1;
        } else {
          out[4 * slot] = r0;
          out[4 * slot + 1] = r1;
          out[4 * slot + 2] = r2;
          out[4 * slot + 3] = r3;
        }
      }
}

// stats[0] = largest list, stats[1] = occupied cells
__global__ void grid_finish_kernel(GridParams g) {
  const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned nb = g.bdim[0] * g.bdim[1] * g.bdim[2];
  if (b >= nb)
    return;
  const unsigned bx = b % g.bdim[0], by = (b / g.bdim[0]) % g.bdim[1], bz = b / (g.bdim[0] * g.bdim[1]);
  unsigned long long mask = 0ull;
  unsigned mx = 0, occ = 0;
  for (unsigned k = 0; k < 64u; ++k) {
    const unsigned x = 4u * bx + (k & 3u), y = 4u * by + ((k >> 2) & 3u), z = 4u * bz + (k >> 4);
    if (x >= g.dim[0] || y >= g.dim[1] || z >= g.dim[2])
      continue;
    const size_t cell = ((size_t)z * g.dim[1] + y) * g.dim[0] + x;
    const unsigned first = g.cellStart[cell], cnt = g.cellFill[cell];
    g.cellHdr[cell] = (first << 6) | (cnt < 63u ? cnt : 63u);
    if (cnt) {
      mask |= 1ull << k;
      ++occ;
    }
    mx = max(mx, cnt);
  }
  g.brickMask[b] = mask;
  if (mx)
    atomicMax(&g.stats[0], mx);
  if (occ)
    atomicAdd(&g.stats[1], occ);
}

hipError_t launch_grid_count(const GridParams &g, hipStream_t st) {
  hipLaunchKernelGGL(grid_count_kernel, dim3((g.n + 255) / 256), dim3(256), 0, st, g);
  return hipGetLastError();
}

hipError_t launch_grid_fill(const GridParams &g, hipStream_t st) {
  hipLaunchKernelGGL(grid_fill_kernel, dim3((g.n + 255) / 256), dim3(256), 0, st, g);
  const unsigned nb = g.bdim[0] * g.bdim[1] * g.bdim[2];
  hipLaunchKernelGGL(grid_finish_kernel, dim3((nb + 63) / 64), dim3(64), 0, st, g);
  return hipGetLastError();
}

} // namespace vr

// vr_trace_wf.hip — the staged instantiations of trace_kernel (vr_trace_kernel.hpp): the generation-by-generation
// path of the general kernel.  Generation 0 (sorted primaries) runs as FIRST: one segment per ray, survivors go
// to a queue.  Generations 1 .. G run as INTERSECT (closest hits of a queue's rays, few registers, many waves) +
// SHADE (the state machine on full wavefronts, survivors to the next queue); what is left after G generations
// runs to its end in RESUME.  A translation unit of its own so that the 4 x 12 extra kernels compile beside
// vr_trace.hip's.
#include "vr_trace_kernel.hpp"

namespace vr {

template <int D, int GEO, int PARTICLE>
static hipError_t launch_stage_t(const TraceParams &p, int stage, unsigned grid, hipStream_t s) {
  switch (stage) {
  case 1: hipLaunchKernelGGL((trace_kernel<D, GEO, PARTICLE, 0, 1>), dim3(grid), dim3(VR_BLOCK), 0, s, p); break;
  case 2: hipLaunchKernelGGL((trace_kernel<D, GEO, PARTICLE == P_EXT ? 0 : PARTICLE, 0, 2>), dim3(grid), dim3(VR_BLOCK), 0, s, p); break;
  case 3: hipLaunchKernelGGL((trace_kernel<D, GEO, PARTICLE, 0, 3>), dim3(grid), dim3(VR_BLOCK), 0, s, p); break;
  default: hipLaunchKernelGGL((trace_kernel<D, GEO, PARTICLE, 0, 4>), dim3(grid), dim3(VR_BLOCK), 0, s, p); break;
  }
  return hipGetLastError();
}

template <int D, int GEO, int PARTICLE> static int occ_stage_t(int stage) {
  int nb = 0;
  hipError_t e;
  switch (stage) {
  case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, GEO, PARTICLE, 0, 1>, VR_BLOCK, 0); break;
  case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, GEO, PARTICLE == P_EXT ? 0 : PARTICLE, 0, 2>, VR_BLOCK, 0); break;
  case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, GEO, PARTICLE, 0, 3>, VR_BLOCK, 0); break;
  default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<D, GEO, PARTICLE, 0, 4>, VR_BLOCK, 0); break;
  }
  return e == hipSuccess ? nb : 2;
}

template <class F> static auto dispatch_stage_variant(int D, int geo, int particle, F &&f) {
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

hipError_t launch_trace_stage(const TraceParams &p, int D, int geo, int particle, int stage, unsigned grid, hipStream_t s) {
  return dispatch_stage_variant(D, geo, particle, [&](auto d, auto g, auto pt) {
    return launch_stage_t<decltype(d)::value, decltype(g)::value, decltype(pt)::value>(p, stage, grid, s);
  });
}

int trace_stage_blocks_per_cu(int D, int geo, int particle, int stage) {
  return dispatch_stage_variant(D, geo, particle, [&](auto d, auto g, auto pt) {
    return occ_stage_t<decltype(d)::value, decltype(g)::value, decltype(pt)::value>(stage);
  });
}

} // namespace vr

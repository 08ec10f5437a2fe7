// vr_kernels.hpp — host-visible launchers of the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "vr_area.hpp"
#include "vr_types.hpp"

namespace vr {

hipError_t launch_gen(const TraceParams &p, int D, bool keepRng, unsigned maxBlocks, hipStream_t s);
hipError_t launch_scan(unsigned *data, unsigned n, unsigned *tmp, hipStream_t s);
hipError_t launch_trace(const TraceParams &p, int D, int geo, int particle, int mode, unsigned grid,
                        hipStream_t s);
// resident 256-thread blocks per CU of the trace kernel instantiation (occupancy API)
int trace_blocks_per_cu(int D, int geo, int particle, int mode, unsigned smallBytes);
hipError_t launch_debug_intersect(const TraceParams &p, int geo, const float *org, const float *dir,
                                  const float *tnear, unsigned n, int *geomID, unsigned *primID, float *t, int ordered,
                                  unsigned walkStackWaves, hipStream_t s);
hipError_t launch_debug_process_hit(const TraceParams &p, int D, const float *org, const float *dir, const float *tfar,
                                    const unsigned *prim, unsigned n, float *outOrg, float *outDir, int *outReflect,
                                    hipStream_t s);
hipError_t launch_debug_rng(unsigned seed32, unsigned count, unsigned long long *scratch, unsigned long long *out,
                            hipStream_t s);
// device-side setup (vr_setup.hip)
hipError_t launch_setup_bvh(const SetupParams &s, unsigned *scanTmp, hipStream_t st);
hipError_t launch_fit_bvh(const SetupParams &s, hipStream_t st);
hipError_t launch_bvh_check(const SetupParams &s, unsigned *bad, hipStream_t st);
hipError_t launch_smooth_flux(const float *fluxIn, float *fluxOut, const float *normal3, const uint32_t *nbOff,
                              const uint32_t *nbIds, const uint32_t *order, const uint32_t *leafOfOrig, unsigned n,
                              unsigned *overflow, hipStream_t st);
hipError_t launch_smooth_wide(const float *fluxIn, float *fluxOut, const float *normal3, const SetupParams &s, float dist,
                              unsigned *overflow, hipStream_t st);
hipError_t launch_quantize_nodes(const float *nodes, unsigned numNodes, const float *base3, const float *scale3,
                                 uint32_t *qnodes, uint32_t *pnodes, hipStream_t st);
hipError_t launch_setup_neighbors(const SetupParams &s, int pass, hipStream_t st);
hipError_t launch_gather_flux(const unsigned long long *acc, unsigned stride, unsigned replicas,
                              const unsigned *leafOfOrig, unsigned n, unsigned long long *outAcc, unsigned headroomBits,
                              unsigned long long *overflowFlag, hipStream_t s);

// 64-ary box tree for the packet query (vr_setup.hip)
size_t wide_tree_entries(unsigned n);
hipError_t launch_wide_tree(const SetupParams &s, unsigned *out3, hipStream_t st);
hipError_t launch_height_field(const HeightFieldParams &q, hipStream_t st);
hipError_t launch_relief_field(const ReliefParams &q, hipStream_t st);
hipError_t launch_disk4(const float *points3, unsigned n, float radius, int D, float *disk4, hipStream_t st);
// post-processing on the device (vr_setup.hip)
hipError_t launch_disk_areas(const float *disk4, const float *normal3, unsigned n, const AreaParams &p, float *out,
                             hipStream_t st);
hipError_t launch_flux_from_acc(const unsigned long long *acc, unsigned n, float *flux, hipStream_t st);
hipError_t launch_normalize_flux(float *flux, const float *area, unsigned n, int geo, int normType, float normFactor,
                                 double totalDiskArea, unsigned *maxOrd, hipStream_t st);
// issue-ceiling microbenchmarks (vr_bench.hip); out: one {cycles, realtime, sink} triple of u64 per wave
hipError_t launch_issue_kernel(int kind, unsigned blocks, unsigned iters, void *out, hipStream_t s);

} // namespace vr

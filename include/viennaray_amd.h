/* viennaray_amd.h — C ABI of the MI355X-native flux ray-tracing core.
 *
 * This is the drop-in boundary for ONE path of ViennaTools/ViennaRay: the work
 * done by Trace<T,D>::apply() (reference: include/viennaray/rayTrace.hpp:15-180,
 * rayTraceDisk.hpp:19-57, rayTraceTriangle.hpp:19-61 and the ray loop in
 * rayTraceKernel.hpp:32-426).  The reference has no FFI (header-only C++
 * templates); the entry points below are what a `viennaray::TraceDisk` /
 * `TraceTriangle` facade binds to (the headers under include/viennaray_amd/ are that facade,
 * INTEGRATION.md shows the binding).  Everything is `extern "C"`, plain pointers
 * and sizes; the library owns all HIP state.  NumericType on the device is
 * float (Embree is float internally: rayUtil.hpp:96-97).
 *
 * Ownership: every host pointer is read during the call and never retained
 * (the reference copies geometry into Embree buffers: rayGeometryDisk.hpp:123-176).
 * Status: every function returning int returns VR_OK (0) or a negative VR_E_*;
 * vr_last_error() gives the message.  HIP failures never abort the process.
 * Threading: one context per host thread (the reference's Trace objects are not
 * thread-safe either); a context binds one HIP device.
 */
#ifndef VIENNARAY_AMD_H
#define VIENNARAY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vr_context vr_context;

enum {
  VR_OK = 0,
  VR_E_INVALID = -1,  /* bad argument / missing geometry or particle      */
  VR_E_HIP = -2,      /* HIP runtime error (no device, OOM, launch failure) */
  VR_E_STATE = -3     /* call order (e.g. get_flux before apply)           */
};

/* rayBoundary.hpp:10-14 */
enum { VR_REFLECTIVE_BOUNDARY = 0, VR_PERIODIC_BOUNDARY = 1, VR_IGNORE_BOUNDARY = 2 };
/* rayUtil.hpp:40-47 */
enum { VR_POS_X = 0, VR_NEG_X = 1, VR_POS_Y = 2, VR_NEG_Y = 3, VR_POS_Z = 4, VR_NEG_Z = 5 };
/* rayUtil.hpp:38 */
enum { VR_NORM_SOURCE = 0, VR_NORM_MAX = 1 };
/* particle kinds of the device registry (viennaray_amd/csrc/vr_particles.hpp):
 * 0, 1     the reference's built-ins DiffuseParticle / SpecularParticle (rayParticle.hpp:126-204)
 * 2        CONED_COSINE: surfaceReflection = ReflectionConedCosine(coneAngle) (rayReflection.hpp:52-120),
 *          collision like SpecularParticle
 * 3        DIFFUSE_COSINE: a DiffuseParticle with TWO data labels: label 0 += w, label 1 += w * max(0, -d.n)
 * 4        COVERAGE_STICKING: a DiffuseParticle whose sticking falls with the coverage of the primitive it meets:
 *          sticking * (1 - globalData.vector(params[0])[primID]) — reads Trace::setGlobalData (vr_set_global_data)
 * A model is a struct of __device__ functions {sticking, reflect, collide} appended to the registry's type list;
 * its position is its id, its parameters travel in vr_particle::params. */
enum { VR_PARTICLE_DIFFUSE = 0, VR_PARTICLE_SPECULAR = 1, VR_PARTICLE_CONED_COSINE = 2, VR_PARTICLE_DIFFUSE_COSINE = 3,
       VR_PARTICLE_COVERAGE_STICKING = 4 };
/* geometry kinds (rayGeometry.hpp:9) */
enum { VR_GEOMETRY_DISK = 0, VR_GEOMETRY_TRIANGLE = 1 };

/* rayUtil.hpp:65-76 (+ raysTerminated, a local of TraceKernel::apply) */
typedef struct vr_trace_info {
  uint64_t numRays;
  uint64_t totalRaysTraced;
  uint64_t nonGeometryHits;
  uint64_t geometryHits;
  uint64_t particleHits;
  uint64_t boundaryHits;
  uint64_t reflections;
  uint64_t raysTerminated;
  double time;          /* seconds: BVH build + ray loop, like the reference (Q10) */
  double timeBuild;     /* seconds: BVH build part of `time`                       */
  double timeTrace;     /* seconds: device pipeline gen+sort+trace (HIP events)     */
  double timeTraceKernel; /* seconds: the trace kernel(s) alone (HIP events)        */
  int32_t warning;
  int32_t error;
  uint64_t rngFullStates; /* diagnostic: rays that drew more than 156 numbers and continued
                             on the full 312-word engine state       */
  double timeGenKernel;   /* seconds: the ray generator kernel(s) alone (HIP events)  */
  uint32_t bvhRefits;     /* diagnostic: 1 if the resident BVH failed its first consistency check and was
                             re-fitted with agent-scope fences (expected 0; non-zero is a finding, not a
                             state to live with: the tests and smoke() assert 0)               */
  uint32_t bvhBuilds;     /* scene builds this context has run so far (a re-apply on unchanged geometry adds none) */
} vr_trace_info;

/* gpu::Particle-style POD (rayParticle.hpp:208-218): a built-in particle.
 * DIFFUSE ignores sourcePower (rayParticle.hpp:158 returns 1).             */
typedef struct vr_particle {
  int32_t kind;            /* VR_PARTICLE_*                                   */
  float sticking;          /* stickingProbability_                            */
  float sourcePower;       /* cosine exponent n of the source (SPECULAR only) */
  int32_t numMaterialSticking;      /* 0 = none                               */
  const int32_t *materialIds;       /* [numMaterialSticking] material id ...  */
  const float *materialSticking;    /* ... -> sticking override               */
  float coneAngle;                  /* CONED_COSINE: maxConeAngle in radians  */
  float meanFreePath;               /* getMeanFreePath(); <= 0: no scattering (rayParticle.hpp:113) */
  float params[8];                  /* the model's own parameters (registry models beyond the built-ins)      */
} vr_particle;

/* ---- life cycle (Trace::Trace / ~Trace, rayTrace.hpp:17-29) ------------- */
int vr_create(vr_context **out, int device);
void vr_destroy(vr_context *ctx);
const char *vr_last_error(const vr_context *ctx);
/* static: 1 if a HIP device is usable, 0 otherwise (never throws)           */
int vr_device_available(void);
const char *vr_version(void);

/* ---- geometry ------------------------------------------------------------
 * TraceDisk::setGeometry(points, normals, gridDelta[, diskRadius])
 * (rayTraceDisk.hpp:62-92).  points/normals: n x 3 floats (for D==2 the z
 * column is ignored, rayGeometryDisk.hpp:148-176).  diskRadius <= 0 selects
 * gridDelta * DiskFactor<D> (rayUtil.hpp:99-101).                           */
int vr_set_disks(vr_context *ctx, const float *points, const float *normals,
                 uint32_t n, float gridDelta, float diskRadius, int D);
/* TraceTriangle::setGeometry(TriangleMesh) (rayTraceTriangle.hpp:71-74;
 * normals per rayMesh.hpp:99-112).  verts: nverts x 3, tris: ntris x 3.      */
int vr_set_triangles(vr_context *ctx, const float *verts, uint32_t nverts,
                     const uint32_t *tris, uint32_t ntris, float gridDelta, int D);
/* setMaterialIds (rayGeometry.hpp:17-24)                                    */
int vr_set_material_ids(vr_context *ctx, const int32_t *ids, uint32_t n);

/* ---- configuration (rayTrace.hpp:41-121) -------------------------------- */
int vr_set_boundary_conditions(vr_context *ctx, const int32_t *bcs, int n /* = D */);
int vr_set_source_direction(vr_context *ctx, int traceDirection);
int vr_set_primary_direction(vr_context *ctx, const float *dir3 /* NULL = off */);
int vr_set_particle(vr_context *ctx, const vr_particle *particle);
/* Several particles in ONE apply(), as the reference's gpu::Trace does (gpu/raygTrace.hpp:163-248: one launch per
 * particle, all with the seed of that apply): particle i's data labels follow particle i-1's in
 * vr_get_flux_data / vr_num_data; particles with the same source distribution share one generator pass (the rays
 * of ray index idx are the same for them).  runNumber advances once.                                         */
int vr_set_particles(vr_context *ctx, const vr_particle *particles, uint32_t n);
/* OPEN registration (the reference's GPU path registers user callables per particle at run time,
 * gpu/raygCallableConfig.hpp:7-18, gpu/raygTrace.hpp:163-248): `source` is HIP source text defining
 *     struct VrUserModel { static constexpr int kNumData; static constexpr bool kNeedsFull;
 *                          __device__ static float sticking(const ModelCtx &, unsigned primID, float base);
 *                          template <int D> __device__ static V3 reflect(const ModelCtx &, const V3 &rayDir, const V3 &n, Rng &, unsigned &);
 *                          template <class Credit> __device__ static void collide(const ModelCtx &, float w, const V3 &rayDir,
 *                                                                               const V3 &n, unsigned primID, Credit &&credit); };
 * (usually derived from a built-in model of viennaray_amd/csrc/vr_particles.hpp, overriding one member).  The library
 * compiles the extended trace kernels around it for gfx950 (hipcc --genco, cached by content under VR_CACHE_DIR or
 * /tmp) and loads the code object; *kind (>= VR_PARTICLE_USER_BASE) is then a valid vr_particle::kind of this context.
 * numData = VrUserModel::kNumData (1..4); flags: VR_MODEL_NEEDS_FULL = VrUserModel::kNeedsFull (the model is to be
 * combined with WDIST crediting / mean-free-path scattering, or brings heavy code of its own).  Needs hipcc at run time. */
enum { VR_PARTICLE_USER_BASE = 1000, VR_MODEL_NEEDS_FULL = 1 };
int vr_register_particle_model(vr_context *ctx, const char *name, const char *source, int numData, int flags, int32_t *kind);
/* Trace::setGlobalData (rayTrace.hpp:137-145; handed to every surfaceCollision / surfaceReflection,
 * rayParticle.hpp:21-81): vector `vecIdx` (indexed by primitive id) and the scalars of the caller's TracingData,
 * copied to HBM and readable by the device particle models.  data == NULL drops the vector and those behind it. */
int vr_set_global_data(vr_context *ctx, uint32_t vecIdx, const float *data, uint32_t n);
int vr_set_global_scalars(vr_context *ctx, const float *data, uint32_t n);
/* VIENNARAY_USE_WDIST (CMakeLists.txt:15, rayTraceKernel.hpp:258-296) as a run-time switch: a hit's
 * weight is shared among the credited disks by inverse impact distance                       */
int vr_set_use_wdist(vr_context *ctx, int on);
/* setSource(SourceGrid) (raySourceGrid.hpp): explicit origins, direction from the particle's cosine
 * power; numRays = n * numRaysPerPoint.  n == 0 = resetSource() (rayTrace.hpp:53-61)            */
int vr_set_source_grid(vr_context *ctx, const float *points3, uint32_t n);
/* setSource(any other Source) (raySource.hpp:10-19): the facade runs the callback on the host and
 * hands over ray idx -> origin, direction, engine outputs consumed.  n == 0 = resetSource()      */
int vr_set_host_rays(vr_context *ctx, const float *org3, const float *dir3, const uint32_t *draws, uint64_t n);
/* Source::getInitialRayWeight(idx) (raySource.hpp:18; rayTraceKernel.hpp:124,316-328,435-460) of the rays given
 * with vr_set_host_rays: start weight of ray idx and scale of the roulette's thresholds.  n == 0: all 1          */
int vr_set_host_ray_weights(vr_context *ctx, const float *weights, uint64_t n);
/* Source::getSourceArea() (raySource.hpp:17) for normalizeFlux(SOURCE) (rayTraceDisk.hpp:127,
 * rayTraceTriangle.hpp:113); area <= 0: SourceRandom's (raySourceRandom.hpp:40-47, the bbox source face)       */
int vr_set_source_area(vr_context *ctx, float area);
int vr_set_number_of_rays_per_point(vr_context *ctx, uint64_t n);
int vr_set_number_of_rays_fixed(vr_context *ctx, uint64_t n);
int vr_set_max_reflections(vr_context *ctx, uint32_t n);
int vr_set_max_boundary_hits(vr_context *ctx, uint32_t n);
int vr_set_rng_seed(vr_context *ctx, uint32_t seed);
int vr_set_use_random_seeds(vr_context *ctx, int useRandom);
/* KernelConfig::runNumber (rayUtil.hpp:93); apply() increments it           */
int vr_set_run_number(vr_context *ctx, uint32_t runNumber);
int vr_get_run_number(const vr_context *ctx, uint32_t *runNumber);

/* Multi-GPU sharding hook (not in the reference): trace only the global ray
 * indices [first, first+count).  count == 0 restores "all rays".  Ray idx
 * stays global, so the union over ranks reproduces the single-device stream
 * (rayTraceKernel.hpp:118-121).                                             */
int vr_set_ray_range(vr_context *ctx, uint64_t first, uint64_t count);
/* Number of ranks whose flux accumulators the caller is going to SUM (default 1; vr_apply_sharded sets it by itself).
 * The accumulators are int64 fixed point (weight * 2^VR_FLUX_FRAC_BITS): one primitive and data label holds
 * 2^23 = 8.39e6 weight units per apply(), divided by `world` rounded up to a power of two so that the signed sum over
 * the ranks cannot wrap either.  Beyond that vr_apply_finish / vr_apply FAILS (VR_E_STATE, TraceInfo.error = 1,
 * "flux accumulator overflow") instead of returning wrapped flux — the reference's float sums (rayTraceKernel.hpp:
 * 348-360, rayParticle.hpp:148-156) stall near 2^24 at that point.                                                  */
int vr_set_world_size(vr_context *ctx, uint32_t world);

/* Not in the reference (its ray loop allocates nothing per apply): apply() once per time step with a ray
 * count that follows the moving surface re-sizes the HBM ray stream; reserve it for the largest count
 * expected.  (Without a reservation the stream grows by half again when it must, and is kept.)             */
int vr_reserve_rays(vr_context *ctx, uint64_t n);

/* ---- run (Trace::apply) -------------------------------------------------- */
int vr_apply(vr_context *ctx);
/* Same, split for benchmarking: build = bbox/boundary/areas/BVH/uploads,
 * launch = zero accumulators + enqueue the trace kernel on the context's
 * stream (asynchronous), finish = wait + read counters.                     */
int vr_apply_prepare(vr_context *ctx);
int vr_apply_launch(vr_context *ctx);
int vr_apply_finish(vr_context *ctx);

/* Multi-GPU apply() (the reference has no distributed layer): one process per GPU,
 * geometry and BVH replicated, rank r traces the r-th contiguous share of the global ray indices
 * (ray idx keeps its global value, so the union reproduces the single-device stream,
 * rayTraceKernel.hpp:118-121), and the per-primitive int64 accumulators plus the seven counters
 * are summed over the ranks by `reduce` — an in-place sum all-reduce of `count` int64 in DEVICE
 * memory enqueued on `hipStream` (0 = ok).  vr_rccl_allreduce (viennaray_amd_rccl.h) is that
 * callback on RCCL over xGMI.  Afterwards every rank holds the full result of Trace::apply().   */
typedef int (*vr_allreduce_fn)(void *user, void *devInt64, size_t count, void *hipStream);
int vr_apply_sharded(vr_context *ctx, int rank, int world, vr_allreduce_fn reduce, void *user);

/* ---- results ------------------------------------------------------------- */
uint32_t vr_num_primitives(const vr_context *ctx);
/* getLocalData().getVectorData(0) (rayTrace.hpp:135): raw, un-normalised     */
int vr_get_flux(vr_context *ctx, float *out, uint32_t n);
int vr_get_flux_f64(vr_context *ctx, double *out, uint32_t n);
/* getLocalData().getVectorData(dataIdx) for particles with several data labels
 * (AbstractParticle::getLocalDataLabels, rayParticle.hpp:75-78; rayTraceDisk.hpp:40-47)      */
uint32_t vr_num_data(const vr_context *ctx);
int vr_get_flux_data(vr_context *ctx, uint32_t dataIdx, float *out, uint32_t n);
int vr_get_trace_info(const vr_context *ctx, vr_trace_info *out);
/* the counters of particle `particleIdx` of a multi-particle apply (vr_get_trace_info holds their sums)   */
int vr_get_particle_trace_info(const vr_context *ctx, uint32_t particleIdx, vr_trace_info *out);
/* which trace_kernel variant the last vr_apply_prepare selected: 0 general (reflection, roulette,
 * RNG), 1 absorbing + flat scene, 2 absorbing + structured scene, 3 general + flat scene,
 * 4 general, scene of a few hundred primitives resident in LDS             */
int vr_get_trace_mode(const vr_context *ctx, int32_t *mode);
/* normalizeFlux / smoothFlux (rayTraceDisk.hpp:103-193, rayTraceTriangle.hpp:92-136), in place on
 * a caller buffer; both run as HIP kernels on the resident areas / neighbourhood (upload, kernel,
 * download)                                                                                   */
int vr_normalize_flux(vr_context *ctx, float *flux, uint32_t n, int normType);
/* getLocalData().getVectorData(0) + normalizeFlux fused on the device: the raw flux never visits
 * the host (accumulators -> float -> flux * sourceArea / (numRays * area), one download)     */
int vr_get_flux_normalized(vr_context *ctx, float *out, uint32_t n, int normType);
int vr_smooth_flux(vr_context *ctx, float *flux, uint32_t n, int numNeighbors);
/* geometry-derived values the reference exposes to its tests                */
int vr_get_disk_areas(vr_context *ctx, float *out, uint32_t n);
int vr_get_bounding_box(vr_context *ctx, float *out6 /* min xyz, max xyz, adjusted */);
float vr_get_source_area(vr_context *ctx);
float vr_get_disk_radius(const vr_context *ctx);
int vr_get_neighbor_counts(vr_context *ctx, uint32_t *out, uint32_t n);

/* ---- device accumulators for collectives (multi-GPU) ----------------------
 * The per-primitive accumulator is an int64 fixed-point sum (weight * 2^VR_FLUX_FRAC_BITS),
 * so sums are order-independent and an integer all-reduce is exact.  The
 * pointer is DEVICE memory of `n` int64, in the caller's primitive order; it is
 * valid after vr_apply_finish() until the next launch.  vr_flux_accumulators_from
 * replaces the device contents (e.g. after an all-reduce done elsewhere).    */
#define VR_FLUX_FRAC_BITS 40
int vr_flux_accumulators(vr_context *ctx, void **devPtr, uint32_t *n);
/* Let the caller own the accumulator buffer instead (e.g. a torch int64 tensor
 * handed to torch.distributed/RCCL): DEVICE pointer to n int64; NULL restores
 * the library-owned buffer.  Must stay valid until replaced.  The binding survives
 * vr_set_particle(s) as long as numPrims x data labels is unchanged; a call that changes the
 * number of data labels, and vr_set_disks / vr_set_triangles, drop it (the library's own
 * buffer is used again: bind anew).                                          */
int vr_bind_flux_accumulators(vr_context *ctx, void *devPtr, uint32_t n);
int vr_add_trace_info(vr_context *ctx, const vr_trace_info *other);
/* stream the context launches on (hipStream_t as void*)                      */
void *vr_stream(vr_context *ctx);

/* ---- diagnostics used by the parity tests --------------------------------- */
/* closest hit of explicit rays against {boundary, geometry} (rtcIntersect1 stand-in,
 * rayTraceKernel.hpp:163-167).  geomID: 0 boundary, 1 geometry, -1 miss.     */
int vr_debug_intersect(vr_context *ctx, const float *org, const float *dir,
                       const float *tnear, uint32_t nrays, int32_t *geomID,
                       uint32_t *primID, float *t);
/* Boundary::processHit (rayBoundary.hpp:29-127) for hand-built hits, as the reference's
 * tests/boundaryHit and tests/boundaryHit2D feed it: ray (org, dir) meets wall triangle primID
 * (0..7) at tfar                                                                              */
int vr_debug_process_hit(vr_context *ctx, const float *org, const float *dir, const float *tfar,
                         const uint32_t *primID, uint32_t n, float *outOrg, float *outDir, int32_t *outReflect);
/* first (origin, direction) of global ray indices idx[] for kernel seed `seed`
 * (raySourceRandom.hpp:25-36 after rayTraceKernel.hpp:120-121)               */
int vr_debug_source_sample(vr_context *ctx, const uint64_t *idx, uint32_t n,
                           uint32_t seed, float *org, float *dir);
/* first `count` raw mt19937_64 outputs of the per-ray engine of ray idx      */
int vr_debug_rng_outputs(vr_context *ctx, uint64_t idx, uint32_t seed,
                         uint32_t count, uint64_t *out);
/* BVH statistics: nodes, leaves, max depth */
int vr_debug_bvh_stats(vr_context *ctx, uint32_t *out3);
/* consistency of the resident device-built BVH: number of internal nodes whose box is not
 * exactly the union of their children's or whose subtree size is inconsistent (expected 0)  */
int vr_debug_bvh_check(vr_context *ctx, uint32_t *violations);

/* Measurement aid (bench.py's roofline): the instruction-issue ceiling of the device for one of
 * the instruction mixes the hot kernels are made of (0 f32 VALU independent, 1 f32 VALU dependent
 * chain, 2 mt19937_64 seeding steps, 3 SALU, 4 packet-traversal VALU+SALU mix,
 * 5 independent VALU+SALU mix), at `wavesPerSimd`
 * resident waves per SIMD.  out4 = {counted instructions / s, sustained clock Hz, seconds, count} */
int vr_debug_issue_rate(vr_context *ctx, int kind, int wavesPerSimd, uint32_t iters, double *out4);

#ifdef __cplusplus
}
#endif
#endif /* VIENNARAY_AMD_H */

// vr_api.cpp — the C ABI (include/viennaray_amd.h) on top of the host setup
// (vr_host.cpp) and the HIP kernels (vr_trace.hip).
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/viennaray_amd.h"
#include "vr_host.hpp"
#include "vr_kernels.hpp"
#include "vr_particles.hpp"
#include "vr_types.hpp"

using namespace vr;

namespace {
template <class T> struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap && p)
      return hipSuccess;
    if (p)
      (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T));
    if (e == hipSuccess)
      cap = std::max<size_t>(n, 1);
    return e;
  }
  // buffers whose size follows the ray count of an apply(): grown by half again, so a simulation whose
  // ray count creeps up from step to step re-allocates O(log) times, not every step (hipMalloc of a
  // multi-GB ray stream costs tens of ms)
  hipError_t ensure_grow(size_t n) {
    if (n <= cap && p)
      return hipSuccess;
    const size_t want = std::max(n, cap + cap / 2);
    hipError_t e = ensure(want);
    if (e != hipSuccess && want > n) { // (no room for the head-room: the exact size)
      (void)hipGetLastError();
      e = ensure(n);
    }
    return e;
  }
  void release() {
    if (p)
      (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); } // (vr_destroy selects the device before the context goes away)
};
} // namespace

// one entry of vr_set_particles (a deep copy of the caller's vr_particle)
struct ParticleSpec {
  int kind = 0;
  float sticking = 1.f, sourcePower = 1.f, coneAngle = 0.f, meanFreePath = -1.f;
  float params[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int userModel = -1; // kind >= VR_PARTICLE_USER_BASE: index of the run-time model
  std::vector<int32_t> matIds;
  std::vector<float> matVals;
};
// a particle model registered at run time (vr_register_particle_model): its own code object with the extended trace
// kernels, the model compiled in as entry VR_BUILTIN_MODELS of that module's registry
struct UserModel {
  std::string name;
  hipModule_t module = nullptr;
  int numData = 1;
  bool needsFull = false;
  std::map<int, hipFunction_t> kernels; // key: D * 100 + geo * 10 + mode
};

// the prepared launch of one particle of a multi-particle apply()
struct ParticleLaunch {
  TraceParams params{};
  unsigned grid = 0;
  int traceMode = 0, kernelParticle = 0;
  bool absorb = false;
  uint32_t numData = 1, dataBase = 0;
  hipFunction_t userKernel = nullptr; // the trace kernel of a run-time model (nullptr: a kernel of the library)
  float *primSticking = nullptr; // owned (hipMalloc): this particle's per-primitive sticking, leaf order
  bool relief = false;           // a second launch (looseMode) traces the loose bins
  int looseMode = 0;
  unsigned looseGrid = 0;
  vr_trace_info info{};
};

struct vr_context {
  int device = 0;
  int numCUs = 256;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;

  HostGeometry geo;
  bool geometryDirty = true; // BVH / uploads need rebuilding
  bool configDirty = true;   // bbox / walls / areas / sticking map need recomputing
  Bvh bvh;
  std::vector<uint32_t> leafOfOrig;
  std::vector<float> diskAreas;       // host mirror of dAreas (disks), downloaded on demand
  bool diskAreasHostValid = false;

  // Trace<T,D> configuration (rayTrace.hpp:157-179, rayUtil.hpp:83-94)
  int bcs[3] = {0, 0, 0};
  int sourceDirection = -1; // -1: default by D (POS_Y for 2-D, POS_Z for 3-D)
  bool usePrimaryDirection = false;
  float primaryDirection[3] = {0, 0, 0};
  bool haveParticle = false;
  int particleKind = 0;
  float coneAngle = 0.f, meanFreePath = -1.f;
  bool useWdist = false;
  uint32_t numData = 1;           // data labels of the (active) particle
  uint32_t totalData = 1;         // ... of all particles of the apply: accumulator planes, TracingData vectors
  uint32_t dataBase = 0;          // first plane of the active particle
  uint32_t counterSlot = 0;       // ... and its block of 80 counter words
  uint32_t accPlanes = 0;         // planes the accumulator buffers currently hold
  float particleParams[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int userModel = -1;             // index into userModels when the active particle is a run-time model
  hipFunction_t userKernel = nullptr; // ... and the kernel vr_apply_prepare picked from its module
  std::vector<UserModel> userModels;
  bool particleDirty = true;      // the sticking map needs recomputing
  std::vector<ParticleSpec> specs;      // vr_set_particles: > 1 entries = a multi-particle apply
  std::vector<ParticleLaunch> launches; // prepared by vr_apply_prepare when specs.size() > 1
  // Trace::setGlobalData: vectors (padded to one stride) and scalars, resident in HBM
  std::vector<std::vector<float>> globalVecs;
  std::vector<float> globalScalars;
  bool globalDirty = false;
  uint32_t globalStride = 0;
  DevBuf<float> dGlobalVec, dGlobalScalars;
  // sources other than SourceRandom
  std::vector<float> gridPoints;  // SourceGrid origins (raySourceGrid.hpp)
  std::vector<float> hostOrg, hostDir;
  std::vector<uint32_t> hostDraws;
  std::vector<float> hostWeights;  // Source::getInitialRayWeight(idx) of the host rays (empty: 1)
  float sourceAreaOverride = 0.f;  // Source::getSourceArea() of a user source (<= 0: SourceRandom's, the bbox face)
  uint64_t reserveRays = 0;        // vr_reserve_rays: the ray-stream buffers hold at least this many rays
  bool sourceDirty = false;
  DevBuf<float> dGrid, dHostOrg, dHostDir, dHostWeights;
  DevBuf<uint32_t> dHostDraws;
  float sticking = 1.f, sourcePower = 1.f;
  std::vector<int32_t> matStickIds;
  std::vector<float> matStickVals;
  uint64_t numRaysPerPoint = 1000, numRaysFixed = 0;
  uint32_t maxReflections = 0xFFFFFFFFu, maxBoundaryHits = 1000;
  uint32_t rngSeed = 0;
  bool useRandomSeed = true;
  uint32_t runNumber = 1;
  uint64_t rayFirst = 0, rayCount = 0;
  bool haveSharedSeed = false; // vr_apply_sharded + useRandomSeed: rank 0's draw, handed round by the all-reduce;
                               // a multi-particle apply with random seeds: its one draw
  bool keepSharedSeed = false; // (the sharded entry point clears the seed itself)
  size_t numGenLaunches = 0, numTraceLaunches = 0;
  uint32_t sharedSeed = 0;

  // derived at prepare()
  float bbLo[3], bbHi[3];
  std::array<int, 5> ts{};
  int boundaryConds[2] = {0, 0};
  float sourceArea = 0.f;
  uint64_t numRaysLast = 0;
  bool prepared = false, launched = false, haveResult = false;
  TraceParams params{};
  unsigned grid = 0;

  vr_trace_info info{};
  double buildSeconds = 0.0;

  // device buffers
  DevBuf<float> dNodes, dPrims, dPrimSticking;
  DevBuf<float> dAreas, dFluxTmp;     // exposed area per primitive (caller's order); normalisation scratch
  DevBuf<uint32_t> dNormMax;          // flux_max_kernel's reduction word
  bool areasValid = false;
  DevBuf<uint32_t> dNbOff, dNbIds, dLeafOfOrig;
  DevBuf<uint32_t> dNbTmp; // the one-pass neighbourhood query's fixed-stride lists (build scratch)
  uint32_t nbTotal = 0;               // entries of the resident neighbourhood CSR
  DevBuf<unsigned long long> dFluxAcc, dFluxOrig, dCounters, dScratch;
  DevBuf<unsigned long long> dWorkQ;  // span cursors of the trace kernel's per-XCD queues
  size_t scratchWaves = 0;
  // flux accumulators are replicated accReplicas times (power of two, stride accStride
  // elements); a block credits replica blockIdx & (accReplicas-1): small scenes would
  // otherwise serialise every credit of the chip on a handful of cache lines
  uint32_t accReplicas = 1, accStride = 0;
  // device-side setup (vr_setup.hip)
  DevBuf<float> dDisk4, dNormal3, dPoints3, dVerts, dBox, dSBox, dNodeBox;
  DevBuf<uint32_t> dTris, dBounds, dValsA, dValsB, dSortTable, dRangeLo, dRangeHi, dChildL, dChildR, dParentInt,
      dParentLeaf, dArrive, dOrder, dSubSize, dQNodes, dPNodes, dWalkStack;
  size_t walkStackWaves = 0;
  DevBuf<float> dNodesPre, dWide;
  uint32_t wideRoot[3] = {0, 0, 0};  // 64-ary tree: root's first child, count | flag, primitive base
  bool haveWide = false;
  float sceneLo[3] = {0, 0, 0}, sceneHi[3] = {0, 0, 0};
  uint32_t numNodes = 0;         // traversal nodes emitted by the builder
  float qbase[3] = {0, 0, 0}, qscale[3] = {0, 0, 0}; // frame of the 16-byte nodes
  float keyCoord = 0.f;          // sort plane of the ray stream on the tracing axis (host_sort_plane)
  float keyShare = 1.f;          // share of the surface shown to the source that lies in that plane
  int traceMode = 0;             // trace_kernel MODE of the prepared launch
  int kernelParticle = 0;        // PARTICLE template id of the prepared launch (P_EXT: extended kernel)
  SetupParams lastSetup{};       // buffers of the resident device build (vr_debug_bvh_check)
  bool haveSetup = false;
  int builtOrderAxis = -1;       // child order of the resident BVH (source side first)
  int bvhRefits = 0;             // 1 if the last build had to be re-fitted with agent-scope fences
  uint32_t bvhBuilds = 0;        // scene builds of this context
  float builtOrderSign = 0.f;
  DevBuf<unsigned long long> dKeysA, dKeysB;
  bool hostOrderValid = false;   // c->bvh.order mirrors dOrder
  bool hostNeighborsValid = false;
  // ray stream (one batch)
  DevBuf<float> dSlotRec, dWalls;
  DevBuf<uint32_t> dBinCount;
  size_t slotStride = 0; // record slots of the ray-stream buffer (bins + overflow region)
  uint32_t raysPerBin = 40;
  DevBuf<uint32_t> dScanTmp;
  uint32_t batchCap = 0;      // rays per batch the buffers hold
  uint32_t numBins = 0;
  uint64_t rayFirstLaunch = 0, rayEndLaunch = 0;
  bool absorb = true;
  float wallsHost[96] = {0};        // the eight wall triangles (made with the bounding box)
  std::vector<float> frameHostAll;  // wall table + scalar frame of every particle of the apply (staging of their uploads)
  // relief field over the source plane (ReliefParams): scenes that are flat with relief
  DevBuf<uint32_t> dRfRawLo, dRfRawHi, dRfStats;
  DevBuf<float> dRfFine, dRfCoarse;
  ReliefParams rf{};
  uint32_t rfBuild = 0xFFFFFFFFu;
  int rfAxes[4] = {-1, -1, -1, -1};
  float rfLooseShare = 1.f;
  DevBuf<float> dSpillRec;     // the general relief kernel's spill queue (TraceParams::spillRec), 16 floats per ray of a batch
  DevBuf<uint32_t> dSpillCount;
  bool reliefScene = false;    // the prepared launch: MODE 5 / 6 over the tight bins + looseMode over the loose ones
  int looseMode = 0;
  unsigned looseGrid = 0;
  DevBuf<uint32_t> dHfRaw;    // height field over the source plane (HeightFieldParams): built for particles that reflect
  DevBuf<float> dHf;
  HeightFieldParams hf{};
  uint32_t hfBuild = 0xFFFFFFFFu; // the bvhBuilds count and source frame it was made for
  int hfAxes[4] = {-1, -1, -1, -1};
  bool recExtra = false;      // non-absorbing particle under a tilted / grid / host source: the records' side array
  DevBuf<float> dRecExtra;
  std::vector<hipEvent_t> evK; // trace-kernel event pairs, one per batch
  std::vector<hipEvent_t> evG; // generator event pairs, one per batch
  double traceKernelSeconds = 0.0;
  bool havePrimSticking = false;
  uint32_t worldSize = 1;     // ranks whose accumulators will be summed (vr_set_world_size): head-room of the overflow check
  unsigned long long *boundFlux = nullptr; // caller-owned accumulator buffer
  uint32_t boundFluxN = 0;
  unsigned long long *fluxOut() { return boundFlux ? boundFlux : dFluxOrig.p; }
};

#define VR_HIP(ctx, call)                                                                                              \
  do {                                                                                                                 \
    hipError_t e__ = (call);                                                                                           \
    if (e__ != hipSuccess) {                                                                                           \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                                                 \
      return VR_E_HIP;                                                                                                 \
    }                                                                                                                  \
  } while (0)

static int fail(vr_context *c, int code, const char *msg) {
  if (c)
    c->err = msg;
  return code;
}

static void size_bins(int D, uint64_t count, uint32_t perBin, TraceParams &p, uint32_t &numBins);
static void size_loose(int D, TraceParams &p);

extern "C" {

const char *vr_version(void) { return "viennaray_amd 0.1 (gfx950)"; }

int vr_device_available(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  return (e == hipSuccess && n > 0) ? 1 : 0;
}

int vr_create(vr_context **out, int device) {
  if (!out)
    return VR_E_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
    return VR_E_HIP;
  vr_context *c = new vr_context();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    delete c;
    return VR_E_HIP;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess)
    c->numCUs = prop.multiProcessorCount;
  *out = c;
  return VR_OK;
}

void vr_destroy(vr_context *c) {
  if (!c)
    return;
  (void)hipSetDevice(c->device);
  if (c->stream)
    (void)hipStreamSynchronize(c->stream);
  for (auto &L : c->launches)
    if (L.primSticking)
      (void)hipFree(L.primSticking);
  for (auto &um : c->userModels)
    if (um.module)
      (void)hipModuleUnload(um.module);
  c->dNodes.release();
  c->dPrims.release();
  c->dPrimSticking.release();
  c->dNbOff.release();
  c->dNbTmp.release();
  c->dNbIds.release();
  c->dHfRaw.release();
  c->dHf.release();
  c->dLeafOfOrig.release();
  c->dFluxAcc.release();
  c->dFluxOrig.release();
  c->dCounters.release();
  c->dScratch.release();
  c->dWalls.release();
  c->dSlotRec.release();
  c->dBinCount.release();
  c->dScanTmp.release();
  for (auto e : c->evK)
    (void)hipEventDestroy(e);
  for (auto e : c->evG)
    (void)hipEventDestroy(e);
  if (c->ev0)
    (void)hipEventDestroy(c->ev0);
  if (c->ev1)
    (void)hipEventDestroy(c->ev1);
  if (c->stream)
    (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *vr_last_error(const vr_context *c) { return c ? c->err.c_str() : "null context"; }

// ---- geometry ---------------------------------------------------------------
int vr_set_disks(vr_context *c, const float *points, const float *normals, uint32_t n, float gridDelta,
                 float diskRadius, int D) {
  if (!c || !points || !normals || (D != 2 && D != 3) || n >= (1u << 27))
    return fail(c, VR_E_INVALID, "vr_set_disks: bad argument");
  host_set_disks(c->geo, points, normals, n, gridDelta, diskRadius, D);
  c->hostNeighborsValid = false;
  c->areasValid = false;
  c->boundFlux = nullptr;
  c->geometryDirty = true;
  c->configDirty = true;
  c->prepared = c->haveResult = false;
  return VR_OK;
}

int vr_set_triangles(vr_context *c, const float *verts, uint32_t nverts, const uint32_t *tris, uint32_t ntris,
                     float gridDelta, int D) {
  if (!c || !verts || !tris || (D != 2 && D != 3) || ntris >= (1u << 27))
    return fail(c, VR_E_INVALID, "vr_set_triangles: bad argument");
  for (size_t i = 0; i < (size_t)ntris * 3; ++i)
    if (tris[i] >= nverts)
      return fail(c, VR_E_INVALID, "vr_set_triangles: vertex index out of range");
  host_set_triangles(c->geo, verts, nverts, tris, ntris, gridDelta, D);
  c->areasValid = false;
  c->boundFlux = nullptr;
  c->geometryDirty = true;
  c->configDirty = true;
  c->prepared = c->haveResult = false;
  return VR_OK;
}

int vr_set_material_ids(vr_context *c, const int32_t *ids, uint32_t n) {
  if (!c || !ids)
    return fail(c, VR_E_INVALID, "vr_set_material_ids: bad argument");
  c->geo.materialIds.assign(ids, ids + n);
  c->prepared = false;
  c->configDirty = true;
  return VR_OK;
}

// ---- configuration ----------------------------------------------------------
int vr_set_boundary_conditions(vr_context *c, const int32_t *bcs, int n) {
  if (!c || !bcs || n < 1 || n > 3)
    return fail(c, VR_E_INVALID, "vr_set_boundary_conditions: bad argument");
  for (int i = 0; i < n; ++i) {
    if (bcs[i] < 0 || bcs[i] > 2)
      return fail(c, VR_E_INVALID, "vr_set_boundary_conditions: unknown condition");
    c->bcs[i] = bcs[i];
  }
  c->prepared = false;
  c->configDirty = true;
  return VR_OK;
}
int vr_set_source_direction(vr_context *c, int d) {
  if (!c || d < 0 || d > 5)
    return fail(c, VR_E_INVALID, "vr_set_source_direction: bad argument");
  c->sourceDirection = d;
  c->prepared = false;
  c->configDirty = true;
  return VR_OK;
}
int vr_set_primary_direction(vr_context *c, const float *d) {
  if (!c)
    return VR_E_INVALID;
  if (d) {
    std::memcpy(c->primaryDirection, d, 12);
    c->usePrimaryDirection = true;
  } else {
    c->usePrimaryDirection = false;
  }
  c->prepared = false;
  c->configDirty = true;
  return VR_OK;
}
static bool spec_from_pod(const vr_context *c, const vr_particle *p, ParticleSpec &sp) {
  if (!p || p->kind < 0)
    return false;
  const bool user = p->kind >= VR_PARTICLE_USER_BASE;
  if (user ? (size_t)(p->kind - VR_PARTICLE_USER_BASE) >= c->userModels.size() : p->kind >= Particles::count)
    return false;
  sp = ParticleSpec{};
  sp.kind = p->kind;
  sp.userModel = user ? p->kind - VR_PARTICLE_USER_BASE : -1;
  sp.sticking = p->sticking;
  // rayParticle.hpp:158,199: only SpecularParticle-like particles carry a source power of their own
  sp.sourcePower = (p->kind == VR_PARTICLE_DIFFUSE || p->kind == VR_PARTICLE_DIFFUSE_COSINE ||
                    p->kind == VR_PARTICLE_COVERAGE_STICKING)
                       ? 1.f
                       : p->sourcePower;
  sp.coneAngle = p->coneAngle;
  sp.meanFreePath = p->meanFreePath;
  std::memcpy(sp.params, p->params, sizeof(sp.params));
  if (p->kind == VR_PARTICLE_CONED_COSINE)
    sp.params[0] = p->coneAngle; // (the model reads its cone angle from params[0])
  if (p->numMaterialSticking > 0 && p->materialIds && p->materialSticking) {
    sp.matIds.assign(p->materialIds, p->materialIds + p->numMaterialSticking);
    sp.matVals.assign(p->materialSticking, p->materialSticking + p->numMaterialSticking);
  }
  return true;
}

// make `sp` the particle the next prepare works on
static void activate_particle(vr_context *c, const ParticleSpec &sp) {
  c->userModel = sp.userModel;
  // (inside its own code object a run-time model is the registry's last entry)
  c->particleKind = sp.userModel >= 0 ? VR_BUILTIN_MODELS : sp.kind;
  c->sticking = sp.sticking;
  c->sourcePower = sp.sourcePower;
  c->coneAngle = sp.coneAngle;
  c->meanFreePath = sp.meanFreePath;
  std::memcpy(c->particleParams, sp.params, sizeof(sp.params));
  c->numData = sp.userModel >= 0 ? (uint32_t)c->userModels[sp.userModel].numData : (uint32_t)Particles::numData(sp.kind);
  c->matStickIds = sp.matIds;
  c->matStickVals = sp.matVals;
  c->particleDirty = true;
}

int vr_set_particles(vr_context *c, const vr_particle *list, uint32_t n) {
  if (!c || !list || n == 0 || n > 64)
    return fail(c, VR_E_INVALID, "vr_set_particles: between 1 and 64 particles");
  std::vector<ParticleSpec> specs(n);
  uint32_t total = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (!spec_from_pod(c, &list[i], specs[i]))
      return fail(c, VR_E_INVALID, "vr_set_particle: unknown particle kind (not in the device registry, not registered)");
    total += specs[i].userModel >= 0 ? (uint32_t)c->userModels[specs[i].userModel].numData
                                     : (uint32_t)Particles::numData(specs[i].kind);
  }
  c->specs = std::move(specs);
  activate_particle(c, c->specs[0]);
  c->totalData = total;
  c->dataBase = 0;
  c->counterSlot = 0;
  // a caller's accumulator buffer (vr_bind_flux_accumulators) stays bound while it still has the right size: numPrims x
  // data labels of ALL particles; it is dropped only when the number of planes changed
  if (c->boundFlux && c->boundFluxN != c->geo.numPrims * total) {
    c->boundFlux = nullptr;
    c->boundFluxN = 0;
  }
  c->haveParticle = true;
  c->prepared = false;
  return VR_OK;
}

int vr_set_particle(vr_context *c, const vr_particle *p) { return vr_set_particles(c, p, 1); }

// ---- run-time particle models -----------------------------------------------------------------------------------
static uint64_t fnv1a(uint64_t h, const void *data, size_t n) {
  const unsigned char *b = (const unsigned char *)data;
  for (size_t i = 0; i < n; ++i) {
    h ^= b[i];
    h *= 1099511628211ull;
  }
  return h;
}

static bool slurp(const std::string &path, std::string &out) {
  std::ifstream f(path, std::ios::binary);
  if (!f)
    return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return true;
}

// the kernel sources the library was built from: next to the library (in-tree layout viennaray_amd/csrc), or VR_CSRC_DIR
static std::string csrc_dir() {
  if (const char *e = std::getenv("VR_CSRC_DIR"))
    return e;
  Dl_info info;
  if (dladdr((const void *)&vr_version, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t k = p.find_last_of('/');
    return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/csrc";
  }
  return "csrc";
}

// POSIX cksum (CRC-32, polynomial 0x04C11DB7, the length appended) of a file's bytes: what the Makefile records of every
// kernel source at build time (VR_SRC_CKSUM) — a run-time model must be compiled from THOSE sources: its kernels take the
// library's TraceParams by value.
static uint32_t posix_cksum(const std::string &data) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i << 24;
      for (int k = 0; k < 8; ++k)
        c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : (c << 1);
      table[i] = c;
    }
    init = true;
  }
  uint32_t crc = 0;
  for (unsigned char b : data)
    crc = (crc << 8) ^ table[((crc >> 24) ^ b) & 0xFFu];
  for (size_t n = data.size(); n; n >>= 8)
    crc = (crc << 8) ^ table[((crc >> 24) ^ (n & 0xFFu)) & 0xFFu];
  return ~crc;
}

// the compiler's identity (`hipcc --version`, first lines), part of the cache key: a code object does not survive a
// toolchain upgrade
static std::string compiler_identity(const std::string &hipcc) {
  std::string out;
  if (hipcc.find('\'') != std::string::npos)
    return out;
  if (FILE *f = popen(("'" + hipcc + "' --version 2>/dev/null").c_str(), "r")) {
    char buf[256];
    while (out.size() < 2048 && std::fgets(buf, sizeof(buf), f))
      out += buf;
    pclose(f);
  }
  return out;
}

// The reference's GPU path registers user callables per particle at run time (gpu/raygCallableConfig.hpp:7-18: OptiX
// direct callables named in the particle).  Here the caller hands over the SOURCE of a model — `struct VrUserModel` with
// the registry's shape (vr_particles.hpp: sticking / reflect / collide, kNumData, kNeedsFull), usually a few lines on top of
// one of the built-in models — and the library compiles the extended trace kernels around it for gfx950 (hipcc --genco,
// cached by content) and loads them.  The returned kind goes into vr_particle::kind like a built-in one.
int vr_register_particle_model(vr_context *c, const char *name, const char *source, int numData, int flags, int32_t *kindOut) {
  if (!c || !source || !kindOut || numData < 1 || numData > VR_MAX_LABELS)
    return fail(c, VR_E_INVALID, "vr_register_particle_model: bad argument (1 .. 4 data labels)");
  VR_HIP(c, hipSetDevice(c->device));
  const std::string csrc = csrc_dir();
  const bool full = (flags & VR_MODEL_NEEDS_FULL) != 0;
  const std::string hipcc = std::getenv("VR_HIPCC") ? std::getenv("VR_HIPCC") : "/opt/rocm/bin/hipcc";
  const std::string ccFlags = " --genco --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function";
  uint64_t h = 1469598103934665603ull;
  h = fnv1a(h, source, std::strlen(source));
  h = fnv1a(h, &numData, sizeof(numData));
  h = fnv1a(h, &full, sizeof(full));
  std::string cksums;
  for (const char *fn : {"vr_trace.hip", "vr_device.hpp", "vr_particles.hpp", "vr_types.hpp", "vr_libm.hpp", "vr_kernels.hpp"}) {
    std::string text;
    if (!slurp(csrc + "/" + fn, text))
      return fail(c, VR_E_STATE, ("vr_register_particle_model: kernel source not found: " + csrc + "/" + fn +
                                  " (the sources ship next to the library; VR_CSRC_DIR overrides)").c_str());
    h = fnv1a(h, text.data(), text.size());
    cksums += std::to_string(posix_cksum(text)) + "-";
  }
#ifdef VR_SRC_CKSUM
  // The kernels of the module take the library's TraceParams by value and read the LDS frame the host fills: sources that
  // are not the ones this library was built from (an edited checkout without a rebuild, a wrong VR_CSRC_DIR, an installed
  // library next to a newer tree) would end in a GPU memory fault, not in an error code.  Refused here.
  if (cksums != VR_SRC_CKSUM)
    return fail(c, VR_E_STATE, ("vr_register_particle_model: the kernel sources in " + csrc + " are not the ones this library was built "
                                "from (checksums " + cksums + " against " VR_SRC_CKSUM "): rebuild the library, or point VR_CSRC_DIR at its sources").c_str());
#endif
  { // (a code object does not survive a change of the compiler or of its flags)
    const std::string id = compiler_identity(hipcc) + ccFlags;
    h = fnv1a(h, id.data(), id.size());
  }
  // The cache holds code that hipModuleLoad will run: a directory of the caller's own, mode 0700, and checked — a
  // predictable name under /tmp that another local user created first could hold a planted code object.
  std::string cache;
  if (const char *e = std::getenv("VR_CACHE_DIR"))
    cache = e;
  else if (const char *x = std::getenv("XDG_CACHE_HOME"); x && *x)
    cache = std::string(x) + "/viennaray_amd";
  else if (const char *hm = std::getenv("HOME"); hm && *hm && std::string(hm) != "/") {
    (void)mkdir((std::string(hm) + "/.cache").c_str(), 0700);
    cache = std::string(hm) + "/.cache/viennaray_amd";
  } else
    cache = "/tmp/viennaray_amd_cache_" + std::to_string((unsigned)getuid());
  (void)mkdir(cache.c_str(), 0700);
  {
    struct stat ds;
    if (lstat(cache.c_str(), &ds) != 0 || !S_ISDIR(ds.st_mode) || ds.st_uid != getuid() || (ds.st_mode & 077) != 0)
      return fail(c, VR_E_STATE, ("vr_register_particle_model: the code-object cache " + cache + " must be a directory (no symbolic link) "
                                  "owned by this user with mode 0700 - refused; set VR_CACHE_DIR to a private directory").c_str());
  }
  char hex[32];
  std::snprintf(hex, sizeof(hex), "%016llx", (unsigned long long)h);
  const std::string base = cache + "/model_" + hex, hsaco = base + ".hsaco";
  struct stat st;
  if (stat(hsaco.c_str(), &st) != 0 || st.st_size == 0) {
    // every file of this compilation under a name of this process's own (several ranks register the same model on a cold
    // cache at once); the code object then moves into place atomically
    const std::string mine = base + ".p" + std::to_string((int)getpid());
    const std::string tmp = mine + ".hsaco";
    { std::ofstream f(mine + "_model.hpp"); f << source << "\n"; }
    {
      std::ofstream f(mine + ".hip");
      f << "// generated by vr_register_particle_model\n#define VR_USER_MODULE 1\n#define VR_USER_NUM_DATA " << numData
        << "\n#define VR_USER_MODEL_FILE \"" << mine << "_model.hpp\"\n#include <cstddef>\n#include \"" << csrc << "/vr_trace.hip\"\n"
        << "static_assert(vr::VrUserModel::kNeedsFull == " << (full ? "true" : "false")
        << ", \"kNeedsFull differs from the VR_MODEL_NEEDS_FULL flag given at registration\");\n"
        // the launch parameters and the LDS frame as THIS library lays them out
        << "static_assert(sizeof(vr::TraceParams) == " << sizeof(TraceParams) << " && offsetof(vr::TraceParams, globalVec) == "
        << offsetof(TraceParams, globalVec) << " && offsetof(vr::TraceParams, counters) == " << offsetof(TraceParams, counters)
        << " && offsetof(vr::TraceParams, pqMargin) == " << offsetof(TraceParams, pqMargin) << " && vr::VR_WALL_TABLE == "
        << VR_WALL_TABLE << ", \"vr::TraceParams / the launch frame differ from the loaded library's: these kernel sources are not its own\");\n";
    }
    auto quoted = [](const std::string &path) { return "'" + path + "'"; }; // (paths with blanks; a quote in a path is refused below)
    if ((cache + csrc + hipcc).find('\'') != std::string::npos)
      return fail(c, VR_E_INVALID, "vr_register_particle_model: the cache / source directory must not contain a quote character");
    const std::string cmd = quoted(hipcc) + ccFlags + " -I" + quoted(csrc) + " " + quoted(mine + ".hip") + " -o " + quoted(tmp) + " > " +
                            quoted(mine + ".log") + " 2>&1";
    const int rc = std::system(cmd.c_str());
    (void)unlink((mine + ".hip").c_str());
    (void)unlink((mine + "_model.hpp").c_str());
    if (rc != 0) {
      std::string all, log;
      (void)slurp(mine + ".log", all);
      (void)std::rename((mine + ".log").c_str(), (base + ".log").c_str()); // (kept for the caller to read)
      { // the compiler's error lines (and the source line under each), not the tail of its output
        std::istringstream in(all);
        std::string line;
        int keep = 0;
        while (std::getline(in, line) && log.size() < 1500) {
          if (line.find("error") != std::string::npos)
            keep = 3;
          if (keep-- > 0)
            log += line + "\n";
        }
        if (log.empty())
          log = all.size() > 1500 ? all.substr(all.size() - 1500) : all;
      }
      (void)unlink(tmp.c_str());
      return fail(c, VR_E_INVALID, ("vr_register_particle_model: the model did not compile (" + base + ".log):\n" + log).c_str());
    }
    (void)unlink((mine + ".log").c_str());
    if (std::rename(tmp.c_str(), hsaco.c_str()) != 0)
      return fail(c, VR_E_STATE, "vr_register_particle_model: cannot write the code object cache");
  }
  UserModel um;
  um.name = name ? name : "";
  um.numData = numData;
  um.needsFull = full;
  VR_HIP(c, hipModuleLoad(&um.module, hsaco.c_str()));
  const int P = full ? (int)P_EXT_FULL : (int)P_EXT;
  for (int D = 2; D <= 3; ++D)
    for (int geo = 0; geo <= 1; ++geo)
      for (int mode : {0, 3, 4}) {
        if (mode == 3 && (geo != 0 || full))
          continue;
        char sym[128];
        std::snprintf(sym, sizeof(sym), "_ZN2vr12trace_kernelILi%dELi%dELi%dELi%dEEEvNS_11TraceParamsE", D, geo, P, mode);
        hipFunction_t f = nullptr;
        if (hipModuleGetFunction(&f, um.module, sym) != hipSuccess || !f) {
          (void)hipModuleUnload(um.module);
          return fail(c, VR_E_STATE, (std::string("vr_register_particle_model: kernel missing from the code object: ") + sym).c_str());
        }
        um.kernels[D * 100 + geo * 10 + mode] = f;
      }
  c->userModels.push_back(std::move(um));
  *kindOut = VR_PARTICLE_USER_BASE + (int32_t)c->userModels.size() - 1;
  return VR_OK;
}



// Trace::setGlobalData (rayTrace.hpp:137-145): vector `vecIdx` of the borrowed TracingData (data == NULL or n == 0
// drops it and every vector behind it).  The particle models index it by the primitive id of the caller's geometry.
int vr_set_global_data(vr_context *c, uint32_t vecIdx, const float *data, uint32_t n) {
  if (!c || vecIdx >= 16 || (n && !data))
    return fail(c, VR_E_INVALID, "vr_set_global_data: bad argument (at most 16 vectors)");
  if (!data || n == 0) {
    if (vecIdx < c->globalVecs.size())
      c->globalVecs.resize(vecIdx);
  } else {
    if (c->globalVecs.size() <= vecIdx)
      c->globalVecs.resize(vecIdx + 1);
    c->globalVecs[vecIdx].assign(data, data + n);
  }
  c->globalDirty = true;
  c->prepared = false;
  return VR_OK;
}
int vr_set_global_scalars(vr_context *c, const float *data, uint32_t n) {
  if (!c || (n && !data))
    return fail(c, VR_E_INVALID, "vr_set_global_scalars: bad argument");
  c->globalScalars.assign(data, data + n);
  c->globalDirty = true;
  c->prepared = false;
  return VR_OK;
}
int vr_set_use_wdist(vr_context *c, int on) {
  if (!c)
    return VR_E_INVALID;
  c->useWdist = on != 0;
  c->prepared = false;
  return VR_OK;
}
// Source = SourceGrid(points, particle's cosine power) (raySourceGrid.hpp); n == 0: back to SourceRandom
int vr_set_source_grid(vr_context *c, const float *points3, uint32_t n) {
  if (!c || (n && !points3))
    return fail(c, VR_E_INVALID, "vr_set_source_grid: bad argument");
  c->gridPoints.assign(points3, points3 + (size_t)n * 3);
  c->hostOrg.clear();
  c->hostDir.clear();
  c->hostDraws.clear();
  c->hostWeights.clear();
  c->sourceDirty = true;
  c->prepared = false;
  return VR_OK;
}
// Rays of a host-side Source callback for the NEXT applies: ray idx starts at org3[3 idx] towards
// dir3[3 idx] having consumed draws[idx] outputs of its engine (NULL: none).  n == 0: back to SourceRandom.
int vr_set_host_rays(vr_context *c, const float *org3, const float *dir3, const uint32_t *draws, uint64_t n) {
  if (!c || (n && (!org3 || !dir3)) || n > 0xFFFFFFFFull)
    return fail(c, VR_E_INVALID, "vr_set_host_rays: bad argument");
  c->hostOrg.assign(org3, org3 + (size_t)n * 3);
  c->hostDir.assign(dir3, dir3 + (size_t)n * 3);
  if (draws && n)
    c->hostDraws.assign(draws, draws + (size_t)n);
  else
    c->hostDraws.clear();
  c->hostWeights.clear();
  c->gridPoints.clear();
  c->sourceDirty = true;
  c->prepared = false;
  return VR_OK;
}
// Source::getInitialRayWeight(idx) (raySource.hpp:18, rayTraceKernel.hpp:124) of the rays handed over with
// vr_set_host_rays: the weight a ray starts with and the scale of the roulette's thresholds.  n == 0: all 1.
int vr_set_host_ray_weights(vr_context *c, const float *weights, uint64_t n) {
  if (!c || (n && !weights))
    return fail(c, VR_E_INVALID, "vr_set_host_ray_weights: bad argument");
  if (n && n != c->hostOrg.size() / 3)
    return fail(c, VR_E_INVALID, "vr_set_host_ray_weights: one weight per host ray (call vr_set_host_rays first)");
  c->hostWeights.assign(weights, weights + (size_t)n);
  c->sourceDirty = true;
  c->prepared = false;
  return VR_OK;
}
// Source::getSourceArea() (raySource.hpp:17) of a user source, used by normalizeFlux(SOURCE)
// (rayTraceDisk.hpp:127); area <= 0 restores SourceRandom's (the source face of the bounding box)
int vr_set_source_area(vr_context *c, float area) {
  if (!c)
    return VR_E_INVALID;
  c->sourceAreaOverride = area > 0.f ? area : 0.f;
  return VR_OK;
}
// apply() is called once per time step with a ray count that follows the surface: reserve the ray-stream
// buffers for the largest count expected (they also grow by half again on their own and are kept)
int vr_reserve_rays(vr_context *c, uint64_t n) {
  if (!c)
    return VR_E_INVALID;
  c->reserveRays = n;
  c->prepared = false;
  return VR_OK;
}
int vr_set_number_of_rays_per_point(vr_context *c, uint64_t n) {
  if (!c)
    return VR_E_INVALID;
  c->numRaysPerPoint = n;
  c->numRaysFixed = 0;
  return VR_OK;
}
int vr_set_number_of_rays_fixed(vr_context *c, uint64_t n) {
  if (!c)
    return VR_E_INVALID;
  c->numRaysFixed = n;
  c->numRaysPerPoint = 0;
  return VR_OK;
}
int vr_set_max_reflections(vr_context *c, uint32_t n) {
  if (!c)
    return VR_E_INVALID;
  c->maxReflections = n;
  return VR_OK;
}
int vr_set_max_boundary_hits(vr_context *c, uint32_t n) {
  if (!c)
    return VR_E_INVALID;
  c->maxBoundaryHits = n;
  return VR_OK;
}
int vr_set_rng_seed(vr_context *c, uint32_t s) {
  if (!c)
    return VR_E_INVALID;
  c->rngSeed = s;
  c->useRandomSeed = false;
  return VR_OK;
}
int vr_set_use_random_seeds(vr_context *c, int b) {
  if (!c)
    return VR_E_INVALID;
  c->useRandomSeed = b != 0;
  return VR_OK;
}
int vr_set_run_number(vr_context *c, uint32_t r) {
  if (!c)
    return VR_E_INVALID;
  c->runNumber = r;
  return VR_OK;
}
int vr_set_world_size(vr_context *c, uint32_t world) {
  if (!c || world == 0 || world > (1u << 20))
    return fail(c, VR_E_INVALID, "vr_set_world_size: 1 .. 2^20 ranks");
  c->worldSize = world;
  return VR_OK;
}

int vr_set_ray_range(vr_context *c, uint64_t first, uint64_t count) {
  if (!c)
    return VR_E_INVALID;
  c->rayFirst = first;
  c->rayCount = count;
  return VR_OK;
}

// ---- scene build (device LBVH + neighbourhood; VR_HOST_BUILD=1 selects the host builder) -------
static int ensure_host_order(vr_context *c) {
  if (c->hostOrderValid)
    return VR_OK;
  const uint32_t N = c->geo.numPrims;
  c->bvh.order.resize(N);
  VR_HIP(c, hipMemcpy(c->bvh.order.data(), c->dOrder.p, (size_t)N * 4, hipMemcpyDeviceToHost));
  c->hostOrderValid = true;
  return VR_OK;
}

// neighbourhood CSR in ORIGINAL ids on the host (smoothFlux, neighbour counts), lazily
static int ensure_host_neighbors(vr_context *c) {
  if (c->hostNeighborsValid)
    return VR_OK;
  HostGeometry &g = c->geo;
  const uint32_t N = g.numPrims;
  if (g.geo != 0) {
    g.nbOff.assign((size_t)N + 1, 0u);
    g.nbIds.clear();
  } else if (!c->geometryDirty && c->dNbOff.p) {
    int r = ensure_host_order(c);
    if (r != VR_OK)
      return r;
    std::vector<uint32_t> off((size_t)N + 1);
    VR_HIP(c, hipMemcpy(off.data(), c->dNbOff.p, ((size_t)N + 1) * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> ids(off[N]);
    if (off[N])
      VR_HIP(c, hipMemcpy(ids.data(), c->dNbIds.p, (size_t)off[N] * 4, hipMemcpyDeviceToHost));
    g.nbOff.assign((size_t)N + 1, 0u);
    for (uint32_t q = 0; q < N; ++q)
      g.nbOff[c->bvh.order[q] + 1] = off[q + 1] - off[q];
    for (uint32_t i = 0; i < N; ++i)
      g.nbOff[i + 1] += g.nbOff[i];
    g.nbIds.resize(off[N]);
    for (uint32_t q = 0; q < N; ++q) {
      uint32_t w = g.nbOff[c->bvh.order[q]];
      for (uint32_t j = off[q]; j < off[q + 1]; ++j)
        g.nbIds[w++] = c->bvh.order[ids[j]];
      std::sort(g.nbIds.begin() + g.nbOff[c->bvh.order[q]], g.nbIds.begin() + w);
    }
  } else {
    host_neighbors(g.D, g.points3.data(), N, 2 * g.diskRadius, g.minC, g.nbOff, g.nbIds);
  }
  c->hostNeighborsValid = true;
  return VR_OK;
}

// 16-byte nodes for the per-lane traversal: frame from the root box (which holds every
// padded primitive box), two cells of margin so the outward rounding never clamps
static int quantize_scene(vr_context *c, const float *preNodes, const float *root8) {
  const float lo[3] = {root8[0], root8[1], root8[2]}, hi[3] = {root8[4], root8[5], root8[6]};
  for (int k = 0; k < 3; ++k) {
    c->sceneLo[k] = lo[k];
    c->sceneHi[k] = hi[k];
    const float ext = hi[k] - lo[k];
    c->qscale[k] = ext > 0.f ? 65531.0f / ext : 0.f;
    c->qbase[k] = ext > 0.f ? lo[k] - 2.0f / c->qscale[k] : lo[k];
  }
  VR_HIP(c, c->dQNodes.ensure((size_t)c->numNodes * 4));
  VR_HIP(c, c->dPNodes.ensure((size_t)std::max<uint32_t>(c->numNodes, 1u) * 8));
  VR_HIP(c, launch_quantize_nodes(preNodes, c->numNodes, c->qbase, c->qscale, c->dQNodes.p, c->dPNodes.p, c->stream));
  VR_HIP(c, hipStreamSynchronize(c->stream));
  return VR_OK;
}

static int build_scene(vr_context *c) {
  HostGeometry &g = c->geo;
  const uint32_t N = g.numPrims;
  const bool disk = g.geo == 0;
  c->hostOrderValid = false;
  c->hostNeighborsValid = false;
  VR_HIP(c, c->dLeafOfOrig.ensure(N));
  VR_HIP(c, c->dOrder.ensure(N));
  {
    uint32_t R = 1;
    if (const char *e = std::getenv("VR_ACC_REPLICAS"))
      R = (uint32_t)std::max(1, std::atoi(e));
    else
      while (R < 64u && (size_t)N * (2u * R) <= (1u << 21))
        R *= 2u;
    while (R & (R - 1u)) // power of two
      R &= R - 1u;
    c->accReplicas = R;
    c->accStride = (N + 15u) & ~15u; // replicas start on 128-byte lines
    c->accPlanes = 0;                // (buffers are sized per data label in vr_apply_prepare)
  }
  VR_HIP(c, c->dCounters.ensure(80));
  VR_HIP(c, c->dNbOff.ensure((size_t)N + 1));
  const char *hb = std::getenv("VR_HOST_BUILD");
  if (hb && std::atoi(hb)) {
    // host builder (validation path): LBVH + CSR on the CPU, uploaded
    if (disk)
      host_neighbors(g.D, g.points3.data(), N, 2 * g.diskRadius, g.minC, g.nbOff, g.nbIds);
    else
      g.nbOff.assign((size_t)N + 1, 0u), g.nbIds.clear();
    c->hostNeighborsValid = true;
    host_build_bvh(g, c->bvh);
    c->hostOrderValid = true;
    std::vector<float> prims;
    host_pack_prims(g, c->bvh, prims);
    c->leafOfOrig.resize(N);
    for (uint32_t q = 0; q < N; ++q)
      c->leafOfOrig[c->bvh.order[q]] = q;
    std::vector<uint32_t> off((size_t)N + 1, 0u), ids(g.nbIds.size());
    for (uint32_t q = 0; q < N; ++q) {
      const uint32_t o = c->bvh.order[q];
      off[q + 1] = off[q] + (g.nbOff[o + 1] - g.nbOff[o]);
    }
    for (uint32_t q = 0; q < N; ++q) {
      const uint32_t o = c->bvh.order[q];
      uint32_t w = off[q];
      for (uint32_t j = g.nbOff[o]; j < g.nbOff[o + 1]; ++j)
        ids[w++] = c->leafOfOrig[g.nbIds[j]];
    }
    VR_HIP(c, c->dNodes.ensure(c->bvh.nodes.size()));
    VR_HIP(c, c->dPrims.ensure(prims.size()));
    VR_HIP(c, c->dNbIds.ensure(ids.size()));
    c->nbTotal = (uint32_t)ids.size();
    VR_HIP(c, hipMemcpyAsync(c->dNodes.p, c->bvh.nodes.data(), c->bvh.nodes.size() * 4, hipMemcpyHostToDevice, c->stream));
    VR_HIP(c, hipMemcpyAsync(c->dPrims.p, prims.data(), prims.size() * 4, hipMemcpyHostToDevice, c->stream));
    VR_HIP(c, hipMemcpyAsync(c->dNbOff.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, c->stream));
    if (!ids.empty())
      VR_HIP(c, hipMemcpyAsync(c->dNbIds.p, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, c->stream));
    VR_HIP(c, hipMemcpyAsync(c->dLeafOfOrig.p, c->leafOfOrig.data(), (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
    VR_HIP(c, hipMemcpyAsync(c->dOrder.p, c->bvh.order.data(), (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
    VR_HIP(c, hipStreamSynchronize(c->stream));
    c->numNodes = c->bvh.numNodes;
    c->haveWide = false; // (validation path: walks only)
    return quantize_scene(c, c->dNodes.p, c->bvh.nodes.data()); // (host builder: pre-order already)
  }

  // ---- device builder ----
  SetupParams s{};
  // (triangles: a leaf of up to 3 — their test is 64 bytes and ~60 instructions per primitive; measured 4 -> 3:
  //  trenchMesh 0.1 29.6 -> 28.4 ms, C4 21.4 -> 20.6; disks: 2 .. 4 within 2 %, 6 and 8 slower)
  s.leafMax = g.geo == 1 ? 3u : (uint32_t)VR_LEAF_MAX;
  if (const char *e = std::getenv("VR_LEAF_MAX"))
    s.leafMax = (uint32_t)std::min(15, std::max(1, std::atoi(e)));
  s.orderAxis = c->ts[0];                   // rays travel along this axis ...
  s.orderSign = c->ts[3] ? 1.f : -1.f;      // ... from its max (min) side: that child first
  if (const char *e = std::getenv("VR_NO_CHILD_ORDER"))
    if (std::atoi(e))
      s.orderSign = 0.f;
  s.n = N;
  s.geo = g.geo;
  s.D = g.D;
  s.nbDist = 2 * g.diskRadius;
  s.mortonAniso = VR_MORTON_ANISO;
  if (const char *e = std::getenv("VR_MORTON_ANISO"))
    s.mortonAniso = std::max(1.f, (float)std::atof(e));
  VR_HIP(c, c->dNormal3.ensure((size_t)N * 3));
  VR_HIP(c, hipMemcpyAsync(c->dNormal3.p, g.normal3.data(), (size_t)N * 12, hipMemcpyHostToDevice, c->stream));
  if (disk) {
    VR_HIP(c, c->dDisk4.ensure((size_t)N * 4));
    VR_HIP(c, c->dPoints3.ensure((size_t)N * 3));
    VR_HIP(c, hipMemcpyAsync(c->dPoints3.p, g.points3.data(), (size_t)N * 12, hipMemcpyHostToDevice, c->stream));
    VR_HIP(c, launch_disk4(c->dPoints3.p, N, g.diskRadius, g.D, c->dDisk4.p, c->stream)); // (= g.disk4, made on the device)
  } else {
    VR_HIP(c, c->dVerts.ensure(g.verts.size()));
    VR_HIP(c, c->dTris.ensure(g.tris.size()));
    VR_HIP(c, hipMemcpyAsync(c->dVerts.p, g.verts.data(), g.verts.size() * 4, hipMemcpyHostToDevice, c->stream));
    VR_HIP(c, hipMemcpyAsync(c->dTris.p, g.tris.data(), g.tris.size() * 4, hipMemcpyHostToDevice, c->stream));
  }
  const size_t tiles = ((size_t)N + 1023) / 1024;
  VR_HIP(c, c->dBox.ensure((size_t)N * 6));
  VR_HIP(c, c->dSBox.ensure((size_t)N * 6));
  VR_HIP(c, c->dNodeBox.ensure((size_t)N * 6));
  VR_HIP(c, c->dBounds.ensure(8));
  VR_HIP(c, c->dKeysA.ensure(N));
  VR_HIP(c, c->dKeysB.ensure(N));
  VR_HIP(c, c->dValsA.ensure(N));
  VR_HIP(c, c->dValsB.ensure(N));
  VR_HIP(c, c->dSortTable.ensure(256 * tiles));
  VR_HIP(c, c->dRangeLo.ensure(N));
  VR_HIP(c, c->dRangeHi.ensure(N));
  VR_HIP(c, c->dChildL.ensure(N));
  VR_HIP(c, c->dChildR.ensure(N));
  VR_HIP(c, c->dParentInt.ensure(N));
  VR_HIP(c, c->dParentLeaf.ensure(N));
  VR_HIP(c, c->dArrive.ensure(N));
  VR_HIP(c, c->dSubSize.ensure(N));
  VR_HIP(c, c->dNodes.ensure(((size_t)2 * N) * 8));
  VR_HIP(c, c->dNodesPre.ensure(((size_t)2 * N) * 8));
  VR_HIP(c, c->dPrims.ensure((size_t)N * (disk ? 8 : 16)));
  VR_HIP(c, c->dScanTmp.ensure(2 * ((256 * tiles + (size_t)N + 1) / 2048 + 4) + 64));
  s.disk4 = c->dDisk4.p;
  s.normal3 = c->dNormal3.p;
  s.points3 = c->dPoints3.p;
  s.verts = c->dVerts.p;
  s.tris = c->dTris.p;
  s.box = c->dBox.p;
  s.sbox = c->dSBox.p;
  s.bounds = c->dBounds.p;
  s.keysA = c->dKeysA.p;
  s.keysB = c->dKeysB.p;
  s.valsA = c->dValsA.p;
  s.valsB = c->dValsB.p;
  s.sortTable = c->dSortTable.p;
  s.rangeLo = c->dRangeLo.p;
  s.rangeHi = c->dRangeHi.p;
  s.childL = c->dChildL.p;
  s.childR = c->dChildR.p;
  s.parentInt = c->dParentInt.p;
  s.parentLeaf = c->dParentLeaf.p;
  s.arrive = c->dArrive.p;
  s.nodeBox = c->dNodeBox.p;
  s.subSize = c->dSubSize.p;
  s.nodes = c->dNodes.p;
  s.nodesPre = c->dNodesPre.p;
  s.prims = c->dPrims.p;
  s.leafOfOrig = c->dLeafOfOrig.p;
  s.order = c->dOrder.p;
  s.nbOff = c->dNbOff.p;
  s.nbIds = nullptr;
  VR_HIP(c, c->dWide.ensure(wide_tree_entries(N) * 8));
  s.wide = c->dWide.p;
  VR_HIP(c, launch_setup_bvh(s, c->dScanTmp.p, c->stream));
  VR_HIP(c, launch_wide_tree(s, c->wideRoot, c->stream));
  c->haveWide = true;
  // every build is verified (one small kernel; its counter is read back with the syncs below):
  // the fit's cross-workgroup hand-over is the one place the build relies on memory ordering
  VR_HIP(c, hipMemsetAsync(c->dBounds.p + 6, 0, 4, c->stream));
  VR_HIP(c, launch_bvh_check(s, c->dBounds.p + 6, c->stream));
  c->lastSetup = s;
  c->haveSetup = true;
  if (disk) {
    // neighbourhood: ONE query that counts and keeps up to VR_NB_KEEP ids per primitive -> scan -> pack (the query, a
    // range walk of the BVH per primitive, is the most expensive kernel of a build: 0.4 ms per 10^6 disks; counting and
    // filling in two passes walked twice).  A primitive with more neighbours: the two-pass path.
    VR_HIP(c, c->dNbTmp.ensure((size_t)N * VR_NB_KEEP + 1));
    s.nbTmp = c->dNbTmp.p;
    VR_HIP(c, hipMemsetAsync(c->dNbTmp.p + (size_t)N * VR_NB_KEEP, 0, 4, c->stream));
    VR_HIP(c, hipMemsetAsync(c->dNbOff.p + N, 0, 4, c->stream));
    VR_HIP(c, launch_setup_neighbors(s, 2, c->stream));
    VR_HIP(c, launch_scan(c->dNbOff.p, N + 1, c->dScanTmp.p, c->stream));
    uint32_t total = 0, overflow = 0;
    VR_HIP(c, hipMemcpyAsync(&total, c->dNbOff.p + N, 4, hipMemcpyDeviceToHost, c->stream));
    VR_HIP(c, hipMemcpyAsync(&overflow, c->dNbTmp.p + (size_t)N * VR_NB_KEEP, 4, hipMemcpyDeviceToHost, c->stream));
    VR_HIP(c, hipStreamSynchronize(c->stream));
    VR_HIP(c, c->dNbIds.ensure(total));
    c->nbTotal = total;
    s.nbIds = c->dNbIds.p;
    VR_HIP(c, launch_setup_neighbors(s, (overflow || std::getenv("VR_NB_TWO_PASS")) ? 1 : 3, c->stream));
  } else {
    c->nbTotal = 0;
    VR_HIP(c, hipMemsetAsync(c->dNbOff.p, 0, ((size_t)N + 1) * 4, c->stream));
    VR_HIP(c, c->dNbIds.ensure(1));
  }
  float root8[8];
  uint32_t sz = 0, bad = 0;
  VR_HIP(c, hipMemcpyAsync(root8, c->dNodesPre.p, sizeof(root8), hipMemcpyDeviceToHost, c->stream));
  VR_HIP(c, hipMemcpyAsync(&sz, c->dSubSize.p, 4, hipMemcpyDeviceToHost, c->stream));
  VR_HIP(c, hipMemcpyAsync(&bad, c->dBounds.p + 6, 4, hipMemcpyDeviceToHost, c->stream));
  VR_HIP(c, hipStreamSynchronize(c->stream));
  c->bvhRefits = 0;
  if (bad != 0) {
    // never observed; the textbook agent-scope fences cost 3 ms per 10^6 primitives
    s.strictFence = 1;
    s.nbIds = nullptr;
    VR_HIP(c, launch_fit_bvh(s, c->stream));
    VR_HIP(c, hipMemsetAsync(c->dBounds.p + 6, 0, 4, c->stream));
    VR_HIP(c, launch_bvh_check(s, c->dBounds.p + 6, c->stream));
    VR_HIP(c, hipMemcpyAsync(root8, c->dNodesPre.p, sizeof(root8), hipMemcpyDeviceToHost, c->stream));
    VR_HIP(c, hipMemcpyAsync(&sz, c->dSubSize.p, 4, hipMemcpyDeviceToHost, c->stream));
    VR_HIP(c, hipMemcpyAsync(&bad, c->dBounds.p + 6, 4, hipMemcpyDeviceToHost, c->stream));
    VR_HIP(c, hipStreamSynchronize(c->stream));
    c->bvhRefits = 1;
    if (bad != 0)
      return fail(c, VR_E_HIP, "device BVH build failed its consistency check twice");
    if (disk) { // the neighbourhood was queried on the inconsistent tree: redo it
      s.nbIds = nullptr;
      VR_HIP(c, hipMemsetAsync(c->dNbOff.p + N, 0, 4, c->stream));
      VR_HIP(c, launch_setup_neighbors(s, 0, c->stream));
      VR_HIP(c, launch_scan(c->dNbOff.p, N + 1, c->dScanTmp.p, c->stream));
      uint32_t total = 0;
      VR_HIP(c, hipMemcpyAsync(&total, c->dNbOff.p + N, 4, hipMemcpyDeviceToHost, c->stream));
      VR_HIP(c, hipStreamSynchronize(c->stream));
      VR_HIP(c, c->dNbIds.ensure(total));
      c->nbTotal = total;
      s.nbIds = c->dNbIds.p;
      VR_HIP(c, launch_setup_neighbors(s, 1, c->stream));
      VR_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->lastSetup = s;
  }
  c->numNodes = sz & 0x7FFFFFFFu;
  c->bvh.numNodes = c->numNodes;
  c->bvh.numLeaves = 0;
  c->bvh.maxDepth = 0;
  const int rq = quantize_scene(c, c->dNodesPre.p, root8);
  c->dNodesPre.release(); // (build-time scratch)
  return rq;
}

// ---- run ----------------------------------------------------------------------
static int effective_direction(const vr_context *c) {
  if (c->sourceDirection >= 0)
    return c->sourceDirection;
  return c->geo.D == 2 ? VR_POS_Y : VR_POS_Z; // rayTrace.hpp:166-167
}

// rayTraceKernel.hpp:57-61: numRaysFixed, or source.getNumPoints() * numRaysPerPoint
// (SourceRandom: the geometry's points; SourceGrid: the grid's; host rays: exactly those given)
static uint64_t rays_of_apply(const vr_context *c) {
  if (!c->hostOrg.empty())
    return c->hostOrg.size() / 3;
  const uint64_t srcPoints = !c->gridPoints.empty() ? c->gridPoints.size() / 3 : c->geo.numPrims;
  return c->numRaysFixed == 0 ? srcPoints * c->numRaysPerPoint : c->numRaysFixed;
}

// everything one particle's launch needs (scene build and areas only when they changed)
static int prepare_one(vr_context *c) {
  VR_HIP(c, hipSetDevice(c->device));
  c->info = vr_trace_info{};
  // checkSettings (rayTraceDisk.hpp:196-217): the reference logs and carries
  // on; with nothing to trace we stop and report through the error flag.
  if (!c->haveParticle) {
    c->info.error = 1;
    return fail(c, VR_E_INVALID, "No particle was specified in rayTrace. Aborting.");
  }
  if (c->geo.numPrims == 0) {
    c->info.error = 1;
    return fail(c, VR_E_INVALID, "No geometry was passed to rayTrace. Aborting.");
  }
  const int D = c->geo.D;
  const int dir = effective_direction(c);
  if (D == 2 && (dir == VR_POS_Z || dir == VR_NEG_Z)) {
    c->info.error = 1;
    return fail(c, VR_E_INVALID, "Invalid source direction in 2D geometry. Aborting.");
  }
  if (c->geo.geo == 0 && c->geo.diskRadius > c->geo.gridDelta)
    c->info.warning = 1;

  const auto t0 = std::chrono::steady_clock::now();
  TraceParams &p = c->params;
  const bool redoConfig = c->configDirty || c->geometryDirty;
  if (redoConfig) {
    // bounding box, trace settings, boundary (rayTraceDisk.hpp:21-27)
    for (int k = 0; k < 3; ++k) {
      c->bbLo[k] = c->geo.minC[k];
      c->bbHi[k] = c->geo.maxC[k];
    }
    host_adjust_bbox(c->bbLo, c->bbHi, D, dir, c->geo.geo == 0 ? c->geo.diskRadius : c->geo.gridDelta);
    c->ts = host_trace_settings(dir);
    {
      Tri walls[8];
      host_build_walls(c->bbLo, c->bbHi, c->ts[1], c->ts[2], walls);
      float *tbl = c->wallsHost; // (uploaded with the launch's scalar frame: end of this function)
      for (int i = 0; i < 8; ++i) {
        std::memcpy(tbl + 12 * i, walls[i].v0, 12);
        std::memcpy(tbl + 12 * i + 3, walls[i].e1, 12);
        std::memcpy(tbl + 12 * i + 6, walls[i].e2, 12);
        std::memcpy(tbl + 12 * i + 9, walls[i].Ng, 12);
      }
    }
    // rayBoundary.hpp:23-25: conditions are picked by AXIS
    c->boundaryConds[0] = c->bcs[c->ts[1]];
    c->boundaryConds[1] = (D == 2 && c->ts[2] >= 2) ? 0 : c->bcs[c->ts[2]];
    // SourceRandom::getSourceArea (raySourceRandom.hpp:40-47)
    const int f = c->ts[1], s = c->ts[2];
    c->sourceArea = D == 2 ? (c->bbHi[f] - c->bbLo[f]) : (c->bbHi[f] - c->bbLo[f]) * (c->bbHi[s] - c->bbLo[s]);
    c->keyCoord = host_sort_plane(c->geo, c->ts[0], c->ts[3] ? c->geo.minC[c->ts[0]] : c->geo.maxC[c->ts[0]],
                                 &c->keyShare);
  }

  const uint32_t N = c->geo.numPrims;
  // the BVH's child order follows the source side: a new source direction rebuilds it
  if (c->builtOrderAxis != c->ts[0] || c->builtOrderSign != (c->ts[3] ? 1.f : -1.f))
    c->geometryDirty = true;
  if (c->geometryDirty) {
    int r = build_scene(c);
    if (r != VR_OK)
      return r;
    c->geometryDirty = false;
    ++c->bvhBuilds;
    c->builtOrderAxis = c->ts[0];
    c->builtOrderSign = c->ts[3] ? 1.f : -1.f;
  }
  // exposed area of every primitive, resident on the device for normalizeFlux
  // (computeDiskAreas, rayGeometryDisk.hpp:266-354: one thread per disk; triangle areas come
  // with the mesh, rayGeometryTriangle.hpp:145-176)
  if (redoConfig || !c->areasValid) {
    VR_HIP(c, c->dAreas.ensure(N));
    c->diskAreasHostValid = false;
    if (c->geo.geo == 0) {
      AreaParams ap{};
      ap.D = D;
      ap.firstDir = c->ts[1];
      ap.secondDir = c->ts[2];
      // rayGeometryDisk.hpp:281-284 indexes the 2-entry BC array by AXIS; axis 2 is out of
      // range there, entry 1 is used for it
      ap.bcFirst = c->boundaryConds[c->ts[1] > 1 ? 1 : c->ts[1]];
      ap.bcSecond = c->boundaryConds[c->ts[2] > 1 ? 1 : c->ts[2]];
      for (int k = 0; k < 3; ++k) {
        ap.minC[k] = c->geo.minC[k];
        ap.maxC[k] = c->geo.maxC[k];
      }
      const char *hb = std::getenv("VR_HOST_BUILD");
      if (hb && std::atoi(hb)) {
        host_disk_areas(c->geo, ap, c->diskAreas);
        VR_HIP(c, hipMemcpy(c->dAreas.p, c->diskAreas.data(), (size_t)N * 4, hipMemcpyHostToDevice));
        c->diskAreasHostValid = true;
      } else {
        VR_HIP(c, launch_disk_areas(c->dDisk4.p, c->dNormal3.p, N, ap, c->dAreas.p, c->stream));
      }
    } else {
      VR_HIP(c, hipMemcpyAsync(c->dAreas.p, c->geo.triAreas.data(), (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
      VR_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->areasValid = true;
  }
  // per-primitive sticking from the material map (gpu::Particle-style, rayParticle.hpp:208-218)
  const bool redoSticking = redoConfig || c->particleDirty;
  if (redoSticking)
    c->havePrimSticking = false;
  if (redoSticking && !c->matStickIds.empty()) {
    int ro = ensure_host_order(c);
    if (ro != VR_OK)
      return ro;
    std::vector<float> ps(N);
    for (uint32_t q = 0; q < N; ++q) {
      const uint32_t o = c->bvh.order[q];
      const int mat = o < c->geo.materialIds.size() ? c->geo.materialIds[o] : 0;
      float s = c->sticking;
      for (size_t m = 0; m < c->matStickIds.size(); ++m)
        if (c->matStickIds[m] == mat)
          s = c->matStickVals[m];
      ps[q] = s;
    }
    VR_HIP(c, c->dPrimSticking.ensure(N));
    VR_HIP(c, hipMemcpy(c->dPrimSticking.p, ps.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    c->havePrimSticking = true;
  }
  const float *dStick = c->havePrimSticking ? c->dPrimSticking.p : nullptr;
  c->configDirty = false;
  c->particleDirty = false;
  // Trace::setGlobalData: every vector padded to one stride, one upload
  if (c->globalDirty) {
    uint32_t stride = 0;
    for (const auto &v : c->globalVecs)
      stride = std::max<uint32_t>(stride, (uint32_t)v.size());
    c->globalStride = stride;
    if (stride && !c->globalVecs.empty()) {
      std::vector<float> flat((size_t)stride * c->globalVecs.size(), 0.f);
      for (size_t v = 0; v < c->globalVecs.size(); ++v)
        std::copy(c->globalVecs[v].begin(), c->globalVecs[v].end(), flat.begin() + v * stride);
      VR_HIP(c, c->dGlobalVec.ensure(flat.size()));
      VR_HIP(c, hipMemcpy(c->dGlobalVec.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
    }
    if (!c->globalScalars.empty()) {
      VR_HIP(c, c->dGlobalScalars.ensure(c->globalScalars.size()));
      VR_HIP(c, hipMemcpy(c->dGlobalScalars.p, c->globalScalars.data(), c->globalScalars.size() * 4, hipMemcpyHostToDevice));
    }
    c->globalDirty = false;
  }

  const uint64_t numRays = rays_of_apply(c);
  c->numRaysLast = numRays;
  uint64_t first = 0, last = numRays;
  if (c->rayCount) {
    first = std::min(c->rayFirst, numRays);
    last = std::min(numRays, first + c->rayCount);
  }
  c->rayFirstLaunch = first;
  c->rayEndLaunch = last;
  uint32_t seed = c->runNumber + c->rngSeed; // rayTraceKernel.hpp:100
  if (c->haveSharedSeed) { // (vr_apply_sharded with random seeds: the one seed every rank agreed on)
    seed = c->sharedSeed;
  } else if (c->useRandomSeed) {
    std::random_device rd;
    seed = (uint32_t)rd();
  }
  // ABSORB: every hit takes the whole weight -> nothing after the first
  // surface hit is observable
  c->absorb = c->sticking >= 1.f;
  for (float v : c->matStickVals)
    c->absorb = c->absorb && v >= 1.f;
  // the extended kernel (vr_particles.hpp) serves everything beyond the two built-in particles
  const bool extended = c->particleKind >= VR_PARTICLE_CONED_COSINE || c->useWdist || c->meanFreePath > 0.f;
  if (extended)
    c->absorb = false;
  if (!c->hostOrg.empty() && !c->hostWeights.empty())
    c->absorb = false; // (the absorbing kernels credit unit weights)
  // (the rare, register-hungry options — coned-cosine model, WDIST crediting, mean free path — have an instantiation
  //  of their own: multi-label and per-material particles should not pay for them)
  bool extFull = Particles::needsFull(c->particleKind) || c->useWdist || c->meanFreePath > 0.f;
  if (c->userModel >= 0) {
    const UserModel &um = c->userModels[c->userModel];
    if (extFull && !um.needsFull)
      return fail(c, VR_E_INVALID, "this particle model was registered without VR_MODEL_NEEDS_FULL: its code object has no "
                                   "kernel with WDIST crediting / mean-free-path scattering");
    extFull = um.needsFull;
  }
  c->kernelParticle = extended ? (extFull ? (int)P_EXT_FULL : (int)P_EXT) : c->particleKind;
  // a scene of a few hundred primitives goes into LDS as a whole (MODE 4: the general kernel — also for
  // absorbing particles — of whatever particle): pair nodes, records, neighbourhood, accumulators (one plane
  // per data label), per-material sticking
  bool smallScene = false;
  {
    const uint32_t recB = c->geo.geo == 0 ? 32u : 64u;
    uint32_t off[6], o = 0, nbTotal = 0;
    if (c->geo.geo == 0)
      nbTotal = c->nbTotal;
    auto put = [&](int k, size_t bytes) {
      off[k] = o;
      o += (uint32_t)((bytes + 15) & ~(size_t)15);
    };
    put(0, (size_t)c->numNodes * 32);
    put(1, (size_t)N * recB);
    put(2, ((size_t)N + 1) * 4);
    put(3, (size_t)nbTotal * 4);
    put(4, (size_t)N * 8 * c->numData);
    put(5, c->havePrimSticking ? (size_t)N * 4 : 0);
    smallScene = o <= VR_SMALL_LDS && c->numNodes > 0;
    if (const char *e = std::getenv("VR_SMALL_SCENE"))
      smallScene = smallScene && std::atoi(e) != 0;
    for (int k = 0; k < 6; ++k)
      p.smallOff[k] = off[k];
    p.smallNb = nbTotal;
    p.smallBytes = (o + 255u) & ~255u;
    if (smallScene)
      c->absorb = false; // (ray records with the RNG cursors: the general kernel reads them)
  }
  // ---- flat WITH RELIEF?  (DESIGN.md 5.2 "relief packets")  The flat-scene kernels owe their speed to the packet
  // query, and the query clips its rays to the SCENE box: half a grid cell of relief lets the grazing rays of a wave
  // stretch its box over hundreds of cells.  Where the scene is thin along the source axis and the relief field says that
  // few rays would be grazing ones (ReliefParams::stats), the rays are sorted by their predicted first hit, the grazing ones are filed apart
  // (bin_of_relief, vr_trace.hip) and the query clips to the LOCAL relief (relief_clip, vr_device.hpp): MODE 5 / 6.
  const bool flatScene = c->keyShare >= 0.95f && (c->sceneHi[c->ts[0]] - c->sceneLo[c->ts[0]]) <= 0.25f * c->geo.gridDelta;
  c->reliefScene = false;
  {
    float maxThick = 8.f, travel = 1.5f;
    if (const char *e = std::getenv("VR_RELIEF_MAX_THICK"))
      maxThick = (float)std::atof(e);
    if (const char *e = std::getenv("VR_RELIEF_TRAVEL"))
      travel = std::max(0.05f, (float)std::atof(e));
    const float thickScene = c->sceneHi[c->ts[0]] - c->sceneLo[c->ts[0]];
    const bool plainSource = !c->usePrimaryDirection && c->gridPoints.empty() && c->hostOrg.empty();
    const bool kernelOk = c->absorb || (c->geo.geo == 0 && c->kernelParticle <= (int)P_EXT);
    bool want = !flatScene && !smallScene && plainSource && kernelOk && c->userModel < 0 && c->geo.gridDelta > 0.f &&
                thickScene <= maxThick * c->geo.gridDelta && !std::getenv("VR_NO_RELIEF");
    if (want) {
      const bool stale = c->rfBuild != c->bvhBuilds || c->rfAxes[0] != c->ts[0] || c->rfAxes[1] != c->ts[1] ||
                         c->rfAxes[2] != c->ts[2] || c->rfAxes[3] != c->ts[3] || c->rf.travel != travel * c->geo.gridDelta;
      if (stale) {
        ReliefParams &q = c->rf;
        q.prims = c->dPrims.p;
        q.n = N;
        q.geo = c->geo.geo;
        q.ax = c->ts[0];
        q.a1 = c->ts[1];
        q.a2 = c->ts[2];
        const float ext1 = c->sceneHi[q.a1] - c->sceneLo[q.a1], ext2 = D == 3 ? c->sceneHi[q.a2] - c->sceneLo[q.a2] : 0.f;
        float cells = 1.f; // fine tile side in grid cells
        if (const char *e = std::getenv("VR_RELIEF_TILE"))
          cells = std::max(0.25f, (float)std::atof(e));
        float tile = std::max(cells * c->geo.gridDelta, std::max(ext1, ext2) / 1024.f);
        q.tile = tile;
        q.invTile = 1.f / tile;
        q.lo1 = c->sceneLo[q.a1];
        q.lo2 = D == 3 ? c->sceneLo[q.a2] : 0.f;
        q.nx = std::max(1, std::min(1024, (int)std::ceil(ext1 / tile)));
        q.ny = D == 3 ? std::max(1, std::min(1024, (int)std::ceil(ext2 / tile))) : 1;
        q.k = std::max(2, (std::max(q.nx, q.ny) + 255) / 256);
        if (const char *e = std::getenv("VR_RELIEF_COARSE_K"))
          q.k = std::max(1, std::atoi(e));
        q.cnx = (q.nx + q.k - 1) / q.k;
        q.cny = (q.ny + q.k - 1) / q.k;
        float scale = 1e-3f;
        for (int k = 0; k < 3; ++k)
          scale = std::max(scale, std::max(std::fabs(c->sceneLo[k]), std::fabs(c->sceneHi[k])));
        q.pad = 1e-5f * scale;
        q.travel = travel * c->geo.gridDelta;
        q.emptyMid = c->keyCoord;
        VR_HIP(c, c->dRfRawLo.ensure((size_t)q.nx * q.ny));
        VR_HIP(c, c->dRfRawHi.ensure((size_t)q.nx * q.ny));
        VR_HIP(c, c->dRfFine.ensure((size_t)q.nx * q.ny * 2));
        VR_HIP(c, c->dRfCoarse.ensure((size_t)q.cnx * q.cny * 2));
        VR_HIP(c, c->dRfStats.ensure(2));
        q.rawLo = c->dRfRawLo.p;
        q.rawHi = c->dRfRawHi.p;
        q.fine = c->dRfFine.p;
        q.coarse = c->dRfCoarse.p;
        q.stats = c->dRfStats.p;
        VR_HIP(c, launch_relief_field(q, c->stream));
        uint32_t st[2] = {0, 0};
        VR_HIP(c, hipMemcpyAsync(st, q.stats, sizeof(st), hipMemcpyDeviceToHost, c->stream));
        VR_HIP(c, hipStreamSynchronize(c->stream));
        c->rfLooseShare = st[0] ? (float)st[1] / 4096.f / (float)st[0] : 1.f;
        c->rfBuild = c->bvhBuilds;
        for (int k = 0; k < 4; ++k)
          c->rfAxes[k] = c->ts[k];
      }
      // (the share of a cosine source's rays that the generator would file as loose, from the coarse tiles' thickness:
      //  where most rays are loose the structured-scene kernels do the work anyway, without the second launch)
      float share = 0.3f;
      if (const char *e = std::getenv("VR_RELIEF_SHARE"))
        share = (float)std::atof(e);
      c->reliefScene = c->rfLooseShare <= share;
    }
    p.reliefCoarse = c->reliefScene ? c->rf.coarse : nullptr;
    p.rcLo1 = c->rf.lo1;
    p.rcLo2 = c->rf.lo2;
    p.rcInvT = c->reliefScene ? c->rf.invTile / (float)c->rf.k : 0.f;
    p.rcNx = c->rf.cnx;
    p.rcNy = c->rf.cny;
    p.reliefTravel = travel * c->geo.gridDelta;
    {
      // the tile walk of relief_clip starts where the ray enters the SCENE box: a ray that would cross more than `steps`
      // tiles on its way through it is filed as loose too
      float steps = 6.f;
      if (const char *e = std::getenv("VR_RELIEF_STEPS"))
        steps = std::max(1.f, (float)std::atof(e));
      p.reliefTanMax = thickScene > 0.f ? steps * c->rf.tile / thickScene : 3.0e38f;
    }
    p.reliefLookups = 1; // (2: tight launch -0.1 ms, generator +0.4 ms per 1e8 rays — a random 8-byte gather per ray is a 128-byte line from L2)
    if (const char *e = std::getenv("VR_RELIEF_LOOKUPS"))
      p.reliefLookups = std::min(2, std::max(0, std::atoi(e)));
  }
  // accumulators: one plane per data label, each replicated accReplicas times
  if (c->accPlanes != c->totalData) {
    VR_HIP(c, c->dFluxAcc.ensure((size_t)c->accStride * c->accReplicas * c->totalData));
    VR_HIP(c, c->dFluxOrig.ensure((size_t)N * c->totalData));
    c->accPlanes = c->totalData;
  }
  if (c->boundFlux && c->boundFluxN != N * c->totalData)
    return fail(c, VR_E_STATE, "bound accumulator buffer does not hold numPrims x numData int64");
  // source data
  if (c->sourceDirty) {
    if (!c->gridPoints.empty()) {
      VR_HIP(c, c->dGrid.ensure(c->gridPoints.size()));
      VR_HIP(c, hipMemcpy(c->dGrid.p, c->gridPoints.data(), c->gridPoints.size() * 4, hipMemcpyHostToDevice));
    }
    if (!c->hostOrg.empty()) {
      VR_HIP(c, c->dHostOrg.ensure(c->hostOrg.size()));
      VR_HIP(c, c->dHostDir.ensure(c->hostDir.size()));
      VR_HIP(c, hipMemcpy(c->dHostOrg.p, c->hostOrg.data(), c->hostOrg.size() * 4, hipMemcpyHostToDevice));
      VR_HIP(c, hipMemcpy(c->dHostDir.p, c->hostDir.data(), c->hostDir.size() * 4, hipMemcpyHostToDevice));
      if (!c->hostDraws.empty()) {
        VR_HIP(c, c->dHostDraws.ensure(c->hostDraws.size()));
        VR_HIP(c, hipMemcpy(c->dHostDraws.p, c->hostDraws.data(), c->hostDraws.size() * 4, hipMemcpyHostToDevice));
      }
      if (!c->hostWeights.empty()) {
        VR_HIP(c, c->dHostWeights.ensure(c->hostWeights.size()));
        VR_HIP(c, hipMemcpy(c->dHostWeights.p, c->hostWeights.data(), c->hostWeights.size() * 4, hipMemcpyHostToDevice));
      }
    }
    c->sourceDirty = false;
  }

  // ---- ray stream: one batch of up to 2^27 rays; larger launches run several batches ----------
  const uint64_t span = last - first;
  uint32_t cap = (uint32_t)std::min<uint64_t>(span, 1ull << 27);
  if (const char *e = std::getenv("VR_BATCH_RAYS"))
    cap = (uint32_t)std::min<uint64_t>(span, std::max<long long>(256, std::atoll(e)));
  cap = std::max<uint32_t>(cap, 1u);
  c->batchCap = cap;
  // (Overlapping the generator of batch b+1 on a second stream with the tracer of batch b was measured slower in every
  //  round — 13.4 against 11.3 ms per C2 step in round 3: both kernels want the same issue slots and smaller batches
  //  sort less coherently — and is gone from the code.)
  // sort bins: far-plane cells holding ~40 rays each, VR_BIN_CAP slots (measured: 64 / 32 -> 128 / 40: generator
  // 5.0 -> 4.75 ms, C2 +2.5 %)
  {
    uint32_t binCap = VR_BIN_CAP, perBin = 40;
    if (const char *e = std::getenv("VR_BIN_CAP"))
      binCap = (uint32_t)std::max(8, std::atoi(e));
    if (const char *e = std::getenv("VR_RAYS_PER_BIN"))
      perBin = (uint32_t)std::max(1, std::atoi(e));
    p.binCap = binCap;
    c->raysPerBin = perBin;
    uint32_t nb;
    size_bins(D, cap, perBin, p, nb);
    c->numBins = nb;
    size_t slots = (size_t)nb * binCap + cap; // bins + overflow region
    size_t cntWords = (size_t)nb + 1;
    if (c->reliefScene) { // the loose bins with an overflow region of their own, behind the tight ones (size_loose)
      TraceParams q = p;
      q.numBins = nb;
      q.ovCap = cap;
      size_loose(D, q);
      slots = (size_t)q.looseSlotBase + (size_t)q.looseNumBins * binCap + cap;
      cntWords = (size_t)q.looseCntBase + q.looseNumBins + 1;
      // (the loose launch numbers its slots from looseSlotBase on, and bit 31 of such a number marks a spill-queue record)
      if (slots >= (1ull << 32) || (size_t)q.looseNumBins * binCap + cap >= (1ull << 31))
        return fail(c, VR_E_STATE, "ray stream too large for 32-bit record slots (relief bins)");
    }
    c->slotStride = slots;
    // 32-byte records for every particle (vr_types.hpp); a non-absorbing particle under a source whose origin plane or
    // draw count varies (tilted, grid, host rays) adds 16 bytes per ray in a side array
    c->recExtra = !c->absorb && (c->usePrimaryDirection || !c->gridPoints.empty() || !c->hostOrg.empty());
    const size_t recFloats = 8;
    if (c->recExtra)
      VR_HIP(c, c->dRecExtra.ensure_grow((size_t)cap * 4));
    size_t slotsWant = slots, binsWant = cntWords;
    if (c->reserveRays > span) { // vr_reserve_rays: room for the largest apply() announced
      TraceParams q = p;
      uint32_t nbR = 0;
      const uint32_t capR = (uint32_t)std::min<uint64_t>(c->reserveRays, 1ull << 27);
      size_bins(D, capR, perBin, q, nbR);
      size_t sR = (size_t)nbR * binCap + capR, bR = (size_t)nbR + 1;
      if (c->reliefScene) {
        q.numBins = nbR;
        q.ovCap = capR;
        size_loose(D, q);
        sR = (size_t)q.looseSlotBase + (size_t)q.looseNumBins * binCap + capR;
        bR = (size_t)q.looseCntBase + q.looseNumBins + 1;
      }
      slotsWant = std::max(slotsWant, sR);
      binsWant = std::max(binsWant, bR);
    }
    VR_HIP(c, c->dSlotRec.ensure_grow(slotsWant * recFloats));
    VR_HIP(c, c->dBinCount.ensure_grow(binsWant));
  }

  // launch geometry of the persistent kernels
  {
    // absorbing particles: a (nearly) flat surface is served by packets alone; a structured one
    // ends most rounds in per-lane walks and wants the straggler carry-over (MODE 2)
    // general particles on a flat surface of disks: the general kernel with the packet query's crediting (MODE 3)
    // (the lean extended kernel P_EXT — data labels, per-material sticking, global data — has the packet query's
    //  crediting too; P_EXT_FULL, the instantiation with the rare options, stays on MODE 0)
    // "flat": 95 % of the surface shown to the source lies in one plane AND the scene box is thin along the source
    // axis — the packet query clips its rays to that box, and a box half a grid cell thick already lets the few
    // grazing rays of a wave stretch its query over dozens of primitives (a 10^6-disk plane with ONE 50 x 50 bump of
    // 0.3 cells: the absorbing kernel 6.4 -> 8.3 ms, the general one 11 -> 18; the kernels for structured scenes are
    // then 2 - 6 % ahead of the flat ones.  DESIGN.md section 10: a flat layer + relief decomposition would close this)
    c->traceMode = !c->absorb ? ((flatScene && c->geo.geo == 0 && c->kernelParticle <= (int)P_EXT) ? 3 : 0)
                              : (flatScene ? 1 : 2);
    c->looseMode = c->traceMode;
    if (c->reliefScene) { // flat with relief: the flat-scene kernels on the tight bins, the structured-scene ones on the loose
      c->traceMode = c->absorb ? 5 : 6;
      if (!c->absorb && !std::getenv("VR_NO_SPILL"))
        c->looseMode = 7; // ... which also resume the rays the tight general kernel spills (TraceParams::spillRec)
    }
    if (const char *e = std::getenv("VR_GENERAL_FLAT"))
      if (!c->absorb && c->geo.geo == 0 && c->kernelParticle <= (int)P_EXT)
        c->traceMode = std::atoi(e) ? 3 : 0;
    if (const char *e = std::getenv("VR_ABSORB_CARRY"))
      if (c->absorb)
        c->traceMode = std::atoi(e) ? 2 : 1;
    if (smallScene)
      c->traceMode = 4;
    if (c->traceMode != 5 && c->traceMode != 6) { // (an environment switch above took the mode back)
      c->reliefScene = false;
      p.reliefCoarse = nullptr;
    }
    int blocks = 1;
    c->userKernel = nullptr;
    if (c->userModel >= 0) { // the kernel of the model's own code object
      const UserModel &um = c->userModels[c->userModel];
      auto it = um.kernels.find(D * 100 + c->geo.geo * 10 + c->traceMode);
      if (it == um.kernels.end())
        return fail(c, VR_E_STATE, "run-time particle model: no kernel for this geometry / mode in its code object");
      c->userKernel = it->second;
      int nb = 0;
      if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, c->userKernel, VR_BLOCK, c->traceMode == 4 ? p.smallBytes : 0) != hipSuccess)
        nb = 2;
      blocks = std::max(1, nb);
    } else {
      blocks = std::max(1, trace_blocks_per_cu(D, c->geo.geo, c->kernelParticle, c->traceMode, p.smallBytes));
    }
    // a small launch does better on fewer persistent waves: every wave pays its start-up and its tail.  Best grid on
    // P(100), blocks per CU (tools/small_launch.py): 3 10^5 rays 1, 6 10^5 2, 10^6 3, 2 - 3 10^6 4, 10^7 and more all of
    // them — about sqrt(rays / 10^5).  10^6 rays: 0.69 -> 0.49 ms (absorbing 0.47 -> 0.31)
    if (c->traceMode != 4)
      blocks = std::min(blocks, std::max(1, (int)std::lround(std::sqrt((double)cap / 1e5))));
    if (const char *e = std::getenv("VR_TRACE_BLOCKS"))
      blocks = std::max(1, std::atoi(e));
    c->grid = (unsigned)c->numCUs * (unsigned)blocks;
    c->looseGrid = 0;
    if (c->reliefScene) { // (the loose bins hold about a tenth of the rays)
      int lb = std::max(1, trace_blocks_per_cu(D, c->geo.geo, c->kernelParticle, c->looseMode, 0));
      lb = std::min(lb, std::max(1, (int)std::lround(std::sqrt((double)cap / 1e6))));
      if (const char *e = std::getenv("VR_LOOSE_BLOCKS"))
        lb = std::max(1, std::atoi(e));
      c->looseGrid = (unsigned)c->numCUs * (unsigned)lb;
    }
  }
  // deep part of the per-lane walk's stack (entries beyond the LDS-resident ones), one slab per resident wave
  {
    const size_t waves = (size_t)std::max<unsigned>(c->grid, (unsigned)c->numCUs * 8u) * (VR_BLOCK / 64);
    if (waves > c->walkStackWaves) {
      VR_HIP(c, c->dWalkStack.ensure(waves * (size_t)VR_STACK_GLOBAL * 64u));
      c->walkStackWaves = waves;
    }
  }
  // tier-2 RNG slabs (312 x 64 words per resident wave): only a kernel that can draw more than
  // 156 numbers per ray touches them — the general trace kernel and the tilted-source generator
  {
    size_t waves = 0;
    if (!c->absorb)
      waves = (size_t)std::max(c->grid, c->looseGrid) * (VR_BLOCK / 64);
    if (c->usePrimaryDirection || !c->hostOrg.empty())
      waves = std::max(waves, (size_t)c->numCUs * 8u * (VR_BLOCK / 64)); // launch_gen's grid bound
    if (waves > c->scratchWaves) {
      VR_HIP(c, c->dScratch.ensure(waves * 312u * 64u));
      c->scratchWaves = waves;
    }
  }

  p.nodes = c->dNodes.p;
  p.qnodes = c->dQNodes.p;
  p.pnodes = c->dPNodes.p;
  p.walkStack = c->dWalkStack.p;
  p.numNodes = c->numNodes;
  for (int k = 0; k < 3; ++k) {
    p.qbase[k] = c->qbase[k];
    p.qscale[k] = c->qscale[k];
  }
  p.prims = c->dPrims.p;
  p.wide = c->haveWide ? c->dWide.p : nullptr;
  p.wideTopFirst = c->wideRoot[0];
  p.wideTopCount = c->wideRoot[1];
  p.widePrimBase = c->wideRoot[2];
  p.pqMaxFrontier = 12;
  if (const char *e = std::getenv("VR_PQ_FRONTIER"))
    p.pqMaxFrontier = (uint32_t)std::min(24, std::max(1, std::atoi(e))); // (<= 24: the cached frontier shares the lists with its box)
  p.pqMaxCand = 24;
  if (const char *e = std::getenv("VR_PQ_CAND"))
    p.pqMaxCand = (uint32_t)std::min(24, std::max(1, std::atoi(e))); // (2 * pqMaxCand + 1 records fit VR_PQ_CANDS)
  {
    float scale = 1e-3f;
    for (int k = 0; k < 3; ++k) {
      p.sceneLo[k] = c->sceneLo[k];
      p.sceneHi[k] = c->sceneHi[k];
      scale = std::max(scale, std::max(std::fabs(c->sceneLo[k]), std::fabs(c->sceneHi[k])));
    }
    p.nbDist = 2 * c->geo.diskRadius;
    p.geoD = D;
    p.pqPad = 1e-5f * scale; // >> the rounding of the clip (1e-7 relative); the boxes carry their own 4e-6 pad
  }
  p.nbOff = c->dNbOff.p;
  p.nbIds = c->dNbIds.p;
  p.primSticking = dStick;
  { // wall table + scalar frame: one slot per particle of the apply (each kernel stages its own launch's frame in LDS)
    const size_t nSlots = std::max<size_t>(1, c->specs.size());
    if (c->dWalls.cap < nSlots * VR_WALL_TABLE)
      VR_HIP(c, c->dWalls.ensure(nSlots * VR_WALL_TABLE));
    if (c->frameHostAll.size() < nSlots * VR_WALL_TABLE)
      c->frameHostAll.assign(nSlots * VR_WALL_TABLE, 0.f);
  }
  p.wallTable = c->dWalls.p + (size_t)c->counterSlot * VR_WALL_TABLE;
  p.planeStride = c->accStride * c->accReplicas;
  p.fluxAcc = c->dFluxAcc.p + (size_t)c->dataBase * p.planeStride; // (this particle's planes)
  p.accStride = c->accStride;
  p.numData = c->numData;
  p.particleKind = c->particleKind;
  p.meanFreePath = c->meanFreePath;
  std::memcpy(p.particleParams, c->particleParams, sizeof(p.particleParams));
  p.globalVec = (c->globalStride && !c->globalVecs.empty()) ? c->dGlobalVec.p : nullptr;
  p.globalScalars = c->globalScalars.empty() ? nullptr : c->dGlobalScalars.p;
  p.numGlobalVec = p.globalVec ? (uint32_t)c->globalVecs.size() : 0u;
  p.globalStride = c->globalStride;
  p.numGlobalScalars = (uint32_t)c->globalScalars.size();
  p.useWdist = c->useWdist ? 1 : 0;
  p.gridPoints = c->gridPoints.empty() ? nullptr : c->dGrid.p;
  p.gridCount = (uint32_t)(c->gridPoints.size() / 3);
  p.eeGrid = 2.f / (c->sourcePower + 1); // raySourceGrid.hpp:22
  p.hostOrg = c->hostOrg.empty() ? nullptr : c->dHostOrg.p;
  p.hostDir = c->hostOrg.empty() ? nullptr : c->dHostDir.p;
  p.hostDraws = c->hostDraws.empty() ? nullptr : c->dHostDraws.p;
  p.hostWeights = (c->hostOrg.empty() || c->hostWeights.empty()) ? nullptr : c->dHostWeights.p;
  p.accMask = c->accReplicas - 1u;
  VR_HIP(c, c->dCounters.ensure(80 * std::max<size_t>(1, c->specs.size())));
  p.counters = c->dCounters.p + 80 * (size_t)c->counterSlot;
  VR_HIP(c, c->dWorkQ.ensure(VR_QUEUES * VR_QUEUE_STRIDE));
  p.workCounter = c->dWorkQ.p;
  p.numQueues = VR_QUEUES;
  p.recExtra = c->recExtra ? c->dRecExtra.p : nullptr;
  p.spillRec = nullptr;
  p.spillCount = nullptr;
  if (c->reliefScene && c->looseMode == 7) {
    // (a record per ray of a batch + the unused end of every wave's last 64-record block)
    VR_HIP(c, c->dSpillRec.ensure_grow(((size_t)c->batchCap + (size_t)c->grid * (VR_BLOCK / 64) * 64u) * 16));
    VR_HIP(c, c->dSpillCount.ensure(1));
    p.spillRec = c->dSpillRec.p;
    p.spillCount = c->dSpillCount.p;
  }
  {
    // the packet query's search margin (frontier reuse over neighbouring rounds, flat-scene kernels): in units of the
    // neighbourhood distance 2 r (disks) / 1.7 grid cells (triangles); pqMaxFrontier <= 24 entries fit the cached lists
    float mg = 1.5f;
    if (const char *e = std::getenv("VR_PQ_MARGIN"))
      mg = std::max(0.f, (float)std::atof(e));
    p.pqMargin = mg * (c->geo.geo == 0 ? 2.f * c->geo.diskRadius : 1.7f * c->geo.gridDelta);
  }
  p.rngScratch = c->dScratch.p;
  p.slotRec = c->dSlotRec.p;
  p.binCount = c->dBinCount.p;
  p.idxList = nullptr;
  p.batchFirst = first;
  p.batchCount = 0;
  p.ovCap = c->batchCap;
  p.numBins = c->numBins;
  p.seed = seed;
  p.numPrims = N;
  p.maxReflections = c->maxReflections;
  p.maxBoundaryHits = c->maxBoundaryHits;
  p.chunk = 64;
  p.rayDir = c->ts[0];
  p.firstDir = c->ts[1];
  p.secondDir = c->ts[2];
  p.minMax = c->ts[3];
  p.posNeg = (float)c->ts[4];
  p.ee = 1.f / (c->sourcePower + 1); // raySourceRandom.hpp:21
  p.sticking = c->sticking;
  p.bc0 = c->boundaryConds[0];
  p.bc1 = c->boundaryConds[1];
  p.useBasis = c->usePrimaryDirection ? 1 : 0;
  if (c->usePrimaryDirection)
    host_orthonormal_basis(c->primaryDirection, p.basis);
  else
    std::memset(p.basis, 0, sizeof(p.basis));
  p.srcCoord = c->ts[3] ? c->bbHi[c->ts[0]] : c->bbLo[c->ts[0]];
  p.lo1 = c->bbLo[c->ts[1]];
  p.hi1 = c->bbHi[c->ts[1]];
  p.lo2 = c->bbLo[c->ts[2]];
  p.hi2 = c->bbHi[c->ts[2]];
  {
    const float lr = c->bbLo[c->ts[0]], hr = c->bbHi[c->ts[0]];
    float scale = 0.f;
    for (int k = 0; k < 3; ++k)
      scale = std::max(scale, std::max(std::fabs(c->bbLo[k]), std::fabs(c->bbHi[k])));
    const float margin = 1e-3f * std::max(scale, hr - lr) + 1e-6f;
    p.wallLoR = lr - margin;
    p.wallHiR = hr + margin;
  }
  p.keyCoord = c->keyCoord;
  if (const char *e = std::getenv("VR_KEY_COORD"))
    p.keyCoord = (float)std::atof(e);
  p.invExt1 = (p.hi1 > p.lo1) ? 1.f / (p.hi1 - p.lo1) : 0.f;
  p.invExt2 = (p.hi2 > p.lo2) ? 1.f / (p.hi2 - p.lo2) : 0.f;
  p.packetBudget = 128;
  if (const char *e = std::getenv("VR_PACKET_BUDGET"))
    p.packetBudget = (uint32_t)std::max(0, std::atoi(e));
  // (share of parked lanes at which the pending leaves are tested: sweep 10 / 18 / 25 / 35 / 50 — disks flat between
  //  18 and 35; triangles, whose leaf test is the longer one, 10: trenchMesh 0.1 28.3 -> 27.5 ms, C4 20.6 -> 20.2)
  p.walkPark = c->geo.geo == 1 ? 10 : 25;
  if (const char *e = std::getenv("VR_WALK_PARK"))
    p.walkPark = (uint32_t)std::min(100, std::max(1, std::atoi(e)));
  p.walkExit = 16; // (sweep 12 .. 36: 12 - 20 within 1 %, 36 slower by 4 - 7 %)
  if (const char *e = std::getenv("VR_WALK_EXIT"))
    p.walkExit = (uint32_t)std::min(64, std::max(1, std::atoi(e)));
  p.packetRatio = 3;
  if (const char *e = std::getenv("VR_PACKET_RATIO"))
    p.packetRatio = (uint32_t)std::max(1, std::atoi(e));
  p.debugFlags = 0;
  if (const char *e = std::getenv("VR_DEBUG_FLAGS"))
    p.debugFlags = (uint32_t)std::atoi(e);
  { // the launch's scalar frame, staged in LDS by the trace kernels (VR_F_*, vr_device.hpp)
    float *const slotHost = c->frameHostAll.data() + (size_t)c->counterSlot * VR_WALL_TABLE;
    std::memcpy(slotHost, c->wallsHost, sizeof(c->wallsHost));
    float *f = slotHost + 96;
    auto bits = [](int32_t v) {
      float r;
      std::memcpy(&r, &v, 4);
      return r;
    };
    f[0] = p.srcCoord;
    f[1] = bits(p.rayDir);
    f[2] = bits(p.firstDir);
    f[3] = bits(p.secondDir);
    f[4] = f[5] = 0.f; // (the records' side-array address: written by the kernel from its own argument)
    f[6] = p.lo1;
    f[7] = p.hi1;
    f[8] = p.lo2;
    f[9] = p.hi2;
    f[10] = p.wallLoR;
    f[11] = p.wallHiR;
    for (int k = 0; k < 3; ++k) {
      f[12 + k] = p.sceneLo[k];
      f[15 + k] = p.sceneHi[k];
    }
    f[18] = p.pqPad;
    f[19] = bits(p.bc0);
    f[20] = bits(p.bc1);
    f[21] = p.nbDist;
    for (int k = 22; k < VR_WALL_TABLE - 96; ++k)
      f[k] = 0.f;
    if (c->reliefScene) { // the relief field's fine tiles (VR_F_RF_*: relief_clip, vr_device.hpp)
      const ReliefParams &q = c->rf;
      f[32] = q.lo1;
      f[33] = q.lo2;
      f[34] = q.invTile;
      f[35] = q.tile;
      f[36] = bits(q.nx);
      f[37] = bits(q.ny);
      const uint64_t addr = (uint64_t)(uintptr_t)q.fine;
      f[38] = bits((int32_t)(uint32_t)(addr & 0xFFFFFFFFull));
      f[39] = bits((int32_t)(uint32_t)(addr >> 32));
    }
    // height field over the source plane: for particles that go on after a hit ("segments that rise clear", vr_trace.hip)
    if (!c->absorb && c->geo.numPrims && !std::getenv("VR_NO_HEIGHT_FIELD")) {
      const bool stale = c->hfBuild != c->bvhBuilds || c->hfAxes[0] != c->ts[0] || c->hfAxes[1] != c->ts[1] ||
                         c->hfAxes[2] != c->ts[2] || c->hfAxes[3] != c->ts[3];
      if (stale) {
        HeightFieldParams &q = c->hf;
        q.prims = c->dPrims.p;
        q.n = c->geo.numPrims;
        q.geo = c->geo.geo;
        q.ax = c->ts[0];
        q.a1 = c->ts[1];
        q.a2 = c->ts[2];
        q.sign = c->ts[3] ? 1.f : -1.f; // (ts[3]: the source plane lies at the max side)
        const float ext1 = c->sceneHi[q.a1] - c->sceneLo[q.a1], ext2 = D == 3 ? c->sceneHi[q.a2] - c->sceneLo[q.a2] : 0.f;
        float cells = 4.f; // (tile side in grid cells; sweep 2 / 3 / 4 / 6 / 8: see DESIGN.md 7)
        if (const char *e = std::getenv("VR_HF_TILE"))
          cells = std::max(0.25f, (float)std::atof(e));
        float tile = std::max(cells * c->geo.gridDelta, std::max(ext1, ext2) / 256.f);
        if (!(tile > 0.f))
          tile = 1.f;
        q.lo1 = c->sceneLo[q.a1];
        q.lo2 = D == 3 ? c->sceneLo[q.a2] : 0.f;
        q.invTile = 1.f / tile;
        q.nx = std::max(1, std::min(256, (int)std::ceil(ext1 / tile)));
        q.ny = D == 3 ? std::max(1, std::min(256, (int)std::ceil(ext2 / tile))) : 1;
        float scale = 1e-3f;
        for (int k = 0; k < 3; ++k)
          scale = std::max(scale, std::max(std::fabs(c->sceneLo[k]), std::fabs(c->sceneHi[k])));
        q.pad = 8e-7f * scale; // (a dozen ulp of the largest coordinate: see DESIGN.md 5.2)
        VR_HIP(c, c->dHfRaw.ensure((size_t)q.nx * q.ny));
        VR_HIP(c, c->dHf.ensure((size_t)q.nx * q.ny));
        q.raw = c->dHfRaw.p;
        q.field = c->dHf.p;
        VR_HIP(c, launch_height_field(q, c->stream));
        c->hfBuild = c->bvhBuilds;
        for (int k = 0; k < 4; ++k)
          c->hfAxes[k] = c->ts[k];
      }
      const HeightFieldParams &q = c->hf;
      f[22] = q.lo1;
      f[23] = q.lo2;
      f[24] = q.invTile;
      f[25] = 1.f / q.invTile;
      f[26] = (q.sign > 0.f ? c->sceneHi[q.ax] : -c->sceneLo[q.ax]); // above this nothing is left (the BVH's root box)
      f[27] = q.sign;
      f[28] = bits(q.nx);
      f[29] = bits(q.ny);
      const uint64_t addr = (uint64_t)(uintptr_t)q.field;
      f[30] = bits((int32_t)(uint32_t)(addr & 0xFFFFFFFFull));
      f[31] = bits((int32_t)(uint32_t)(addr >> 32));
    }
    VR_HIP(c, hipMemcpyAsync(c->dWalls.p + (size_t)c->counterSlot * VR_WALL_TABLE, slotHost, VR_WALL_TABLE * 4, hipMemcpyHostToDevice, c->stream));
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (redoConfig)
    c->buildSeconds = secs; // a cheap re-prepare (new seed / ray range only) keeps the last build time
  c->prepared = true;
  c->launched = false;
  c->haveResult = false;
  return VR_OK;
}

// sort-bin grid for a batch of `count` rays: far-plane cells holding ~perBin rays each
static void size_bins(int D, uint64_t count, uint32_t perBin, TraceParams &p, uint32_t &numBins) {
  const uint64_t target = std::max<uint64_t>(count / std::max<uint32_t>(perBin, 1u), 1);
  if (D == 2) {
    p.binT1 = (int)std::min<uint64_t>(target, 1u << 22);
    p.binT2 = 1;
    p.binTiles = 1;
    numBins = (uint32_t)p.binT1;
  } else {
    p.binT1 = p.binT2 = (int)std::min<double>(4096.0, std::max(1.0, std::ceil(std::sqrt((double)target))));
    p.binTiles = (p.binT1 + 7) / 8;
    numBins = (uint32_t)p.binTiles * (uint32_t)p.binTiles * 64u;
  }
}

// The LOOSE bins of a scene with relief (TraceParams, round 4): a grid a third as fine per axis as the tight one p.binT*
// describes — they hold the grazing rays, about a tenth of all — whose cursors and record slots (+ an overflow region of
// p.ovCap slots) lie behind the tight bins' in the same two buffers.
static void size_loose(int D, TraceParams &p) {
  p.looseT1 = std::max(1, p.binT1 / 3);
  if (D == 2) {
    p.looseT2 = 1;
    p.looseTiles = 1;
    p.looseNumBins = (uint32_t)p.looseT1;
  } else {
    p.looseT2 = std::max(1, p.binT2 / 3);
    p.looseTiles = (p.looseT1 + 7) / 8;
    p.looseNumBins = (uint32_t)p.looseTiles * (uint32_t)((p.looseT2 + 7) / 8) * 64u;
  }
  p.looseCntBase = (p.numBins + 1u + 3u) & ~3u;
  p.looseSlotBase = p.numBins * p.binCap + p.ovCap;
}

// what one trace launch of a batch needs beyond the prepared parameters
struct LaunchDesc {
  const TraceParams *params;
  unsigned grid;
  int traceMode, kernelParticle;
  bool absorb;
  hipFunction_t userKernel; // a run-time model's kernel, or nullptr
  bool relief = false;      // flat with relief: a second launch (looseMode, looseGrid) traces the loose bins
  int looseMode = 0;
  unsigned looseGrid = 0;
};

static hipEvent_t &event_at(std::vector<hipEvent_t> &v, size_t i, vr_context *c, int &rc) {
  while (v.size() <= i) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) {
      rc = fail(c, VR_E_HIP, "hipEventCreate failed");
      static hipEvent_t none = nullptr;
      return none;
    }
    v.push_back(e);
  }
  return v[i];
}

// the batch's own fields of a particle's launch parameters: sort bins, spans per queue grab, queues
static TraceParams batch_params(vr_context *c, const LaunchDesc &L, uint64_t first, uint32_t count) {
  TraceParams p = *L.params;
  p.batchFirst = first;
  p.batchCount = count;
  uint32_t nbBatch = c->numBins;
  size_bins(c->geo.D, count, c->raysPerBin, p, nbBatch); // (<= the grid the buffers were sized for)
  nbBatch = std::min(nbBatch, c->numBins);
  p.numBins = nbBatch;
  if (L.relief)
    size_loose(c->geo.D, p); // (p.ovCap = the batch capacity: the tight bins' overflow region keeps its full size)
  else
    p.reliefCoarse = nullptr;
  {
    // bins per queue grab: ~1024 rays for big batches, but never so many that a small
    // batch (a short last one, a small launch) is handed to a few waves only
    const uint64_t waves = std::min<uint64_t>(L.grid, ((uint64_t)count + 255) / 256) * (VR_BLOCK / 64);
    // (a grab of the queue costs two dependent trips to memory: the packet kernels want long spans; the
    //  general kernel's rounds are long and its bounce chains uneven: shorter spans balance its tail)
    // (a round that straddles two spans mixes rays of two places: its packet query gives up — every failed query of a flat
    //  plane is one of these, 4.9 % of the rounds at 32 bins, 2 % at 64 — which costs the absorbing kernel nothing
    //  measurable but the general flat-scene kernels 3 % (their failed round also loses its follow-up segments))
    uint64_t spanBins = L.traceMode == 0 ? 16 : ((L.traceMode == 3 || L.traceMode == 6) ? 64 : 32);
    if (const char *e = std::getenv("VR_SPAN_BINS"))
      spanBins = (uint64_t)std::min(64, std::max(1, std::atoi(e)));
    p.chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(spanBins, nbBatch / std::max<uint64_t>(waves * 2, 1)));
  }
  p.workCounter = c->dWorkQ.p;
  // One queue per XCD pays where neighbouring rounds share primitive records that do not fit an XCD's 4 MiB L2 and the
  // work per bin is even: flat scenes of more than ~10^5 primitives (measured, VR_QUEUES=1 / 8 on one box: C2 sticking
  // 0.1 15.62 -> 15.16 ms, C2 1.0 6.67 -> 6.62, plane 100^2 +-0; L2 hit rate of the C2 launch 74 -> 84 %, fabric reads
  // 9.0 -> 5.2 GB).  A structured scene is L2 resident anyway and its bins differ in cost — an eighth of the trench is
  // not an eighth of the work: trench3D +3 %, C5 +6 %: one queue.
  const bool flat = L.traceMode == 3; // (the absorbing kernels have the single queue compiled in)
  p.numQueues = (flat && c->geo.numPrims > (1u << 17) && nbBatch >= 64u * VR_QUEUES * p.chunk) ? VR_QUEUES : 1u;
  if (const char *e = std::getenv("VR_QUEUES"))
    p.numQueues = std::atoi(e) >= (int)VR_QUEUES ? VR_QUEUES : 1u;
  return p;
}

// One batch of the ray stream: ONE generator pass straight into the sort bins, then the trace kernel of every
// particle of `group` over the same records (particles of a group share source distribution and record format).
static int run_batch(vr_context *c, const std::vector<LaunchDesc> &group, uint64_t first, uint32_t count, size_t &genNo,
                     size_t &traceNo) {
  int rc = VR_OK;
  const bool keepRng = !group[0].absorb; // records carry the RNG cursors
  const TraceParams pg = batch_params(c, group[0], first, count);
  VR_HIP(c, hipMemsetAsync(pg.binCount, 0, (pg.reliefCoarse ? (size_t)pg.looseCntBase + pg.looseNumBins + 1 : (size_t)pg.numBins + 1) * 4, c->stream));
  hipEvent_t g0 = event_at(c->evG, 2 * genNo, c, rc), g1 = event_at(c->evG, 2 * genNo + 1, c, rc);
  if (rc != VR_OK)
    return rc;
  VR_HIP(c, hipEventRecord(g0, c->stream));
  VR_HIP(c, launch_gen(pg, c->geo.D, keepRng, (unsigned)c->numCUs * 8u, c->stream));
  VR_HIP(c, hipEventRecord(g1, c->stream));
  ++genNo;
  for (const LaunchDesc &L : group) {
    const TraceParams p = &L == &group[0] ? pg : batch_params(c, L, first, count);
    VR_HIP(c, hipMemsetAsync(p.workCounter, 0, VR_QUEUES * VR_QUEUE_STRIDE * 8, c->stream));
    if (p.spillCount)
      VR_HIP(c, hipMemsetAsync(p.spillCount, 0, 4, c->stream));
    hipEvent_t k0 = event_at(c->evK, 2 * traceNo, c, rc), k1 = event_at(c->evK, 2 * traceNo + 1, c, rc);
    if (rc != VR_OK)
      return rc;
    VR_HIP(c, hipEventRecord(k0, c->stream));
    // a small batch does not need the whole persistent grid: one wave per 64 rays is plenty
    const unsigned gridBatch = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(L.grid, ((uint64_t)count + 255) / 256));
    if (L.relief && std::getenv("VR_SKIP_TIGHT")) { // (diagnostics: the loose launch alone — the result is incomplete)
    } else if (L.userKernel) {
      TraceParams pk = p;
      void *args[] = {&pk};
      VR_HIP(c, hipModuleLaunchKernel(L.userKernel, gridBatch, 1, 1, VR_BLOCK, 1, 1, L.traceMode == 4 ? p.smallBytes : 0, c->stream,
                                      args, nullptr));
    } else {
      VR_HIP(c, launch_trace(p, c->geo.D, c->geo.geo, L.kernelParticle, L.traceMode, gridBatch, c->stream));
    }
    VR_HIP(c, hipEventRecord(k1, c->stream));
    ++traceNo;
    if (L.relief && !std::getenv("VR_SKIP_LOOSE")) { // (VR_SKIP_LOOSE: diagnostics, the tight launch alone — the result is incomplete)
      // the loose bins (the grazing rays, filed apart by the generator): the kernel for structured scenes over the second
      // set of bins — the same buffers from their loose parts on, a single queue
      TraceParams q = p;
      q.binCount = p.binCount + p.looseCntBase;
      q.slotRec = p.slotRec + (size_t)p.looseSlotBase * 8;
      q.numBins = p.looseNumBins;
      q.reliefCoarse = nullptr;
      q.numQueues = 1;
      {
        const uint64_t waves = std::min<uint64_t>(L.looseGrid, ((uint64_t)count / 8 + 255) / 256) * (VR_BLOCK / 64);
        q.chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(16, q.numBins / std::max<uint64_t>(waves * 2, 1)));
      }
      VR_HIP(c, hipMemsetAsync(q.workCounter, 0, VR_QUEUES * VR_QUEUE_STRIDE * 8, c->stream));
      hipEvent_t l0 = event_at(c->evK, 2 * traceNo, c, rc), l1 = event_at(c->evK, 2 * traceNo + 1, c, rc);
      if (rc != VR_OK)
        return rc;
      VR_HIP(c, hipEventRecord(l0, c->stream));
      const unsigned gridLoose = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(L.looseGrid, ((uint64_t)count / 4 + 255) / 256));
      VR_HIP(c, launch_trace(q, c->geo.D, c->geo.geo, L.kernelParticle, L.looseMode, gridLoose, c->stream));
      VR_HIP(c, hipEventRecord(l1, c->stream));
      ++traceNo;
    }
  }
  return VR_OK;
}

int vr_apply_launch(vr_context *c) {
  if (!c)
    return VR_E_INVALID;
  if (!c->prepared)
    return fail(c, VR_E_STATE, "vr_apply_launch: call vr_apply_prepare first");
  VR_HIP(c, hipSetDevice(c->device));
  const uint32_t N = c->geo.numPrims;
  const size_t nPart = std::max<size_t>(1, c->specs.size());
  VR_HIP(c, hipMemsetAsync(c->dFluxAcc.p, 0, (size_t)c->accStride * c->accReplicas * c->totalData * 8, c->stream));
  VR_HIP(c, hipMemsetAsync(c->dCounters.p, 0, 80 * nPart * 8, c->stream));
  VR_HIP(c, hipEventRecord(c->ev0, c->stream));
  // groups of particles that can share a generator pass: the same source distribution (cosine power: the rays of
  // index idx are then identical, gpu/raygTrace.hpp launches every particle with the apply's one seed) and the
  // same record format (with / without the RNG cursors)
  std::vector<std::vector<LaunchDesc>> groups;
  if (nPart == 1) {
    groups.push_back({LaunchDesc{&c->params, c->grid, c->traceMode, c->kernelParticle, c->absorb, c->userKernel, c->reliefScene,
                                 c->looseMode, c->looseGrid}});
  } else {
    for (const ParticleLaunch &L : c->launches) {
      const LaunchDesc d{&L.params, L.grid, L.traceMode, L.kernelParticle, L.absorb, L.userKernel, L.relief, L.looseMode, L.looseGrid};
      bool placed = false;
      for (auto &g : groups)
        if (g[0].absorb == d.absorb && g[0].params->ee == d.params->ee && g[0].params->eeGrid == d.params->eeGrid &&
            g[0].relief == d.relief) { // (relief: the generator's bins are laid out differently)
          g.push_back(d);
          placed = true;
          break;
        }
      if (!placed)
        groups.push_back({d});
    }
  }
  c->numGenLaunches = c->numTraceLaunches = 0;
  for (const auto &g : groups)
    for (uint64_t f = c->rayFirstLaunch; f < c->rayEndLaunch; f += c->batchCap) {
      const uint32_t cnt = (uint32_t)std::min<uint64_t>(c->batchCap, c->rayEndLaunch - f);
      int r = run_batch(c, g, f, cnt, c->numGenLaunches, c->numTraceLaunches);
      if (r != VR_OK)
        return r;
    }
  VR_HIP(c, hipEventRecord(c->ev1, c->stream));
  {
    unsigned headroom = 0; // the sums of `worldSize` ranks must still fit a signed int64 (vr_set_world_size)
    while ((1u << headroom) < c->worldSize)
      ++headroom;
    for (uint32_t l = 0; l < c->totalData; ++l)
      VR_HIP(c, launch_gather_flux(c->dFluxAcc.p + (size_t)l * c->accStride * c->accReplicas, c->accStride, c->accReplicas,
                                   c->dLeafOfOrig.p, N, c->fluxOut() + (size_t)l * N, headroom, c->dCounters.p + 61, c->stream));
  }
  c->launched = true;
  return VR_OK;
}

static void info_from_counters(vr_trace_info &i, const unsigned long long *cnt) {
  i.totalRaysTraced = cnt[C_TRACES];
  i.nonGeometryHits = cnt[C_NONGEO];
  i.geometryHits = cnt[C_GEO];
  i.particleHits = cnt[C_PARTICLE];
  i.boundaryHits = cnt[C_BOUNDARY];
  i.reflections = cnt[C_REFLECTIONS];
  i.raysTerminated = cnt[C_TERMINATED];
  i.rngFullStates = cnt[C_TIER2];
}

int vr_apply_finish(vr_context *c) {
  if (!c)
    return VR_E_INVALID;
  if (!c->launched)
    return fail(c, VR_E_STATE, "vr_apply_finish: nothing launched");
  VR_HIP(c, hipSetDevice(c->device));
  VR_HIP(c, hipStreamSynchronize(c->stream));
  const size_t nPart = std::max<size_t>(1, c->specs.size());
  std::vector<unsigned long long> all(80 * nPart); // per particle: [0..7] TraceInfo counters, [60] the walk's stack-overflow flag
  VR_HIP(c, hipMemcpy(all.data(), c->dCounters.p, all.size() * 8, hipMemcpyDeviceToHost));
  const unsigned long long *cnt = all.data();
#ifdef VR_DIAG
  { // lane-occupancy diagnostics of a -DVR_DIAG build (see vr_trace.hip)
    unsigned long long dg[32];
    VR_HIP(c, hipMemcpy(dg, c->dCounters.p + 16, sizeof(dg), hipMemcpyDeviceToHost));
    static const char *names[16] = {"rounds", "walk steps", "leaf prim tests", "packet visits", "packet prim tests",
                                    "state machine", "neighbour iters", "walk steps: unfinished", "refill reps", "wall init",
                                    "roulette", "credit", "pq attempts", "pq done", "visits: no child hit", "visits: both hit"};
    for (int k = 0; k < 16; ++k)
      if (dg[2 * k])
        std::fprintf(stderr, "diag %-18s wave-iters %12llu  lane-iters %14llu  (%.1f lanes)\n", names[k], dg[2 * k],
                     dg[2 * k + 1], (double)dg[2 * k + 1] / (double)dg[2 * k]);
    unsigned long long ph[16];
    VR_HIP(c, hipMemcpy(ph, c->dCounters.p + 64, sizeof(ph), hipMemcpyDeviceToHost));
    static const char *pn[16] = {"refill", "packets", "walk: search", "walk: leaf tests", "walls", "state machine + credit",
                                 "packet-query credit", "tail", "  (of state machine) neighbour loop", "  (of state machine) reflection + roulette; (absorbing kernels: of packets) packet query: record loads + box tests of the last level", "  (of state machine) from its start to the back-face test (vote, miss / wall branches, normal fetch)", "  (of state machine) boundary hit",
                                 "  (of state machine) up to the aggregation vote", "  (of state machine) up to the end counters", "  (of packets) packet query: descent of the 64-ary tree", "  (of packets) packet query: exact tests of the candidates"};
    double tot = 0;
    for (int k = 0; k < 8; ++k)
      tot += (double)ph[k];
    for (int k = 0; k < 16; ++k)
      if (ph[k])
        std::fprintf(stderr, "phase %-24s %5.1f %% of wave time\n", pn[k], 100.0 * (double)ph[k] / tot);
  }
#endif
  // A flux accumulator ran out of range (gather_flux_kernel): 2^23 = 8.39e6 weight units per primitive and data label
  // in one apply() — divided by the rank count rounded up to a power of two — is what int64 at 2^-40 holds (signed: the
  // multi-GPU all-reduce).  The reference's float sums stall near 2^24; these would wrap: the apply fails instead.
  if (all[61]) {
    c->launched = false;
    c->prepared = false;
    c->info.error = 1;
    ++c->runNumber; // (the apply happened, like one that ends in the reference's error flag: the seeds move on)
    return fail(c, VR_E_STATE, "flux accumulator overflow: a primitive collected more than 2^23 (8.39e6) weight units per rank-power-of-two "
                               "in one apply() (int64 fixed point, 2^-40 per unit) - result discarded; trace fewer rays per apply() "
                               "and sum the normalised results");
  }
  for (size_t q = 0; q < nPart; ++q) {
    // the walk's stack ran out (a tree deeper than SD + VR_STACK_GLOBAL levels of deferred children): the
    // result would be wrong, so the apply fails
    if (all[80 * q + 60]) {
      c->launched = false;
      c->prepared = false;
      return fail(c, VR_E_STATE, "BVH traversal stack overflow (degenerate tree), or a rank of a sharded apply failed: result discarded");
    }
  }
#ifdef VR_SELFCHECK
  {
    unsigned long long sc[12];
    VR_HIP(c, hipMemcpy(sc, c->dCounters.p + 48, sizeof(sc), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "[vr] self-check: %llu segments disagree with the escape-link walk\n", sc[0]);
    if (sc[0]) {
      float v[8];
      for (int k = 0; k < 8; ++k) {
        const uint32_t u = (uint32_t)sc[2 + k];
        std::memcpy(&v[k], &u, 4);
      }
      std::fprintf(stderr, "[vr]   first: o %.9g %.9g %.9g d %.9g %.9g %.9g  t %.9g pos %u geom %d | ref t %.9g pos %u geom %d\n",
                   v[0], v[1], v[2], v[3], v[4], v[5], v[6], (unsigned)(sc[10] >> 32), (int)(sc[11] >> 32), v[7],
                   (unsigned)(sc[10] & 0xFFFFFFFFu), (int)(sc[11] & 0xFFFFFFFFu));
    }
  }
#endif
  float ms = 0.f;
  VR_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  vr_trace_info &i = c->info;
  i.numRays = c->numRaysLast;
  if (nPart == 1) {
    info_from_counters(i, cnt);
  } else {
    // per particle, and their sums in the context's TraceInfo
    vr_trace_info sum{};
    for (size_t q = 0; q < nPart; ++q) {
      vr_trace_info &pi = c->launches[q].info;
      pi = vr_trace_info{};
      pi.numRays = c->numRaysLast;
      info_from_counters(pi, all.data() + 80 * q);
      sum.totalRaysTraced += pi.totalRaysTraced;
      sum.nonGeometryHits += pi.nonGeometryHits;
      sum.geometryHits += pi.geometryHits;
      sum.particleHits += pi.particleHits;
      sum.boundaryHits += pi.boundaryHits;
      sum.reflections += pi.reflections;
      sum.raysTerminated += pi.raysTerminated;
      sum.rngFullStates += pi.rngFullStates;
    }
    const uint64_t nr = i.numRays;
    const int32_t w = i.warning, e = i.error;
    i = sum;
    i.numRays = nr;
    i.warning = w;
    i.error = e;
  }
  i.timeTrace = ms * 1e-3;
  double kms = 0.0;
  for (size_t b = 0; b < c->numTraceLaunches; ++b) {
    float m = 0.f;
    VR_HIP(c, hipEventElapsedTime(&m, c->evK[2 * b], c->evK[2 * b + 1]));
    kms += m;
    if (std::getenv("VR_PRINT_LAUNCHES")) // (diagnostics: a scene with relief runs two trace launches per batch)
      std::fprintf(stderr, "[vr] trace launch %zu: %.3f ms\n", b, m);
  }
  i.timeTraceKernel = kms * 1e-3;
  if (std::getenv("VR_PRINT_LAUNCHES") && c->params.spillCount) { // (diagnostics: rays the tight general relief kernel handed over)
    uint32_t sp = 0;
    if (hipMemcpy(&sp, c->params.spillCount, 4, hipMemcpyDeviceToHost) == hipSuccess)
      std::fprintf(stderr, "[vr] spilled rays (last batch): %u\n", sp);
  }
  double gms = 0.0;
  for (size_t b = 0; b < c->numGenLaunches; ++b) {
    float m = 0.f;
    VR_HIP(c, hipEventElapsedTime(&m, c->evG[2 * b], c->evG[2 * b + 1]));
    gms += m;
  }
  i.timeGenKernel = gms * 1e-3;
  i.timeBuild = c->buildSeconds;
  i.time = i.timeBuild + i.timeTrace;
  i.bvhRefits = (uint32_t)c->bvhRefits;
  i.bvhBuilds = c->bvhBuilds;
  for (auto &L : c->launches) {
    L.info.timeTrace = i.timeTrace;
    L.info.time = i.time;
    L.info.timeBuild = i.timeBuild;
  }
  ++c->runNumber; // rayTraceDisk.hpp:54
  c->haveSharedSeed = c->keepSharedSeed && c->haveSharedSeed;
  c->launched = false;
  c->prepared = false;
  c->haveResult = true;
  return VR_OK;
}

// Trace::apply() set-up.  One particle: prepare_one.  Several (vr_set_particles): every particle is prepared in
// turn — its kernel variant, launch geometry, per-material sticking, accumulator planes and counter block — with
// ONE seed for the whole apply (gpu/raygTrace.hpp:163-248).
int vr_apply_prepare(vr_context *c) {
  if (!c)
    return VR_E_INVALID;
  if (c->specs.size() <= 1) {
    c->dataBase = 0;
    c->counterSlot = 0;
    return prepare_one(c);
  }
  if (c->useRandomSeed && !c->haveSharedSeed) { // one draw for all particles of this apply
    std::random_device rd;
    c->sharedSeed = (uint32_t)rd();
    c->haveSharedSeed = true;
    c->keepSharedSeed = false;
  }
  for (auto &L : c->launches)
    if (L.primSticking) {
      (void)hipFree(L.primSticking);
      L.primSticking = nullptr;
    }
  c->launches.assign(c->specs.size(), ParticleLaunch{});
  uint32_t base = 0;
  for (size_t q = 0; q < c->specs.size(); ++q) {
    activate_particle(c, c->specs[q]);
    c->dataBase = base;
    c->counterSlot = (uint32_t)q;
    const int r = prepare_one(c);
    if (r != VR_OK) {
      activate_particle(c, c->specs[0]);
      return r;
    }
    ParticleLaunch &L = c->launches[q];
    L.params = c->params;
    L.grid = c->grid;
    L.traceMode = c->traceMode;
    L.kernelParticle = c->kernelParticle;
    L.absorb = c->absorb;
    L.numData = c->numData;
    L.dataBase = base;
    L.userKernel = c->userKernel;
    L.relief = c->reliefScene;
    L.looseMode = c->looseMode;
    L.looseGrid = c->looseGrid;
    if (c->havePrimSticking) { // this particle's sticking map: the next prepare would overwrite the shared buffer
      L.primSticking = c->dPrimSticking.p;
      L.params.primSticking = L.primSticking;
      c->dPrimSticking.p = nullptr;
      c->dPrimSticking.cap = 0;
      c->havePrimSticking = false;
    }
    base += c->numData;
  }
  // the shared buffers were (re-)sized by each prepare in turn and only ever grow — a later particle may have moved
  // one (records with RNG cursors after records without, a larger grid's walk stacks): everybody gets the final addresses
  for (size_t q = 0; q < c->launches.size(); ++q) {
    TraceParams &lp = c->launches[q].params;
    lp.slotRec = c->dSlotRec.p;
    lp.binCount = c->dBinCount.p;
    lp.walkStack = c->dWalkStack.p;
    lp.rngScratch = c->dScratch.p;
    lp.workCounter = c->dWorkQ.p;
    lp.recExtra = lp.recExtra ? c->dRecExtra.p : nullptr;
    lp.spillRec = lp.spillRec ? c->dSpillRec.p : nullptr;
    lp.spillCount = lp.spillCount ? c->dSpillCount.p : nullptr;
    lp.counters = c->dCounters.p + 80 * q;
    lp.fluxAcc = c->dFluxAcc.p + (size_t)c->launches[q].dataBase * lp.planeStride;
  }
  activate_particle(c, c->specs[0]);
  c->dataBase = 0;
  c->counterSlot = 0;
  c->prepared = true;
  return VR_OK;
}

int vr_apply(vr_context *c) {
  int r = vr_apply_prepare(c);
  if (r != VR_OK)
    return r;
  r = vr_apply_launch(c);
  if (r != VR_OK)
    return r;
  return vr_apply_finish(c);
}

// Multi-GPU apply() behind the C ABI: this rank traces its contiguous share of the
// global ray indices, then the per-primitive int64 accumulators (exact, order-independent) and the
// seven counters are summed over all ranks by the caller's collective — RCCL over xGMI through
// vr_rccl_allreduce (libviennaray_amd_rccl.so), or anything else with the same signature.  Every
// rank ends with the full flux, bit-identical to the single-device run; runNumber advances on
// every rank (also one whose share is empty), so later applies keep using the same seeds.
int vr_apply_sharded(vr_context *c, int rank, int world, vr_allreduce_fn reduce, void *user) {
  if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !reduce))
    return fail(c, VR_E_INVALID, "vr_apply_sharded: bad argument");
  VR_HIP(c, hipSetDevice(c->device));
  const uint64_t total = rays_of_apply(c);
  const uint64_t first = total * (uint64_t)rank / (uint64_t)world;
  const uint64_t last = total * (uint64_t)(rank + 1) / (uint64_t)world;
  const uint32_t N = c->geo.numPrims;
  VR_HIP(c, c->dCounters.ensure(80 * std::max<size_t>(1, c->specs.size())));
  c->haveSharedSeed = false;
  if (world > 1 && c->useRandomSeed) {
    // setUseRandomSeeds(true): every rank would draw its own seed and the shards would belong to different
    // streams.  Rank 0 draws, the others contribute 0, and the all-reduce hands the seed round.
    unsigned long long word = 0;
    if (rank == 0) {
      std::random_device rd;
      word = (uint32_t)rd();
    }
    VR_HIP(c, hipMemcpyAsync(c->dCounters.p + 63, &word, 8, hipMemcpyHostToDevice, c->stream));
    if (reduce(user, c->dCounters.p + 63, 1, (void *)c->stream) != 0)
      return fail(c, VR_E_HIP, "vr_apply_sharded: the all-reduce callback failed (seed)");
    VR_HIP(c, hipMemcpyAsync(&word, c->dCounters.p + 63, 8, hipMemcpyDeviceToHost, c->stream));
    VR_HIP(c, hipStreamSynchronize(c->stream));
    c->sharedSeed = (uint32_t)word;
    c->haveSharedSeed = true;
    c->keepSharedSeed = true; // (cleared below, after the launch)
  }
  int r = VR_OK;
  const uint32_t worldBefore = c->worldSize;
  c->worldSize = std::max<uint32_t>(c->worldSize, (uint32_t)world); // (head-room of the overflow check: the sums of all ranks fit int64)
  if (last > first) {
    c->rayFirst = first;
    c->rayCount = last - first;
    r = vr_apply_prepare(c);
    if (r == VR_OK)
      r = vr_apply_launch(c);
  } else {
    // an empty share: nothing to trace, but the scene is prepared like everywhere else (numRays, areas,
    // accumulator planes: the collective below must see the same buffer sizes on every rank)
    c->rayFirst = total; // (an empty range behind the last ray)
    c->rayCount = 1;
    r = vr_apply_prepare(c);
    if (r == VR_OK) {
      VR_HIP(c, hipMemsetAsync(c->fluxOut(), 0, (size_t)N * c->totalData * 8, c->stream));
      VR_HIP(c, hipMemsetAsync(c->dCounters.p, 0, 80 * 8 * std::max<size_t>(1, c->specs.size()), c->stream));
      VR_HIP(c, hipEventRecord(c->ev0, c->stream));
      VR_HIP(c, hipEventRecord(c->ev1, c->stream));
      c->numGenLaunches = c->numTraceLaunches = 0;
      c->launched = true;
    }
  }
  c->rayFirst = 0;
  c->rayCount = 0;
  c->haveSharedSeed = false;
  c->keepSharedSeed = false;
  c->worldSize = worldBefore;
  if (world > 1) {
    // A rank that failed above still enters the collectives when it can (zeros and a raised failure word) —
    // the others would hang in them otherwise.  The TraceInfo counters [0..7] AND the failure word [60] (the
    // walk's stack overflow, or this) travel together: every rank fails together, none returns VR_OK
    // holding sums that include a discarded share.
    const std::string firstErr = c->err;
    const bool haveBuf = c->boundFlux ? c->boundFluxN == N * c->totalData : c->dFluxOrig.cap >= (size_t)N * c->totalData;
    if (r != VR_OK) {
      if (!haveBuf)
        return r; // (failed before the accumulators existed: a configuration error, the same on every rank)
      const unsigned long long one = 1;
      (void)hipMemsetAsync(c->fluxOut(), 0, (size_t)N * c->totalData * 8, c->stream);
      (void)hipMemsetAsync(c->dCounters.p, 0, 80 * 8 * std::max<size_t>(1, c->specs.size()), c->stream);
      (void)hipMemcpyAsync(c->dCounters.p + 60, &one, 8, hipMemcpyHostToDevice, c->stream);
    }
    if (reduce(user, c->fluxOut(), (size_t)N * c->totalData, (void *)c->stream) != 0 ||
        reduce(user, c->dCounters.p, 80 * std::max<size_t>(1, c->specs.size()), (void *)c->stream) != 0)
      return fail(c, VR_E_HIP, "vr_apply_sharded: the all-reduce callback failed");
    if (r != VR_OK) {
      (void)hipStreamSynchronize(c->stream);
      c->err = firstErr;
      return r;
    }
  } else if (r != VR_OK) {
    return r;
  }
  return vr_apply_finish(c);
}

// ---- results --------------------------------------------------------------------
uint32_t vr_num_primitives(const vr_context *c) { return c ? c->geo.numPrims : 0; }

uint32_t vr_num_data(const vr_context *c) { return c ? c->totalData : 0; }

static int get_flux_plane_f64(vr_context *c, uint32_t dataIdx, double *out, uint32_t n);

int vr_get_flux_f64(vr_context *c, double *out, uint32_t n) { return get_flux_plane_f64(c, 0, out, n); }

// getLocalData().getVectorData(dataIdx): the particle's data label `dataIdx`, as the reference's float vector.
// The int64 fixed-point sums become floats on the device (float(double(acc) * 2^-40), what the host conversion
// did): one 4-byte-per-primitive download instead of 8 bytes and two host passes.
int vr_get_flux_data(vr_context *c, uint32_t dataIdx, float *out, uint32_t n) {
  if (!c || !out)
    return VR_E_INVALID;
  if (!c->haveResult)
    return fail(c, VR_E_STATE, "vr_get_flux: no result (call vr_apply)");
  if (n != c->geo.numPrims)
    return fail(c, VR_E_INVALID, "vr_get_flux: size mismatch");
  if (dataIdx >= c->totalData)
    return fail(c, VR_E_INVALID, "vr_get_flux_data: the particle has no such data label");
  VR_HIP(c, hipSetDevice(c->device));
  VR_HIP(c, c->dFluxTmp.ensure(n));
  VR_HIP(c, launch_flux_from_acc(c->fluxOut() + (size_t)dataIdx * n, n, c->dFluxTmp.p, c->stream));
  VR_HIP(c, hipMemcpyAsync(out, c->dFluxTmp.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  VR_HIP(c, hipStreamSynchronize(c->stream));
  return VR_OK;
}

static int get_flux_plane_f64(vr_context *c, uint32_t dataIdx, double *out, uint32_t n) {
  if (!c || !out)
    return VR_E_INVALID;
  if (!c->haveResult)
    return fail(c, VR_E_STATE, "vr_get_flux: no result (call vr_apply)");
  if (n != c->geo.numPrims)
    return fail(c, VR_E_INVALID, "vr_get_flux: size mismatch");
  if (dataIdx >= c->totalData)
    return fail(c, VR_E_INVALID, "vr_get_flux_data: the particle has no such data label");
  VR_HIP(c, hipSetDevice(c->device));
  std::vector<unsigned long long> acc(n);
  VR_HIP(c, hipMemcpy(acc.data(), c->fluxOut() + (size_t)dataIdx * n, (size_t)n * 8, hipMemcpyDeviceToHost));
  const double scale = std::ldexp(1.0, -VR_FLUX_FRAC_BITS);
  for (uint32_t i = 0; i < n; ++i)
    out[i] = (double)acc[i] * scale;
  return VR_OK;
}

int vr_get_flux(vr_context *c, float *out, uint32_t n) { return vr_get_flux_data(c, 0, out, n); }

int vr_get_trace_info(const vr_context *c, vr_trace_info *out) {
  if (!c || !out)
    return VR_E_INVALID;
  *out = c->info;
  return VR_OK;
}

int vr_get_particle_trace_info(const vr_context *c, uint32_t q, vr_trace_info *out) {
  if (!c || !out)
    return VR_E_INVALID;
  if (c->specs.size() <= 1) {
    if (q != 0)
      return VR_E_INVALID;
    *out = c->info;
    return VR_OK;
  }
  if (q >= c->launches.size())
    return VR_E_INVALID;
  *out = c->launches[q].info;
  return VR_OK;
}

int vr_get_trace_mode(const vr_context *c, int32_t *mode) {
  if (!c || !mode)
    return VR_E_INVALID;
  *mode = c->traceMode;
  return VR_OK;
}

int vr_add_trace_info(vr_context *c, const vr_trace_info *o) {
  if (!c || !o)
    return VR_E_INVALID;
  vr_trace_info &i = c->info;
  i.totalRaysTraced += o->totalRaysTraced;
  i.nonGeometryHits += o->nonGeometryHits;
  i.geometryHits += o->geometryHits;
  i.particleHits += o->particleHits;
  i.boundaryHits += o->boundaryHits;
  i.reflections += o->reflections;
  i.raysTerminated += o->raysTerminated;
  i.rngFullStates += o->rngFullStates;
  return VR_OK;
}

int vr_flux_accumulators(vr_context *c, void **devPtr, uint32_t *n) {
  if (!c || !devPtr)
    return VR_E_INVALID;
  if (!c->haveResult)
    return fail(c, VR_E_STATE, "vr_flux_accumulators: no result");
  *devPtr = c->fluxOut();
  if (n)
    *n = c->geo.numPrims * c->totalData;
  return VR_OK;
}

int vr_bind_flux_accumulators(vr_context *c, void *devPtr, uint32_t n) {
  if (!c)
    return VR_E_INVALID;
  if (devPtr && n != c->geo.numPrims * c->totalData)
    return fail(c, VR_E_INVALID, "vr_bind_flux_accumulators: size mismatch (numPrims x data labels; set geometry and particle first)");
  c->boundFlux = (unsigned long long *)devPtr;
  c->boundFluxN = devPtr ? n : 0;
  return VR_OK;
}

void *vr_stream(vr_context *c) { return c ? (void *)c->stream : nullptr; }

// normalizeFlux on the device (rayTraceDisk.hpp:103-142, rayTraceTriangle.hpp:92-130;
// gpu/kernels/normKernels.cu:58-74): dFluxTmp holds the flux in the caller's order
static int normalize_on_device(vr_context *c, uint32_t n, int normType) {
  const bool disk = c->geo.geo == 0;
  if (!c->areasValid)
    return fail(c, VR_E_STATE, "vr_normalize_flux: call vr_apply first (primitive areas)");
  float normFactor = 0.f;
  if (normType == VR_NORM_SOURCE) {
    if (c->numRaysLast == 0)
      return fail(c, VR_E_STATE, "No source was specified in rayTrace for the normalization.");
    normFactor = (c->sourceAreaOverride > 0.f ? c->sourceAreaOverride : c->sourceArea) / c->numRaysLast;
  } else if (normType != VR_NORM_MAX) {
    return VR_OK; // `default: break;` in the reference
  }
  const double totalDiskArea = c->geo.diskRadius * c->geo.diskRadius * M_PI;
  VR_HIP(c, c->dNormMax.ensure(1)); // (a word of its own: the builder's scratch does not exist under VR_HOST_BUILD)
  VR_HIP(c, launch_normalize_flux(c->dFluxTmp.p, c->dAreas.p, n, disk ? 0 : 1, normType, normFactor, totalDiskArea,
                                  c->dNormMax.p, c->stream));
  return VR_OK;
}

int vr_normalize_flux(vr_context *c, float *flux, uint32_t n, int normType) {
  if (!c || !flux || n != c->geo.numPrims)
    return fail(c, VR_E_INVALID, "vr_normalize_flux: bad argument");
  VR_HIP(c, hipSetDevice(c->device));
  VR_HIP(c, c->dFluxTmp.ensure(n));
  VR_HIP(c, hipMemcpyAsync(c->dFluxTmp.p, flux, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  int r = normalize_on_device(c, n, normType);
  if (r != VR_OK)
    return r;
  VR_HIP(c, hipMemcpyAsync(flux, c->dFluxTmp.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  VR_HIP(c, hipStreamSynchronize(c->stream));
  return VR_OK;
}

// getLocalData().getVectorData(0) followed by normalizeFlux, without the raw flux ever
// visiting the host: int64 accumulators -> float -> normalised, one download
int vr_get_flux_normalized(vr_context *c, float *out, uint32_t n, int normType) {
  if (!c || !out)
    return VR_E_INVALID;
  if (!c->haveResult)
    return fail(c, VR_E_STATE, "vr_get_flux_normalized: no result (call vr_apply)");
  if (n != c->geo.numPrims)
    return fail(c, VR_E_INVALID, "vr_get_flux_normalized: size mismatch");
  VR_HIP(c, hipSetDevice(c->device));
  VR_HIP(c, c->dFluxTmp.ensure(n));
  VR_HIP(c, launch_flux_from_acc(c->fluxOut(), n, c->dFluxTmp.p, c->stream));
  int r = normalize_on_device(c, n, normType);
  if (r != VR_OK)
    return r;
  VR_HIP(c, hipMemcpyAsync(out, c->dFluxTmp.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  VR_HIP(c, hipStreamSynchronize(c->stream));
  return VR_OK;
}

// rayTraceDisk.hpp:146-193 (triangle version is a no-op: rayTraceTriangle.hpp:134-136)
int vr_smooth_flux(vr_context *c, float *flux, uint32_t n, int numNeighbors) {
  if (!c || !flux || n != c->geo.numPrims)
    return fail(c, VR_E_INVALID, "vr_smooth_flux: bad argument");
  if (c->geo.geo != 0 || numNeighbors < 1)
    return VR_OK;
  // device path: the geometry's own neighbourhood (numNeighbors == 1, what every reference example asks for) is
  // resident with the device-built scene; a wider one (k > 1) is a range query of radius k * 2 r over the resident BVH,
  // fused with the averaging.  No download of any neighbourhood.
  if (c->haveSetup && !c->geometryDirty && !std::getenv("VR_HOST_SMOOTH")) {
    VR_HIP(c, hipSetDevice(c->device));
    DevBuf<float> dIn, dOut;
    DevBuf<uint32_t> dOv;
    VR_HIP(c, dIn.ensure(n));
    VR_HIP(c, dOut.ensure(n));
    VR_HIP(c, dOv.ensure(1));
    uint32_t ov = 0;
    VR_HIP(c, hipMemcpyAsync(dIn.p, flux, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    VR_HIP(c, hipMemsetAsync(dOv.p, 0, 4, c->stream));
    if (numNeighbors == 1)
      VR_HIP(c, launch_smooth_flux(dIn.p, dOut.p, c->dNormal3.p, c->dNbOff.p, c->dNbIds.p, c->dOrder.p,
                                   c->dLeafOfOrig.p, n, dOv.p, c->stream));
    else
      VR_HIP(c, launch_smooth_wide(dIn.p, dOut.p, c->dNormal3.p, c->lastSetup, numNeighbors * 2 * c->geo.diskRadius,
                                   dOv.p, c->stream));
    VR_HIP(c, hipMemcpyAsync(&ov, dOv.p, 4, hipMemcpyDeviceToHost, c->stream));
    VR_HIP(c, hipStreamSynchronize(c->stream));
    if (ov == 0) {
      VR_HIP(c, hipMemcpy(flux, dOut.p, (size_t)n * 4, hipMemcpyDeviceToHost));
      return VR_OK;
    } // (some neighbourhood longer than the kernel's buffer: host path below)
  }
  if (numNeighbors == 1) {
    int r = ensure_host_neighbors(c);
    if (r != VR_OK)
      return r;
  }
  const std::vector<uint32_t> *off = &c->geo.nbOff, *ids = &c->geo.nbIds;
  std::vector<uint32_t> woff, wids;
  if (numNeighbors != 1) {
    std::vector<float> pts((size_t)n * 3);
    for (uint32_t i = 0; i < n; ++i)
      std::memcpy(&pts[3 * (size_t)i], &c->geo.disk4[4 * (size_t)i], 12);
    host_neighbors(c->geo.D, pts.data(), n, numNeighbors * 2 * c->geo.diskRadius, c->geo.minC, woff, wids);
    off = &woff;
    ids = &wids;
  }
  std::vector<float> old(flux, flux + n);
  const float *nr = c->geo.normal3.data();
  for (uint32_t i = 0; i < n; ++i) {
    float vv = old[i];
    float sum = 1.f;
    for (uint32_t j = (*off)[i]; j < (*off)[i + 1]; ++j) {
      const uint32_t nb = (*ids)[j];
      const float w = (nr[3 * (size_t)i] * nr[3 * (size_t)nb] + nr[3 * (size_t)i + 1] * nr[3 * (size_t)nb + 1]) +
                      nr[3 * (size_t)i + 2] * nr[3 * (size_t)nb + 2];
      if (w > 0.f) {
        vv += old[nb] * w;
        sum += w;
      }
    }
    flux[i] = vv / sum;
  }
  return VR_OK;
}

int vr_get_disk_areas(vr_context *c, float *out, uint32_t n) {
  if (!c || !out || n != c->geo.numPrims || c->geo.geo != 0 || !c->areasValid)
    return fail(c, VR_E_STATE, "vr_get_disk_areas: not available");
  if (!c->diskAreasHostValid) {
    VR_HIP(c, hipSetDevice(c->device));
    c->diskAreas.resize(n);
    VR_HIP(c, hipStreamSynchronize(c->stream));
    VR_HIP(c, hipMemcpy(c->diskAreas.data(), c->dAreas.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    c->diskAreasHostValid = true;
  }
  std::memcpy(out, c->diskAreas.data(), (size_t)n * 4);
  return VR_OK;
}
int vr_get_bounding_box(vr_context *c, float *out6) {
  if (!c || !out6)
    return VR_E_INVALID;
  for (int k = 0; k < 3; ++k) {
    out6[k] = c->bbLo[k];
    out6[k + 3] = c->bbHi[k];
  }
  return VR_OK;
}
float vr_get_source_area(vr_context *c) {
  return c ? (c->sourceAreaOverride > 0.f ? c->sourceAreaOverride : c->sourceArea) : 0.f;
}
float vr_get_disk_radius(const vr_context *c) { return c ? c->geo.diskRadius : 0.f; }
int vr_get_neighbor_counts(vr_context *c, uint32_t *out, uint32_t n) {
  if (!c || !out || n != c->geo.numPrims)
    return VR_E_INVALID;
  int r = ensure_host_neighbors(c);
  if (r != VR_OK)
    return r;
  for (uint32_t i = 0; i < n; ++i)
    out[i] = c->geo.nbOff[i + 1] - c->geo.nbOff[i];
  return VR_OK;
}

// ---- diagnostics ------------------------------------------------------------------
int vr_debug_intersect(vr_context *c, const float *org, const float *dir, const float *tnear, uint32_t n,
                       int32_t *geomID, uint32_t *primID, float *t) {
  if (!c || !org || !dir || !tnear || !geomID || !primID || !t)
    return VR_E_INVALID;
  if (!c->prepared) {
    int r = vr_apply_prepare(c);
    if (r != VR_OK)
      return r;
  }
  DevBuf<float> dO, dD, dT, dt;
  DevBuf<int> dG;
  DevBuf<uint32_t> dP;
  VR_HIP(c, dO.ensure((size_t)n * 3));
  VR_HIP(c, dD.ensure((size_t)n * 3));
  VR_HIP(c, dT.ensure(n));
  VR_HIP(c, dt.ensure(n));
  VR_HIP(c, dG.ensure(n));
  VR_HIP(c, dP.ensure(n));
  VR_HIP(c, hipMemcpy(dO.p, org, (size_t)n * 12, hipMemcpyHostToDevice));
  VR_HIP(c, hipMemcpy(dD.p, dir, (size_t)n * 12, hipMemcpyHostToDevice));
  VR_HIP(c, hipMemcpy(dT.p, tnear, (size_t)n * 4, hipMemcpyHostToDevice));
  // the ordered (pair-node, stack) walk of the trace kernels; VR_DEBUG_WALK=0: the escape-link walk it replaced
  int ordered = 1;
  if (const char *e = std::getenv("VR_DEBUG_WALK"))
    ordered = std::atoi(e) != 0;
  // (64-thread blocks running concurrently must not share a slab: at most walkStackWaves blocks per launch)
  const uint32_t chunk = (uint32_t)std::max<size_t>(c->walkStackWaves, 1) * 64u;
  for (uint32_t f0 = 0; f0 < n; f0 += chunk) {
    const uint32_t m = std::min(chunk, n - f0);
    VR_HIP(c, launch_debug_intersect(c->params, c->geo.geo, dO.p + 3 * (size_t)f0, dD.p + 3 * (size_t)f0, dT.p + f0, m,
                                     dG.p + f0, dP.p + f0, dt.p + f0, ordered, (unsigned)std::max<size_t>(c->walkStackWaves, 1), c->stream));
  }
  VR_HIP(c, hipStreamSynchronize(c->stream));
  VR_HIP(c, hipMemcpy(geomID, dG.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  VR_HIP(c, hipMemcpy(primID, dP.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  VR_HIP(c, hipMemcpy(t, dt.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  dO.release();
  dD.release();
  dT.release();
  dt.release();
  dG.release();
  dP.release();
  return VR_OK;
}

// Boundary::processHit on the device for hand-built hits (tests/boundaryHit, boundaryHit2D): ray (org, dir) meets wall
// triangle primID at parameter tfar -> new origin, new (projected) direction, reflect flag
int vr_debug_process_hit(vr_context *c, const float *org, const float *dir, const float *tfar, const uint32_t *primID,
                         uint32_t n, float *outOrg, float *outDir, int32_t *outReflect) {
  if (!c || !org || !dir || !tfar || !primID || !outOrg || !outDir || !outReflect)
    return VR_E_INVALID;
  for (uint32_t i = 0; i < n; ++i)
    if (primID[i] > 7u)
      return fail(c, VR_E_INVALID, "vr_debug_process_hit: boundary primID must be 0..7");
  if (!c->prepared) {
    int r = vr_apply_prepare(c);
    if (r != VR_OK)
      return r;
  }
  DevBuf<float> dO, dD, dT, dOo, dDo;
  DevBuf<uint32_t> dP;
  DevBuf<int> dR;
  VR_HIP(c, dO.ensure((size_t)n * 3));
  VR_HIP(c, dD.ensure((size_t)n * 3));
  VR_HIP(c, dT.ensure(n));
  VR_HIP(c, dP.ensure(n));
  VR_HIP(c, dOo.ensure((size_t)n * 3));
  VR_HIP(c, dDo.ensure((size_t)n * 3));
  VR_HIP(c, dR.ensure(n));
  VR_HIP(c, hipMemcpy(dO.p, org, (size_t)n * 12, hipMemcpyHostToDevice));
  VR_HIP(c, hipMemcpy(dD.p, dir, (size_t)n * 12, hipMemcpyHostToDevice));
  VR_HIP(c, hipMemcpy(dT.p, tfar, (size_t)n * 4, hipMemcpyHostToDevice));
  VR_HIP(c, hipMemcpy(dP.p, primID, (size_t)n * 4, hipMemcpyHostToDevice));
  VR_HIP(c, launch_debug_process_hit(c->params, c->geo.D, dO.p, dD.p, dT.p, dP.p, n, dOo.p, dDo.p, dR.p, c->stream));
  VR_HIP(c, hipStreamSynchronize(c->stream));
  VR_HIP(c, hipMemcpy(outOrg, dOo.p, (size_t)n * 12, hipMemcpyDeviceToHost));
  VR_HIP(c, hipMemcpy(outDir, dDo.p, (size_t)n * 12, hipMemcpyDeviceToHost));
  VR_HIP(c, hipMemcpy(outReflect, dR.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return VR_OK;
}

int vr_debug_source_sample(vr_context *c, const uint64_t *idx, uint32_t n, uint32_t seed, float *org, float *dir) {
  if (!c || !idx || !org || !dir)
    return VR_E_INVALID;
  if (!c->prepared) {
    int r = vr_apply_prepare(c);
    if (r != VR_OK)
      return r;
  }
  if (n > c->slotStride)
    return fail(c, VR_E_INVALID, "vr_debug_source_sample: more rays than one batch holds");
  TraceParams p = c->params;
  p.seed = seed;
  p.batchCount = n;
  p.binCount = nullptr; // no binning: record i goes to slot i
  DevBuf<unsigned long long> dI;
  VR_HIP(c, dI.ensure(n));
  VR_HIP(c, hipMemcpy(dI.p, idx, (size_t)n * 8, hipMemcpyHostToDevice));
  p.idxList = dI.p;
  VR_HIP(c, launch_gen(p, c->geo.D, false, (unsigned)c->numCUs * 8u, c->stream));
  VR_HIP(c, hipStreamSynchronize(c->stream));
  std::vector<float> A((size_t)n * 4), B((size_t)n * 4);
  A.resize((size_t)n * 8);
  VR_HIP(c, hipMemcpy(A.data(), c->dSlotRec.p, (size_t)n * 32, hipMemcpyDeviceToHost));
  (void)B;
  for (uint32_t i = 0; i < n; ++i) {
    const float *r = &A[8 * (size_t)i];
    org[3 * i] = r[0];
    org[3 * i + 1] = r[1];
    org[3 * i + 2] = r[2];
    dir[3 * i] = r[3];
    dir[3 * i + 1] = r[4];
    dir[3 * i + 2] = r[5];
  }
  dI.release();
  return VR_OK;
}

int vr_debug_rng_outputs(vr_context *c, uint64_t idx, uint32_t seed, uint32_t count, uint64_t *out) {
  if (!c || !out)
    return VR_E_INVALID;
  VR_HIP(c, hipSetDevice(c->device));
  // tea<3>(idx, seed) on the host (same mix as vr_device.hpp)
  unsigned v0 = (unsigned)idx, v1 = seed, s0 = 0;
  for (int n = 0; n < 3; ++n) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  DevBuf<unsigned long long> dS, dOut;
  VR_HIP(c, dS.ensure(312u * 64u));
  VR_HIP(c, dOut.ensure(count));
  VR_HIP(c, launch_debug_rng(v0, count, dS.p, dOut.p, c->stream));
  VR_HIP(c, hipStreamSynchronize(c->stream));
  VR_HIP(c, hipMemcpy(out, dOut.p, (size_t)count * 8, hipMemcpyDeviceToHost));
  dS.release();
  dOut.release();
  return VR_OK;
}

int vr_debug_bvh_check(vr_context *c, uint32_t *violations) {
  if (!c || !violations)
    return VR_E_INVALID;
  if (!c->haveSetup || c->geometryDirty)
    return fail(c, VR_E_STATE, "vr_debug_bvh_check: no device-built BVH resident (call vr_apply_prepare)");
  VR_HIP(c, hipSetDevice(c->device));
  DevBuf<uint32_t> dBad;
  VR_HIP(c, dBad.ensure(1));
  VR_HIP(c, hipMemsetAsync(dBad.p, 0, 4, c->stream));
  VR_HIP(c, launch_bvh_check(c->lastSetup, dBad.p, c->stream));
  VR_HIP(c, hipMemcpyAsync(violations, dBad.p, 4, hipMemcpyDeviceToHost, c->stream));
  VR_HIP(c, hipStreamSynchronize(c->stream));
  return VR_OK;
}

int vr_get_run_number(const vr_context *c, uint32_t *out) {
  if (!c || !out)
    return VR_E_INVALID;
  *out = c->runNumber;
  return VR_OK;
}

// Measured instruction-issue ceiling (vr_bench.hip): `wavesPerSimd` blocks of 256 threads per CU
// (= that many waves on every SIMD) run `iters` passes of the chosen mix.
// out4 = {counted instructions per second, sustained clock in Hz (median over waves),
//         seconds (HIP events), counted instructions}
int vr_debug_issue_rate(vr_context *c, int kind, int wavesPerSimd, uint32_t iters, double *out4) {
  if (!c || !out4 || kind < 0 || kind > 6 || wavesPerSimd < 1 || wavesPerSimd > 8 || iters == 0)
    return fail(c, VR_E_INVALID, "vr_debug_issue_rate: bad argument");
  VR_HIP(c, hipSetDevice(c->device));
  const unsigned blocks = (unsigned)c->numCUs * (unsigned)wavesPerSimd;
  const size_t waves = (size_t)blocks * 4;
  DevBuf<unsigned long long> dOut;
  VR_HIP(c, dOut.ensure(waves * 3));
  VR_HIP(c, hipMemsetAsync(dOut.p, 0, waves * 24, c->stream));
  VR_HIP(c, launch_issue_kernel(kind, blocks, std::max<uint32_t>(iters / 16, 1), dOut.p, c->stream)); // warm-up, clocks up
  VR_HIP(c, hipEventRecord(c->ev0, c->stream));
  VR_HIP(c, launch_issue_kernel(kind, blocks, iters, dOut.p, c->stream));
  VR_HIP(c, hipEventRecord(c->ev1, c->stream));
  VR_HIP(c, hipStreamSynchronize(c->stream));
  float ms = 0.f;
  VR_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  std::vector<unsigned long long> h(waves * 3);
  VR_HIP(c, hipMemcpy(h.data(), dOut.p, waves * 24, hipMemcpyDeviceToHost));
  std::vector<double> clk;
  for (size_t w = 0; w < waves; ++w)
    if (h[3 * w + 1])
      clk.push_back((double)h[3 * w] / (double)h[3 * w + 1] * 1e8);
  std::sort(clk.begin(), clk.end());
  const double perPass = (kind == 4 || kind == 5) ? 24.0 : 32.0;
  const double counted = (double)waves * (double)iters * perPass;
  out4[0] = counted / (ms * 1e-3);
  out4[1] = clk.empty() ? 0.0 : clk[clk.size() / 2];
  out4[2] = ms * 1e-3;
  out4[3] = counted;
  return VR_OK;
}

int vr_debug_bvh_stats(vr_context *c, uint32_t *out3) {
  if (!c || !out3)
    return VR_E_INVALID;
  out3[0] = c->bvh.numNodes;
  out3[1] = c->bvh.numLeaves;
  out3[2] = c->bvh.maxDepth;
  return VR_OK;
}

} // extern "C"

"""Host-side mirror of the reference front-end for the accelerated path.

Same names, argument meaning and error behaviour as `viennaray::Trace<T,D>`,
`TraceDisk`, `TraceTriangle`, `DiffuseParticle`, `SpecularParticle`,
`TracingData` (reference: include/viennaray/rayTrace.hpp:15-180,
rayTraceDisk.hpp:13-224, rayTraceTriangle.hpp:13-154, rayParticle.hpp:126-204,
rayTracingData.hpp:16-219), driving the C ABI in include/viennaray_amd.h.
The C++ façade with the reference's exact spelling lives in
include/viennaray_amd/*.hpp; this module exists so the parity tests read like
the reference's tests.
"""
import ctypes as C
import enum

import numpy as np

from . import capi
from .capi import VrError, TraceInfoPOD, ParticlePOD


class BoundaryCondition(enum.IntEnum):  # rayBoundary.hpp:10-14
    REFLECTIVE_BOUNDARY = 0
    PERIODIC_BOUNDARY = 1
    IGNORE_BOUNDARY = 2


class TraceDirection(enum.IntEnum):  # rayUtil.hpp:40-47
    POS_X = 0
    NEG_X = 1
    POS_Y = 2
    NEG_Y = 3
    POS_Z = 4
    NEG_Z = 5


class NormalizationType(enum.IntEnum):  # rayUtil.hpp:38
    SOURCE = 0
    MAX = 1


class TracingDataMergeEnum(enum.IntEnum):  # rayTracingData.hpp:10-14
    SUM = 0
    APPEND = 1
    AVERAGE = 2


class DiffuseParticle:
    """rayParticle.hpp:126-163"""
    kind = 0

    def __init__(self, stickingProbability, dataLabel, materialSticking=None):
        self.stickingProbability = float(stickingProbability)
        self.dataLabel = dataLabel
        self.materialSticking = dict(materialSticking or {})

    def getSourceDistributionPower(self):
        return 1.0

    def getLocalDataLabels(self):
        return [self.dataLabel]


class SpecularParticle:
    """rayParticle.hpp:165-204"""
    kind = 1

    def __init__(self, stickingProbability, sourcePower, dataLabel, materialSticking=None):
        self.stickingProbability = float(stickingProbability)
        self.sourcePower = float(sourcePower)
        self.dataLabel = dataLabel
        self.materialSticking = dict(materialSticking or {})

    def getSourceDistributionPower(self):
        return self.sourcePower

    def getLocalDataLabels(self):
        return [self.dataLabel]


class ConedCosineParticle:
    """Plug-in particle of the device registry (vr_particles.hpp): collects like SpecularParticle,
    reflects with ReflectionConedCosine(maxConeAngle) (rayReflection.hpp:52-120)."""
    kind = 2

    def __init__(self, stickingProbability, sourcePower, coneAngle, dataLabel, materialSticking=None,
                 meanFreePath=-1.0):
        self.stickingProbability = float(stickingProbability)
        self.sourcePower = float(sourcePower)
        self.coneAngle = float(coneAngle)
        self.dataLabel = dataLabel
        self.materialSticking = dict(materialSticking or {})
        self.meanFreePath = float(meanFreePath)

    def getSourceDistributionPower(self):
        return self.sourcePower

    def getMeanFreePath(self):
        return self.meanFreePath

    def getLocalDataLabels(self):
        return [self.dataLabel]


class DiffuseCosineParticle:
    """Plug-in particle with TWO data labels: label 0 += w (like DiffuseParticle), label 1 += w * max(0, -d.n)."""
    kind = 3

    def __init__(self, stickingProbability, dataLabel, cosineLabel, materialSticking=None, meanFreePath=-1.0):
        self.stickingProbability = float(stickingProbability)
        self.dataLabels = [dataLabel, cosineLabel]
        self.materialSticking = dict(materialSticking or {})
        self.meanFreePath = float(meanFreePath)

    def getSourceDistributionPower(self):
        return 1.0

    def getMeanFreePath(self):
        return self.meanFreePath

    def getLocalDataLabels(self):
        return list(self.dataLabels)


class CoverageStickingParticle:
    """Plug-in particle reading Trace.setGlobalData: a DiffuseParticle whose sticking falls with the coverage of the
    primitive it meets, sticking * (1 - globalData.getVectorData(coverageVector)[primID]) — the host form returns
    that as the first member of surfaceReflection (rayParticle.hpp:44-50)."""
    kind = 4

    def __init__(self, stickingProbability, dataLabel, coverageVector=0, materialSticking=None):
        self.stickingProbability = float(stickingProbability)
        self.dataLabel = dataLabel
        self.coverageVector = int(coverageVector)
        self.materialSticking = dict(materialSticking or {})
        self.params = [float(self.coverageVector)]

    def getSourceDistributionPower(self):
        return 1.0

    def getLocalDataLabels(self):
        return [self.dataLabel]


class UserModelParticle:
    """A particle whose device model was registered at run time (Trace.registerParticleModel): `kind` is what the
    registration returned, `dataLabels` one label per data label of the model, `params` the model's parameters."""

    def __init__(self, kind, stickingProbability, dataLabels, sourcePower=1.0, params=(), materialSticking=None,
                 meanFreePath=-1.0):
        self.kind = int(kind)
        self.stickingProbability = float(stickingProbability)
        self.dataLabels = list(dataLabels)
        self.sourcePower = float(sourcePower)
        self.params = [float(v) for v in params]
        self.materialSticking = dict(materialSticking or {})
        self.meanFreePath = float(meanFreePath)

    def getSourceDistributionPower(self):
        return self.sourcePower

    def getLocalDataLabels(self):
        return list(self.dataLabels)


class SourceGrid:
    """raySourceGrid.hpp: explicit ray origins (createSourceGrid, rayUtil.hpp:564-611); the direction
    comes from the particle's cosine power."""

    def __init__(self, points):
        self.points = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)

    def getNumPoints(self):
        return self.points.shape[0]


class TracingData:
    """rayTracingData.hpp:16-219: labelled vectors and scalars with a merge type each.  The
    device path fills vector 0 of Trace.getLocalData() (merge type SUM)."""

    def __init__(self):
        self._vectors, self._vlabels, self._vmerge = [], [], []
        self._scalars, self._slabels, self._smerge = [], [], []

    def setNumberOfVectorData(self, n):
        self._vectors = [np.zeros(0, dtype=np.float32) for _ in range(n)]
        self._vlabels = ["vectorData"] * n
        self._vmerge = [TracingDataMergeEnum.SUM] * n

    def setNumberOfScalarData(self, n):
        self._scalars = [0.0] * n
        self._slabels = ["scalarData"] * n
        self._smerge = [TracingDataMergeEnum.SUM] * n

    def setScalarData(self, num, value, label="scalarData"):
        self._scalars[num] = float(value)
        self._slabels[num] = label

    def setVectorData(self, num, data, label="vectorData", value=None):
        """setVectorData(num, array, label) or setVectorData(num, size, label, value=v)"""
        if value is not None:
            data = np.full(int(data), value, dtype=np.float32)
        self._vectors[num] = np.asarray(data, dtype=np.float32)
        self._vlabels[num] = label

    def appendVectorData(self, num, data):
        self._vectors[num] = np.concatenate([self._vectors[num], np.asarray(data, dtype=np.float32)])

    def resizeAllVectorData(self, size, val=0.0):
        self._vectors = [np.full(int(size), val, dtype=np.float32) for _ in self._vectors]

    def setVectorMergeType(self, num, m):
        self._vmerge[num] = TracingDataMergeEnum(m)

    def setScalarMergeType(self, num, m):
        self._smerge[num] = TracingDataMergeEnum(m)

    def getVectorData(self, key=0):
        if isinstance(key, str):
            i = self.getVectorDataIndex(key)
            if i < 0:
                raise KeyError("Can not find vector data label in TracingData.")
            key = i
        return self._vectors[key]

    def getScalarData(self, key=0):
        if isinstance(key, str):
            i = self.getScalarDataIndex(key)
            if i < 0:
                raise KeyError("Can not find scalar data label in TracingData.")
            key = i
        return self._scalars[key]

    def getVectorDataLabel(self, i):
        return self._vlabels[i] if i < len(self._vlabels) else ""

    def getScalarDataLabel(self, i):
        return self._slabels[i] if i < len(self._slabels) else ""

    def getVectorDataIndex(self, label):
        return self._vlabels.index(label) if label in self._vlabels else -1

    def getScalarDataIndex(self, label):
        return self._slabels.index(label) if label in self._slabels else -1

    def getVectorMergeType(self, num):
        return self._vmerge[num]

    def getScalarMergeType(self, num):
        return self._smerge[num]


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Trace:
    """rayTrace.hpp:15-180 (NumericType = float)."""

    def __init__(self, D=3, device=0):
        self.D = D
        self._L = capi.load()
        h = C.c_void_p()
        rc = self._L.vr_create(C.byref(h), device)
        if rc != capi.VR_OK:
            raise VrError("vr_create failed: no usable HIP device (the accelerated path has no CPU fallback)")
        self._h = h
        self._particle = None
        self._localData = TracingData()
        self._n = 0

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.vr_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _check(self, rc):
        if rc != capi.VR_OK:
            raise VrError(self._L.vr_last_error(self._h).decode())

    # --- setters (rayTrace.hpp:41-121) --------------------------------------
    def _pod(self, particle, keep):
        pod = ParticlePOD()
        pod.kind = particle.kind
        pod.sticking = particle.stickingProbability
        pod.sourcePower = particle.getSourceDistributionPower()
        pod.coneAngle = getattr(particle, "coneAngle", 0.0)
        pod.meanFreePath = getattr(particle, "meanFreePath", -1.0)
        for k, v in enumerate(getattr(particle, "params", [])[:8]):
            pod.params[k] = v
        ms = particle.materialSticking
        if ms:
            ids = (C.c_int32 * len(ms))(*ms.keys())
            vals = (C.c_float * len(ms))(*ms.values())
            pod.numMaterialSticking = len(ms)
            pod.materialIds = ids
            pod.materialSticking = vals
            keep.append((ids, vals))
        return pod

    def setParticleType(self, particle):
        self._particle = particle
        self._particles = [particle]
        keep = []
        pod = self._pod(particle, keep)
        self._check(self._L.vr_set_particle(self._h, C.byref(pod)))
        del keep

    def setParticleTypes(self, particles):
        """Several particles traced in ONE apply() (the reference's gpu::Trace keeps a particle list,
        gpu/raygTrace.hpp:163-248): every particle sees the same seed, particles with the same source
        distribution share one generator pass; getLocalData() holds particle 0's labels, then particle 1's, ..."""
        particles = list(particles)
        keep = []
        arr = (ParticlePOD * len(particles))(*[self._pod(q, keep) for q in particles])
        self._check(self._L.vr_set_particles(self._h, arr, len(particles)))
        self._particle = particles[0]
        self._particles = particles
        del keep

    def registerParticleModel(self, source, numData=1, needsFull=False, name="user"):
        """vr_register_particle_model: HIP source of `struct VrUserModel` -> the kind id of a UserModelParticle"""
        k = C.c_int32(0)
        self._check(self._L.vr_register_particle_model(self._h, name.encode(), source.encode(), int(numData),
                                                       1 if needsFull else 0, C.byref(k)))
        return int(k.value)

    def setGlobalData(self, data):
        """rayTrace.hpp:141: a TracingData (or a list of per-primitive arrays) the device particle models may read"""
        vecs = data._vectors if isinstance(data, TracingData) else list(data)
        self._check(self._L.vr_set_global_data(self._h, 0, None, 0))
        for k, v in enumerate(vecs):
            a = np.ascontiguousarray(v, dtype=np.float32)
            self._check(self._L.vr_set_global_data(self._h, k, _fptr(a), a.size))
        if isinstance(data, TracingData) and data._scalars:
            sc = np.ascontiguousarray(data._scalars, dtype=np.float32)
            self._check(self._L.vr_set_global_scalars(self._h, _fptr(sc), sc.size))
        else:  # (no scalars in the new data: the previous ones must not linger on the device)
            self._check(self._L.vr_set_global_scalars(self._h, None, 0))
        self._globalData = data

    def getGlobalData(self):
        return getattr(self, "_globalData", None)

    def getParticleTraceInfo(self, q):
        pod = TraceInfoPOD()
        self._check(self._L.vr_get_particle_trace_info(self._h, int(q), C.byref(pod)))
        return pod

    def setUseWdist(self, on=True):
        """VIENNARAY_USE_WDIST as a run-time switch (rayTraceKernel.hpp:258-296)"""
        self._check(self._L.vr_set_use_wdist(self._h, int(bool(on))))

    def setSource(self, source):
        """rayTrace.hpp:53-56.  SourceGrid runs natively in the generator kernel; any other object with
        getOriginAndDirection(idx, rng) is a host callback: it is evaluated here for every ray (rng = a
        counting stand-in of the per-ray engine) and the rays are handed to the device."""
        if isinstance(source, SourceGrid):
            self._check(self._L.vr_set_source_grid(self._h, _fptr(source.points), source.points.shape[0]))
            return
        raise VrError("setSource: host-callback sources go through setHostRays(origins, directions, draws)")

    def setHostRays(self, origins, directions, draws=None, weights=None, sourceArea=None):
        """Rays of a host-callback Source (raySource.hpp:10-19): origin, direction, engine outputs consumed and —
        if the source overrides them — getInitialRayWeight(idx) per ray and getSourceArea()."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        assert o.shape == d.shape
        k = None
        if draws is not None:
            k = np.ascontiguousarray(draws, dtype=np.uint32)
            assert k.size == o.shape[0]
        self._check(self._L.vr_set_host_rays(self._h, _fptr(o), _fptr(d),
                                             k.ctypes.data_as(C.POINTER(C.c_uint32)) if k is not None else None,
                                             o.shape[0]))
        if weights is not None:
            w = np.ascontiguousarray(weights, dtype=np.float32)
            assert w.size == o.shape[0]
            self._check(self._L.vr_set_host_ray_weights(self._h, _fptr(w), w.size))
        if sourceArea is not None or hasattr(self._L, "vr_set_source_area"):
            self._check(self._L.vr_set_source_area(self._h, float(sourceArea) if sourceArea is not None else 0.0))

    def reserveRays(self, n):
        """Size the HBM ray stream for applies of up to n rays (apply() per time step with a growing count)."""
        self._check(self._L.vr_reserve_rays(self._h, int(n)))

    def resetSource(self):
        """rayTrace.hpp:58-61"""
        self._check(self._L.vr_set_source_grid(self._h, None, 0))
        self._check(self._L.vr_set_source_area(self._h, 0.0))

    def setBoundaryConditions(self, bcs):
        a = (C.c_int32 * len(bcs))(*[int(b) for b in bcs])
        self._check(self._L.vr_set_boundary_conditions(self._h, a, len(bcs)))

    def setNumberOfRaysPerPoint(self, n):
        self._check(self._L.vr_set_number_of_rays_per_point(self._h, int(n)))

    def setNumberOfRaysFixed(self, n):
        self._check(self._L.vr_set_number_of_rays_fixed(self._h, int(n)))

    def setMaxReflections(self, n):
        self._check(self._L.vr_set_max_reflections(self._h, int(n)))

    def setMaxBoundaryHits(self, n):
        self._check(self._L.vr_set_max_boundary_hits(self._h, int(n)))

    def setSourceDirection(self, d):
        self._check(self._L.vr_set_source_direction(self._h, int(d)))

    def setPrimaryDirection(self, d):
        a = np.ascontiguousarray(d, dtype=np.float32)
        self._check(self._L.vr_set_primary_direction(self._h, _fptr(a)))

    def setUseRandomSeeds(self, b):
        self._check(self._L.vr_set_use_random_seeds(self._h, int(bool(b))))

    def setRngSeed(self, s):
        self._check(self._L.vr_set_rng_seed(self._h, int(s)))

    def setMaterialIds(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        self._check(self._L.vr_set_material_ids(self._h, a.ctypes.data_as(C.POINTER(C.c_int32)), a.size))

    # --- extensions used by the multi-GPU driver and the tests ---------------
    def setRunNumber(self, r):
        self._check(self._L.vr_set_run_number(self._h, int(r)))

    def getRunNumber(self):
        v = C.c_uint32(0)
        self._check(self._L.vr_get_run_number(self._h, C.byref(v)))
        return int(v.value)

    def skipApply(self):
        """What an apply() does to the run number (rayTraceDisk.hpp:54) without tracing: a rank
        whose ray shard is empty stays in step with the others' seeds."""
        self.setRunNumber(self.getRunNumber() + 1)

    def setRayRange(self, first, count):
        self._check(self._L.vr_set_ray_range(self._h, int(first), int(count)))

    def setWorldSize(self, world):
        """Ranks whose accumulators will be summed: head-room of the accumulator-overflow check (vr_set_world_size)."""
        self._check(self._L.vr_set_world_size(self._h, int(world)))

    # --- run ---------------------------------------------------------------
    def apply(self):
        if self._particle is None:
            # checkSettings (rayTraceDisk.hpp:197-200)
            raise VrError("No particle was specified in rayTrace. Aborting.")
        self._check(self._L.vr_apply(self._h))
        self._collect()

    def applySharded(self, rank, world, reduce_fn=None, user=None):
        """vr_apply_sharded: this rank's share of the rays, then `reduce_fn(user, devPtr, count, stream)` (a
        ctypes function pointer, e.g. vr_rccl_allreduce) sums accumulators and counters over the ranks."""
        if self._particle is None:
            raise VrError("No particle was specified in rayTrace. Aborting.")
        fn = C.cast(reduce_fn, C.c_void_p) if reduce_fn is not None else None
        self._check(self._L.vr_apply_sharded(self._h, int(rank), int(world), fn, user))
        self._collect()

    def applyPrepare(self):
        self._check(self._L.vr_apply_prepare(self._h))

    def applyLaunch(self):
        self._check(self._L.vr_apply_launch(self._h))

    def applyFinish(self, collect=True):
        self._check(self._L.vr_apply_finish(self._h))
        if collect:
            self._collect()

    def _collect(self):
        labels = [l for q in getattr(self, "_particles", [self._particle]) for l in q.getLocalDataLabels()]
        self._localData.setNumberOfVectorData(len(labels))
        for l, label in enumerate(labels):
            out = np.empty(self._n, dtype=np.float32)
            self._check(self._L.vr_get_flux_data(self._h, l, _fptr(out), self._n))
            self._localData.setVectorData(l, out, label)

    def numData(self):
        return int(self._L.vr_num_data(self._h))

    def getLocalData(self):
        return self._localData

    def getFluxF64(self):
        out = np.empty(self._n, dtype=np.float64)
        self._check(self._L.vr_get_flux_f64(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), self._n))
        return out

    def getRayTraceInfo(self):
        pod = TraceInfoPOD()
        self._check(self._L.vr_get_trace_info(self._h, C.byref(pod)))
        return pod

    def traceMode(self):
        """trace_kernel variant of the last prepare: 0 general, 1 absorbing/flat, 2 absorbing/structured"""
        v = C.c_int32(0)
        self._check(self._L.vr_get_trace_mode(self._h, C.byref(v)))
        return int(v.value)

    def normalizeFlux(self, flux, norm=NormalizationType.SOURCE):
        f = np.ascontiguousarray(flux, dtype=np.float32).copy()
        self._check(self._L.vr_normalize_flux(self._h, _fptr(f), f.size, int(norm)))
        return f

    def getFluxNormalized(self, norm=NormalizationType.SOURCE):
        """raw flux -> normalizeFlux fused on the device (one download)"""
        out = np.empty(self._n, dtype=np.float32)
        self._check(self._L.vr_get_flux_normalized(self._h, _fptr(out), self._n, int(norm)))
        return out

    def smoothFlux(self, flux, numNeighbors=1):
        f = np.ascontiguousarray(flux, dtype=np.float32).copy()
        self._check(self._L.vr_smooth_flux(self._h, _fptr(f), f.size, int(numNeighbors)))
        return f

    # --- geometry-derived values -----------------------------------------------
    def getBoundingBox(self):
        out = np.empty(6, dtype=np.float32)
        self._check(self._L.vr_get_bounding_box(self._h, _fptr(out)))
        return out.reshape(2, 3)

    def getSourceArea(self):
        return self._L.vr_get_source_area(self._h)

    def fluxAccumulators(self):
        """(device pointer, n) of the int64 fixed-point accumulators."""
        p = C.c_void_p()
        n = C.c_uint32()
        self._check(self._L.vr_flux_accumulators(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bindFluxAccumulators(self, dev_ptr, n):
        """Use a caller-owned device buffer of n int64 (e.g. a torch tensor)."""
        self._check(self._L.vr_bind_flux_accumulators(self._h, C.c_void_p(dev_ptr), int(n)))

    # --- diagnostics ---------------------------------------------------------------
    def debugIntersect(self, org, dirn, tnear=1e-4):
        o = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirn, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        tn = np.full(n, tnear, dtype=np.float32)
        g = np.empty(n, dtype=np.int32)
        p = np.empty(n, dtype=np.uint32)
        t = np.empty(n, dtype=np.float32)
        self._check(self._L.vr_debug_intersect(self._h, _fptr(o), _fptr(d), _fptr(tn), n,
                                               g.ctypes.data_as(C.POINTER(C.c_int32)),
                                               p.ctypes.data_as(C.POINTER(C.c_uint32)), _fptr(t)))
        return g, p, t

    def debugProcessHit(self, org, dirn, tfar, primID):
        """Boundary::processHit on the device: (new origins, new directions, reflect flags)"""
        o = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirn, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(tfar, dtype=np.float32), (n,)))
        p = np.ascontiguousarray(np.broadcast_to(np.asarray(primID, dtype=np.uint32), (n,)))
        oo, do = np.empty_like(o), np.empty_like(d)
        r = np.empty(n, dtype=np.int32)
        self._check(self._L.vr_debug_process_hit(self._h, _fptr(o), _fptr(d), _fptr(t),
                                                 p.ctypes.data_as(C.POINTER(C.c_uint32)), n, _fptr(oo), _fptr(do),
                                                 r.ctypes.data_as(C.POINTER(C.c_int32))))
        return oo, do, r.astype(bool)

    def debugSourceSample(self, idx, seed):
        i = np.ascontiguousarray(idx, dtype=np.uint64)
        o = np.empty((i.size, 3), dtype=np.float32)
        d = np.empty((i.size, 3), dtype=np.float32)
        self._check(self._L.vr_debug_source_sample(self._h, i.ctypes.data_as(C.POINTER(C.c_uint64)), i.size,
                                                   int(seed), _fptr(o), _fptr(d)))
        return o, d

    def debugRngOutputs(self, idx, seed, count):
        out = np.empty(count, dtype=np.uint64)
        self._check(self._L.vr_debug_rng_outputs(self._h, int(idx), int(seed), count,
                                                 out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def debugIssueRate(self, kind, waves_per_simd, iters=20000):
        """Measured issue ceiling: dict(rate [counted instr/s], clock_hz, seconds, count)."""
        out = (C.c_double * 4)()
        self._check(self._L.vr_debug_issue_rate(self._h, int(kind), int(waves_per_simd), int(iters), out))
        return dict(rate=out[0], clock_hz=out[1], seconds=out[2], count=out[3])

    def debugBvhCheck(self):
        v = C.c_uint32(0)
        self._check(self._L.vr_debug_bvh_check(self._h, C.byref(v)))
        return int(v.value)

    def debugBvhStats(self):
        a = (C.c_uint32 * 3)()
        self._check(self._L.vr_debug_bvh_stats(self._h, a))
        return dict(nodes=a[0], leaves=a[1], maxDepth=a[2])


class TraceDisk(Trace):
    """rayTraceDisk.hpp:13-224"""

    def setGeometry(self, points, normals, gridDelta, diskRadius=0.0):
        p = np.ascontiguousarray(points, dtype=np.float32)
        n = np.ascontiguousarray(normals, dtype=np.float32)
        if p.shape[1] == 2:  # 2-D points: z := 0 (rayGeometryDisk.hpp:148-151)
            p = np.concatenate([p, np.zeros((p.shape[0], 1), np.float32)], axis=1)
            n = np.concatenate([n, np.zeros((n.shape[0], 1), np.float32)], axis=1)
        p = np.ascontiguousarray(p)
        n = np.ascontiguousarray(n)
        assert p.shape == n.shape, "Geometry: Points/Normals size mismatch"
        self._n = p.shape[0]
        self._check(self._L.vr_set_disks(self._h, _fptr(p), _fptr(n), self._n, float(gridDelta),
                                         float(diskRadius), self.D))

    def getDiskAreas(self):
        out = np.empty(self._n, dtype=np.float32)
        self._check(self._L.vr_get_disk_areas(self._h, _fptr(out), self._n))
        return out

    def getDiskRadius(self):
        return self._L.vr_get_disk_radius(self._h)

    def getNeighborCounts(self):
        out = np.empty(self._n, dtype=np.uint32)
        self._check(self._L.vr_get_neighbor_counts(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), self._n))
        return out


class TraceTriangle(Trace):
    """rayTraceTriangle.hpp:13-154"""

    def setLineGeometry(self, nodes, lines, gridDelta):
        """setGeometry(LineMesh) (rayTraceTriangle.hpp:76-81, D == 2): lines -> triangle strips"""
        from . import io
        v, t, _ = io.lines_to_triangles(nodes, lines, gridDelta)
        self.setGeometry(v, t, gridDelta)

    def setGeometry(self, points, triangles, gridDelta):
        v = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3)
        self._n = t.shape[0]
        self._check(self._L.vr_set_triangles(self._h, _fptr(v), v.shape[0],
                                             t.ctypes.data_as(C.POINTER(C.c_uint32)), self._n,
                                             float(gridDelta), self.D))

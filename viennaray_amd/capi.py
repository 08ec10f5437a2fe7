"""ctypes binding of the C ABI declared in include/viennaray_amd.h.

The shared library is built in-tree (viennaray_amd/libviennaray_amd.so, see
viennaray_amd/csrc/Makefile).  There is NO CPU fallback: if the library is
missing or no HIP device is usable, calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VR_LIB_PATH") or os.path.join(_HERE, "libviennaray_amd.so")  # (override: A/B runs of two builds)

VR_OK, VR_E_INVALID, VR_E_HIP, VR_E_STATE = 0, -1, -2, -3


class VrError(RuntimeError):
    pass


class TraceInfoPOD(C.Structure):
    """vr_trace_info (include/viennaray_amd.h) == rayUtil.hpp:65-76"""
    _fields_ = [("numRays", C.c_uint64), ("totalRaysTraced", C.c_uint64),
                ("nonGeometryHits", C.c_uint64), ("geometryHits", C.c_uint64),
                ("particleHits", C.c_uint64), ("boundaryHits", C.c_uint64),
                ("reflections", C.c_uint64), ("raysTerminated", C.c_uint64),
                ("time", C.c_double), ("timeBuild", C.c_double), ("timeTrace", C.c_double),
                ("timeTraceKernel", C.c_double),
                ("warning", C.c_int32), ("error", C.c_int32), ("rngFullStates", C.c_uint64),
                ("timeGenKernel", C.c_double), ("bvhRefits", C.c_uint32), ("bvhBuilds", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class ParticlePOD(C.Structure):
    _fields_ = [("kind", C.c_int32), ("sticking", C.c_float), ("sourcePower", C.c_float),
                ("numMaterialSticking", C.c_int32), ("materialIds", C.POINTER(C.c_int32)),
                ("materialSticking", C.POINTER(C.c_float)), ("coneAngle", C.c_float), ("meanFreePath", C.c_float),
                ("params", C.c_float * 8)]


# every symbol include/viennaray_amd.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_fp = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)
SIGNATURES = {
    "vr_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "vr_destroy": (None, [_vp]),
    "vr_last_error": (C.c_char_p, [_vp]),
    "vr_device_available": (C.c_int, []),
    "vr_version": (C.c_char_p, []),
    "vr_set_disks": (C.c_int, [_vp, _fp, _fp, C.c_uint32, C.c_float, C.c_float, C.c_int]),
    "vr_set_triangles": (C.c_int, [_vp, _fp, C.c_uint32, _u32p, C.c_uint32, C.c_float, C.c_int]),
    "vr_set_material_ids": (C.c_int, [_vp, _i32p, C.c_uint32]),
    "vr_set_boundary_conditions": (C.c_int, [_vp, _i32p, C.c_int]),
    "vr_set_source_direction": (C.c_int, [_vp, C.c_int]),
    "vr_set_primary_direction": (C.c_int, [_vp, _fp]),
    "vr_set_particle": (C.c_int, [_vp, C.POINTER(ParticlePOD)]),
    "vr_set_particles": (C.c_int, [_vp, C.POINTER(ParticlePOD), C.c_uint32]),
    "vr_register_particle_model": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int, _i32p]),
    "vr_set_global_data": (C.c_int, [_vp, C.c_uint32, _fp, C.c_uint32]),
    "vr_set_global_scalars": (C.c_int, [_vp, _fp, C.c_uint32]),
    "vr_get_particle_trace_info": (C.c_int, [_vp, C.c_uint32, C.POINTER(TraceInfoPOD)]),
    "vr_set_use_wdist": (C.c_int, [_vp, C.c_int]),
    "vr_set_source_grid": (C.c_int, [_vp, _fp, C.c_uint32]),
    "vr_set_host_rays": (C.c_int, [_vp, _fp, _fp, _u32p, C.c_uint64]),
    "vr_set_host_ray_weights": (C.c_int, [_vp, _fp, C.c_uint64]),
    "vr_set_source_area": (C.c_int, [_vp, C.c_float]),
    "vr_reserve_rays": (C.c_int, [_vp, C.c_uint64]),
    "vr_set_number_of_rays_per_point": (C.c_int, [_vp, C.c_uint64]),
    "vr_set_number_of_rays_fixed": (C.c_int, [_vp, C.c_uint64]),
    "vr_set_max_reflections": (C.c_int, [_vp, C.c_uint32]),
    "vr_set_max_boundary_hits": (C.c_int, [_vp, C.c_uint32]),
    "vr_set_rng_seed": (C.c_int, [_vp, C.c_uint32]),
    "vr_set_use_random_seeds": (C.c_int, [_vp, C.c_int]),
    "vr_set_run_number": (C.c_int, [_vp, C.c_uint32]),
    "vr_get_run_number": (C.c_int, [_vp, _u32p]),
    "vr_set_ray_range": (C.c_int, [_vp, C.c_uint64, C.c_uint64]),
    "vr_set_world_size": (C.c_int, [_vp, C.c_uint32]),
    "vr_apply": (C.c_int, [_vp]),
    "vr_apply_prepare": (C.c_int, [_vp]),
    "vr_apply_launch": (C.c_int, [_vp]),
    "vr_apply_finish": (C.c_int, [_vp]),
    "vr_apply_sharded": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "vr_num_primitives": (C.c_uint32, [_vp]),
    "vr_get_flux": (C.c_int, [_vp, _fp, C.c_uint32]),
    "vr_get_flux_f64": (C.c_int, [_vp, C.POINTER(C.c_double), C.c_uint32]),
    "vr_num_data": (C.c_uint32, [_vp]),
    "vr_get_flux_data": (C.c_int, [_vp, C.c_uint32, _fp, C.c_uint32]),
    "vr_get_trace_info": (C.c_int, [_vp, C.POINTER(TraceInfoPOD)]),
    "vr_get_trace_mode": (C.c_int, [_vp, _i32p]),
    "vr_normalize_flux": (C.c_int, [_vp, _fp, C.c_uint32, C.c_int]),
    "vr_get_flux_normalized": (C.c_int, [_vp, _fp, C.c_uint32, C.c_int]),
    "vr_smooth_flux": (C.c_int, [_vp, _fp, C.c_uint32, C.c_int]),
    "vr_get_disk_areas": (C.c_int, [_vp, _fp, C.c_uint32]),
    "vr_get_bounding_box": (C.c_int, [_vp, _fp]),
    "vr_get_source_area": (C.c_float, [_vp]),
    "vr_get_disk_radius": (C.c_float, [_vp]),
    "vr_get_neighbor_counts": (C.c_int, [_vp, _u32p, C.c_uint32]),
    "vr_flux_accumulators": (C.c_int, [_vp, C.POINTER(_vp), _u32p]),
    "vr_bind_flux_accumulators": (C.c_int, [_vp, _vp, C.c_uint32]),
    "vr_add_trace_info": (C.c_int, [_vp, C.POINTER(TraceInfoPOD)]),
    "vr_stream": (_vp, [_vp]),
    "vr_debug_intersect": (C.c_int, [_vp, _fp, _fp, _fp, C.c_uint32, _i32p, _u32p, _fp]),
    "vr_debug_process_hit": (C.c_int, [_vp, _fp, _fp, _fp, _u32p, C.c_uint32, _fp, _fp, _i32p]),
    "vr_debug_source_sample": (C.c_int, [_vp, _u64p, C.c_uint32, C.c_uint32, _fp, _fp]),
    "vr_debug_rng_outputs": (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_uint32, _u64p]),
    "vr_debug_issue_rate": (C.c_int, [_vp, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_double)]),
    "vr_debug_bvh_stats": (C.c_int, [_vp, _u32p]),
    "vr_debug_bvh_check": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
}

_lib = None


def load():
    """Load the HIP library; raises VrError if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VrError(f"{LIB_PATH} not found: build it with `make -C viennaray_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("VR_LIB_PATH") and not hasattr(L, name):
                continue  # (an A/B run against an older build of the library: it lacks the newer entry points)
            fn = getattr(L, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def device_available():
    return bool(load().vr_device_available())

"""ctypes binding of the CPU oracle (oracle/vr_oracle.cpp).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never from the viennaray_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvr_oracle.so")

DIFFUSE, SPECULAR, CONED_COSINE, DIFFUSE_COSINE, COVERAGE_STICKING = 0, 1, 2, 3, 4
REFLECTIVE, PERIODIC, IGNORE = 0, 1, 2
POS_X, NEG_X, POS_Y, NEG_Y, POS_Z, NEG_Z = range(6)

INFO_FIELDS = ("numRays", "totalRaysTraced", "nonGeometryHits", "geometryHits",
               "particleHits", "boundaryHits", "reflections", "raysTerminated")


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "vr_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        up = C.POINTER(C.c_uint)
        u64p = C.POINTER(C.c_uint64)
        vp = C.c_void_p
        L.orc_create.restype = vp
        L.orc_destroy.argtypes = [vp]
        L.orc_set_disks.argtypes = [vp, fp, fp, C.c_uint, C.c_float, C.c_float, C.c_int]
        L.orc_set_triangles.argtypes = [vp, fp, C.c_uint, up, C.c_uint, C.c_float, C.c_int]
        L.orc_set_material_ids.argtypes = [vp, C.POINTER(C.c_int), C.c_uint]
        L.orc_set_boundary_conditions.argtypes = [vp, C.POINTER(C.c_int), C.c_int]
        L.orc_set_source_direction.argtypes = [vp, C.c_int]
        L.orc_set_primary_direction.argtypes = [vp, fp]
        L.orc_set_particle.argtypes = [vp, C.c_int, C.c_float, C.c_float]
        L.orc_set_particle_ex.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_set_material_sticking.argtypes = [vp, C.POINTER(C.c_int), fp, C.c_int]
        L.orc_set_wdist.argtypes = [vp, C.c_int]
        L.orc_set_global_data.argtypes = [vp, C.c_uint, fp, C.c_uint]
        L.orc_set_source_grid.argtypes = [vp, fp, C.c_uint]
        L.orc_set_host_rays.argtypes = [vp, fp, fp, C.c_uint]
        L.orc_set_host_ray_weights.argtypes = [vp, fp, C.c_uint]
        L.orc_set_host_ray_draws.argtypes = [vp, C.POINTER(C.c_uint), C.c_uint]
        L.orc_set_source_area.argtypes = [vp, C.c_float]
        L.orc_create_source_grid.argtypes = [vp, C.c_uint64, C.c_float, fp, C.c_uint]
        L.orc_create_source_grid.restype = C.c_uint
        L.orc_num_data.argtypes = [vp]
        L.orc_num_data.restype = C.c_int
        L.orc_get_flux_data.argtypes = [vp, C.c_int, fp]
        L.orc_set_num_rays_per_point.argtypes = [vp, C.c_uint64]
        L.orc_set_num_rays_fixed.argtypes = [vp, C.c_uint64]
        L.orc_set_max_reflections.argtypes = [vp, C.c_uint]
        L.orc_set_max_boundary_hits.argtypes = [vp, C.c_uint]
        L.orc_set_rng_seed.argtypes = [vp, C.c_uint]
        L.orc_set_use_random_seeds.argtypes = [vp, C.c_int]
        L.orc_set_run_number.argtypes = [vp, C.c_uint]
        L.orc_set_ray_range.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.orc_set_lazy_rng.argtypes = [vp, C.c_int]
        L.orc_set_event_capacity.argtypes = [vp, C.c_uint64]
        L.orc_prepare.argtypes = [vp]
        L.orc_apply.argtypes = [vp, C.c_int]
        L.orc_num_prims.argtypes = [vp]
        L.orc_num_prims.restype = C.c_uint
        L.orc_get_flux.argtypes = [vp, fp]
        L.orc_get_info.argtypes = [vp, u64p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.orc_get_bbox.argtypes = [vp, fp]
        L.orc_get_geometry_bbox.argtypes = [vp, fp]
        L.orc_get_source_area.argtypes = [vp]
        L.orc_get_source_area.restype = C.c_float
        L.orc_get_disk_radius.argtypes = [vp]
        L.orc_get_disk_radius.restype = C.c_float
        L.orc_get_disk_areas.argtypes = [vp, fp]
        L.orc_get_tri_areas.argtypes = [vp, fp]
        L.orc_get_normals.argtypes = [vp, fp]
        L.orc_neighbor_count.argtypes = [vp, C.c_uint]
        L.orc_neighbor_count.restype = C.c_uint
        L.orc_get_neighbors.argtypes = [vp, C.c_uint, up]
        L.orc_normalize_flux.argtypes = [vp, fp, C.c_int]
        L.orc_smooth_flux.argtypes = [vp, fp, C.c_int]
        L.orc_num_events.argtypes = [vp]
        L.orc_num_events.restype = C.c_uint64
        L.orc_get_events.argtypes = [vp, u64p, C.POINTER(C.c_int), up, fp, fp]
        L.orc_tea3.argtypes = [C.c_uint, C.c_uint]
        L.orc_tea3.restype = C.c_uint
        L.orc_mt64_outputs.argtypes = [C.c_uint64, C.c_int, u64p, C.c_int]
        L.orc_uniform_float.argtypes = [u64p, C.c_int, C.c_float, C.c_float, fp]
        L.orc_uniform_double.argtypes = [u64p, C.c_int, C.POINTER(C.c_double)]
        L.orc_source_sample.argtypes = [vp, C.c_uint64, C.c_uint, fp, fp]
        L.orc_intersect1.argtypes = [vp, fp, fp, C.c_float, C.c_int, up, fp, fp]
        L.orc_intersect1.restype = C.c_int
        L.orc_boundary_process_hit.argtypes = [vp, fp, fp, C.c_float, C.c_uint, fp, fp]
        L.orc_boundary_process_hit.restype = C.c_int
        L.orc_wall_normal.argtypes = [vp, C.c_uint, fp]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def tea3(v0, v1):
    return lib().orc_tea3(v0 & 0xFFFFFFFF, v1 & 0xFFFFFFFF)


def mt64_outputs(seed, n, lazy=False):
    out = np.empty(n, dtype=np.uint64)
    lib().orc_mt64_outputs(seed, n, out.ctypes.data_as(C.POINTER(C.c_uint64)), int(lazy))
    return out


def uniform_float(raw, a=0.0, b=1.0):
    raw = np.ascontiguousarray(raw, dtype=np.uint64)
    out = np.empty(raw.size, dtype=np.float32)
    lib().orc_uniform_float(raw.ctypes.data_as(C.POINTER(C.c_uint64)), raw.size, a, b, _fp(out))
    return out


def uniform_double(raw):
    raw = np.ascontiguousarray(raw, dtype=np.uint64)
    out = np.empty(raw.size, dtype=np.float64)
    lib().orc_uniform_double(raw.ctypes.data_as(C.POINTER(C.c_uint64)), raw.size,
                             out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def max_threads():
    return lib().orc_max_threads()


class Oracle:
    """Mirror of the reference front-end on top of the oracle library."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_create())
        self.n = 0

    def __del__(self):
        try:
            if self.h:
                self.L.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # geometry -----------------------------------------------------------
    def set_disks(self, points, normals, grid_delta, D=3, radius=0.0):
        p = _f32(points).reshape(-1, 3)
        n = _f32(normals).reshape(-1, 3)
        assert p.shape == n.shape
        self.n = p.shape[0]
        self.L.orc_set_disks(self.h, _fp(p), _fp(n), self.n, grid_delta, radius, D)

    def set_triangles(self, verts, tris, grid_delta, D=3):
        v = _f32(verts).reshape(-1, 3)
        t = np.ascontiguousarray(tris, dtype=np.uint32).reshape(-1, 3)
        self.n = t.shape[0]
        self.L.orc_set_triangles(self.h, _fp(v), v.shape[0],
                                 t.ctypes.data_as(C.POINTER(C.c_uint)), self.n, grid_delta, D)

    def set_material_ids(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        self.L.orc_set_material_ids(self.h, a.ctypes.data_as(C.POINTER(C.c_int)), a.size)

    # configuration ------------------------------------------------------
    def set_boundary_conditions(self, bcs):
        a = (C.c_int * len(bcs))(*[int(b) for b in bcs])
        self.L.orc_set_boundary_conditions(self.h, a, len(bcs))

    def set_source_direction(self, d):
        self.L.orc_set_source_direction(self.h, int(d))

    def set_primary_direction(self, d):
        if d is None:
            self.L.orc_set_primary_direction(self.h, None)
        else:
            a = _f32(d)
            self.L.orc_set_primary_direction(self.h, _fp(a))

    def set_particle(self, kind, sticking, source_power=1.0):
        self.L.orc_set_particle(self.h, kind, sticking, source_power)

    def set_particle_ex(self, kind, sticking, source_power=1.0, cone_angle=0.0, mean_free_path=-1.0):
        self.L.orc_set_particle_ex(self.h, kind, sticking, source_power, cone_angle, mean_free_path)

    def set_material_sticking(self, mapping):
        ids = (C.c_int * len(mapping))(*mapping.keys())
        vals = (C.c_float * len(mapping))(*mapping.values())
        self.L.orc_set_material_sticking(self.h, ids, vals, len(mapping))

    def set_global_data(self, idx, data):
        """Trace::setGlobalData: vector `idx` (None drops it)"""
        if data is None:
            self.L.orc_set_global_data(self.h, idx, None, 0)
        else:
            a = _f32(data)
            self.L.orc_set_global_data(self.h, idx, _fp(a), a.size)

    def set_wdist(self, on=True):
        self.L.orc_set_wdist(self.h, int(on))

    def set_source_grid(self, pts):
        if pts is None:
            self.L.orc_set_source_grid(self.h, None, 0)
        else:
            a = _f32(pts).reshape(-1, 3)
            self.L.orc_set_source_grid(self.h, _fp(a), a.shape[0])

    def set_host_rays(self, org, dirn, weights=None, source_area=None):
        o, d = _f32(org).reshape(-1, 3), _f32(dirn).reshape(-1, 3)
        self.L.orc_set_host_rays(self.h, _fp(o), _fp(d), o.shape[0])
        if weights is not None:
            w = _f32(weights)
            assert w.size == o.shape[0]
            self.L.orc_set_host_ray_weights(self.h, _fp(w), w.size)
        self.L.orc_set_source_area(self.h, float(source_area) if source_area is not None else 0.0)

    def set_host_ray_draws(self, draws):
        k = np.ascontiguousarray(draws, dtype=np.uint32)
        self.L.orc_set_host_ray_draws(self.h, k.ctypes.data_as(C.POINTER(C.c_uint)), k.size)

    def create_source_grid(self, num_points, grid_delta):
        out = np.empty((int(num_points) * 2 + 64, 3), dtype=np.float32)
        n = self.L.orc_create_source_grid(self.h, int(num_points), grid_delta, _fp(out), out.shape[0])
        return out[:n].copy()

    def num_data(self):
        return self.L.orc_num_data(self.h)

    def flux_data(self, idx):
        out = np.empty(self.n, dtype=np.float32)
        self.L.orc_get_flux_data(self.h, idx, _fp(out))
        return out

    def set_num_rays_per_point(self, n):
        self.L.orc_set_num_rays_per_point(self.h, n)

    def set_num_rays_fixed(self, n):
        self.L.orc_set_num_rays_fixed(self.h, n)

    def set_max_reflections(self, n):
        self.L.orc_set_max_reflections(self.h, n)

    def set_max_boundary_hits(self, n):
        self.L.orc_set_max_boundary_hits(self.h, n)

    def set_rng_seed(self, s):
        self.L.orc_set_rng_seed(self.h, s)

    def set_run_number(self, r):
        self.L.orc_set_run_number(self.h, r)

    def set_ray_range(self, first, count):
        self.L.orc_set_ray_range(self.h, first, count)

    def set_lazy_rng(self, b=True):
        self.L.orc_set_lazy_rng(self.h, int(b))

    def set_event_capacity(self, n):
        self.L.orc_set_event_capacity(self.h, n)

    # run ----------------------------------------------------------------
    def prepare(self):
        self.L.orc_prepare(self.h)

    def apply(self, threads=1):
        self.L.orc_apply(self.h, threads)

    def flux(self):
        out = np.empty(self.n, dtype=np.float32)
        self.L.orc_get_flux(self.h, _fp(out))
        return out

    def info(self):
        a = (C.c_uint64 * 8)()
        t = C.c_double()
        we = (C.c_int * 2)()
        self.L.orc_get_info(self.h, a, C.byref(t), we)
        d = {k: int(a[i]) for i, k in enumerate(INFO_FIELDS)}
        d["time"] = t.value
        d["warning"] = bool(we[0])
        d["error"] = bool(we[1])
        return d

    def bbox(self):
        out = np.empty(6, dtype=np.float32)
        self.L.orc_get_bbox(self.h, _fp(out))
        return out.reshape(2, 3)

    def geometry_bbox(self):
        out = np.empty(6, dtype=np.float32)
        self.L.orc_get_geometry_bbox(self.h, _fp(out))
        return out.reshape(2, 3)

    def source_area(self):
        return self.L.orc_get_source_area(self.h)

    def disk_radius(self):
        return self.L.orc_get_disk_radius(self.h)

    def disk_areas(self):
        out = np.empty(self.n, dtype=np.float32)
        self.L.orc_get_disk_areas(self.h, _fp(out))
        return out

    def tri_areas(self):
        out = np.empty(self.n, dtype=np.float32)
        self.L.orc_get_tri_areas(self.h, _fp(out))
        return out

    def normals(self):
        out = np.empty((self.n, 3), dtype=np.float32)
        self.L.orc_get_normals(self.h, _fp(out))
        return out

    def neighbors(self, idx):
        k = self.L.orc_neighbor_count(self.h, idx)
        out = np.empty(k, dtype=np.uint32)
        if k:
            self.L.orc_get_neighbors(self.h, idx, out.ctypes.data_as(C.POINTER(C.c_uint)))
        return out

    def neighbor_counts(self):
        return np.array([self.L.orc_neighbor_count(self.h, i) for i in range(self.n)])

    def normalize_flux(self, flux, norm=0):
        f = _f32(flux).copy()
        self.L.orc_normalize_flux(self.h, _fp(f), norm)
        return f

    def smooth_flux(self, flux, num_neighbors=1):
        f = _f32(flux).copy()
        self.L.orc_smooth_flux(self.h, _fp(f), num_neighbors)
        return f

    def events(self):
        n = self.L.orc_num_events(self.h)
        ray = np.empty(n, dtype=np.uint64)
        kind = np.empty(n, dtype=np.int32)
        prim = np.empty(n, dtype=np.uint32)
        t = np.empty(n, dtype=np.float32)
        w = np.empty(n, dtype=np.float32)
        if n:
            self.L.orc_get_events(self.h, ray.ctypes.data_as(C.POINTER(C.c_uint64)),
                                  kind.ctypes.data_as(C.POINTER(C.c_int)),
                                  prim.ctypes.data_as(C.POINTER(C.c_uint)), _fp(t), _fp(w))
        return dict(ray=ray, kind=kind, prim=prim, t=t, weight=w)

    # components ---------------------------------------------------------
    def source_sample(self, idx, seed):
        o = np.empty(3, dtype=np.float32)
        d = np.empty(3, dtype=np.float32)
        self.L.orc_source_sample(self.h, idx, seed, _fp(o), _fp(d))
        return o, d

    def intersect1(self, org, dirn, tnear=1e-4, brute=False):
        o, d = _f32(org), _f32(dirn)
        prim = C.c_uint()
        t = C.c_float()
        ng = np.empty(3, dtype=np.float32)
        g = self.L.orc_intersect1(self.h, _fp(o), _fp(d), tnear, int(brute),
                                  C.byref(prim), C.byref(t), _fp(ng))
        if g < 0:
            return dict(geomID=-1, primID=None, t=None, Ng=None)
        return dict(geomID=g, primID=prim.value, t=t.value, Ng=ng)

    def boundary_process_hit(self, org, dirn, tfar, prim_id, ray_direction=None):
        o, d = _f32(org).copy(), _f32(dirn).copy()
        rd = _f32(dirn if ray_direction is None else ray_direction).copy()
        ng = np.empty(3, dtype=np.float32)
        self.L.orc_wall_normal(self.h, prim_id, _fp(ng))
        r = self.L.orc_boundary_process_hit(self.h, _fp(o), _fp(d), tfar, prim_id, _fp(ng), _fp(rd))
        return dict(reflect=bool(r), org=o, dir=d, rayDirection=rd, Ng=ng)

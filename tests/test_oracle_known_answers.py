"""Pins the CPU oracle against every known answer the reference's own tests
hold for this path.  Each test cites the reference test it
restates.  CPU only."""
import numpy as np
import pytest

from oracle import pyoracle as po
from viennaray_amd import io
from helpers import DISK_FACTOR_3D, sphere3d, trench2d


def make_plane(extent, delta, direction=(0, 1, 2), radius=None, D=3):
    pts, nrm = io.create_plane_grid(delta, extent, direction)
    o = po.Oracle()
    o.set_disks(pts, nrm, delta, D, radius=0.0 if radius is None else radius)
    return o, pts, nrm


def test_intersection_known_answers():
    """tests/intersectionTest/intersectionTest.cpp:91-92,126-127"""
    r = np.float32(0.5 * DISK_FACTOR_3D)
    o, pts, _ = make_plane(10, 0.5, radius=r)
    assert pts.shape[0] == 41 * 41
    o.set_source_direction(po.POS_Z)
    o.prepare()
    for brute in (False, True):
        h = o.intersect1([0, 0, 2 * r], [0, 0, -1], tnear=0.0, brute=brute)
        assert h["geomID"] == 1 and h["primID"] == 840
        d = np.array([0, 2, -1.0])
        d /= np.linalg.norm(d)
        h = o.intersect1([0, 9, 2 * r], d, tnear=0.0, brute=brute)
        assert h["geomID"] == 0 and h["primID"] == 7


def test_bvh_matches_brute_force_random_rays():
    _, pts, nrm = trench2d()
    gd, p3, n3 = sphere3d()
    o = po.Oracle()
    o.set_disks(p3, n3, gd, 3)
    o.set_source_direction(po.POS_Z)
    o.prepare()
    rng = np.random.default_rng(3)
    for _ in range(400):
        org = rng.uniform(-1.5, 1.5, 3)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        a = o.intersect1(org, d)
        b = o.intersect1(org, d, brute=True)
        assert a["geomID"] == b["geomID"] and a["primID"] == b["primID"]
        if a["geomID"] >= 0:
            assert a["t"] == b["t"]


def test_boundary_hit_3d():
    """tests/boundaryHit/boundaryHit.cpp:68-76,128-136,188-196"""
    eps = 1e-6
    # reflective x wall, POS_Z
    o, _, _ = make_plane(1.0, 0.1, (0, 1, 2), radius=np.float32(0.1))
    o.set_boundary_conditions([po.REFLECTIVE, po.PERIODIC, po.PERIODIC])
    o.set_source_direction(po.POS_Z)
    # the reference test pads the bbox with gridDelta (= the disk radius here)
    o.prepare()
    d = np.array([0.5, 0.0, -0.25], dtype=np.float32)
    dist = np.float32(np.linalg.norm(d))
    dn = d / dist
    r = o.boundary_process_hit([0.5, 0.5, 0.5], dn, dist, 2)
    assert r["reflect"]
    assert np.allclose(r["org"], [1.0, 0.5, 0.25], atol=eps)
    assert np.allclose(r["dir"], [-dn[0], dn[1], dn[2]], atol=eps)
    assert np.allclose(r["Ng"] / np.linalg.norm(r["Ng"]), [-1, 0, 0])

    # reflective z wall, POS_Y, plane in x-z
    o, _, _ = make_plane(1.0, 0.1, (0, 2, 1), radius=np.float32(0.1))
    o.set_boundary_conditions([po.PERIODIC, po.PERIODIC, po.REFLECTIVE])
    o.set_source_direction(po.POS_Y)
    o.prepare()
    d = np.array([0.0, -0.25, 0.5], dtype=np.float32)
    dist = np.float32(np.linalg.norm(d))
    dn = d / dist
    r = o.boundary_process_hit([0.5, 0.5, 0.5], dn, dist, 6)
    assert r["reflect"]
    assert np.allclose(r["org"], [0.5, 0.25, 1.0], atol=eps)
    assert np.allclose(r["dir"], [dn[0], dn[1], -dn[2]], atol=eps)
    assert np.allclose(r["Ng"] / np.linalg.norm(r["Ng"]), [0, 0, -1])

    # periodic x wall
    o, _, _ = make_plane(1.0, 0.1, (0, 1, 2), radius=np.float32(0.1))
    o.set_boundary_conditions([po.PERIODIC] * 3)
    o.set_source_direction(po.POS_Z)
    o.prepare()
    d = np.array([0.5, 0.0, -0.25], dtype=np.float32)
    dist = np.float32(np.linalg.norm(d))
    dn = d / dist
    r = o.boundary_process_hit([0.5, 0.5, 0.5], dn, dist, 2)
    assert r["reflect"]
    assert np.allclose(r["org"], [-1.0, 0.5, 0.25], atol=eps)
    assert np.allclose(r["dir"], dn, atol=eps)


def _line_2d(axis):
    vals = np.arange(-2, 2.0001, 0.5, dtype=np.float32)
    pts = np.zeros((vals.size, 3), dtype=np.float32)
    pts[:, axis] = vals
    nrm = np.zeros_like(pts)
    nrm[:, 1 - axis] = 1
    return pts, nrm


def test_boundary_hit_2d():
    """tests/boundaryHit2D/boundaryHit2D.cpp:73-79,122-128,185-190,234-239"""
    eps = 1e-6
    pts, nrm = _line_2d(1)  # points along y, normals +x, traced from POS_X
    for bc, ynew in ((po.REFLECTIVE, 2.0), (po.PERIODIC, -2.0)):
        o = po.Oracle()
        o.set_disks(pts, nrm, 0.5, 2, radius=np.float32(0.5))
        o.set_boundary_conditions([po.REFLECTIVE, bc])
        o.set_source_direction(po.POS_X)
        o.prepare()
        d = np.array([-0.5, 1.0, 0.0], dtype=np.float32)
        dist = np.float32(np.linalg.norm(d))
        dn = d / dist
        r = o.boundary_process_hit([1.0, 1.0, 0.0], dn, dist, 3)
        assert r["reflect"]
        assert np.allclose(r["org"], [0.5, ynew, 0.0], atol=eps)
        exp = [dn[0], -dn[1], 0] if bc == po.REFLECTIVE else dn
        assert np.allclose(r["dir"], exp, atol=eps)
    pts, nrm = _line_2d(0)  # points along x, normals +y, traced from POS_Y
    for bc, xnew in ((po.REFLECTIVE, 2.0), (po.PERIODIC, -2.0)):
        o = po.Oracle()
        o.set_disks(pts, nrm, 0.5, 2, radius=np.float32(0.5))
        o.set_boundary_conditions([bc, po.REFLECTIVE])
        o.set_source_direction(po.POS_Y)
        o.prepare()
        d = np.array([1.0, -0.5, 0.0], dtype=np.float32)
        dist = np.float32(np.linalg.norm(d))
        dn = d / dist
        r = o.boundary_process_hit([1.0, 1.0, 0.0], dn, dist, 3)
        assert np.allclose(r["org"], [xnew, 0.5, 0.0], atol=eps)


def test_build_boundary_bbox():
    """tests/buildBoundary/buildBoundary.cpp:34-39, createGeometry.cpp:27-33"""
    gd, p3, n3 = sphere3d()
    o = po.Oracle()
    o.set_disks(p3, n3, gd, 3)
    g = o.geometry_bbox()
    assert np.allclose(g[0], [-1, -1, -1], atol=1e-6) and np.allclose(g[1], [1, 1, 1], atol=1e-6)
    r = o.disk_radius()
    for direction, axis, side in ((po.POS_X, 0, 1), (po.NEG_X, 0, 0), (po.POS_Y, 1, 1),
                                  (po.NEG_Y, 1, 0), (po.POS_Z, 2, 1), (po.NEG_Z, 2, 0)):
        o.set_source_direction(direction)
        o.prepare()
        b = o.bbox()
        assert (b[0] <= b[1]).all()
        exp = g.copy()
        exp[side, axis] += (2 * r) if side else (-2 * r)
        assert np.allclose(b, exp, atol=1e-6)


@pytest.mark.parametrize("direction,axis,sign", [(po.POS_X, 0, -1), (po.NEG_X, 0, 1), (po.POS_Y, 1, -1),
                                                 (po.NEG_Y, 1, 1), (po.POS_Z, 2, -1), (po.NEG_Z, 2, 1)])
def test_create_ray_planes_and_signs(direction, axis, sign):
    """tests/createRay/createRay.cpp:55-56..170-193: origin on the extended
    face, direction component sign; tilted primary direction still points in."""
    gd, p3, n3 = sphere3d()
    o = po.Oracle()
    o.set_disks(p3, n3, gd, 3)
    o.set_source_direction(direction)
    o.set_particle(po.SPECULAR, 1.0, 2.0)
    o.prepare()
    b = o.bbox()
    face = b[1, axis] if sign < 0 else b[0, axis]
    for idx in range(200):
        org, d = o.source_sample(idx, 31)
        assert org[axis] == face
        assert (b[0] <= org).all() and (org <= b[1]).all()
        assert d[axis] * sign > 0
        assert abs(np.linalg.norm(d.astype(np.float64)) - 1) < 1e-5
    prim = np.zeros(3, dtype=np.float32)
    prim[axis] = sign
    prim[(axis + 1) % 3] = 1.0
    o.set_primary_direction(prim)
    o.prepare()
    for idx in range(200):
        org, d = o.source_sample(idx, 31)
        assert d[axis] * sign >= 0
        assert abs(np.linalg.norm(d.astype(np.float64)) - 1) < 1e-5


def test_disk_areas():
    """tests/diskAreas/diskAreas.cpp:58-61,76-96 (full / half / quarter)"""
    o, pts, _ = make_plane(2, 1.0)
    o.set_source_direction(po.POS_Z)
    o.set_num_rays_per_point(1)
    o.set_rng_seed(0)
    o.apply(1)
    areas = o.disk_areas()
    r = o.disk_radius()
    whole = r * r * np.pi
    g = o.geometry_bbox()
    for i, p in enumerate(pts):
        onx = abs(p[0] - g[0, 0]) < 1e-6 or abs(p[0] - g[1, 0]) < 1e-6
        ony = abs(p[1] - g[0, 1]) < 1e-6 or abs(p[1] - g[1, 1]) < 1e-6
        exp = whole / 4 if (onx and ony) else whole / 2 if (onx or ony) else whole
        assert abs(areas[i] - exp) <= 1e-6 * max(1, exp) + 2e-6
    assert o.info()["numRays"] == 25


def test_point_neighborhood_3d():
    """tests/pointNeighborhood/pointNeighborhood.cpp:51 — 8 / 5 / 3"""
    delta = np.float32(0.5)
    o, pts, _ = make_plane(10, 0.5, radius=np.float32(delta - np.float32(1e-6)))
    cnt = o.neighbor_counts()
    g = o.geometry_bbox()
    for i, p in enumerate(pts):
        ex = p[0] in (g[0, 0], g[1, 0])
        ey = p[1] in (g[0, 1], g[1, 1])
        assert cnt[i] == (3 if ex and ey else 5 if ex or ey else 8)
    # symmetry
    for i in (0, 40, 840, 1680):
        for j in o.neighbors(i):
            assert i in o.neighbors(int(j))


def test_point_neighborhood_2d():
    """tests/pointNeighborhood2D/pointNeighborhood2D.cpp:43,46 — 2 / 1"""
    pts, nrm = _line_2d(0)
    o = po.Oracle()
    o.set_disks(pts, nrm, 0.5, 2, radius=np.float32(0.5 - 1e-6))
    cnt = o.neighbor_counts()
    assert cnt[0] == 1 and cnt[-1] == 1 and (cnt[1:-1] == 2).all()


def _rng_seed_run(threads):
    o, pts, _ = make_plane(5, 0.5)
    o.set_particle(po.DIFFUSE, 1.0)
    o.set_num_rays_per_point(10)
    o.set_rng_seed(12345)
    o.apply(threads)
    return o.flux(), o.info()


def test_rng_seed_determinism():
    """tests/rngSeed/rngSeed.cpp:48-51 — same seed => bit-identical flux (4 threads)"""
    f1, i1 = _rng_seed_run(4)
    f2, i2 = _rng_seed_run(4)
    f3, _ = _rng_seed_run(1)
    assert (f1 == f2).all() and (f1 == f3).all()
    assert i1["numRays"] == 441 * 10
    assert f1.sum() > 0


def test_trace_interface():
    """tests/traceInterface/traceInterface.cpp:60,67"""
    o, pts, _ = make_plane(5, 0.5)
    o.set_particle(po.DIFFUSE, 1.0)
    o.set_boundary_conditions([po.REFLECTIVE] * 3)
    o.set_source_direction(po.POS_Z)
    o.set_num_rays_per_point(10)
    o.L.orc_set_use_random_seeds(o.h, 0)
    o.set_material_ids(np.zeros(len(pts), dtype=np.int32))
    o.apply(4)
    flux = o.flux()
    assert flux.size == len(pts)
    flux = o.normalize_flux(flux)
    flux = o.smooth_flux(flux, 2)
    assert flux.size == len(pts) and np.isfinite(flux).all()
    assert o.info()["numRays"] == 4410


def test_smoothing_orthogonal_normals():
    """tests/smoothing/smoothing.cpp:43,50"""
    pts = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [0, 1, 0], [1, 1, 0], [2, 1, 0]], dtype=np.float32)
    nrm = np.array([[0, 0, 1]] * 3 + [[0, 1, 0]] * 3, dtype=np.float32)
    o = po.Oracle()
    o.set_disks(pts, nrm, 1.0, 3)
    out = o.smooth_flux(np.array([1, 1, 1, 0, 0, 0], dtype=np.float32), 1)
    assert np.allclose(out[:3], 1.0, atol=1e-6) and np.allclose(out[3:], 0.0, atol=1e-6)


def test_trace_2d_runs():
    """tests/trace2D/trace2D.cpp — 2-D trench, reflective BCs, runs to completion"""
    gd, pts, nrm = trench2d()
    o = po.Oracle()
    o.set_disks(pts, nrm, gd, 2)
    o.set_boundary_conditions([po.REFLECTIVE, po.REFLECTIVE])
    o.set_source_direction(po.POS_Y)
    o.set_particle(po.DIFFUSE, 0.1)
    o.set_num_rays_per_point(50)
    o.set_rng_seed(1)
    o.apply(2)
    i = o.info()
    assert i["numRays"] == 239 * 50 and i["geometryHits"] > 0 and i["reflections"] > 0


def test_analytic_plane_flux_is_one():
    """G6: flat plane, cosine source, sticking 1, periodic walls =>
    SOURCE-normalised interior flux = 1 within Monte-Carlo error, and (almost)
    every ray deposits."""
    pts, nrm = io.plane_grid(40, 1.0)
    o = po.Oracle()
    o.set_disks(pts, nrm, 1.0, 3)
    o.set_boundary_conditions([po.PERIODIC] * 3)
    o.set_particle(po.DIFFUSE, 1.0)
    o.set_num_rays_fixed(400000)
    o.set_rng_seed(12345)
    o.set_lazy_rng(True)
    o.apply(po.max_threads())
    info = o.info()
    # a wrapped ray restarting within tnear=1e-4 of the plane tunnels through it
    #  -> a handful of misses are legitimate reference behaviour
    assert info["nonGeometryHits"] < 40
    assert info["geometryHits"] + info["nonGeometryHits"] + info["raysTerminated"] == 400000
    f = o.normalize_flux(o.flux())
    # ~250 rays per unit area, each disk sees ~2.36x that
    assert abs(f.mean() - 1.0) < 0.01
    assert f.std() < 0.1

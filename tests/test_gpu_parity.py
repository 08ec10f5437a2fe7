"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
the same seeded inputs.  Integer work (RNG words, ray origins AND directions, hit
ids, every TraceInfo counter, sticking-1 flux) is bit-exact; fractional-weight flux
differs only by float summation order and is compared by L2-relative error against
the tolerance BASELINE.json's north_star states (1e-4; measured <= 3e-7)."""
import os

import numpy as np
import pytest

import viennaray_amd as vr
from viennaray_amd import BoundaryCondition as BC, TraceDirection as TD
from oracle import pyoracle as po
from helpers import l2_rel, sphere3d, trench2d, trench3d, trench_mesh, DISK_FACTOR_3D

pytestmark = pytest.mark.gpu

FLUX_TOL = 1e-4  # north_star: "flux within 1e-4 relative of reference"
INFO_KEYS = ("numRays", "totalRaysTraced", "nonGeometryHits", "geometryHits", "particleHits",
             "boundaryHits", "reflections", "raysTerminated")


def info_dict(t):
    i = t.getRayTraceInfo()
    return {k: int(getattr(i, k)) for k in INFO_KEYS}


def make_pair_disks(pts, nrm, gd, D, bcs, direction, particle, rays_fixed=None, rays_pp=None, seed=12345,
                    radius=0.0, primary=None):
    t = vr.TraceDisk(D)
    t.setGeometry(pts, nrm, gd, radius)
    t.setBoundaryConditions(bcs)
    t.setSourceDirection(direction)
    o = po.Oracle()
    o.set_disks(pts, nrm, gd, D, radius=radius)
    o.set_boundary_conditions([int(b) for b in bcs])
    o.set_source_direction(int(direction))
    kind, sticking, power = particle
    if kind == "diffuse":
        t.setParticleType(vr.DiffuseParticle(sticking, "flux"))
        o.set_particle(po.DIFFUSE, sticking)
    else:
        t.setParticleType(vr.SpecularParticle(sticking, power, "flux"))
        o.set_particle(po.SPECULAR, sticking, power)
    if rays_fixed:
        t.setNumberOfRaysFixed(rays_fixed)
        o.set_num_rays_fixed(rays_fixed)
    else:
        t.setNumberOfRaysPerPoint(rays_pp)
        o.set_num_rays_per_point(rays_pp)
    t.setRngSeed(seed)
    o.set_rng_seed(seed)
    if primary is not None:
        t.setPrimaryDirection(primary)
        o.set_primary_direction(primary)
    o.set_lazy_rng(True)
    return t, o


def compare(t, o, exact_flux=False, counter_slack=0):
    t.apply()
    o.apply(po.max_threads())
    f = t.getLocalData().getVectorData(0)
    r = o.flux()
    gi, oi = info_dict(t), o.info()
    err = l2_rel(f, r)
    for k in INFO_KEYS:
        assert abs(gi[k] - oi[k]) <= counter_slack, (k, gi[k], oi[k], err)
    if exact_flux and counter_slack == 0:
        assert (f == r).all(), err
    assert err <= FLUX_TOL, err
    # with identical rays (exact RNG + exact libm) only the float summation order differs
    assert err <= 5e-6, err
    # SOURCE-normalised flux (BASELINE metric iii)
    fn, rn = t.normalizeFlux(f), o.normalize_flux(r)
    assert l2_rel(fn, rn) <= FLUX_TOL
    return err, gi


# ---------------------------------------------------------------------------
def test_rng_stream_bit_exact():
    t = vr.TraceDisk(3)
    for idx, seed in ((0, 12346), (1, 12346), (123456789, 7), (2**32 - 1, 0xFFFFFFFF)):
        got = t.debugRngOutputs(idx, seed, 700)  # tier 1, tier 2 and a second block twist
        exp = po.mt64_outputs(po.tea3(idx, seed), 700)
        assert (got == exp).all()


def test_source_sample_matches_oracle():
    gd, p, n = sphere3d()
    for direction in (TD.POS_Z, TD.NEG_X, TD.POS_Y):
        for power, primary in ((1.0, None), (50.0, None), (5.0, [0.2, 0.1, -1.0])):
            t = vr.TraceDisk(3)
            t.setGeometry(p, n, gd)
            t.setSourceDirection(direction)
            t.setParticleType(vr.SpecularParticle(1.0, power, "f"))
            o = po.Oracle()
            o.set_disks(p, n, gd, 3)
            o.set_source_direction(int(direction))
            o.set_particle(po.SPECULAR, 1.0, power)
            if primary is not None:
                # tilt relative to the tracing axis so the rejection loop is exercised
                prim = np.roll(np.array(primary, dtype=np.float32), {TD.POS_Z: 0, TD.NEG_X: 1, TD.POS_Y: 2}[direction])
                if direction == TD.NEG_X:
                    prim[0] = abs(prim[0])
                t.setPrimaryDirection(prim)
                o.set_primary_direction(prim)
            o.prepare()
            idx = np.arange(4096, dtype=np.uint64)
            org, d = t.debugSourceSample(idx, 31)
            eo = np.empty_like(org)
            ed = np.empty_like(d)
            for i in range(idx.size):
                eo[i], ed[i] = o.source_sample(int(idx[i]), 31)
            assert (org == eo).all()  # origins: pure IEEE arithmetic -> bit exact
            # directions: glibc's sincosf/powf are reproduced bit for bit on the device
            # (vr_libm.hpp, tests/test_libm_exact.py)
            assert (d.view(np.uint32) == ed.view(np.uint32)).all()


def test_intersection_known_answers_gpu():
    """tests/intersectionTest/intersectionTest.cpp:91-92,126-127 on the HIP path"""
    r = np.float32(0.5 * DISK_FACTOR_3D)
    pts, nrm = vr.io.create_plane_grid(0.5, 10)
    t = vr.TraceDisk(3)
    t.setGeometry(pts, nrm, 0.5, r)
    t.setParticleType(vr.DiffuseParticle(1.0, "f"))
    d = np.array([0, 2, -1.0])
    d /= np.linalg.norm(d)
    g, p, tt = t.debugIntersect([[0, 0, 2 * r], [0, 9, 2 * r]], [[0, 0, -1], d], tnear=0.0)
    assert g[0] == 1 and p[0] == 840
    assert g[1] == 0 and p[1] == 7


@pytest.mark.parametrize("geom", ["sphere", "trench3d", "trench2d", "mesh"])
def test_closest_hit_matches_oracle(geom):
    rng = np.random.default_rng(5)
    if geom == "mesh":
        gd, v, tri = trench_mesh()
        t = vr.TraceTriangle(3)
        t.setGeometry(v, tri, gd)
        o = po.Oracle()
        o.set_triangles(v, tri, gd, 3)
        lo, hi = v.min(0), v.max(0)
        D = 3
    else:
        gd, p, n = {"sphere": sphere3d, "trench3d": trench3d, "trench2d": trench2d}[geom]()
        D = 2 if geom == "trench2d" else 3
        t = vr.TraceDisk(D)
        t.setGeometry(p, n, gd)
        o = po.Oracle()
        o.set_disks(p, n, gd, D)
        lo, hi = p.min(0), p.max(0)
        if D == 2:
            t.setSourceDirection(TD.POS_Y)
            o.set_source_direction(po.POS_Y)
    t.setParticleType(vr.DiffuseParticle(1.0, "f"))
    o.prepare()
    nr = 3000
    org = rng.uniform(lo - 0.5, hi + 0.5, size=(nr, 3)).astype(np.float32)
    d = rng.normal(size=(nr, 3))
    if D == 2:
        org[:, 2] = 0
        d[:, 2] = 0
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    g, prim, tt = t.debugIntersect(org, d)
    for i in range(nr):
        h = o.intersect1(org[i], d[i])
        assert g[i] == h["geomID"], i
        if h["geomID"] >= 0:
            assert prim[i] == h["primID"] and tt[i] == np.float32(h["t"]), (i, prim[i], h)


@pytest.mark.parametrize("geom", ["sphere", "trench3d", "trench2d", "mesh"])
def test_ordered_walk_equals_escape_link_walk(geom, monkeypatch):
    """The trace kernels' ordered per-lane walk (pair nodes, near child first, LDS stack) and the escape-link
    walk it replaced find the same closest hit, bit for bit, on primary-like rays and on rays that start ON the
    surface (bounces): the closest-hit rule does not depend on the traversal order."""
    rng = np.random.default_rng(11)
    if geom == "mesh":
        gd, v, tri = trench_mesh()
        t = vr.TraceTriangle(3)
        t.setGeometry(v, tri, gd)
        lo, hi, D = v.min(0), v.max(0), 3
    else:
        gd, p, n = {"sphere": sphere3d, "trench3d": trench3d, "trench2d": trench2d}[geom]()
        D = 2 if geom == "trench2d" else 3
        t = vr.TraceDisk(D)
        t.setGeometry(p, n, gd)
        lo, hi = p.min(0), p.max(0)
        if D == 2:
            t.setSourceDirection(TD.POS_Y)
    t.setParticleType(vr.DiffuseParticle(0.1, "f"))
    up = 1 if D == 2 else 2
    nr = 200_000
    o = rng.uniform(lo, hi, size=(nr, 3)).astype(np.float32)
    o[:, up] = hi[up] + gd
    d = rng.normal(size=(nr, 3)) * 0.3
    d[:, up] = -1.0
    if D == 2:
        o[:, 2] = 0
        d[:, 2] = 0
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)

    def both(o, d):
        monkeypatch.setenv("VR_DEBUG_WALK", "0")
        g0, p0, t0 = t.debugIntersect(o, d)
        monkeypatch.setenv("VR_DEBUG_WALK", "1")
        g1, p1, t1 = t.debugIntersect(o, d)
        assert np.array_equal(g0, g1)
        m = g0 >= 0
        assert np.array_equal(p0[m], p1[m]) and np.array_equal(t0[m].view(np.uint32), t1[m].view(np.uint32))
        return g0, t0

    g, tt = both(o, d)
    m = g == 1
    assert m.sum() > nr // 4
    o2 = (o[m] + d[m] * tt[m, None]).astype(np.float32)
    d2 = rng.normal(size=o2.shape)
    if D == 2:
        d2[:, 2] = 0
    d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
    both(o2, d2)
    d3 = np.zeros_like(o2)
    d3[:, 0] = rng.choice([-1.0, 1.0], size=len(o2))  # axis-parallel, zero components (inverse = +-1e30)
    both(o2, d3.astype(np.float32))


def test_ordered_walk_deep_tree_uses_the_global_stack(monkeypatch):
    """A caterpillar tree (disk centres in geometric progression along the diagonal: every Morton prefix
    splits off one disk) is ~60 levels deep; rays along the diagonal meet every box, so the deferred children
    outgrow the 12 LDS-resident stack entries and the walk continues in its global slab.  Same hits as the
    escape-link walk, and apply() runs (an overflow would fail it)."""
    k = np.arange(0, 22)
    c = (2.0 ** -k)[:, None] * np.ones((1, 3))
    # a sheet of disks in front of the diagonal so that random rays also find work
    g = np.stack(np.meshgrid(np.linspace(0, 1, 24), np.linspace(0, 1, 24), [-0.05]), -1).reshape(-1, 3)
    pts = np.concatenate([c, g]).astype(np.float32)
    nrm = np.tile(np.array([[0, 0, 1.0]], dtype=np.float32), (len(pts), 1))
    nrm[: len(k)] = np.array([1, 1, 1.0], dtype=np.float32) / np.sqrt(3.0)
    t = vr.TraceDisk(3)
    t.setGeometry(pts, nrm, 0.04)
    t.setParticleType(vr.DiffuseParticle(0.5, "f"))
    rng = np.random.default_rng(3)
    n = 20000
    o = np.tile(np.array([[1.5, 1.5, 1.5]], dtype=np.float32), (n, 1)) + rng.normal(size=(n, 3)).astype(np.float32) * 0.01
    d = -np.ones((n, 3)) + rng.normal(size=(n, 3)) * 0.02
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o2 = rng.uniform(-0.2, 1.2, size=(n, 3)).astype(np.float32)
    d2 = rng.normal(size=(n, 3))
    d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
    for oo, dd in ((o, d), (o2, d2), (-o + 0.5, -d)):
        monkeypatch.setenv("VR_DEBUG_WALK", "0")
        g0, p0, t0 = t.debugIntersect(oo, dd)
        monkeypatch.setenv("VR_DEBUG_WALK", "1")
        g1, p1, t1 = t.debugIntersect(oo, dd)
        assert np.array_equal(g0, g1)
        m = g0 >= 0
        assert np.array_equal(p0[m], p1[m]) and np.array_equal(t0[m].view(np.uint32), t1[m].view(np.uint32))
    assert (g0 == 1).sum() > 0
    t.setNumberOfRaysPerPoint(200)
    t.setRngSeed(5)
    t.apply()
    assert t.getRayTraceInfo().totalRaysTraced > 0


# ---------------------------------------------------------------------------
def test_rng_seed_config_bit_exact():
    """tests/rngSeed/rngSeed.cpp geometry: 21x21 plane, sticking 1, 10 rays/point.
    Integer weights -> the flux must equal the oracle's exactly, twice."""
    pts, nrm = vr.io.create_plane_grid(0.5, 5)
    t, o = make_pair_disks(pts, nrm, 0.5, 3, [BC.REFLECTIVE_BOUNDARY] * 3, TD.POS_Z, ("diffuse", 1.0, 1), rays_pp=10)
    compare(t, o, exact_flux=True, counter_slack=0)
    f1 = t.getLocalData().getVectorData(0).copy()
    t2, _ = make_pair_disks(pts, nrm, 0.5, 3, [BC.REFLECTIVE_BOUNDARY] * 3, TD.POS_Z, ("diffuse", 1.0, 1), rays_pp=10)
    t2.apply()
    assert (t2.getLocalData().getVectorData(0) == f1).all()
    assert t.getRayTraceInfo().numRays == 4410


@pytest.mark.parametrize("bc", [BC.PERIODIC_BOUNDARY, BC.REFLECTIVE_BOUNDARY, BC.IGNORE_BOUNDARY])
@pytest.mark.parametrize("sticking", [1.0, 0.1])
def test_c1_plane_100x100(bc, sticking):
    """BASELINE config C1 (P(100), 1e6 rays) on the HIP path vs the oracle."""
    pts, nrm = vr.io.plane_grid(100, 1.0)
    t, o = make_pair_disks(pts, nrm, 1.0, 3, [bc] * 3, TD.POS_Z, ("diffuse", sticking, 1), rays_fixed=1000000)
    err, gi = compare(t, o, counter_slack=0)
    print("C1", bc, sticking, "L2", err, gi)


def test_trench3d_bounces():
    """examples/disk3D geometry (28 919 disks), sticking 0.1: long bounce chains,
    tier-2 RNG, back-face pass-through, roulette."""
    gd, p, n = trench3d()
    t, o = make_pair_disks(p, n, gd, 3, [BC.PERIODIC_BOUNDARY] * 3, TD.POS_Z, ("diffuse", 0.1, 1), rays_pp=20)
    err, gi = compare(t, o, counter_slack=0)
    print("trench3d L2", err, gi)


@pytest.mark.parametrize("bc", [BC.PERIODIC_BOUNDARY, BC.REFLECTIVE_BOUNDARY])
def test_trench2d(bc):
    """BASELINE config C5 geometry (239 disks, D=2, POS_Y)."""
    gd, p, n = trench2d()
    t, o = make_pair_disks(p, n, gd, 2, [bc, bc], TD.POS_Y, ("diffuse", 0.1, 1), rays_pp=2000)
    err, gi = compare(t, o, counter_slack=0)
    print("trench2d", bc, "L2", err, gi)


def test_trench2d_specular():
    gd, p, n = trench2d()
    t, o = make_pair_disks(p, n, gd, 2, [BC.REFLECTIVE_BOUNDARY] * 2, TD.POS_Y, ("specular", 0.2, 20.0), rays_pp=1000)
    err, gi = compare(t, o, counter_slack=0)
    print("trench2d specular L2", err, gi)


def test_sphere_all_directions():
    gd, p, n = sphere3d()
    for direction in TD:
        t, o = make_pair_disks(p, n, gd, 3, [BC.REFLECTIVE_BOUNDARY] * 3, direction, ("diffuse", 0.3, 1), rays_pp=1000)
        err, gi = compare(t, o, counter_slack=0)


def test_tilted_primary_direction():
    pts, nrm = vr.io.plane_grid(40, 1.0)
    t, o = make_pair_disks(pts, nrm, 1.0, 3, [BC.PERIODIC_BOUNDARY] * 3, TD.POS_Z, ("specular", 0.5, 30.0),
                           rays_fixed=200000, primary=[0.3, 0.1, -1.0])
    compare(t, o, counter_slack=0)


def test_triangle_mesh_specular():
    """BASELINE config C4 geometry: trenchMesh.dat, specular, power 50, reflective."""
    gd, v, tri = trench_mesh()
    t = vr.TraceTriangle(3)
    t.setGeometry(v, tri, gd)
    t.setParticleType(vr.SpecularParticle(0.1, 50.0, "flux"))
    t.setNumberOfRaysPerPoint(40)
    t.setRngSeed(12345)
    o = po.Oracle()
    o.set_triangles(v, tri, gd, 3)
    o.set_particle(po.SPECULAR, 0.1, 50.0)
    o.set_num_rays_per_point(40)
    o.set_rng_seed(12345)
    o.set_lazy_rng(True)
    err, gi = compare(t, o, counter_slack=0)
    print("mesh specular L2", err, gi)


def test_triangle_mesh_diffuse():
    """examples/triangle3D: diffuse sticking 0.1."""
    gd, v, tri = trench_mesh()
    t = vr.TraceTriangle(3)
    t.setGeometry(v, tri, gd)
    t.setParticleType(vr.DiffuseParticle(0.1, "flux"))
    t.setNumberOfRaysPerPoint(20)
    t.setRngSeed(7)
    o = po.Oracle()
    o.set_triangles(v, tri, gd, 3)
    o.set_particle(po.DIFFUSE, 0.1)
    o.set_num_rays_per_point(20)
    o.set_rng_seed(7)
    o.set_lazy_rng(True)
    err, gi = compare(t, o, counter_slack=0)
    print("mesh diffuse L2", err, gi)


def test_ray_range_shards_sum_to_whole():
    """Multi-GPU contract: tracing disjoint global ray-index ranges and adding the
    integer accumulators reproduces the single-launch result exactly."""
    pts, nrm = vr.io.plane_grid(64, 1.0)

    def run(first, count):
        t = vr.TraceDisk(3)
        t.setGeometry(pts, nrm, 1.0)
        t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
        t.setParticleType(vr.DiffuseParticle(0.3, "flux"))
        t.setNumberOfRaysFixed(300000)
        t.setRngSeed(99)
        t.setRayRange(first, count)
        t.apply()
        return t.getFluxF64(), info_dict(t)

    whole, wi = run(0, 0)
    a, ai = run(0, 100000)
    b, bi = run(100000, 120000)
    c, ci = run(220000, 80000)
    assert (a + b + c == whole).all()
    for k in INFO_KEYS[1:]:
        assert ai[k] + bi[k] + ci[k] == wi[k]


def test_material_sticking_map():
    """per-material sticking (gpu::Particle::materialSticking, rayParticle.hpp:208-218): the oracle
    models the map as a particle whose surfaceReflection looks the material up"""
    pts, nrm = vr.io.plane_grid(32, 1.0)
    mats = (np.arange(32 * 32) % 2).astype(np.int32)
    t, o = make_pair_disks(pts, nrm, 1.0, 3, [BC.PERIODIC_BOUNDARY] * 3, TD.POS_Z, ("diffuse", 1.0, 1.0),
                           rays_fixed=100000, seed=3)
    t.setMaterialIds(mats)
    o.set_material_ids(mats)
    t.setParticleType(vr.DiffuseParticle(1.0, "flux", materialSticking={1: 0.5}))
    o.set_material_sticking({1: 0.5})
    err, i = compare(t, o)
    # closest disk of material 1 -> the ray survives with weight 0.5 and reflects
    assert 0.3 * 100000 < i["reflections"] < 0.7 * 100000


def test_error_paths():
    t = vr.TraceDisk(3)
    with pytest.raises(vr.VrError):
        t.apply()  # no particle
    t.setParticleType(vr.DiffuseParticle(1.0, "f"))
    with pytest.raises(vr.VrError):
        t.apply()  # no geometry
    pts, nrm = vr.io.plane_grid(8, 1.0)
    t2 = vr.TraceDisk(2)
    t2.setGeometry(pts, nrm, 1.0)
    t2.setParticleType(vr.DiffuseParticle(1.0, "f"))
    t2.setSourceDirection(TD.POS_Z)
    with pytest.raises(vr.VrError):
        t2.apply()  # 2-D with a z source


def test_gpu_shard_driver_matches_plain_apply():
    """viennaray_amd.distributed.GpuShard (torch-owned int64 accumulators) on one
    rank: the sum over 3 emulated shards equals a plain apply()."""
    import torch
    from viennaray_amd import distributed as vd
    pts, nrm = vr.io.plane_grid(48, 1.0)

    def mk():
        t = vr.TraceDisk(3)
        t.setGeometry(pts, nrm, 1.0)
        t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
        t.setParticleType(vr.DiffuseParticle(0.2, "flux"))
        t.setNumberOfRaysFixed(150000)
        t.setRngSeed(5)
        return t

    t = mk()
    t.apply()
    ref = t.getFluxF64()
    t2 = mk()
    shard = vd.GpuShard(t2, "cuda:0")
    total = torch.zeros(len(pts), dtype=torch.int64, device="cuda:0")
    cnt = None
    for r in range(3):
        first, count = vd.ray_shard(150000, r, 3)
        acc, c = shard.trace_local(first, count, run_number=1)
        torch.cuda.synchronize()
        total += acc
        c = np.asarray(c.tolist() if hasattr(c, "tolist") else c, dtype=np.int64)
        cnt = c if cnt is None else cnt + c
    assert (vd.accumulators_to_flux(total) == ref).all()
    gi = info_dict(t)
    for k, v in zip(vd.COUNTER_KEYS, cnt.tolist()):
        assert gi[k] == v


@pytest.mark.parametrize("name,data,rpp,rays", [("disk3D", "trenchGrid3D.dat", 50, 1445950),
                                                ("disk2D", "trenchGrid2D.dat", 500, 119500),
                                                ("triangle3D", "trenchMesh.dat", 20, 256000),
                                                ("triangle2D", "lineMesh.dat", 400, 102400)])
def test_ported_reference_example_runs(name, data, rpp, rays):
    """The reference's four example programs, ported to the C++ façade (same API, other include
    path), run on the fixture copies of the reference's data files."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", name, name)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "examples")])
    cwd = os.path.join(root, "gpurun_out") if os.path.isdir(os.path.join(root, "gpurun_out")) else "/tmp"
    out = subprocess.run([exe, os.path.join(root, "tests", "golden", "data", data), str(rpp)],
                         capture_output=True, text=True, cwd=cwd, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert f"rays {rays}" in out.stdout, out.stdout
    mean = float(out.stdout.split("mean normalised flux")[1].split()[0])
    assert 0.05 < mean < 1.5


@pytest.mark.parametrize("bc", [BC.REFLECTIVE_BOUNDARY, BC.PERIODIC_BOUNDARY])
def test_line_mesh_2d(bc):
    """examples/triangle2D: LineMesh -> triangle strips, TraceTriangle<D=2> (rayTraceTriangle.hpp:76-81)"""
    import os
    from helpers import DATA
    gd, nodes, lines = vr.io.read_line_mesh(os.path.join(DATA, "lineMesh.dat"))
    v, tri, keep = vr.io.lines_to_triangles(nodes, lines, gd)
    assert lines.shape[0] == 130 and keep.size == 128  # 2 zero-length lines dropped
    t = vr.TraceTriangle(2)
    t.setLineGeometry(nodes, lines, gd)
    t.setBoundaryConditions([bc, bc])
    t.setParticleType(vr.DiffuseParticle(0.1, "flux"))
    t.setNumberOfRaysPerPoint(300)
    t.setRngSeed(12345)
    o = po.Oracle()
    o.set_triangles(v, tri, gd, 2)
    o.set_boundary_conditions([int(bc), int(bc)])
    o.set_particle(po.DIFFUSE, 0.1)
    o.set_num_rays_per_point(300)
    o.set_rng_seed(12345)
    o.set_lazy_rng(True)
    err, gi = compare(t, o, counter_slack=0)
    assert gi["numRays"] == 256 * 300 and gi["reflections"] > 0


@pytest.mark.parametrize("geom", ["plane", "sphere", "trench3d", "trench2d"])
def test_device_neighbourhood_matches_oracle(geom):
    """The device-built neighbourhood CSR (BVH range query) holds exactly the
    reference's pairs (rayPointNeighborhood.hpp:287-298); plane: 8 / 5 / 3
    (tests/pointNeighborhood/pointNeighborhood.cpp:51)."""
    D = 3
    radius = 0.0
    if geom == "plane":
        gd = 0.5
        p, n = vr.io.create_plane_grid(0.5, 10)
        radius = float(np.float32(0.5) - np.float32(1e-6))
    else:
        gd, p, n = {"sphere": sphere3d, "trench3d": trench3d, "trench2d": trench2d}[geom]()
        D = 2 if geom == "trench2d" else 3
    t = vr.TraceDisk(D)
    t.setGeometry(p, n, gd, radius)
    t.setParticleType(vr.DiffuseParticle(1.0, "f"))
    t.setNumberOfRaysFixed(256)
    t.setRngSeed(1)
    if D == 2:
        t.setSourceDirection(TD.POS_Y)
    t.apply()  # builds BVH + neighbourhood on the device
    got = t.getNeighborCounts()
    o = po.Oracle()
    o.set_disks(p, n, gd, D, radius=radius)
    exp = o.neighbor_counts()
    assert (got == exp).all()
    if geom == "plane":
        assert sorted(set(got.tolist())) == [3, 5, 8]


def test_host_and_device_builders_agree(monkeypatch):
    """VR_HOST_BUILD=1 selects the host LBVH/CSR builder; the flux must not depend
    on which tree is traversed (closest-hit rule is order independent)."""
    gd, p, n = trench3d()

    def run():
        t = vr.TraceDisk(3)
        t.setGeometry(p, n, gd)
        t.setBoundaryConditions([BC.REFLECTIVE_BOUNDARY] * 3)
        t.setParticleType(vr.DiffuseParticle(0.3, "flux"))
        t.setNumberOfRaysPerPoint(10)
        t.setRngSeed(11)
        t.apply()
        return t.getFluxF64(), info_dict(t), t.debugBvhStats()

    fd, idv, sd = run()
    monkeypatch.setenv("VR_HOST_BUILD", "1")
    fh, ih, sh = run()
    assert (fd == fh).all() and idv == ih
    assert sh["maxDepth"] > 0 and 0 < sd["nodes"] <= 2 * len(p) - 1


SCHEDULING_KNOBS = [
    {"VR_BATCH_RAYS": "40000"},                       # many batches
    {"VR_BIN_CAP": "8", "VR_RAYS_PER_BIN": "16"},     # most rays overflow their bin
    {"VR_RAYS_PER_BIN": "2"},                         # nearly empty bins
    {"VR_RAYS_PER_BIN": "1", "VR_SPAN_BINS": "64"},   # ... whole spans of them: a wave with nothing to do must walk on
    {"VR_SPAN_BINS": "1"}, {"VR_SPAN_BINS": "64"},    # work-queue granularity
    {"VR_SMALL_SCENE": "0"},                          # small scenes from HBM instead of resident in LDS (MODE 4 off)
    {"VR_WALK_EXIT": "1"}, {"VR_WALK_EXIT": "64"},    # no / eager straggler carry-over
    {"VR_WALK_PARK": "1"}, {"VR_WALK_PARK": "100"},   # leaf batching extremes
    {"VR_PACKET_BUDGET": "0"}, {"VR_PACKET_BUDGET": "100000", "VR_PACKET_RATIO": "1000"},
    {"VR_DEBUG_FLAGS": "32"},                         # no packets at all
    {"VR_LEAF_MAX": "1"}, {"VR_LEAF_MAX": "12"},
    {"VR_ACC_REPLICAS": "1"}, {"VR_ACC_REPLICAS": "64"},
    {"VR_NO_CHILD_ORDER": "1"},
    {"VR_TRACE_BLOCKS": "1"},
    {"VR_KEY_COORD": "-7.5"},
    {"VR_ABSORB_CARRY": "0"}, {"VR_ABSORB_CARRY": "1"},
    {"VR_DEBUG_FLAGS": "128"},                        # no packet (box) queries
    {"VR_PQ_CAND": "63", "VR_PQ_FRONTIER": "64"},     # box queries run to the end even where they do not pay
    {"VR_PQ_CAND": "2"},                              # ... or give up half way (found hits stay valid)
    {"VR_PQ_FRONTIER": "1"},
    {"VR_GENERAL_FLAT": "1"}, {"VR_GENERAL_FLAT": "0"},  # packet-query crediting on / off in the general kernel
    {"VR_DEBUG_FLAGS": "256"},                        # no follow-up segments inside a packet-query round
    {"VR_MORTON_ANISO": "1"}, {"VR_MORTON_ANISO": "1000000"},  # Morton grid of cubes / in the scene box's proportions
    {"VR_PQ_MARGIN": "0"}, {"VR_PQ_MARGIN": "0.2"}, {"VR_PQ_MARGIN": "6"},  # packet query: no / short-lived / far-reaching frontier cache
]


@pytest.mark.parametrize("geom,sticking", [("trench3d", 0.15), ("mesh", 0.15), ("trench2d", 0.15), ("trench3d", 1.0),
                                           ("plane", 0.15), ("plane", 1.0)])
def test_scheduling_knobs_do_not_change_results(geom, sticking, monkeypatch):
    """Ray order, batching, bin geometry, packet / per-lane traversal policy, BVH leaf size and
    child order, accumulator replication: none of it may change a single accumulator bit or
    counter (int64 fixed-point sums + the order-independent closest-hit rule)."""
    def run():
        if geom == "mesh":
            gd, v, tri = trench_mesh()
            t = vr.TraceTriangle(3)
            t.setGeometry(v, tri, gd)
            t.setBoundaryConditions([BC.PERIODIC_BOUNDARY, BC.REFLECTIVE_BOUNDARY, BC.REFLECTIVE_BOUNDARY])
            t.setNumberOfRaysPerPoint(12)
        elif geom == "trench2d":
            gd, p, n = trench2d()
            t = vr.TraceDisk(2)
            t.setGeometry(p, n, gd)
            t.setSourceDirection(TD.POS_Y)
            t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 2)
            t.setNumberOfRaysPerPoint(1500)
        elif geom == "plane":  # flat scene: the box query carries every segment
            p, n = vr.io.plane_grid(150, 0.5)
            t = vr.TraceDisk(3)
            t.setGeometry(p, n, 0.5)
            t.setBoundaryConditions([BC.PERIODIC_BOUNDARY, BC.REFLECTIVE_BOUNDARY, BC.REFLECTIVE_BOUNDARY])
            t.setNumberOfRaysPerPoint(12)
        else:
            gd, p, n = trench3d()
            t = vr.TraceDisk(3)
            t.setGeometry(p, n, gd)
            t.setBoundaryConditions([BC.REFLECTIVE_BOUNDARY, BC.PERIODIC_BOUNDARY, BC.PERIODIC_BOUNDARY])
            t.setNumberOfRaysPerPoint(12)
        t.setParticleType(vr.DiffuseParticle(sticking, "flux"))
        t.setRngSeed(99)
        t.apply()
        return t.getFluxF64(), info_dict(t)

    f0, i0 = run()
    assert (i0["reflections"] > 0 or sticking >= 1.0) and i0["boundaryHits"] > 0
    for knobs in SCHEDULING_KNOBS:
        with monkeypatch.context() as m:
            for k, v in knobs.items():
                m.setenv(k, v)
            f, i = run()
        assert i == i0, knobs
        assert (f == f0).all(), knobs


def _rippled_surface(n=120, gd=0.5, amp=0.5, wave=2.0):
    """A gently rippled sheet of disks (z = amp sin(x / wave) cos(y / wave), normals of the height field): flat enough
    for the box query to carry every primary ray, tilted enough for reflected rays to meet neighbouring disks."""
    ax = (np.arange(n) - (n - 1) / 2.0) * gd
    x, y = np.meshgrid(ax, ax, indexing="ij")
    z = amp * np.sin(x / wave) * np.cos(y / wave)
    nx = -amp / wave * np.cos(x / wave) * np.cos(y / wave)
    ny = amp / wave * np.sin(x / wave) * np.sin(y / wave)
    nrm = np.stack([nx, ny, np.ones_like(nx)], -1).reshape(-1, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    pts = np.stack([x, y, z], -1).reshape(-1, 3)
    return pts.astype(np.float32), nrm.astype(np.float32), gd


@pytest.mark.parametrize("particle", [("diffuse", 0.1, 1.0), ("specular", 0.2, 1.0), ("specular", 0.05, 40.0)])
@pytest.mark.parametrize("surface", ["plane", "ripple"])
@pytest.mark.parametrize("bcs", [[BC.REFLECTIVE_BOUNDARY] * 3, [BC.PERIODIC_BOUNDARY, BC.IGNORE_BOUNDARY, BC.REFLECTIVE_BOUNDARY]])
def test_follow_up_segments_inside_a_packet_query_round(surface, particle, bcs, monkeypatch):
    """The general flat-scene kernel (MODE 3) finishes a continuing ray's next segment in the round of the packet query
    that found its hit, where that segment stays inside the query's box and meets none of its candidates.  Same
    accumulator bits and counters as without (VR_DEBUG_FLAGS=256) and as the general kernel (MODE 0); oracle parity on a
    rippled sheet, where some follow-up segments DO meet a neighbouring disk and are left to the next round."""
    if surface == "plane":
        pts, nrm = vr.io.plane_grid(120, 0.5)
        gd = 0.5
    else:
        pts, nrm, gd = _rippled_surface()
    monkeypatch.setenv("VR_GENERAL_FLAT", "1")

    def pair():
        return make_pair_disks(pts, nrm, gd, 3, bcs, TD.POS_Z, particle, rays_pp=40, seed=2024)

    t, o = pair()
    err, info = compare(t, o)
    assert t.traceMode() == 3
    assert info["reflections"] > 0 and info["boundaryHits"] > 0
    if surface == "ripple" and particle[0] == "diffuse" and BC.IGNORE_BOUNDARY not in bcs:  # (reflected rays do meet the surface again)
        assert info["geometryHits"] > 1.01 * info["numRays"]
    f0 = t.getFluxF64()
    for knobs in ({"VR_DEBUG_FLAGS": "256"}, {"VR_GENERAL_FLAT": "0"}):
        with monkeypatch.context() as m:
            for k, v in knobs.items():
                m.setenv(k, v)
            t2, _ = pair()
            t2.apply()
            assert info_dict(t2) == info, knobs
            assert (t2.getFluxF64() == f0).all(), knobs


def _relief_surface(kind, n=140, gd=0.5):
    """Sheets of disks that are flat WITH RELIEF (a few grid cells thick at most): the rippled sheet, a plane with one
    bump a third of a cell high, terraces (steps of 1.5 cells, no risers: the rays see both levels' rims), a tilted sheet."""
    if kind == "ripple":
        return _rippled_surface(n=n, gd=gd, amp=0.5 * gd / 0.5, wave=2.0)
    ax = (np.arange(n) - (n - 1) / 2.0) * gd
    x, y = np.meshgrid(ax, ax, indexing="ij")
    if kind == "bump":
        rr2 = (x * x + y * y) / (6.0 * gd) ** 2
        z = 0.3 * gd * np.exp(-rr2)
        gx, gy = -2.0 * x / (6.0 * gd) ** 2 * z, -2.0 * y / (6.0 * gd) ** 2 * z
    elif kind == "steps":
        z = 1.5 * gd * (np.floor((x - ax[0]) / (9.0 * gd)) % 2)
        gx = gy = np.zeros_like(x)
    else:  # "tilt": a plane tilted by 2 cells over the sheet's width (its normal not along the source axis)
        slope = 2.0 * gd / (n * gd)
        z = slope * x
        gx, gy = np.full_like(x, slope), np.zeros_like(x)
    nrm = np.stack([-gx, -gy, np.ones_like(x)], -1).reshape(-1, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32), nrm.astype(np.float32), gd


RELIEF_KNOBS = [
    {"VR_DEBUG_FLAGS": "512"},     # the packet query clips to the scene box again (grazing rays still filed apart)
    {"VR_NO_RELIEF": "1"},         # the kernels for structured scenes, one set of bins
    {"VR_RELIEF_TRAVEL": "0.2"},   # nearly every ray is filed as loose
    {"VR_RELIEF_TRAVEL": "50"},    # ... none is
    {"VR_RELIEF_TILE": "0.5"}, {"VR_RELIEF_TILE": "3"},
    {"VR_RELIEF_LOOKUPS": "2"}, {"VR_RELIEF_LOOKUPS": "0"}, {"VR_RELIEF_COARSE_K": "1"},
    {"VR_RAYS_PER_BIN": "3"}, {"VR_BATCH_RAYS": "50000"},
    {"VR_BIN_CAP": "8", "VR_RAYS_PER_BIN": "16"},  # most rays overflow their (tight or loose) bin
    {"VR_NO_SPILL": "1"}, {"VR_DEBUG_FLAGS": "8192"},  # continuing rays stay in the tight general kernel (no spill queue)
    {"VR_LOOSE_BLOCKS": "1"},
    {"VR_PQ_MARGIN": "0"}, {"VR_PQ_MARGIN": "5"},  # packet query without / with a far-reaching frontier cache
    {"VR_RELIEF_STEPS": "1"}, {"VR_RELIEF_STEPS": "1000"},  # rays whose stretch through the scene box spans > n tiles are loose / never
]


@pytest.mark.parametrize("particle", [("diffuse", 1.0, 1.0), ("diffuse", 0.1, 1.0), ("specular", 0.2, 30.0)])
@pytest.mark.parametrize("surface", ["ripple", "bump", "steps", "tilt"])
def test_relief_packets_match_the_oracle(surface, particle, monkeypatch):
    """Scenes that are flat with relief run the flat-scene kernels with the packet query clipped to the LOCAL relief
    (MODE 5 absorbing / 6 general): rays sorted by their predicted first hit, grazing rays filed in loose bins that the
    structured-scene kernel traces.  Counters equal and flux as the oracle's on four kinds of relief; the same accumulator
    bits with the clip, the relief path, the loose bins or the field's resolution changed (every knob only re-orders work
    or loosens a conservative bound)."""
    pts, nrm, gd = _relief_surface(surface)
    bcs = [BC.PERIODIC_BOUNDARY, BC.REFLECTIVE_BOUNDARY, BC.REFLECTIVE_BOUNDARY]

    def pair():
        return make_pair_disks(pts, nrm, gd, 3, bcs, TD.POS_Z, particle, rays_pp=40, seed=4242)

    t, o = pair()
    err, info = compare(t, o, exact_flux=particle[1] >= 1.0)
    assert t.traceMode() == (5 if particle[1] >= 1.0 else 6), t.traceMode()
    assert info["boundaryHits"] > 0
    f0 = t.getFluxF64()
    for knobs in RELIEF_KNOBS:
        with monkeypatch.context() as m:
            for k, v in knobs.items():
                m.setenv(k, v)
            t2, _ = pair()
            t2.apply()
            assert info_dict(t2) == info, knobs
            assert (t2.getFluxF64() == f0).all(), knobs
            if "VR_NO_RELIEF" in knobs:
                assert t2.traceMode() in (0, 2)


@pytest.mark.parametrize("direction", [TD.NEG_Z, TD.POS_X])
def test_relief_packets_other_source_sides(direction):
    """The relief field lives over the SOURCE plane: source below the sheet (back faces first), and a sheet seen from +x."""
    pts, nrm, gd = _relief_surface("ripple", n=120)
    if direction == TD.POS_X:  # the same sheet stood up: its height along x
        pts = np.ascontiguousarray(pts[:, [2, 0, 1]])
        nrm = np.ascontiguousarray(nrm[:, [2, 0, 1]])
    for particle in (("diffuse", 1.0, 1.0), ("diffuse", 0.3, 1.0)):
        t, o = make_pair_disks(pts, nrm, gd, 3, [BC.REFLECTIVE_BOUNDARY] * 3, direction, particle, rays_pp=30, seed=7)
        compare(t, o)
        assert t.traceMode() in (5, 6)


def test_relief_packets_2d_and_triangles():
    """D = 2 (a rippled line of disks too long for the LDS-resident kernel: the relief field is one row of tiles) and an
    absorbing particle on a rippled triangle mesh (the absorbing relief kernel serves triangles too)."""
    n, gd = 4000, 0.5
    x = (np.arange(n) - (n - 1) / 2.0) * gd
    y = 0.2 * np.sin(x / 2.0)
    nrm = np.stack([-0.1 * np.cos(x / 2.0), np.ones_like(x), np.zeros_like(x)], -1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    pts = np.stack([x, y, np.zeros_like(x)], -1).astype(np.float32)
    for particle in (("diffuse", 1.0, 1.0), ("diffuse", 0.2, 1.0)):
        t, o = make_pair_disks(pts, nrm.astype(np.float32), gd, 2, [BC.PERIODIC_BOUNDARY] * 2, TD.POS_Y, particle, rays_pp=300, seed=11)
        compare(t, o)
        assert t.traceMode() in (5, 6), t.traceMode()
    # triangles
    m, gd = 90, 0.5
    ax = (np.arange(m) - (m - 1) / 2.0) * gd
    xx, yy = np.meshgrid(ax, ax, indexing="ij")
    zz = 0.5 * np.sin(xx / 2.0) * np.cos(yy / 2.0)
    verts = np.stack([xx, yy, zz], -1).reshape(-1, 3).astype(np.float32)
    idx = np.arange(m * m).reshape(m, m)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    tris = np.concatenate([np.stack([a, b, c], -1), np.stack([a, c, d], -1)]).astype(np.uint32)
    t = vr.TraceTriangle(3)
    t.setGeometry(verts, tris, gd)
    t.setBoundaryConditions([BC.REFLECTIVE_BOUNDARY] * 3)
    t.setParticleType(vr.DiffuseParticle(1.0, "flux"))
    t.setNumberOfRaysPerPoint(30)
    t.setRngSeed(3)
    o = po.Oracle()
    o.set_triangles(verts, tris, gd, 3)
    o.set_boundary_conditions([po.REFLECTIVE] * 3)
    o.set_particle(po.DIFFUSE, 1.0)
    o.set_num_rays_per_point(30)
    o.set_rng_seed(3)
    o.set_lazy_rng(True)
    compare(t, o, exact_flux=True)
    assert t.traceMode() == 5


def test_relief_packets_with_a_particle_list():
    """Several particles in ONE apply on a scene with relief: an absorbing one (MODE 5 + MODE 2 over the loose bins), two
    reflecting ones that share a generator pass (MODE 6 + MODE 7 each, one spill queue used in turn), a two-label registry
    particle and a coned-cosine one (the full extended kernel has no relief variant: the general kernel, bins of its own)
    — every counter and data label against one oracle run per particle."""
    pts, nrm, gd = _relief_surface("ripple", n=120)
    bcs = [BC.PERIODIC_BOUNDARY, BC.REFLECTIVE_BOUNDARY, BC.REFLECTIVE_BOUNDARY]
    plist = [(vr.DiffuseParticle(1.0, "a"), po.DIFFUSE, 1.0, 1.0, 0.0),
             (vr.DiffuseParticle(0.2, "b"), po.DIFFUSE, 0.2, 1.0, 0.0),
             (vr.SpecularParticle(0.3, 1.0, "c"), po.SPECULAR, 0.3, 1.0, 0.0),
             (vr.DiffuseCosineParticle(0.4, "d", "e"), po.DIFFUSE_COSINE, 0.4, 1.0, 0.0),
             (vr.ConedCosineParticle(0.5, 1.0, 0.6, "f"), po.CONED_COSINE, 0.5, 1.0, 0.6)]
    t = vr.TraceDisk(3)
    t.setGeometry(pts, nrm, gd)
    t.setBoundaryConditions(bcs)
    t.setNumberOfRaysPerPoint(25)
    t.setRngSeed(606)
    t.setParticleTypes([q[0] for q in plist])
    t.apply()
    ld = t.getLocalData()
    plane = 0
    for q, (_, okind, s, power, cone) in enumerate(plist):
        o = po.Oracle()
        o.set_disks(pts, nrm, gd, 3)
        o.set_boundary_conditions([int(b) for b in bcs])
        o.set_particle_ex(okind, s, power, cone, -1.0)
        o.set_num_rays_per_point(25)
        o.set_rng_seed(606)
        o.set_lazy_rng(True)
        o.apply(po.max_threads())
        pi, oi = t.getParticleTraceInfo(q), o.info()
        assert {k: int(getattr(pi, k)) for k in INFO_KEYS} == {k: oi[k] for k in INFO_KEYS}, (q, okind)
        for l in range(o.num_data()):
            assert l2_rel(ld.getVectorData(plane), o.flux_data(l)) <= 5e-6, (q, okind, l)
            plane += 1
    assert plane == t.numData()


def test_relief_packets_as_ray_range_shards():
    """A scene with relief traced as three shards of the ray-index range through the multi-GPU driver (bound torch
    accumulators, world size 3: the overflow check's head-room) — tight bins, loose bins and the spill queue of every
    shard — sums to the plain apply() bit for bit, counters included."""
    import torch
    from viennaray_amd import distributed as vd
    pts, nrm, gd = _relief_surface("ripple", n=130)
    for sticking in (1.0, 0.15):
        def mk():
            t = vr.TraceDisk(3)
            t.setGeometry(pts, nrm, gd)
            t.setBoundaryConditions([BC.PERIODIC_BOUNDARY, BC.REFLECTIVE_BOUNDARY, BC.REFLECTIVE_BOUNDARY])
            t.setParticleType(vr.DiffuseParticle(sticking, "flux"))
            t.setNumberOfRaysFixed(700001)
            t.setRngSeed(9)
            return t

        t = mk()
        t.apply()
        assert t.traceMode() in (5, 6)
        ref = t.getFluxF64()
        t2 = mk()
        shard = vd.GpuShard(t2, "cuda:0")
        total = torch.zeros(len(pts), dtype=torch.int64, device="cuda:0")
        cnt = None
        for r in range(3):
            first, count = vd.ray_shard(700001, r, 3)
            acc, c = shard.trace_local(first, count, run_number=1, world=3)
            torch.cuda.synchronize()
            total += acc
            c = np.asarray(c, dtype=np.int64)
            cnt = c if cnt is None else cnt + c
        assert (vd.accumulators_to_flux(total) == ref).all()
        gi = info_dict(t)
        for k, v in zip(vd.COUNTER_KEYS, cnt.tolist()):
            assert gi[k] == v, k


def test_relief_is_for_thin_scenes_only():
    """A trench is not "flat with relief": its scene box is sixty cells deep — the general kernels as before."""
    gd, p, n = trench3d()
    for sticking, mode in ((1.0, 2), (0.2, 0)):
        t = vr.TraceDisk(3)
        t.setGeometry(p, n, gd)
        t.setParticleType(vr.DiffuseParticle(sticking, "flux"))
        t.setNumberOfRaysPerPoint(2)
        t.setRngSeed(1)
        t.apply()
        assert t.traceMode() == mode


def test_relief_state_follows_geometry_and_particle_changes():
    """ONE tracer through a sequence of scenes and particles — rippled sheet (absorbing, then reflecting), the flat plane,
    a trench, a smaller rippled sheet, the first one again: the relief field, the loose bins, the spill queue and the
    launch frames are rebuilt or dropped with the scene; every apply equals a fresh oracle's."""
    gd_t, p_t, n_t = trench3d()
    r1 = _relief_surface("ripple", n=120)
    r2 = _relief_surface("steps", n=70, gd=1.0)
    flat_p, flat_n = vr.io.plane_grid(90, 0.5)
    seq = [(r1, ("diffuse", 1.0, 1.0), 5), (r1, ("diffuse", 0.15, 1.0), 6), ((flat_p, flat_n, 0.5), ("diffuse", 0.15, 1.0), 3),
           ((p_t, n_t, gd_t), ("diffuse", 0.3, 1.0), 0), (r2, ("specular", 0.2, 20.0), 6), (r1, ("diffuse", 1.0, 1.0), 5),
           ((flat_p, flat_n, 0.5), ("diffuse", 1.0, 1.0), 1)]
    t = vr.TraceDisk(3)
    bcs = [BC.REFLECTIVE_BOUNDARY, BC.PERIODIC_BOUNDARY, BC.REFLECTIVE_BOUNDARY]
    t.setBoundaryConditions(bcs)
    t.setSourceDirection(TD.POS_Z)
    for run, ((pts, nrm, gd), particle, mode) in enumerate(seq):
        t.setGeometry(pts, nrm, gd)
        kind, sticking, power = particle
        t.setParticleType(vr.DiffuseParticle(sticking, "flux") if kind == "diffuse" else vr.SpecularParticle(sticking, power, "flux"))
        t.setNumberOfRaysPerPoint(12)
        t.setRngSeed(99)
        t.setRunNumber(1)  # (a fresh tracer's: every apply() moves it on, rayTraceDisk.hpp:54)
        _, o = make_pair_disks(pts, nrm, gd, 3, bcs, TD.POS_Z, particle, rays_pp=12, seed=99)
        print("scene", run, particle, flush=True)
        compare(t, o, exact_flux=sticking >= 1.0)
        assert t.traceMode() == mode, (run, t.traceMode(), mode)


@pytest.mark.parametrize("direction", [TD.POS_Z, TD.NEG_Z])
@pytest.mark.parametrize("particle", [("diffuse", 0.2, 1.0), ("specular", 0.3, 1.0)])
def test_segments_that_rise_clear_respect_nearby_relief(particle, direction, monkeypatch):
    """The general kernels finish a continuing ray in the round of its reflection when the height field over the source
    plane says it rises clear of everything near by.  A plane with a dome in its middle: rays reflected off the plane
    next to the dome DO meet it, rays far from it do not — same counters and flux as the oracle, and the same bits with
    the shortcut switched off (VR_DEBUG_FLAGS=256).  Source above and (back faces first) below."""
    n, gd = 90, 0.5
    ax = (np.arange(n) - (n - 1) / 2.0) * gd
    x, y = np.meshgrid(ax, ax, indexing="ij")
    rr = np.sqrt(x * x + y * y)
    R, H = 8.0, 5.0                      # a paraboloid dome of radius R and height H
    z = np.where(rr < R, H * (1.0 - (rr / R) ** 2), 0.0)
    gx = np.where(rr < R, -2.0 * H * x / R ** 2, 0.0)
    gy = np.where(rr < R, -2.0 * H * y / R ** 2, 0.0)
    nrm = np.stack([-gx, -gy, np.ones_like(x)], -1).reshape(-1, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    pts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    nrm = nrm.astype(np.float32)
    bcs = [BC.REFLECTIVE_BOUNDARY, BC.PERIODIC_BOUNDARY, BC.REFLECTIVE_BOUNDARY]

    def pair():
        return make_pair_disks(pts, nrm, gd, 3, bcs, direction, particle, rays_pp=60, seed=77)

    t, o = pair()
    err, info = compare(t, o)
    assert t.traceMode() == 0
    if direction == TD.POS_Z:
        assert info["geometryHits"] > 1.005 * info["numRays"]  # (reflected rays meet the dome)
    f0 = t.getFluxF64()
    with monkeypatch.context() as m:
        m.setenv("VR_DEBUG_FLAGS", "256")
        t2, _ = pair()
        t2.apply()
        assert info_dict(t2) == info
        assert (t2.getFluxF64() == f0).all()


def test_morton_grid_keeps_its_proportions_bounded(monkeypatch):
    """The LBVH's Morton grid follows the scene box's proportions only up to 2 : 1.  Scaled freely axis by axis, a thin
    sheet (here 400 x 400 cells wide, one cell of relief) spends a third of the code bits on its relief and the tree cuts
    it into contour bands that overlap everywhere in plan: 3.4 x the trace time on a 10^6-disk sheet.  Same bits either
    way; the bounded grid must be clearly faster on the same box."""
    pts, nrm, gd = _rippled_surface(n=400, gd=1.0, amp=0.5, wave=4.0)

    def run():
        t = vr.TraceDisk(3)
        t.setGeometry(pts, nrm, gd)
        t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
        t.setParticleType(vr.DiffuseParticle(0.1, "flux"))
        t.setNumberOfRaysPerPoint(40)
        t.setRngSeed(5)
        best = None
        for _ in range(3):
            t.setRunNumber(1)
            t.apply()
            k = t.getRayTraceInfo().timeTraceKernel
            best = k if best is None else min(best, k)
        return t.getFluxF64(), info_dict(t), best

    monkeypatch.setenv("VR_NO_RELIEF", "1")  # (the walks the Morton grid is about: not the relief packets' flat-scene kernels)
    f0, i0, t0 = run()
    monkeypatch.setenv("VR_MORTON_ANISO", "1000000")
    f1, i1, t1 = run()
    assert i0 == i1 and (f0 == f1).all()
    assert t1 > 1.5 * t0, (t0, t1)


@pytest.mark.parametrize("sticking", [1.0, 0.3])
@pytest.mark.parametrize("geom", ["trench2d", "sphere"])
def test_small_scene_kernel_resident_in_lds(geom, sticking, monkeypatch):
    """Scenes of a few hundred primitives run trace_kernel MODE 4 (pair nodes, records, neighbourhood and flux
    accumulators staged in LDS).  Same accumulator bits and counters as the kernels that read the scene from HBM,
    and the oracle's counters."""
    def run():
        if geom == "trench2d":
            gd, p, n = trench2d()
            t = vr.TraceDisk(2)
            t.setGeometry(p, n, gd)
            t.setSourceDirection(TD.POS_Y)
            t.setBoundaryConditions([BC.REFLECTIVE_BOUNDARY] * 2)
        else:
            gd, p, n = sphere3d()
            t = vr.TraceDisk(3)
            t.setGeometry(p, n, gd)
            t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
        t.setParticleType(vr.DiffuseParticle(sticking, "flux"))
        t.setNumberOfRaysPerPoint(3000)
        t.setRngSeed(321)
        t.apply()
        return t.getFluxF64(), info_dict(t), t.traceMode()

    f1, i1, m1 = run()
    assert m1 == 4
    monkeypatch.setenv("VR_SMALL_SCENE", "0")
    f0, i0, m0 = run()
    assert m0 in (0, 1, 2, 3)
    assert i0 == i1 and (f0 == f1).all()


@pytest.mark.parametrize("variant", ["coned", "two_label", "material", "wdist", "specular"])
def test_small_scene_kernel_serves_every_particle(variant, monkeypatch):
    """MODE 4 with the plug-in particles (extended kernel), two data labels (one accumulator plane each in LDS), the
    per-material sticking map (staged in LDS) and WDIST crediting: every plane bit-identical to the HBM path."""
    gd, p, n = trench2d()

    def run():
        t = vr.TraceDisk(2)
        t.setGeometry(p, n, gd)
        t.setSourceDirection(TD.POS_Y)
        t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 2)
        if variant == "coned":
            t.setParticleType(vr.ConedCosineParticle(0.2, 3.0, 0.8, "flux"))
        elif variant == "two_label":
            t.setParticleType(vr.DiffuseCosineParticle(0.3, "flux", "cosFlux"))
        elif variant == "material":
            t.setMaterialIds((np.arange(len(p)) % 3).astype(np.int32))
            t.setParticleType(vr.DiffuseParticle(0.2, "flux", materialSticking={1: 0.6, 2: 1.0}))
        elif variant == "wdist":
            t.setParticleType(vr.DiffuseParticle(0.25, "flux"))
            t.setUseWdist(True)
        else:
            t.setParticleType(vr.SpecularParticle(0.15, 20.0, "flux"))
        t.setNumberOfRaysPerPoint(2000)
        t.setRngSeed(77)
        t.apply()
        ld = t.getLocalData()
        return [np.array(ld.getVectorData(k), copy=True) for k in range(t.numData())], info_dict(t), t.traceMode()

    f1, i1, m1 = run()
    assert m1 == 4
    monkeypatch.setenv("VR_SMALL_SCENE", "0")
    f0, i0, m0 = run()
    assert m0 != 4 and i0 == i1 and len(f0) == len(f1)
    for a, b in zip(f0, f1):
        assert np.array_equal(a, b) and a.sum() > 0


# ---------------------------------------------------------------------------
# edge cases of the reference's loop (limits, degenerate scenes, tiny launches)
# ---------------------------------------------------------------------------
def test_flat_scene_kernel_in_two_dimensions(monkeypatch):
    """A flat line of disks in 2-D (normals with a z component, which a 2-D trace drops: rayGeometryDisk.hpp:170-176):
    the packet query's crediting (MODE 3) re-derives the neighbour relation from the record centres, the CSR was built
    from the caller's points — the same floats, so MODE 3, MODE 0 and the oracle agree.  (MODE 4 off: the scene is small.)"""
    monkeypatch.setenv("VR_SMALL_SCENE", "0")
    n = 4000
    rng = np.random.default_rng(8)
    pts = np.zeros((n, 3), np.float32)
    pts[:, 0] = (np.arange(n) - n / 2) * 0.5
    nrm = np.tile(np.array([0, 1, 0], np.float32), (n, 1))
    nrm[:, 2] = rng.normal(scale=0.5, size=n)
    res = {}
    for flat in ("1", "0"):
        monkeypatch.setenv("VR_GENERAL_FLAT", flat)
        t, o = make_pair_disks(pts, nrm, 0.5, 2, [BC.PERIODIC_BOUNDARY] * 2, TD.POS_Y, ("diffuse", 0.2, 1.0), rays_pp=200)
        err, gi = compare(t, o)
        res[flat] = (t.traceMode(), gi, t.getFluxF64())
    assert res["1"][0] == 3 and res["0"][0] == 0
    assert res["1"][1] == res["0"][1] and (res["1"][2] == res["0"][2]).all()


def test_max_normalisation_with_the_host_builder(monkeypatch):
    """normalizeFlux(MAX) needs a reduction word of its own: with VR_HOST_BUILD=1 the device builder's scratch
    (where it used to live) does not exist (advisor, round 2)."""
    monkeypatch.setenv("VR_HOST_BUILD", "1")
    gd, p, n = sphere3d()
    t, o = make_pair_disks(p, n, gd, 3, [BC.REFLECTIVE_BOUNDARY] * 3, TD.POS_Z, ("diffuse", 0.5, 1.0), rays_pp=100)
    compare(t, o)
    f = t.getLocalData().getVectorData(0)
    a = t.normalizeFlux(f, vr.NormalizationType.MAX)
    b = o.normalize_flux(o.flux(), 1)
    assert l2_rel(a, b) <= 1e-6 and abs(float(np.nanmax(a)) - float(np.nanmax(b))) <= 1e-5
    g = t.getFluxNormalized(vr.NormalizationType.MAX)
    assert l2_rel(g, b) <= 1e-6


@pytest.mark.parametrize("max_refl,max_bh", [(2, 1000), (1000000, 1), (0, 0)])
def test_reflection_and_boundary_hit_limits(max_refl, max_bh):
    """rayTraceKernel.hpp:206-214 / :320-324: rays stop at maxBoundaryHits / maxReflections"""
    gd, p, n = trench3d()
    t, o = make_pair_disks(p, n, gd, 3, [BC.REFLECTIVE_BOUNDARY] * 3, TD.POS_Z, ("diffuse", 0.05, 1), rays_pp=8)
    t.setMaxReflections(max_refl)
    o.set_max_reflections(max_refl)
    t.setMaxBoundaryHits(max_bh)
    o.set_max_boundary_hits(max_bh)
    err, gi = compare(t, o, counter_slack=0)
    assert gi["raysTerminated"] > 0


@pytest.mark.parametrize("bcs", [[BC.REFLECTIVE_BOUNDARY, BC.IGNORE_BOUNDARY, BC.PERIODIC_BOUNDARY],
                                 [BC.IGNORE_BOUNDARY, BC.PERIODIC_BOUNDARY, BC.REFLECTIVE_BOUNDARY]])
def test_mixed_boundary_conditions(bcs):
    """rayBoundary.hpp:23-25: the condition is picked by AXIS (Q13)"""
    gd, p, n = trench3d()
    t, o = make_pair_disks(p, n, gd, 3, bcs, TD.POS_Z, ("diffuse", 0.2, 1), rays_pp=10)
    compare(t, o, counter_slack=0)


def test_source_below_hits_back_faces():
    """NEG_Z on the trench: every first hit is a back face (pass-through once, then kill, :224-249)"""
    gd, p, n = trench3d()
    t, o = make_pair_disks(p, n, gd, 3, [BC.PERIODIC_BOUNDARY] * 3, TD.NEG_Z, ("diffuse", 0.5, 1), rays_pp=5)
    err, gi = compare(t, o, counter_slack=0)
    assert gi["raysTerminated"] > 0


@pytest.mark.parametrize("npts", [1, 2, 5])
def test_degenerate_scenes(npts):
    """one disk; coincident disks (equal Morton codes, equal t: lower id wins); a short row"""
    if npts == 2:
        pts = np.zeros((2, 3), np.float32)
    else:
        pts = np.stack([np.arange(npts, dtype=np.float32), np.zeros(npts, np.float32), np.zeros(npts, np.float32)], 1)
    nrm = np.tile(np.array([[0, 0, 1]], np.float32), (npts, 1))
    t, o = make_pair_disks(pts, nrm, 1.0, 3, [BC.REFLECTIVE_BOUNDARY] * 3, TD.POS_Z, ("diffuse", 0.3, 1),
                           rays_fixed=5000)
    t.apply()
    o.apply(1)
    assert info_dict(t) == {k: o.info()[k] for k in INFO_KEYS}
    f, r = t.getLocalData().getVectorData(0), o.flux()
    assert l2_rel(f, r) <= 5e-6
    # a bounding box without extent makes the reference's disk-area code return NaN (0/0 normal,
    # rayDiskBoundingBoxIntersector.hpp:124-135): same NaNs, same finite values
    fn, rn = t.normalizeFlux(f), o.normalize_flux(r)
    assert (np.isnan(fn) == np.isnan(rn)).all()
    ok = ~np.isnan(rn)
    assert l2_rel(fn[ok], rn[ok]) <= FLUX_TOL


@pytest.mark.parametrize("rays", [1, 63, 65, 4097])
def test_tiny_launches(rays):
    pts, nrm = vr.io.plane_grid(12, 1.0)
    t, o = make_pair_disks(pts, nrm, 1.0, 3, [BC.PERIODIC_BOUNDARY] * 3, TD.POS_Z, ("diffuse", 0.4, 1),
                           rays_fixed=rays)
    t.apply()
    o.apply(1)
    assert info_dict(t) == {k: o.info()[k] for k in INFO_KEYS}
    assert l2_rel(t.getLocalData().getVectorData(0), o.flux()) <= 5e-6


def test_long_bounce_chains_use_the_full_engine_state():
    """A closed cavity (two facing planes, reflective side walls), sticking 0.02: a ray enters
    through the back of the upper plane (first back hit passes, rayTraceKernel.hpp:235-240) and
    bounces ~100 times until the roulette ends it, i.e. it draws far more than 156 numbers and
    leaves the streaming tier for the 312-word state (tier 2) — the stream must stay exact."""
    lo_p, lo_n = vr.io.plane_grid(12, 1.0)
    hi_p = lo_p.copy()
    hi_p[:, 2] = 3.0
    hi_n = -lo_n
    pts, nrm = np.concatenate([lo_p, hi_p]), np.concatenate([lo_n, hi_n])
    t, o = make_pair_disks(pts, nrm, 1.0, 3, [BC.REFLECTIVE_BOUNDARY] * 3, TD.POS_Z, ("diffuse", 0.02, 1),
                           rays_fixed=3000)
    err, gi = compare(t, o, counter_slack=0)
    assert gi["reflections"] > 50 * gi["numRays"]
    assert t.getRayTraceInfo().rngFullStates > 1000  # the tier was really exercised


def test_device_bvh_is_consistent_after_every_rebuild():
    """The bottom-up fit hands boxes from one workgroup to another with atomic stores and a
    wait instead of an L2 write-back fence (vr_setup.hip, fit_kernel): every internal box must
    still be exactly the union of its children's, on a big scene, on every rebuild."""
    pts, nrm = vr.io.plane_grid(700, 1.0)
    rng = np.random.default_rng(3)
    t = vr.TraceDisk(3)
    t.setParticleType(vr.DiffuseParticle(1.0, "flux"))
    t.setNumberOfRaysFixed(1000)
    for it in range(6):
        p = pts.copy()
        p[:, 2] = rng.normal(scale=0.3, size=p.shape[0]).astype(np.float32)  # a new rough surface
        t.setGeometry(p, nrm, 1.0)
        t.applyPrepare()
        assert t.debugBvhCheck() == 0
        t.applyLaunch()
        t.applyFinish()
        ti = t.getRayTraceInfo()
        # the fast hand-over never needed the fenced re-fit (a non-zero count would be a finding to root-cause
        # from the node bvh_check_kernel reports, not something the retry may hide), one build per new cloud
        assert ti.bvhRefits == 0 and ti.bvhBuilds == it + 1
    gd, v, tri = trench_mesh()
    m = vr.TraceTriangle(3)
    m.setGeometry(v, tri, gd)
    m.setParticleType(vr.DiffuseParticle(1.0, "flux"))
    m.applyPrepare()
    assert m.debugBvhCheck() == 0
    m.applyLaunch()
    m.applyFinish()
    assert m.getRayTraceInfo().bvhRefits == 0


def test_apply_wall_time_is_device_time_when_the_ray_count_grows():
    """apply() once per time step with a ray count that follows the surface: the ray-stream buffers grow
    geometrically and are kept, so the SECOND apply after a 50-fold increase costs its device time plus < 2 ms of
    host work (round 2: 66 ms of re-allocation on trenchGrid3D.dat), and a reservation makes the first one cheap too."""
    import time
    gd, p, n = trench3d()

    def tracer():
        t = vr.TraceDisk(3)
        t.setGeometry(p, n, gd)
        t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
        t.setParticleType(vr.DiffuseParticle(0.1, "flux"))
        t.setRngSeed(12345)
        return t

    def timed(t, rays):
        t.setNumberOfRaysFixed(rays)
        t.setRunNumber(1)
        t0 = time.perf_counter()
        t.apply()
        wall = time.perf_counter() - t0
        return (wall - t.getRayTraceInfo().timeTrace) * 1e3, info_dict(t)

    import gc
    gc.collect()      # (tracers of earlier tests release multi-GB device buffers when collected: not inside a timed call)
    gc.disable()
    try:
        t = tracer()
        timed(t, 1_000_000)
        timed(t, 50_000_000)                  # grows the stream (allocation: not asserted)
        runs = [timed(t, 50_000_000) for _ in range(2)]
        host2, i2 = min(h for h, _ in runs), runs[-1][1]
        assert host2 < 2.0, host2
        host3 = min(timed(t, 55_000_000)[0] for _ in range(2))   # +10 %: inside the head-room of the last growth
        assert host3 < 2.0, host3
        host4 = min(timed(t, 1_000_000)[0] for _ in range(2))    # shrinking keeps the buffers
        assert host4 < 2.0, host4
        r = tracer()
        r.reserveRays(50_000_000)
        timed(r, 1_000_000)                   # allocates the reservation
        host5, i5 = timed(r, 50_000_000)      # first apply at the big count: nothing left to allocate
        assert host5 < 2.0, host5
        assert i5 == i2                       # (and none of this changes a result)
    finally:
        gc.enable()


# ---------------------------------------------------------------------------
# randomised differential test: random scenes / settings, HIP path vs oracle
# ---------------------------------------------------------------------------
def _random_case(seed):
    rng = np.random.default_rng(seed)
    D = int(rng.choice([2, 3], p=[0.3, 0.7]))
    n = int(rng.integers(3, 400))
    gd = float(rng.choice([0.25, 0.5, 1.0]))
    span = gd * max(2.0, np.sqrt(n))
    pts = rng.uniform(-span, span, size=(n, 3)).astype(np.float32)
    # a rough surface facing the source on average, with some disks turned away
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    if D == 2:
        pts[:, 2] = 0
        nrm[:, 2] = 0
        direction = TD(int(rng.choice([int(TD.POS_Y), int(TD.NEG_Y), int(TD.POS_X), int(TD.NEG_X)])))
    else:
        direction = TD(int(rng.integers(0, 6)))
    axis = {TD.POS_X: 0, TD.NEG_X: 0, TD.POS_Y: 1, TD.NEG_Y: 1, TD.POS_Z: 2, TD.NEG_Z: 2}[direction]
    sign = 1.0 if direction in (TD.POS_X, TD.POS_Y, TD.POS_Z) else -1.0
    nrm[:, axis] = sign * (np.abs(nrm[:, axis]) + 0.5)
    flip = rng.random(n) < 0.1
    nrm[flip] *= -1
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    bcs = [BC(int(b)) for b in rng.integers(0, 3, size=3)]
    if rng.random() < 0.5:
        particle = ("diffuse", float(rng.choice([1.0, 0.5, 0.1, 0.02])), 1.0)
    else:
        particle = ("specular", float(rng.choice([1.0, 0.6, 0.2])), float(rng.choice([1.0, 8.0, 100.0])))
    radius = 0.0 if rng.random() < 0.7 else gd * float(rng.uniform(0.3, 1.2))
    primary = None
    if D == 3 and rng.random() < 0.3:  # tilted source: rejection loop in the generator
        primary = rng.normal(scale=0.4, size=3)
        primary[axis] = -sign
        primary = (primary / np.linalg.norm(primary)).astype(np.float32)
    return dict(primary=primary, D=D, pts=pts, nrm=nrm.astype(np.float32), gd=gd, direction=direction, bcs=bcs[:D] if D == 2 else bcs,
                particle=particle, radius=radius, rays=int(rng.integers(1, 40)), seed=int(rng.integers(0, 2**31)),
                max_refl=int(rng.choice([2**32 - 1, 50, 3])), max_bh=int(rng.choice([1000, 5, 0])))


@pytest.mark.parametrize("seed", range(60))
def test_random_scenes_match_oracle(seed):
    c = _random_case(1000 + seed)
    t, o = make_pair_disks(c["pts"], c["nrm"], c["gd"], c["D"], c["bcs"], c["direction"], c["particle"],
                           rays_pp=c["rays"], seed=c["seed"], radius=c["radius"], primary=c["primary"])
    t.setMaxReflections(c["max_refl"])
    o.set_max_reflections(c["max_refl"])
    t.setMaxBoundaryHits(c["max_bh"])
    o.set_max_boundary_hits(c["max_bh"])
    t.apply()
    o.apply(4)
    gi, oi = info_dict(t), o.info()
    assert gi == {k: oi[k] for k in INFO_KEYS}, c["particle"]
    f, r = t.getLocalData().getVectorData(0), o.flux()
    assert l2_rel(f, r) <= 5e-6
    a, b = t.getDiskAreas(), o.disk_areas()  # two independent restatements of the intersector
    assert (np.isnan(a) == np.isnan(b)).all()
    ok = ~np.isnan(b)
    assert np.allclose(a[ok], b[ok], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("seed", range(24))
def test_random_particle_lists_match_oracle(seed):
    """Randomised differential test of the registry and of particle lists: a random scene (as above, incl. tilted
    sources: the records' side array), random global data, material ids, and a list of 1 - 4 random particles —
    built-in, coned-cosine, two-label, coverage-dependent sticking, with and without per-material sticking — traced
    in ONE apply against one oracle run per particle with the same run number: every counter and every data label."""
    c = _random_case(7000 + seed)
    rng = np.random.default_rng(9000 + seed)
    n = len(c["pts"])
    mats = rng.integers(0, 3, size=n).astype(np.int32)
    cov = rng.uniform(0, 1, size=(2, n)).astype(np.float32)
    t = vr.TraceDisk(c["D"])
    t.setGeometry(c["pts"], c["nrm"], c["gd"], c["radius"])
    t.setBoundaryConditions(c["bcs"])
    t.setSourceDirection(c["direction"])
    t.setMaterialIds(mats)
    t.setGlobalData([cov[0], cov[1]])
    t.setNumberOfRaysPerPoint(c["rays"])
    t.setRngSeed(c["seed"])
    t.setMaxReflections(c["max_refl"])
    t.setMaxBoundaryHits(c["max_bh"])
    if c["primary"] is not None:
        t.setPrimaryDirection(c["primary"])
    plist = []
    for _ in range(int(rng.integers(1, 5))):
        kind = int(rng.integers(0, 5))
        s = float(rng.choice([1.0, 0.6, 0.25, 0.05]))
        ms = {int(m): float(rng.choice([1.0, 0.5, 0.1])) for m in rng.choice(3, size=int(rng.integers(0, 3)), replace=False)}
        power = float(rng.choice([1.0, 6.0]))
        if kind == 0:
            plist.append((vr.DiffuseParticle(s, "a", ms), po.DIFFUSE, s, 1.0, 0.0, ms))
        elif kind == 1:
            plist.append((vr.SpecularParticle(s, power, "a", ms), po.SPECULAR, s, power, 0.0, ms))
        elif kind == 2:
            cone = float(rng.choice([0.0, 0.4, 1.1]))
            plist.append((vr.ConedCosineParticle(s, power, cone, "a", ms), po.CONED_COSINE, s, power, cone, ms))
        elif kind == 3:
            plist.append((vr.DiffuseCosineParticle(s, "a", "b", ms), po.DIFFUSE_COSINE, s, 1.0, 0.0, ms))
        else:
            v = int(rng.integers(0, 2))
            plist.append((vr.CoverageStickingParticle(s, "a", v, ms), po.COVERAGE_STICKING, s, 1.0, float(v), ms))
    t.setParticleTypes([q[0] for q in plist])
    t.apply()
    ld = t.getLocalData()
    plane = 0
    for q, (_, okind, s, power, cone, ms) in enumerate(plist):
        o = po.Oracle()
        o.set_disks(c["pts"], c["nrm"], c["gd"], c["D"], radius=c["radius"])
        o.set_boundary_conditions([int(b) for b in c["bcs"]])
        o.set_source_direction(int(c["direction"]))
        o.set_material_ids(mats)
        o.set_global_data(0, cov[0])
        o.set_global_data(1, cov[1])
        o.set_particle_ex(okind, s, power, cone, -1.0)
        if ms:
            o.set_material_sticking(ms)
        o.set_num_rays_per_point(c["rays"])
        o.set_rng_seed(c["seed"])
        o.set_max_reflections(c["max_refl"])
        o.set_max_boundary_hits(c["max_bh"])
        if c["primary"] is not None:
            o.set_primary_direction(c["primary"])
        o.set_lazy_rng(True)
        o.apply(4)
        pi, oi = t.getParticleTraceInfo(q), o.info()
        assert {k: int(getattr(pi, k)) for k in INFO_KEYS} == {k: oi[k] for k in INFO_KEYS}, (q, okind)
        for l in range(o.num_data()):
            assert l2_rel(ld.getVectorData(plane), o.flux_data(l)) <= 5e-6, (q, okind, l)
            plane += 1
    assert plane == t.numData()


@pytest.mark.parametrize("seed", range(16))
def test_random_triangle_surfaces_match_oracle(seed):
    """random rough height fields as triangle meshes, random walls / particle / limits"""
    rng = np.random.default_rng(5000 + seed)
    m = int(rng.integers(2, 14))
    gd = float(rng.choice([0.5, 1.0]))
    xs, ys = np.meshgrid(np.arange(m) * gd, np.arange(m) * gd, indexing="ij")
    z = rng.normal(scale=float(rng.choice([0.0, 0.3, 2.0])) * gd, size=(m, m))
    v = np.stack([xs.ravel(), ys.ravel(), z.ravel()], 1).astype(np.float32)
    tri = []
    for i in range(m - 1):
        for j in range(m - 1):
            a, b, c, d = i * m + j, (i + 1) * m + j, (i + 1) * m + j + 1, i * m + j + 1
            tri += [[a, b, c], [a, c, d]] if rng.random() < 0.5 else [[a, b, d], [b, c, d]]
    tri = np.array(tri, dtype=np.uint32)
    bcs = [BC(int(b)) for b in rng.integers(0, 3, size=3)]
    t = vr.TraceTriangle(3)
    t.setGeometry(v, tri, gd)
    t.setBoundaryConditions(bcs)
    o = po.Oracle()
    o.set_triangles(v, tri, gd, 3)
    o.set_boundary_conditions([int(b) for b in bcs])
    if rng.random() < 0.5:
        st = float(rng.choice([1.0, 0.3, 0.05]))
        t.setParticleType(vr.DiffuseParticle(st, "flux"))
        o.set_particle(po.DIFFUSE, st)
    else:
        st, pw = float(rng.choice([1.0, 0.4])), float(rng.choice([1.0, 30.0]))
        t.setParticleType(vr.SpecularParticle(st, pw, "flux"))
        o.set_particle(po.SPECULAR, st, pw)
    rays, sd = int(rng.integers(1, 60)), int(rng.integers(0, 2**31))
    t.setNumberOfRaysPerPoint(rays)
    o.set_num_rays_per_point(rays)
    t.setRngSeed(sd)
    o.set_rng_seed(sd)
    o.set_lazy_rng(True)
    t.apply()
    o.apply(4)
    gi, oi = info_dict(t), o.info()
    assert gi == {k: oi[k] for k in INFO_KEYS}
    assert l2_rel(t.getLocalData().getVectorData(0), o.flux()) <= 5e-6


@pytest.mark.parametrize("geom", ["trench3d", "sphere", "trench2d"])
def test_device_smoothing_matches_host_and_oracle(geom, monkeypatch):
    """smoothFlux(flux, 1) runs on the device neighbourhood; it sums the neighbours in ascending
    original id like the host path and the oracle, so all three agree bit for bit."""
    gd, p, n = {"sphere": sphere3d, "trench3d": trench3d, "trench2d": trench2d}[geom]()
    D = 2 if geom == "trench2d" else 3
    direction = TD.POS_Y if D == 2 else TD.POS_Z
    t, o = make_pair_disks(p, n, gd, D, [BC.REFLECTIVE_BOUNDARY] * D, direction, ("diffuse", 0.3, 1), rays_pp=50)
    t.apply()
    o.apply(4)
    f = t.getLocalData().getVectorData(0).copy()
    dev = t.smoothFlux(f.copy(), 1)
    monkeypatch.setenv("VR_HOST_SMOOTH", "1")
    host = t.smoothFlux(f.copy(), 1)
    assert dev.tobytes() == host.tobytes()
    assert dev.tobytes() == o.smooth_flux(f.copy(), 1).tobytes()
    assert not np.array_equal(dev, f)  # it did smooth something
    # k > 1 (rayTraceDisk.hpp:146-193 builds a wider PointNeighborhood for the call): a range query of radius k * 2 r
    # over the resident BVH, fused with the averaging — the host path's and the oracle's bits again
    monkeypatch.delenv("VR_HOST_SMOOTH")
    for k in (2, 3):
        devk = t.smoothFlux(f.copy(), k)
        monkeypatch.setenv("VR_HOST_SMOOTH", "1")
        hostk = t.smoothFlux(f.copy(), k)
        monkeypatch.delenv("VR_HOST_SMOOTH")
        assert devk.tobytes() == hostk.tobytes(), k
        assert devk.tobytes() == o.smooth_flux(f.copy(), k).tobytes(), k
        assert not np.array_equal(devk, dev)


# ---------------------------------------------------------------------------
# BASELINE config C2 (the headline workload): P(1000) = 10^6 disks, PERIODIC, cosine source
# ---------------------------------------------------------------------------
def _c2_pair(sticking, total=100_000_000):
    pts, nrm = vr.io.plane_grid(1000, 1.0)
    t, o = make_pair_disks(pts, nrm, 1.0, 3, [BC.PERIODIC_BOUNDARY] * 3, TD.POS_Z, ("diffuse", sticking, 1.0),
                           rays_fixed=total)
    return t, o


@pytest.mark.parametrize("sticking", [1.0, 0.1])
def test_c2_million_disk_plane_matches_oracle(sticking):
    """The first 10^7 rays of the 10^8-ray C2 stream (same global ray indices, same seed) against the
    oracle: every counter equal, flux within the north-star tolerance (bit-equal for sticking 1)."""
    t, o = _c2_pair(sticking)
    t.setRayRange(0, 10_000_000)
    o.set_ray_range(0, 10_000_000)
    err, gi = compare(t, o, exact_flux=(sticking == 1.0))
    assert gi["numRays"] == 100_000_000 and gi["geometryHits"] >= 9_990_000
    assert t.traceMode() == (1 if sticking == 1.0 else 3)


def test_c2_full_size_properties():
    """C2 at its full 10^8 rays (no oracle at this size): conservation of trace segments, integer
    flux for sticking 1, the analytic plane answer (SOURCE-normalised flux = 1), bit-identical
    rerun, and two ray-range shards summing bit-exactly to the whole."""
    pts, nrm = vr.io.plane_grid(1000, 1.0)
    t = vr.TraceDisk(3)
    t.setGeometry(pts, nrm, 1.0)
    t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
    t.setParticleType(vr.DiffuseParticle(1.0, "flux"))
    t.setNumberOfRaysFixed(100_000_000)
    t.setRngSeed(12345)
    t.apply()
    i = info_dict(t)
    whole = t.getFluxF64()
    assert i["numRays"] == 100_000_000
    # a flat plane seen from above: every trace ends on the plane, on a wall (and goes on) or leaves
    assert i["totalRaysTraced"] == i["geometryHits"] + i["nonGeometryHits"] + i["boundaryHits"]
    assert i["geometryHits"] + i["nonGeometryHits"] == 100_000_000 and i["raysTerminated"] == 0
    assert (whole == np.rint(whole)).all()                      # unit weights only
    assert whole.sum() >= i["geometryHits"]                     # closest disk + overlapping neighbours
    norm = t.normalizeFlux(whole.astype(np.float32))
    assert abs(float(norm.mean()) - 1.0) < 5e-3                 # analytic: cosine source over a plane
    assert abs(whole.sum() / i["geometryHits"] - np.pi * 0.75) < 2e-2   # disks covering a point: pi r^2 / delta^2
    # rerun with the same seed: bit-identical (tests/rngSeed/rngSeed.cpp:48-51 at full size)
    t.setRunNumber(1)
    t.apply()
    assert (t.getFluxF64() == whole).all() and info_dict(t) == i
    # two shards of the ray index range
    parts = []
    for first in (0, 50_000_000):
        t.setRunNumber(1)
        t.setRayRange(first, 50_000_000)
        t.apply()
        parts.append((t.getFluxF64(), info_dict(t)))
    assert (parts[0][0] + parts[1][0] == whole).all()
    for k in INFO_KEYS[1:]:
        assert parts[0][1][k] + parts[1][1][k] == i[k], k


@pytest.mark.parametrize("config", ["C1_trench3d", "C4", "C5"])
def test_structured_configs_full_size_properties(config, monkeypatch):
    """The structured BASELINE configs at their FULL ray counts (no oracle in the suite at this size: tools/
    full_parity_case.py did that once per build, profiles/r03_full_parity_cases.txt): segment conservation, a bit-identical
    rerun, two ray-range shards summing bit-exactly to the whole — and the same counters and accumulator bits with the
    height-field and follow-up shortcuts switched off (VR_DEBUG_FLAGS=256)."""
    def tracer():
        if config == "C4":
            gd, v, tri = trench_mesh()
            t = vr.TraceTriangle(3)
            t.setGeometry(v, tri, gd)
            t.setParticleType(vr.SpecularParticle(0.1, 50.0, "flux"))
            t.setNumberOfRaysFixed(100_000_000)
        elif config == "C5":
            gd, p, n = trench2d()
            t = vr.TraceDisk(2)
            t.setGeometry(p, n, gd)
            t.setSourceDirection(TD.POS_Y)
            t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 2)
            t.setParticleType(vr.DiffuseParticle(0.1, "flux"))
            t.setNumberOfRaysFixed(100_000_000)
        else:
            gd, p, n = trench3d()
            t = vr.TraceDisk(3)
            t.setGeometry(p, n, gd)
            t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
            t.setParticleType(vr.DiffuseParticle(0.1, "flux"))
            t.setNumberOfRaysPerPoint(2000)   # the reference example's 5.78e7 rays
        t.setRngSeed(12345)
        return t

    t = tracer()
    t.apply()
    i = info_dict(t)
    whole = t.getFluxF64()
    rays = i["numRays"]
    assert rays >= 50_000_000
    # every trace ends on the geometry, on a wall (and goes on) or leaves; every ray ends exactly once
    assert i["totalRaysTraced"] >= i["geometryHits"] + i["nonGeometryHits"] + i["boundaryHits"]  # (+ back-face passes)
    assert i["reflections"] > rays and whole.sum() > 0
    t.setRunNumber(1)
    t.apply()
    assert (t.getFluxF64() == whole).all() and info_dict(t) == i
    half = rays // 2
    parts = []
    for first, count in ((0, half), (half, rays - half)):
        t.setRunNumber(1)
        t.setRayRange(first, count)
        t.apply()
        parts.append((t.getFluxF64(), info_dict(t)))
    assert (parts[0][0] + parts[1][0] == whole).all()
    for k in INFO_KEYS[1:]:
        assert parts[0][1][k] + parts[1][1][k] == i[k], k
    monkeypatch.setenv("VR_DEBUG_FLAGS", "256")
    t2 = tracer()
    t2.apply()
    assert info_dict(t2) == i and (t2.getFluxF64() == whole).all()


# ---------------------------------------------------------------------------
# BASELINE config C3: the same plane, 10^9 rays in total, index range sharded over 8 GPUs (SURVEY 8e:
# GPU g traces idx in [g * 1.25e8, (g + 1) * 1.25e8); global idx -> tea<3>(idx, seed), rayTraceKernel.hpp:118-121)
# ---------------------------------------------------------------------------
C3_TOTAL = 1_000_000_000


@pytest.mark.parametrize("sticking", [1.0, 0.1])
def test_c3_slice_of_the_last_rank_matches_oracle(sticking):
    """The first 10^6 rays of rank 7's range of the 10^9-ray stream (batchFirst + idxOff just below 2^30)
    against the oracle on the same global indices: every counter equal, flux <= 5e-6."""
    from viennaray_amd import distributed as vd
    first, count = vd.ray_shard(C3_TOTAL, 7, 8)
    assert (first, count) == (875_000_000, 125_000_000)
    t, o = _c2_pair(sticking, total=C3_TOTAL)
    t.setRayRange(first, 1_000_000)
    o.set_ray_range(first, 1_000_000)
    err, gi = compare(t, o, exact_flux=(sticking == 1.0))
    assert gi["numRays"] == C3_TOTAL and gi["geometryHits"] >= 999_000
    # ... and the slice straddling the end of the stream (the last 10^6 indices, then nothing)
    t.setRunNumber(1)
    o.set_run_number(1)
    t.setRayRange(C3_TOTAL - 1_000_000, 5_000_000)
    o.set_ray_range(C3_TOTAL - 1_000_000, 5_000_000)
    err, gi = compare(t, o, exact_flux=(sticking == 1.0))
    if sticking == 1.0:  # one segment chain per ray, and only the 10^6 existing indices were traced
        assert gi["geometryHits"] + gi["nonGeometryHits"] == 1_000_000


def test_c3_whole_shard_of_rank0_properties():
    """Rank 0's whole 1.25e8-ray shard of C3 (one batch, the path each of 8 GPUs takes): segment conservation,
    integer flux, pi * 0.75 credits per hit, the analytic plane answer, and its two halves summing to it."""
    from viennaray_amd import distributed as vd
    first, count = vd.ray_shard(C3_TOTAL, 0, 8)
    pts, nrm = vr.io.plane_grid(1000, 1.0)
    t = vr.TraceDisk(3)
    t.setGeometry(pts, nrm, 1.0)
    t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
    t.setParticleType(vr.DiffuseParticle(1.0, "flux"))
    t.setNumberOfRaysFixed(C3_TOTAL)
    t.setRngSeed(12345)
    t.setRayRange(first, count)
    t.apply()
    i = info_dict(t)
    whole = t.getFluxF64()
    assert i["numRays"] == C3_TOTAL
    assert i["totalRaysTraced"] == i["geometryHits"] + i["nonGeometryHits"] + i["boundaryHits"]
    assert i["geometryHits"] + i["nonGeometryHits"] == count and i["raysTerminated"] == 0
    assert (whole == np.rint(whole)).all()
    assert abs(whole.sum() / i["geometryHits"] - np.pi * 0.75) < 2e-2
    # SOURCE normalisation divides by the 10^9 rays of the whole job: this shard holds 1/8 of the flux
    norm = t.normalizeFlux(whole.astype(np.float32))
    assert abs(float(norm.mean()) * 8.0 - 1.0) < 5e-3
    parts = []
    for f0, n0 in ((first, count // 2), (first + count // 2, count - count // 2)):
        t.setRunNumber(1)
        t.setRayRange(f0, n0)
        t.apply()
        parts.append((t.getFluxF64(), info_dict(t)))
    assert (parts[0][0] + parts[1][0] == whole).all()
    for k in INFO_KEYS[1:]:
        assert parts[0][1][k] + parts[1][1][k] == i[k], k


# ---------------------------------------------------------------------------
# SURVEY 8f N1 on the device: computeDiskAreas + normalizeFlux as HIP kernels
# ---------------------------------------------------------------------------
def test_disk_areas_known_answers_on_the_device():
    """tests/diskAreas/diskAreas.cpp:58-61,76-96 through the product path: full / half / quarter of
    pi r^2 for interior / edge / corner disks, within the reference test's 1e-6."""
    n, gd = 5, 1.0
    pts, nrm = vr.io.plane_grid(n, gd)
    # the reference test uses createPlaneGrid(gridDelta, extent 2): points on [-2, 2]^2
    t = vr.TraceDisk(3)
    t.setGeometry(pts, nrm, gd)
    t.setBoundaryConditions([BC.REFLECTIVE_BOUNDARY] * 3)
    t.setParticleType(vr.DiffuseParticle(1.0, "flux"))
    t.setNumberOfRaysPerPoint(1)
    t.setRngSeed(0)
    t.apply()
    areas = t.getDiskAreas()
    r = t.getDiskRadius()
    whole = r * r * np.pi
    lo, hi = pts.min(0), pts.max(0)
    for i, p in enumerate(pts):
        onx = abs(p[0] - lo[0]) < 1e-6 or abs(p[0] - hi[0]) < 1e-6
        ony = abs(p[1] - lo[1]) < 1e-6 or abs(p[1] - hi[1]) < 1e-6
        exp = whole / 4 if (onx and ony) else whole / 2 if (onx or ony) else whole
        assert abs(areas[i] - exp) <= 1e-6 * max(1, exp) + 2e-6, (i, areas[i], exp)
    assert t.getRayTraceInfo().numRays == 25


@pytest.mark.parametrize("geom", ["trench3d", "sphere", "trench2d", "tilted"])
def test_device_disk_areas_bit_equal_to_oracle(geom):
    """the device intersector (vr_area.hpp, glibc acosf/sinf reproduced) against the oracle's independent
    restatement on the real glibc: same bits, for every boundary-condition / direction combination"""
    if geom == "tilted":  # random orientations: every disk near a wall is clipped at an angle
        rng = np.random.default_rng(7)
        pts, _ = vr.io.plane_grid(24, 0.5)
        nrm = rng.normal(size=pts.shape).astype(np.float32)
        nrm[:, 2] = np.abs(nrm[:, 2]) + 0.2
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        gd, D = 0.5, 3
    else:
        gd, pts, nrm = {"trench3d": trench3d, "sphere": sphere3d, "trench2d": trench2d}[geom]()
        D = 2 if geom == "trench2d" else 3
    dirs = [TD.POS_Y, TD.NEG_X] if D == 2 else [TD.POS_Z, TD.NEG_Z, TD.POS_X, TD.NEG_Y]
    for direction in dirs:
        for bcs in ([BC.REFLECTIVE_BOUNDARY] * 3, [BC.PERIODIC_BOUNDARY, BC.IGNORE_BOUNDARY, BC.REFLECTIVE_BOUNDARY],
                    [BC.IGNORE_BOUNDARY] * 3):
            t, o = make_pair_disks(pts, nrm, gd, D, bcs[:D] if D == 2 else bcs, direction, ("diffuse", 1.0, 1.0),
                                   rays_fixed=64)
            t.apply()
            o.apply(1)
            a, b = t.getDiskAreas(), o.disk_areas()
            assert (np.isnan(a) == np.isnan(b)).all()
            ok = ~np.isnan(b)
            assert (a[ok].view(np.uint32) == b[ok].view(np.uint32)).all(), (geom, direction, bcs,
                                                                            np.abs(a[ok] - b[ok]).max())


@pytest.mark.parametrize("geom", ["trench3d", "mesh", "trench2d"])
def test_device_normalization_matches_oracle(geom):
    """normalizeFlux as a HIP kernel (SOURCE and MAX), on a caller buffer and fused with the flux
    download, against the oracle's host loop: bit-equal given bit-equal inputs"""
    if geom == "mesh":
        gd, v, tri = trench_mesh()
        t = vr.TraceTriangle(3)
        t.setGeometry(v, tri, gd)
        o = po.Oracle()
        o.set_triangles(v, tri, gd, 3)
    else:
        gd, p, n = trench3d() if geom == "trench3d" else trench2d()
        D = 3 if geom == "trench3d" else 2
        t = vr.TraceDisk(D)
        t.setGeometry(p, n, gd)
        o = po.Oracle()
        o.set_disks(p, n, gd, D)
        if D == 2:
            t.setSourceDirection(TD.POS_Y)
            o.set_source_direction(po.POS_Y)
    t.setParticleType(vr.DiffuseParticle(1.0, "flux"))   # unit weights: the raw flux is bit-equal
    o.set_particle(po.DIFFUSE, 1.0)
    for x in (t,):
        x.setNumberOfRaysPerPoint(20)
        x.setRngSeed(99)
    o.set_num_rays_per_point(20)
    o.set_rng_seed(99)
    o.set_lazy_rng(True)
    t.apply()
    o.apply(po.max_threads())
    f, r = t.getLocalData().getVectorData(0), o.flux()
    assert (f == r).all()
    for norm in (vr.NormalizationType.SOURCE, vr.NormalizationType.MAX):
        want = o.normalize_flux(r, int(norm))
        got = t.normalizeFlux(f, norm)
        fused = t.getFluxNormalized(norm)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (geom, norm, np.nanmax(np.abs(got - want)))
        assert ((fused.view(np.uint32) == got.view(np.uint32)) | (np.isnan(fused) & np.isnan(got))).all()


# ---------------------------------------------------------------------------
# particle plug-ins of the device registry, data labels, WDIST crediting,
# mean-free-path scattering, SourceGrid, host-callback sources
# ---------------------------------------------------------------------------
def _plugin_pair(geom, particle, okind, sticking, power=1.0, cone=0.0, mfp=-1.0, rays=30, seed=4242, wdist=False):
    if geom == "mesh":
        gd, v, tri = trench_mesh()
        t = vr.TraceTriangle(3)
        t.setGeometry(v, tri, gd)
        o = po.Oracle()
        o.set_triangles(v, tri, gd, 3)
    else:
        gd, p, n = {"trench3d": trench3d, "trench2d": trench2d, "sphere": sphere3d}[geom]()
        D = 2 if geom == "trench2d" else 3
        t = vr.TraceDisk(D)
        t.setGeometry(p, n, gd)
        o = po.Oracle()
        o.set_disks(p, n, gd, D)
        if D == 2:
            t.setSourceDirection(TD.POS_Y)
            o.set_source_direction(po.POS_Y)
            t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 2)
            o.set_boundary_conditions([po.PERIODIC] * 2)
    t.setParticleType(particle)
    o.set_particle_ex(okind, sticking, power, cone, mfp)
    t.setUseWdist(wdist)
    o.set_wdist(wdist)
    t.setNumberOfRaysPerPoint(rays)
    o.set_num_rays_per_point(rays)
    t.setRngSeed(seed)
    o.set_rng_seed(seed)
    o.set_lazy_rng(True)
    return t, o


@pytest.mark.parametrize("geom", ["trench3d", "mesh", "trench2d"])
@pytest.mark.parametrize("cone", [0.3, 1.2, 0.0, 2.0])
def test_coned_cosine_particle_matches_oracle(geom, cone):
    """ReflectionConedCosine (rayReflection.hpp:52-120) in the extended kernel.  The accept-reject loop
    and the trigonometry run in double on both sides; the device library's sin / cos are not glibc's bit
    for bit, which after the narrowing to float changes a direction with probability ~1e-8 per sample:
    counters are compared exactly (a flip would show) and the flux to the north-star tolerance."""
    t, o = _plugin_pair(geom, vr.ConedCosineParticle(0.2, 3.0, cone, "flux"), po.CONED_COSINE, 0.2, 3.0, cone)
    err, i = compare(t, o)
    assert i["reflections"] > 0


@pytest.mark.parametrize("geom", ["trench3d", "mesh"])
def test_two_label_particle_matches_oracle(geom):
    """a particle with two data labels (AbstractParticle::getLocalDataLabels): both TracingData vectors"""
    t, o = _plugin_pair(geom, vr.DiffuseCosineParticle(0.3, "flux", "cosFlux"), po.DIFFUSE_COSINE, 0.3)
    err, i = compare(t, o)
    assert t.numData() == 2 and o.num_data() == 2
    ld = t.getLocalData()
    assert ld.getVectorDataIndex("cosFlux") == 1
    a, b = ld.getVectorData("cosFlux"), o.flux_data(1)
    assert l2_rel(a, b) <= 5e-6 and a.sum() > 0
    assert (ld.getVectorData("flux") >= a - 1e-3).all()      # cos <= 1


@pytest.mark.parametrize("geom", ["trench3d", "mesh", "trench2d", "plane"])
def test_coverage_dependent_sticking_reads_global_data(geom):
    """globalData on the device (rayParticle.hpp:21-81 hands `const TracingData *globalData` to every callback;
    rayTrace.hpp:141): the registry's CoverageStickingParticle sticks with s0 * (1 - coverage[primID]), coverage =
    vector 1 of the caller's global data (vector 0 is a decoy) — every counter and the flux against the oracle."""
    if geom == "plane":   # a flat scene: MODE 3, the packet query's crediting of registry particles
        pts, nrm = vr.io.plane_grid(60, 1.0)
        t = vr.TraceDisk(3)
        t.setGeometry(pts, nrm, 1.0)
        t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
        o = po.Oracle()
        o.set_disks(pts, nrm, 1.0, 3)
        o.set_boundary_conditions([po.PERIODIC] * 3)
        t.setParticleType(vr.CoverageStickingParticle(0.6, "flux", coverageVector=1))
        o.set_particle_ex(po.COVERAGE_STICKING, 0.6, 1.0, 1.0, -1.0)   # (4th argument: the model's params[0])
        for x, m in ((t, "setNumberOfRaysPerPoint"), (o, "set_num_rays_per_point")):
            getattr(x, m)(300)
        t.setRngSeed(4242)
        o.set_rng_seed(4242)
        o.set_lazy_rng(True)
    else:
        t, o = _plugin_pair(geom, vr.CoverageStickingParticle(0.6, "flux", coverageVector=1), po.COVERAGE_STICKING, 0.6,
                            cone=1.0)
    n = t._n
    rng = np.random.default_rng(17)
    decoy = rng.uniform(0, 1, n).astype(np.float32)
    coverage = rng.uniform(0, 1, n).astype(np.float32)
    coverage[::7] = 1.0      # fully covered: sticking 0, the ray keeps its whole weight
    coverage[3::11] = 0.0
    g = vr.TracingData()
    g.setNumberOfVectorData(2)
    g.setVectorData(0, decoy, "decoy")
    g.setVectorData(1, coverage, "coverage")
    t.setGlobalData(g)
    assert t.getGlobalData() is g
    o.set_global_data(0, decoy)
    o.set_global_data(1, coverage)
    err, gi = compare(t, o)
    assert gi["reflections"] > gi["geometryHits"] // 2
    if geom == "plane":
        assert t.traceMode() == 3
    f_cov = t.getFluxF64()
    # without global data the model reads coverage 0: plain sticking 0.6 — the DiffuseParticle's flux, and a different
    # one wherever a ray can meet the surface twice (on the plane every reflected ray leaves: nothing to see there)
    t.setGlobalData([])
    t.setRunNumber(1)
    t.apply()
    assert geom == "plane" or l2_rel(t.getFluxF64(), f_cov) > 1e-3
    o.set_global_data(0, None)
    o.set_run_number(1)
    o.apply(po.max_threads())
    assert l2_rel(t.getLocalData().getVectorData(0), o.flux()) <= 5e-6


@pytest.mark.parametrize("geom", ["trench3d", "plane", "trench2d", "mesh"])
def test_multi_particle_apply_shares_the_generator_pass(geom):
    """Several particles in ONE apply() (gpu/raygTrace.hpp:163-248: one launch per particle, all with the apply's seed).
    Four particles — two with the source's cosine power 1 (one generator pass for both), an absorbing one, and a
    specular one with power 8 — against four oracle runs with the same run number: every particle's counters and
    every data label."""
    plist = [(vr.DiffuseParticle(0.2, "a"), po.DIFFUSE, 0.2, 1.0), (vr.DiffuseCosineParticle(0.5, "b", "bcos"), po.DIFFUSE_COSINE, 0.5, 1.0),
             (vr.DiffuseParticle(1.0, "c"), po.DIFFUSE, 1.0, 1.0), (vr.SpecularParticle(0.3, 8.0, "d"), po.SPECULAR, 0.3, 8.0)]
    if geom == "plane":
        pts, nrm = vr.io.plane_grid(70, 1.0)
        t = vr.TraceDisk(3)
        t.setGeometry(pts, nrm, 1.0)
        t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)

        def oracle():
            o = po.Oracle()
            o.set_disks(pts, nrm, 1.0, 3)
            o.set_boundary_conditions([po.PERIODIC] * 3)
            return o
    else:
        t, _ = _plugin_pair(geom, plist[0][0], plist[0][1], plist[0][2])

        def oracle():
            return _plugin_pair(geom, plist[0][0], plist[0][1], plist[0][2])[1]
    t.setNumberOfRaysPerPoint(40)
    t.setRngSeed(99)
    t.setParticleTypes([q[0] for q in plist])
    t.apply()
    assert t.numData() == 5
    ld = t.getLocalData()
    labels = [ld.getVectorDataLabel(k) for k in range(5)]
    assert labels == ["a", "b", "bcos", "c", "d"]
    info = info_dict(t)
    sums = {k: 0 for k in INFO_KEYS[1:]}
    plane = 0
    for q, (particle, okind, sticking, power) in enumerate(plist):
        o = oracle()
        o.set_particle_ex(okind, sticking, power, 0.0, -1.0)
        o.set_num_rays_per_point(40)
        o.set_rng_seed(99)
        o.set_lazy_rng(True)
        o.apply(po.max_threads())
        pi = t.getParticleTraceInfo(q)
        oi = o.info()
        for k in INFO_KEYS:
            assert int(getattr(pi, k)) == oi[k], (q, k)
        for k in sums:
            sums[k] += oi[k]
        for l in range(o.num_data()):
            assert l2_rel(ld.getVectorData(plane), o.flux_data(l)) <= 5e-6, (q, l)
            plane += 1
    assert {k: info[k] for k in sums} == sums       # getRayTraceInfo(): the sums over the particles
    assert t.getRunNumber() == 2                     # ONE apply
    # a single-particle apply afterwards is the plain path again
    t.setParticleType(vr.DiffuseParticle(0.2, "a"))
    t.setRunNumber(1)
    t.apply()
    assert t.numData() == 1 and (t.getLocalData().getVectorData("a") == ld.getVectorData(0)).all()


USER_TWO_LABELS = """
// the registry's two-label particle, written as a caller would write it: DiffuseParticle + a second label
struct VrUserModel : ModelDiffuse {
  static constexpr int kNumData = 2;
  template <class Credit>
  __device__ static void collide(const ModelCtx &, float w, const V3 &rayDir, const V3 &n, unsigned, Credit &&credit) {
    credit(0, w);
    const float cosTheta = -vdot(rayDir, n);
    credit(1, w * fmaxf(cosTheta, 0.f));
  }
};
"""
USER_COVERAGE = """
// coverage-dependent sticking with the coverage vector's index in params[0] and a gain in params[1]
struct VrUserModel : ModelDiffuse {
  __device__ static float sticking(const ModelCtx &m, unsigned primID, float base) {
    return base * (1.f - m.global.vector((unsigned)m.params[0], primID)) * m.params[1];
  }
};
"""


@pytest.fixture(scope="module")
def model_cache(tmp_path_factory):
    """one code-object cache for the whole module: each run-time model is compiled once (about 11 s), then found"""
    return str(tmp_path_factory.mktemp("vr_model_cache"))


@pytest.mark.parametrize("geom", ["trench3d", "plane", "trench2d", "mesh"])
def test_particle_model_registered_at_run_time(geom, model_cache, monkeypatch):
    """OPEN registration (gpu/raygCallableConfig.hpp:7-18: the reference's GPU path registers user callables per particle):
    the SOURCE of a model is compiled at run time (hipcc --genco), loaded as a code object and traced.  Two user models
    that restate built-in ones give the built-in particles' flux bit for bit, on every kernel variant (general, flat
    scene with packet-query crediting, LDS-resident scene, triangles) — alone and inside a particle list."""
    tmp_path = model_cache
    monkeypatch.setenv("VR_CACHE_DIR", tmp_path)

    def tracer():
        if geom == "plane":
            pts, nrm = vr.io.plane_grid(70, 1.0)
            t = vr.TraceDisk(3)
            t.setGeometry(pts, nrm, 1.0)
            t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
        else:
            t, _ = _plugin_pair(geom, vr.DiffuseParticle(0.5, "x"), po.DIFFUSE, 0.5)
        t.setNumberOfRaysPerPoint(60)
        t.setRngSeed(31)
        return t
    t = tracer()
    k2 = t.registerParticleModel(USER_TWO_LABELS, numData=2, name="twoLabels")
    kc = t.registerParticleModel(USER_COVERAGE, numData=1, name="coverage")
    assert k2 >= 1000 and kc == k2 + 1
    assert any(f.endswith(".hsaco") for f in os.listdir(tmp_path))
    cov = np.random.default_rng(3).uniform(0, 1, t._n).astype(np.float32)
    t.setGlobalData([cov])

    def run(particles):
        t.setParticleTypes(particles)
        t.setRunNumber(1)
        t.apply()
        ld = t.getLocalData()
        return t.traceMode(), info_dict(t), [ld.getVectorData(k).copy() for k in range(t.numData())]
    m_user, i_user, f_user = run([vr.UserModelParticle(k2, 0.3, ["flux", "cos"])])
    m_ref, i_ref, f_ref = run([vr.DiffuseCosineParticle(0.3, "flux", "cos")])
    assert m_user == m_ref == {"trench3d": 0, "plane": 3, "trench2d": 4, "mesh": 0}[geom]
    assert i_user == i_ref and all((a == b).all() for a, b in zip(f_user, f_ref))
    _, i_user, f_user = run([vr.UserModelParticle(kc, 0.8, ["flux"], params=[0.0, 1.0])])
    _, i_ref, f_ref = run([vr.CoverageStickingParticle(0.8, "flux", 0)])
    assert i_user == i_ref and (f_user[0] == f_ref[0]).all()
    _, i_half, f_half = run([vr.UserModelParticle(kc, 0.8, ["flux"], params=[0.0, 0.5])])   # the model's own parameter
    assert geom == "plane" or i_half != i_ref     # (on the plane every reflected ray leaves: sticking changes nothing there)
    # a list mixing run-time and built-in models: one apply, every label in place
    _, i_mix, f_mix = run([vr.UserModelParticle(k2, 0.3, ["flux", "cos"]), vr.DiffuseParticle(0.4, "d"),
                           vr.UserModelParticle(kc, 0.8, ["c"], params=[0.0, 1.0])])
    assert len(f_mix) == 4 and (f_mix[3] == f_ref[0]).all()
    _, _, f_two = run([vr.DiffuseCosineParticle(0.3, "flux", "cos")])
    assert (f_mix[0] == f_two[0]).all() and (f_mix[1] == f_two[1]).all()
    # a second context finds the code objects in the cache (no second compile: the directory does not grow)
    before = sorted(os.listdir(tmp_path))
    t2 = tracer()
    assert t2.registerParticleModel(USER_TWO_LABELS, numData=2) >= 1000 and sorted(os.listdir(tmp_path)) == before


def test_particle_model_that_does_not_compile_is_refused(tmp_path, monkeypatch):
    monkeypatch.setenv("VR_CACHE_DIR", str(tmp_path))
    t = vr.TraceDisk(3)
    with pytest.raises(vr.VrError) as e:
        t.registerParticleModel("struct VrUserModel : ModelDiffuse { static constexpr int kNumData = nonsense; };", numData=1)
    assert "did not compile" in str(e.value) and "nonsense" in str(e.value)
    with pytest.raises(vr.VrError):   # kNumData must be what the caller announced
        t.registerParticleModel("struct VrUserModel : ModelDiffuse { static constexpr int kNumData = 2; };", numData=1)
    with pytest.raises(vr.VrError):   # an unregistered kind
        t.setParticleType(vr.UserModelParticle(1000, 0.5, ["x"]))


def test_run_time_model_cache_and_sources_are_checked(tmp_path, monkeypatch):
    """The code-object cache holds code that hipModuleLoad runs: a directory that is not the caller's own with mode 0700
    is refused (another local user could have planted a code object under a predictable name).  And a model is compiled
    only from the kernel sources THIS library was built from — its kernels take the library's TraceParams by value, so
    sources with another layout would end in a GPU memory fault, not in an error code (round-3 advisor)."""
    import shutil
    import stat
    model = "struct VrUserModel : ModelDiffuse {};"
    t = vr.TraceDisk(3)
    open_dir = tmp_path / "shared_cache"
    open_dir.mkdir()
    os.chmod(open_dir, 0o755)
    monkeypatch.setenv("VR_CACHE_DIR", str(open_dir))
    with pytest.raises(vr.VrError, match="mode 0700"):
        t.registerParticleModel(model, numData=1)
    link = tmp_path / "link_cache"
    private = tmp_path / "private_cache"
    private.mkdir(mode=0o700)
    os.symlink(private, link)
    monkeypatch.setenv("VR_CACHE_DIR", str(link))
    with pytest.raises(vr.VrError, match="symbolic link"):
        t.registerParticleModel(model, numData=1)
    # kernel sources that differ from the library's own
    monkeypatch.setenv("VR_CACHE_DIR", str(private))
    src = os.path.join(os.path.dirname(vr.LIB_PATH), "csrc")
    other = tmp_path / "csrc"
    other.mkdir()
    for fn in os.listdir(src):
        if fn.endswith((".hip", ".hpp")):
            shutil.copy(os.path.join(src, fn), other / fn)
    with open(other / "vr_types.hpp", "a") as fh:
        fh.write("\n// edited after the library was built\n")
    monkeypatch.setenv("VR_CSRC_DIR", str(other))
    with pytest.raises(vr.VrError, match="not the ones this library was built from"):
        t.registerParticleModel(model, numData=1)
    assert not [f for f in os.listdir(private) if f.endswith(".hsaco")]
    assert stat.S_IMODE(os.stat(private).st_mode) == 0o700


@pytest.mark.parametrize("n,flat", [(21, "1"), (21, "0"), (100, "0"), (8, "1")])
def test_two_label_particle_on_coarse_planes_matches_oracle(n, flat, monkeypatch):
    """Coarse flat scenes are where a wave's credits pile up on a few disks (merged per disk and weight, or summed over the
    wave): the two-label registry particle against the oracle on P(21) / P(100) / P(8) with REFLECTIVE walls, through the
    packet-query crediting (MODE 3), the per-lane neighbour loop (MODE 0) and the LDS-resident kernel (MODE 4).
    Analytic check: a cosine source makes label 1 / label 0 = E[cos] = 2/3."""
    monkeypatch.setenv("VR_GENERAL_FLAT", flat)
    pts, nrm = vr.io.plane_grid(n, 0.5)
    t = vr.TraceDisk(3)
    t.setGeometry(pts, nrm, 0.5)
    t.setParticleType(vr.DiffuseCosineParticle(0.5, "flux", "cos"))
    o = po.Oracle()
    o.set_disks(pts, nrm, 0.5, 3)
    o.set_particle_ex(po.DIFFUSE_COSINE, 0.5, 1.0, 0.0, -1.0)
    for x, m in ((t, "setNumberOfRaysPerPoint"), (o, "set_num_rays_per_point")):
        getattr(x, m)(400)
    t.setRngSeed(21)
    o.set_rng_seed(21)
    o.set_lazy_rng(True)
    err, gi = compare(t, o)
    assert t.traceMode() == (4 if n == 8 else (3 if flat == "1" else 0))
    a, b = t.getLocalData().getVectorData("cos"), o.flux_data(1)
    assert l2_rel(a, b) <= 5e-6
    assert abs(float(a.sum()) / float(t.getLocalData().getVectorData("flux").sum()) - 2.0 / 3.0) < 0.01


def test_registry_particles_use_packet_query_crediting_on_flat_scenes(monkeypatch):
    """the two-label particle on P(100): MODE 3 (packet-query crediting of registry particles) gives what MODE 0 gives"""
    pts, nrm = vr.io.plane_grid(100, 1.0)

    def run():
        t = vr.TraceDisk(3)
        t.setGeometry(pts, nrm, 1.0)
        t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
        t.setParticleType(vr.DiffuseCosineParticle(0.3, "flux", "cos"))
        t.setNumberOfRaysFixed(2_000_000)
        t.setRngSeed(5)
        t.apply()
        return t.traceMode(), info_dict(t), t.getLocalData().getVectorData(0).copy(), t.getLocalData().getVectorData(1).copy()
    m3, i3, a3, b3 = run()
    monkeypatch.setenv("VR_GENERAL_FLAT", "0")
    m0, i0, a0, b0 = run()
    assert (m3, m0) == (3, 0) and i3 == i0
    assert (a3 == a0).all() and l2_rel(b3, b0) <= 1e-6


@pytest.mark.parametrize("geom", ["trench3d", "sphere", "trench2d"])
def test_wdist_crediting_matches_oracle(geom):
    """VIENNARAY_USE_WDIST (rayTraceKernel.hpp:258-296): the hit's weight shared by inverse impact distance"""
    t, o = _plugin_pair(geom, vr.DiffuseParticle(0.4, "flux"), po.DIFFUSE, 0.4, wdist=True)
    t.apply()
    o.apply(po.max_threads())
    gi, oi = info_dict(t), o.info()
    assert gi == {k: oi[k] for k in INFO_KEYS}
    f, r = t.getLocalData().getVectorData(0), o.flux()
    assert l2_rel(f, r) <= 1e-5
    # the shares of one hit sum to numDisksHit * w / ... : total weight is conserved per hit
    t2, o2 = _plugin_pair(geom, vr.DiffuseParticle(0.4, "flux"), po.DIFFUSE, 0.4, wdist=False)
    t2.apply()
    assert abs(f.sum() / t2.getLocalData().getVectorData(0).sum() - 1) < 1e-3


@pytest.mark.parametrize("geom,mfp", [("trench3d", 20.0), ("mesh", 5.0), ("trench2d", 50.0)])
def test_mean_free_path_scattering_matches_oracle(geom, mfp):
    """getMeanFreePath() > 0 (rayTraceKernel.hpp:179-203) with quirk Q1 kept; glibc expf reproduced"""
    part = vr.DiffuseParticle(0.3, "flux")
    part.meanFreePath = mfp
    t, o = _plugin_pair(geom, part, po.DIFFUSE, 0.3, mfp=mfp)
    err, i = compare(t, o)
    assert i["particleHits"] > 0


def test_source_grid_known_answers_and_parity():
    """tests/createSourceGrid/createSourceGrid.cpp:43-46 on the product path (origins on the extended top
    face, directions pointing down) + flux parity of a SourceGrid run against the oracle"""
    gd, p, n = sphere3d()
    t, o = make_pair_disks(p, n, gd, 3, [BC.REFLECTIVE_BOUNDARY] * 3, TD.POS_Z, ("specular", 0.5, 2.0), rays_pp=40)
    grid = o.create_source_grid(len(p), gd)
    # origins lie on the source face of the adjusted bounding box: top of the geometry + 2 x padding
    # (the reference test pads by gridDelta and asserts 1 + 2 gridDelta; apply() pads by the disk radius)
    assert len(grid) > 0 and np.allclose(grid[:, 2], o.bbox()[1, 2], atol=1e-6)
    assert abs(o.bbox()[1, 2] - (1.0 + 2 * o.disk_radius())) < 1e-5
    t.setSource(vr.SourceGrid(grid))
    o.set_source_grid(grid)
    org, d = t.debugSourceSample(np.arange(len(grid), dtype=np.uint64), 1)
    assert (d[:, 2] < 0).all()
    assert np.allclose(org, grid, atol=1e-6)
    err, i = compare(t, o)
    assert i["numRays"] == len(grid) * 40
    # resetSource(): back to the random source, bit-identical to a tracer that never had a grid
    t.resetSource()
    o.set_source_grid(None)
    compare(t, o)


@pytest.mark.parametrize("sticking", [1.0, 0.2])
def test_host_callback_rays_reproduce_the_internal_source(sticking):
    """setSource(custom Source): the rays a host callback produced (here: exactly what SourceRandom would
    have produced, with the 4 engine outputs it consumes) give the bit-identical run — the per-ray RNG
    continues where the callback left it"""
    gd, p, n = trench3d()
    t = vr.TraceDisk(3)
    t.setGeometry(p, n, gd)
    t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
    t.setParticleType(vr.DiffuseParticle(sticking, "flux"))
    t.setNumberOfRaysFixed(200000)
    t.setRngSeed(11)
    t.apply()
    f0, i0 = t.getFluxF64(), info_dict(t)
    org, d = t.debugSourceSample(np.arange(200000, dtype=np.uint64), 12)  # kernel seed = rngSeed + runNumber(1)
    t.setHostRays(org, d, np.full(200000, 4, dtype=np.uint32))
    t.setRunNumber(1)
    t.apply()
    assert info_dict(t) == i0
    assert (t.getFluxF64() == f0).all()


@pytest.mark.parametrize("geom,sticking", [("trench3d", 0.2), ("trench3d", 1.0), ("trench2d", 0.3)])
def test_host_source_initial_weights_and_source_area(geom, sticking):
    """A user Source that overrides getInitialRayWeight(idx) and getSourceArea() (raySource.hpp:17-18): the weight a
    ray starts with scales its credits AND the roulette's thresholds (rayTraceKernel.hpp:124,435-460), the area
    scales normalizeFlux(SOURCE) (rayTraceDisk.hpp:127).  Host rays with weights 0.05 .. 3 against the oracle."""
    D = 2 if geom == "trench2d" else 3
    gd, p, n = trench2d() if D == 2 else trench3d()
    nr = 120_000
    t = vr.TraceDisk(D)
    t.setGeometry(p, n, gd)
    bcs = [BC.REFLECTIVE_BOUNDARY] * D
    t.setBoundaryConditions(bcs)
    if D == 2:
        t.setSourceDirection(TD.POS_Y)
    t.setParticleType(vr.DiffuseParticle(sticking, "flux"))
    t.setNumberOfRaysFixed(nr)
    t.setRngSeed(77)
    org, d = t.debugSourceSample(np.arange(nr, dtype=np.uint64), 78)
    rng = np.random.default_rng(5)
    w = np.exp(rng.uniform(np.log(0.05), np.log(3.0), size=nr)).astype(np.float32)
    draws = np.full(nr, 4 if D == 3 else 3, dtype=np.uint32)
    t.setHostRays(org, d, draws, weights=w, sourceArea=123.5)
    o = po.Oracle()
    o.set_disks(p, n, gd, D)
    o.set_boundary_conditions([int(b) for b in bcs])
    if D == 2:
        o.set_source_direction(po.POS_Y)
    o.set_particle(po.DIFFUSE, sticking)
    o.set_rng_seed(77)
    o.set_host_rays(org, d, weights=w, source_area=123.5)
    o.set_host_ray_draws(draws)
    o.set_lazy_rng(True)
    err, gi = compare(t, o)
    assert t.traceMode() != 1 and t.traceMode() != 2       # weighted rays never run an absorbing kernel
    assert abs(t.getSourceArea() - 123.5) < 1e-4
    # the same rays with unit weights give a different flux (the weights were really used) ...
    t.setHostRays(org, d, draws)
    t.setRunNumber(1)
    t.apply()
    f1 = t.getFluxF64()
    assert l2_rel(f1, o.flux()) > 0.05
    # ... and the bounding-box source area again
    assert abs(t.getSourceArea() - 123.5) > 1.0


# ---------------------------------------------------------------------------
# the reference's hand-placed boundary rays (gpu/tests/boundaries/boundaries.cpp:77-78,105-106,133-134 with
# the rays of gpu/tests/boundaries/TestPipelineTriangle.cu:85-128) traced by the product's full state machine
# ---------------------------------------------------------------------------
def _norm(v):
    v = np.asarray(v, dtype=np.float32)
    return v / np.float32(np.linalg.norm(v.astype(np.float64)))


def _boundary_fixture_3d():
    nodes = np.array([[1, 0, 0], [0, 0, 0], [1, .5, 0], [0, .5, 0], [1, .5, 1], [0, .5, 1], [1, 1, 1], [0, 1, 1]], np.float32)
    tris = np.array([[0, 1, 2], [1, 3, 2], [2, 4, 3], [3, 4, 5], [5, 4, 6], [5, 6, 7]], np.uint32)
    org = np.array([[.5, .5, 1.1], [.5, .5, 1.5]], np.float32)
    d = np.stack([_norm([0, -1, -.5]), _norm([0, .6, -.5])])
    return nodes, tris, org, d


def test_reference_boundary_rays_3d_reflective():
    """boundaries.cpp:77-78: ray 0 reaches triangle 3 (the step's side wall) and ray 1 triangle 5 (the top)
    after one reflection off the y walls each"""
    nodes, tris, org, d = _boundary_fixture_3d()
    t = vr.TraceTriangle(3)
    t.setGeometry(nodes, tris, 0.5)
    t.setBoundaryConditions([BC.REFLECTIVE_BOUNDARY] * 3)
    t.setParticleType(vr.DiffuseParticle(1.0, "flux"))
    t.setRngSeed(1)
    t.setHostRays(org, d)
    t.apply()
    f = t.getLocalData().getVectorData(0)
    i = info_dict(t)
    assert f[3] > 0 and f[5] > 0 and f.sum() == 2            # boundaries.cpp:77-78
    assert i["numRays"] == 2 and i["boundaryHits"] == 2 and i["geometryHits"] == 2 and i["totalRaysTraced"] == 4
    o = po.Oracle()
    o.set_triangles(nodes, tris, 0.5, 3)
    o.set_boundary_conditions([po.REFLECTIVE] * 3)
    o.set_particle(po.DIFFUSE, 1.0)
    o.set_rng_seed(1)
    o.set_host_rays(org, d)
    o.apply(1)
    assert (o.flux() == f).all() and {k: o.info()[k] for k in INFO_KEYS} == i


@pytest.mark.parametrize("periodic", [False, True])
def test_reference_boundary_rays_2d(periodic):
    """boundaries.cpp:105-106 (reflective: triangles 3 and 5) and :133-134 (periodic: 3 and 7), D = 2.
    The reference's OptiX test pipeline credits every triangle hit; the CPU path this library follows kills
    a ray that meets a triangle from behind (rayTraceKernel.hpp:242-249), so the expectation is asserted
    for the rays that arrive from the front and the whole event sequence is compared with the oracle."""
    nodes = np.array([[0, 0, 0], [.5, 0, 0], [.5, 1, 0], [1, 1, 0]], np.float32)
    lines = np.array([[0, 1], [1, 2], [2, 3]], np.uint32)
    v, tri, keep = vr.io.lines_to_triangles(nodes, lines, 0.5)
    nv = len(v)
    v = np.concatenate([v, np.array([[.6, 0, -.25], [.6, 0, .25], [.6, 1, .25], [.6, 1, -.25]], np.float32)])
    tri = np.concatenate([tri, np.array([[nv, nv + 1, nv + 2], [nv, nv + 3, nv + 2]], np.uint32)])
    org = np.array([[.5, 1.1, 0], [.5, 1.5, 0]], np.float32)
    d = np.stack([_norm([-1, -.5, 0]), _norm([.6, -.5, 0])])
    bc = BC.PERIODIC_BOUNDARY if periodic else BC.REFLECTIVE_BOUNDARY
    t = vr.TraceTriangle(2)
    t.setGeometry(v, tri, 0.5)
    t.setSourceDirection(TD.POS_Y)
    t.setBoundaryConditions([bc] * 2)
    t.setParticleType(vr.DiffuseParticle(1.0, "flux"))
    t.setRngSeed(1)
    t.setHostRays(org, d)
    t.apply()
    f = t.getLocalData().getVectorData(0)
    i = info_dict(t)
    o = po.Oracle()
    o.set_triangles(v, tri, 0.5, 2)
    o.set_source_direction(po.POS_Y)
    o.set_boundary_conditions([int(bc)] * 2)
    o.set_particle(po.DIFFUSE, 1.0)
    o.set_rng_seed(1)
    o.set_host_rays(org, d)
    o.set_event_capacity(64)
    o.apply(1)
    assert (o.flux() == f).all() and {k: o.info()[k] for k in INFO_KEYS} == i
    assert i["boundaryHits"] == 2                             # each ray meets one x wall first
    ev = o.events()
    reached = {int(p) for k, p in zip(ev["kind"], ev["prim"]) if k in (3, 4)}  # surface hit or back-face kill
    want = {3, 7} if periodic else {3, 5}
    assert reached <= {2, 3, 4, 5, 6, 7} and (reached & {w for w in want} or reached & {w - 1 for w in want})
    # same primitive PAIR as the reference expects (a line's two triangles share the diagonal at z = 0)
    assert {r // 2 for r in reached} == {w // 2 for w in want}, (reached, want)


# ---------------------------------------------------------------------------
# Boundary::processHit known answers of the reference ON THE DEVICE (vr_debug_process_hit)
# ---------------------------------------------------------------------------
def _plane_tracer(extent, delta, direction, radius, bcs, src):
    pts, nrm = vr.io.create_plane_grid(delta, extent, direction)
    t = vr.TraceDisk(3)
    t.setGeometry(pts, nrm, delta, radius)
    t.setBoundaryConditions(bcs)
    t.setSourceDirection(src)
    t.setParticleType(vr.DiffuseParticle(1.0, "flux"))
    return t


def test_boundary_hit_known_answers_on_the_device():
    """tests/boundaryHit/boundaryHit.cpp:68-76,128-136,188-196"""
    eps = 1e-6
    d = np.array([0.5, 0.0, -0.25], dtype=np.float32)
    dist = np.float32(np.linalg.norm(d))
    dn = d / dist
    # reflective x wall, POS_Z: new origin (1, 0.5, 0.25), x component of the direction mirrored
    t = _plane_tracer(1.0, 0.1, (0, 1, 2), 0.1, [BC.REFLECTIVE_BOUNDARY, BC.PERIODIC_BOUNDARY, BC.PERIODIC_BOUNDARY], TD.POS_Z)
    o, dd, refl = t.debugProcessHit([[0.5, 0.5, 0.5]], [dn], dist, 2)
    assert refl[0] and np.allclose(o[0], [1.0, 0.5, 0.25], atol=eps) and np.allclose(dd[0], [-dn[0], dn[1], dn[2]], atol=eps)
    # periodic x wall: the origin wraps to x = -1, direction unchanged
    t = _plane_tracer(1.0, 0.1, (0, 1, 2), 0.1, [BC.PERIODIC_BOUNDARY] * 3, TD.POS_Z)
    o, dd, refl = t.debugProcessHit([[0.5, 0.5, 0.5]], [dn], dist, 2)
    assert refl[0] and np.allclose(o[0], [-1.0, 0.5, 0.25], atol=eps) and np.allclose(dd[0], dn, atol=eps)
    # ignore: the ray stops
    t = _plane_tracer(1.0, 0.1, (0, 1, 2), 0.1, [BC.IGNORE_BOUNDARY] * 3, TD.POS_Z)
    assert not t.debugProcessHit([[0.5, 0.5, 0.5]], [dn], dist, 2)[2][0]
    # reflective z wall, POS_Y tracing, plane in x-z: new origin (0.5, 0.25, 1.0), z mirrored
    d2 = np.array([0.0, -0.25, 0.5], dtype=np.float32)
    dist2 = np.float32(np.linalg.norm(d2))
    dn2 = d2 / dist2
    t = _plane_tracer(1.0, 0.1, (0, 2, 1), 0.1, [BC.PERIODIC_BOUNDARY, BC.PERIODIC_BOUNDARY, BC.REFLECTIVE_BOUNDARY], TD.POS_Y)
    o, dd, refl = t.debugProcessHit([[0.5, 0.5, 0.5]], [dn2], dist2, 6)
    assert refl[0] and np.allclose(o[0], [0.5, 0.25, 1.0], atol=eps) and np.allclose(dd[0], [dn2[0], dn2[1], -dn2[2]], atol=eps)
    # a wall met from outside lets the ray through (rayBoundary.hpp:33-44): origin = hit point, same direction
    o, dd, refl = t.debugProcessHit([[0.5, 0.5, 1.5]], [[dn2[0], dn2[1], -dn2[2]]], dist2, 6)
    assert refl[0] and np.allclose(o[0], [0.5, 0.25, 1.0], atol=eps) and np.allclose(dd[0], [dn2[0], dn2[1], -dn2[2]], atol=eps)


def test_boundary_hit_2d_known_answers_on_the_device():
    """tests/boundaryHit2D/boundaryHit2D.cpp:73-79,122-128,185-190,234-239"""
    eps = 1e-6
    vals = np.arange(-2, 2.0001, 0.5, dtype=np.float32)

    def line(axis, naxis):
        pts = np.zeros((vals.size, 3), dtype=np.float32)
        pts[:, axis] = vals
        nrm = np.zeros_like(pts)
        nrm[:, naxis] = 1.0
        return pts, nrm

    pts, nrm = line(1, 0)  # points along y, normals +x, traced from POS_X
    for bc, ynew in ((BC.REFLECTIVE_BOUNDARY, 2.0), (BC.PERIODIC_BOUNDARY, -2.0)):
        t = vr.TraceDisk(2)
        t.setGeometry(pts, nrm, 0.5, 0.5)
        t.setBoundaryConditions([BC.REFLECTIVE_BOUNDARY, bc])
        t.setSourceDirection(TD.POS_X)
        t.setParticleType(vr.DiffuseParticle(1.0, "flux"))
        d = np.array([-0.5, 1.0, 0.0], dtype=np.float32)
        dist = np.float32(np.linalg.norm(d))
        dn = d / dist
        o, dd, refl = t.debugProcessHit([[1.0, 1.0, 0.0]], [dn], dist, 3)
        assert refl[0] and np.allclose(o[0], [0.5, ynew, 0.0], atol=eps)
        assert np.allclose(dd[0], [dn[0], -dn[1], 0] if bc == BC.REFLECTIVE_BOUNDARY else dn, atol=eps)
    pts, nrm = line(0, 1)  # points along x, normals +y, traced from POS_Y
    for bc, xnew in ((BC.REFLECTIVE_BOUNDARY, 2.0), (BC.PERIODIC_BOUNDARY, -2.0)):
        t = vr.TraceDisk(2)
        t.setGeometry(pts, nrm, 0.5, 0.5)
        t.setBoundaryConditions([bc, BC.REFLECTIVE_BOUNDARY])
        t.setSourceDirection(TD.POS_Y)
        t.setParticleType(vr.DiffuseParticle(1.0, "flux"))
        d = np.array([1.0, -0.5, 0.0], dtype=np.float32)
        dist = np.float32(np.linalg.norm(d))
        o, dd, refl = t.debugProcessHit([[1.0, 1.0, 0.0]], [d / dist], dist, 3)
        assert refl[0] and np.allclose(o[0], [xnew, 0.5, 0.0], atol=eps)


# ---------------------------------------------------------------------------
# multi-GPU behind the C ABI: vr_apply_sharded + the RCCL callback library
# ---------------------------------------------------------------------------
def _shard_tracer():
    gd, p, n = trench3d()
    t = vr.TraceDisk(3)
    t.setGeometry(p, n, gd)
    t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
    t.setParticleType(vr.DiffuseParticle(0.3, "flux"))
    t.setNumberOfRaysFixed(300001)
    t.setRngSeed(5)
    return t


def test_apply_sharded_ranks_sum_to_the_whole():
    """vr_apply_sharded with a collective that leaves the buffers alone: the ranks' accumulators are the
    shards, their sum is the single-device result bit for bit, the callback saw flux and counters"""
    import ctypes as C
    t = _shard_tracer()
    t.apply()
    whole, iw = t.getFluxF64(), info_dict(t)
    calls = []
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    cb = CB(lambda user, ptr, count, stream: calls.append(int(count)) or 0)
    for world in (2, 3):
        parts, infos = [], []
        for rank in range(world):
            t.setRunNumber(1)
            calls.clear()
            t.applySharded(rank, world, cb, None)
            assert calls == [t._n, 80]   # the flux, then the particle's counter block: [0..7] together with the failure word [60]
            parts.append(t.getFluxF64())
            infos.append(info_dict(t))
        assert (sum(parts) == whole).all()
        for k in INFO_KEYS[1:]:
            assert sum(i[k] for i in infos) == iw[k], k
        assert all(i["numRays"] == 300001 for i in infos)


def test_apply_sharded_over_rccl_single_rank():
    """the RCCL path end to end at world size 1 (one GPU here): communicator, in-place int64 all-reduce on
    the library's stream, result identical to apply()"""
    from viennaray_amd import rccl
    t = _shard_tracer()
    t.apply()
    whole, iw = t.getFluxF64(), info_dict(t)
    comm = rccl.Communicator(0, 1)
    try:
        t.setRunNumber(1)
        t.applySharded(0, 1, comm.allreduce, comm.handle)
        assert (t.getFluxF64() == whole).all() and info_dict(t) == iw
        # force the collective even at world 1: call it through a 2-rank shape is impossible on one GPU, so
        # exercise the callback directly on the accumulators (sum over one rank = identity)
        import ctypes as C
        ptr, n = t.fluxAccumulators()
        assert comm.allreduce(comm.handle, C.c_void_p(ptr), C.c_size_t(n), C.c_void_p(t._L.vr_stream(t._h))) == 0
        import torch
        torch.cuda.synchronize()
        assert (t.getFluxF64() == whole).all()
    finally:
        comm.close()


def test_apply_sharded_over_rccl_two_ranks(tmp_path):
    """vr_apply_sharded over RCCL with TWO ranks (one process per GPU): both ranks end with the single-device flux,
    bit for bit.  Needs two GPUs: skipped on the one-GPU boxes of this pool, runs wherever a node offers two."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import subprocess, sys, os, textwrap
    from helpers import ROOT
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent('''
        import os, sys, numpy as np
        sys.path.insert(0, os.environ["VR_ROOT"]); sys.path.insert(0, os.path.join(os.environ["VR_ROOT"], "tests"))
        import viennaray_amd as vr
        from viennaray_amd import rccl
        from helpers import trench3d
        rank, world = int(sys.argv[1]), 2
        gd, p, n = trench3d()
        t = vr.TraceDisk(3, device=rank)
        t.setGeometry(p, n, gd); t.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 3)
        t.setParticleType(vr.DiffuseParticle(0.2, "flux")); t.setNumberOfRaysPerPoint(40); t.setRngSeed(5)
        import time
        idf = os.environ["VR_ID_FILE"]
        if rank == 0:      # the unique id travels through a file (any side channel will do)
            uid = rccl.Communicator.unique_id()
            open(idf + ".tmp", "wb").write(uid); os.rename(idf + ".tmp", idf)
        else:
            while not os.path.exists(idf):
                time.sleep(0.05)
            uid = open(idf, "rb").read()
        comm = rccl.Communicator(rank, world, uid)
        t.applySharded(rank, world, comm.allreduce, comm.handle)
        np.save(os.environ["VR_OUT"] + str(rank) + ".npy", t.getFluxF64())
        comm.close()
    '''))
    env = dict(os.environ, VR_ROOT=ROOT, VR_ID_FILE=str(tmp_path / "id.bin"), VR_OUT=str(tmp_path / "flux"))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], env=env) for r in range(2)]
    assert all(q.wait(timeout=600) == 0 for q in procs)
    gd, p, n = trench3d()
    t = vr.TraceDisk(3)
    t.setGeometry(p, n, gd)
    t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
    t.setParticleType(vr.DiffuseParticle(0.2, "flux"))
    t.setNumberOfRaysPerPoint(40)
    t.setRngSeed(5)
    t.apply()
    whole = t.getFluxF64()
    for r in range(2):
        assert (np.load(str(tmp_path / "flux") + f"{r}.npy") == whole).all()


def test_accumulator_overflow_is_detected_not_wrapped():
    """The flux accumulators are int64 fixed point (2^-40 per unit): 2^23 = 8.39e6 weight units per primitive and label in
    one apply() (signed, so that the multi-GPU all-reduce cannot wrap either).  The reference's float sums stall near 2^24
    (rayParticle.hpp:148-156, rayTraceKernel.hpp:348-360); these would WRAP — so the apply fails with a message instead.
    A 4-disk plane: 5e7 absorbing rays put ~3e7 units on every disk -> error; 1e7 rays stay below the limit and are
    bit-exact against the oracle; with room left for 4 ranks' sum (vr_set_world_size) the same 1e7 rays are refused."""
    pts, nrm = vr.io.plane_grid(2, 1.0)

    def tracer(rays):
        t = vr.TraceDisk(3)
        t.setGeometry(pts, nrm, 1.0)
        t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
        t.setParticleType(vr.DiffuseParticle(1.0, "flux"))
        t.setNumberOfRaysFixed(rays)
        t.setRngSeed(5)
        return t

    t = tracer(50_000_000)
    with pytest.raises(vr.VrError, match="flux accumulator overflow"):
        t.apply()
    assert t.getRayTraceInfo().error == 1
    assert t.getRunNumber() == 2            # (the apply counted: the next one uses the next seed, like after any apply)
    t = tracer(10_000_000)
    t.apply()
    f = t.getFluxF64()
    assert f.max() < 2.0 ** 23 and f.max() > 2.0 ** 22   # just below the limit
    o = po.Oracle()
    o.set_disks(pts, nrm, 1.0, 3)
    o.set_boundary_conditions([po.PERIODIC] * 3)
    o.set_particle(po.DIFFUSE, 1.0)
    o.set_num_rays_fixed(10_000_000)
    o.set_rng_seed(5)
    o.set_lazy_rng(True)
    o.apply(po.max_threads())
    assert (f == o.flux().astype(np.float64)).all()       # integers below 2^24: the oracle's float sums are exact too
    assert info_dict(t)["geometryHits"] == o.info()["geometryHits"]
    t = tracer(10_000_000)
    t.setWorldSize(4)
    with pytest.raises(vr.VrError, match="flux accumulator overflow"):
        t.apply()

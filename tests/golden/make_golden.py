#!/usr/bin/env python3
"""Generates tests/golden/vectors_r01.npz — SURVEY.md §8c golden vectors G1, G2, G3 (closest
hits), G4.

The reference cannot be built here (Embree / ViennaCore are absent), so these
vectors are outputs of the CPU ORACLE (oracle/vr_oracle.cpp, single thread), frozen: they pin
today's behaviour of the restatement, so that a later change of the oracle or of the HIP
path shows up as a diff instead of silently moving both.  The engine words additionally come
from libstdc++'s own std::mt19937_64 (the oracle's non-lazy path), an implementation this
repo did not write.

    python tests/golden/make_golden.py        # rewrites the .npz next to this file
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle as po  # noqa: E402
from helpers import sphere3d, trench2d, trench3d, trench_mesh  # noqa: E402
from viennaray_amd import io  # noqa: E402

RNG_PAIRS = [(0, 12346), (1, 12346), (99_999_999, 12346), (123456789, 7), (2**32 - 1, 0xFFFFFFFF)]
SRC_CASES = [  # (D, direction, power)
    (3, po.POS_Z, 1.0), (3, po.NEG_Z, 1.0), (3, po.POS_X, 50.0), (3, po.NEG_X, 1.0), (3, po.POS_Y, 50.0),
    (3, po.NEG_Y, 1.0), (2, po.POS_Y, 1.0), (2, po.NEG_X, 50.0),
]
N_SRC = 256
N_HIT = 512


def hit_rays(lo, hi, D, n, seed):
    rng = np.random.default_rng(seed)
    org = rng.uniform(lo - 0.5, hi + 0.5, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    if D == 2:
        org[:, 2] = 0
        d[:, 2] = 0
    return org, (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)


def c1_oracle(seed):
    """C1: P(100), DiffuseParticle sticking 0.1, PERIODIC, 1e6 rays fixed."""
    pts, nrm = io.plane_grid(100, 1.0)
    o = po.Oracle()
    o.set_disks(pts, nrm, 1.0, 3)
    o.set_boundary_conditions([po.PERIODIC] * 3)
    o.set_particle(po.DIFFUSE, 0.1)
    o.set_num_rays_fixed(1_000_000)
    o.set_rng_seed(seed)
    o.apply(1)
    info = o.info()
    keys = ("numRays", "totalRaysTraced", "nonGeometryHits", "geometryHits", "boundaryHits", "reflections",
            "raysTerminated")
    return o.flux().astype(np.float32), np.array([info[k] for k in keys], dtype=np.int64)


def build():
    out = {}
    # G1: tea<3> and the first 16 engine words
    out["g1_pairs"] = np.array(RNG_PAIRS, dtype=np.uint64)
    out["g1_tea3"] = np.array([po.tea3(i, s) for i, s in RNG_PAIRS], dtype=np.uint32)
    out["g1_mt64"] = np.stack([po.mt64_outputs(int(t), 16) for t in out["g1_tea3"]])
    # G2: first N_SRC (origin, direction) pairs, kernel seed 12346
    gd3, p3, n3 = sphere3d()
    gd2, p2, n2 = trench2d()
    for ci, (D, direction, power) in enumerate(SRC_CASES):
        o = po.Oracle()
        if D == 3:
            o.set_disks(p3, n3, gd3, 3)
        else:
            o.set_disks(p2, n2, gd2, 2)
        o.set_source_direction(direction)
        o.set_particle(po.SPECULAR, 1.0, power)
        o.prepare()
        org = np.empty((N_SRC, 3), np.float32)
        d = np.empty((N_SRC, 3), np.float32)
        for i in range(N_SRC):
            org[i], d[i] = o.source_sample(i, 12346)
        out[f"g2_org_{ci}"] = org
        out[f"g2_dir_{ci}"] = d
    out["g2_cases"] = np.array(SRC_CASES, dtype=np.float64)
    # G3 (closest hits): geomID / primID / t of N_HIT random rays
    for name in ("trench3d", "mesh", "trench2d"):
        o = po.Oracle()
        if name == "mesh":
            gd, v, tri = trench_mesh()
            o.set_triangles(v, tri, gd, 3)
            lo, hi, D = v.min(0), v.max(0), 3
        else:
            gd, p, n = (trench3d if name == "trench3d" else trench2d)()
            D = 2 if name == "trench2d" else 3
            o.set_disks(p, n, gd, D)
            if D == 2:
                o.set_source_direction(po.POS_Y)
            lo, hi = p.min(0), p.max(0)
        o.prepare()
        org, d = hit_rays(lo, hi, D, N_HIT, 17)
        g = np.empty(N_HIT, np.int32)
        prim = np.zeros(N_HIT, np.uint32)
        t = np.zeros(N_HIT, np.float32)
        for i in range(N_HIT):
            h = o.intersect1(org[i], d[i])
            g[i] = h["geomID"]
            if g[i] >= 0:
                prim[i], t[i] = h["primID"], h["t"]
        out[f"g3_{name}_org"], out[f"g3_{name}_dir"] = org, d
        out[f"g3_{name}_geom"], out[f"g3_{name}_prim"], out[f"g3_{name}_t"] = g, prim, t
    # G4: raw flux + counters of C1, seeds 1 and 12345
    for seed in (1, 12345):
        f, c = c1_oracle(seed)
        out[f"g4_flux_{seed}"], out[f"g4_info_{seed}"] = f, c
    return out


if __name__ == "__main__":
    vec = build()
    path = os.path.join(HERE, "vectors_r01.npz")
    np.savez_compressed(path, **vec)
    print("wrote", path, os.path.getsize(path), "bytes,", len(vec), "arrays")

#!/usr/bin/env python3
"""Closest hit of the ordered (pair-node, stack) walk against the escape-link walk on millions of rays of the
kinds the tracer produces (primary rays, bounced rays starting ON the surface, mirrored and axis-parallel
directions, origins on lattice points).   usage: tools/walk_fuzz.py <mesh|trench3d|trench2d|sphere> [rays]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import viennaray_amd as vr
from helpers import trench3d, trench2d, trench_mesh, sphere3d

case = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000

D = 3
if case == "mesh":
    gd, v, tri = trench_mesh(); t = vr.TraceTriangle(3); t.setGeometry(v, tri, gd); lo, hi = v.min(0), v.max(0)
else:
    gd, p, nrm = {"trench3d": trench3d, "trench2d": trench2d, "sphere": sphere3d}[case]()
    D = 2 if case == "trench2d" else 3
    t = vr.TraceDisk(D); t.setGeometry(p, nrm, gd); lo, hi = p.min(0), p.max(0)
    if D == 2: t.setSourceDirection(vr.TraceDirection.POS_Y)
t.setParticleType(vr.DiffuseParticle(0.1, "f"))
rng = np.random.default_rng(7)
up = 1 if D == 2 else 2

def both(o, d, tag):
    os.environ["VR_DEBUG_WALK"] = "0"; g0, p0, t0 = t.debugIntersect(o, d)
    os.environ["VR_DEBUG_WALK"] = "1"; g1, p1, t1 = t.debugIntersect(o, d)
    bad = np.nonzero((g0 != g1) | ((g0 >= 0) & ((p0 != p1) | (t0 != t1))))[0]
    print(f"{case} {tag}: rays {len(o)} geometry hits {int((g0 == 1).sum())} mismatches {len(bad)}")
    for i in bad[:8]:
        print("   ray", i, "o", o[i].tolist(), "d", d[i].tolist(), "escape-link walk", (g0[i], p0[i], t0[i]), "ordered walk", (g1[i], p1[i], t1[i]))
    return g0, p0, t0

o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32); o[:, up] = hi[up] + gd
d = rng.normal(size=(n, 3)) * 0.3; d[:, up] = -1.0
if D == 2: o[:, 2] = 0; d[:, 2] = 0
d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
g, pr, tt = both(o, d, "primary")
m = g == 1
o2 = (o[m] + d[m] * tt[m, None]).astype(np.float32)
d2 = rng.normal(size=o2.shape)
if D == 2: d2[:, 2] = 0
d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
both(o2, d2, "bounced, random direction")
d3 = d[m].copy(); d3[:, up] = -d3[:, up]
both(o2, d3, "bounced, mirrored on the horizontal")
d4 = d[m].copy(); d4[:, 0] = -d4[:, 0]
g4, _, t4 = both(o2, d4, "bounced, mirrored on x")
m4 = g4 == 1
o5 = (o2[m4] + d4[m4] * t4[m4, None]).astype(np.float32); d5 = d4[m4].copy(); d5[:, 0] = -d5[:, 0]
both(o5, d5, "third segment, mirrored back")
for ax in range(D):
    d6 = np.zeros_like(o2); d6[:, ax] = rng.choice([-1.0, 1.0], size=len(o2))
    both(o2, d6.astype(np.float32), f"bounced, along axis {ax}")
# origins snapped onto multiples of the grid spacing (cell planes, shared edges)
o7 = (np.round(o2 / gd) * gd).astype(np.float32)
both(o7, d2, "origins on lattice points, random direction")

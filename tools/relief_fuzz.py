#!/usr/bin/env python3
"""Random sheets of disks that are flat with relief (sums of ripples, tilts, terraces, holes; random cell size, walls,
source side, particle) through the HIP path and the CPU oracle: every counter equal, flux as tests/test_gpu_parity.py
compares it.  The fixed cases of the test suite cover four kinds of relief; this walks the space between them.
usage (GPU box): tools/relief_fuzz.py [seconds=300] [seed=1]        exit code 1 on the first mismatch (the case is printed)"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (one HIP runtime per process: before the tracing library)
import viennaray_amd as vr
from tests.test_gpu_parity import make_pair_disks, compare

BC, TD = vr.BoundaryCondition, vr.TraceDirection
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
modes, cases = {}, 0
while time.time() - t0 < budget:
    n = int(rng.integers(50, 180))
    gd = float(rng.choice([0.25, 0.5, 1.0, 2.0]))
    ax = (np.arange(n) - (n - 1) / 2.0) * gd
    x, y = np.meshgrid(ax, ax, indexing="ij")
    z = np.zeros_like(x)
    gx, gy = np.zeros_like(x), np.zeros_like(x)
    desc = []
    for _ in range(int(rng.integers(0, 4))):  # ripples
        amp, wave, ph = float(rng.uniform(0.05, 1.2)) * gd, float(rng.uniform(1.5, 12.0)) * gd, float(rng.uniform(0, 6.28))
        th = float(rng.uniform(0, 3.1416))
        u = (x * np.cos(th) + y * np.sin(th)) / wave + ph
        z += amp * np.sin(u)
        gx += amp / wave * np.cos(u) * np.cos(th)
        gy += amp / wave * np.cos(u) * np.sin(th)
        desc.append(f"ripple(a={amp / gd:.2f},w={wave / gd:.1f})")
    if rng.random() < 0.3:  # a tilt of up to three cells over the sheet
        sl = float(rng.uniform(-3, 3)) * gd / (n * gd)
        z += sl * x
        gx += sl
        desc.append(f"tilt({sl * n:.2f})")
    if rng.random() < 0.3:  # terraces without risers
        h, w = float(rng.uniform(0.3, 2.0)) * gd, float(rng.uniform(4, 20)) * gd
        z += h * (np.floor((y - ax[0]) / w) % 2)
        desc.append(f"steps(h={h / gd:.2f},w={w / gd:.1f})")
    nrm = np.stack([-gx, -gy, np.ones_like(x)], -1).reshape(-1, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    pts = np.stack([x, y, z], -1).reshape(-1, 3)
    if rng.random() < 0.25:  # holes: a tenth of the disks missing
        keep = rng.random(len(pts)) > 0.1
        pts, nrm = pts[keep], nrm[keep]
        desc.append("holes")
    r = rng.random()
    direction, side = TD.POS_Z, ""
    if r < 0.2:    # the source below the sheet: back faces first
        direction, side = TD.NEG_Z, "NEG_Z "
    elif r < 0.35:  # the sheet stood up, its height along x
        pts, nrm = np.ascontiguousarray(pts[:, [2, 0, 1]]), np.ascontiguousarray(nrm[:, [2, 0, 1]])
        direction, side = TD.POS_X, "POS_X "
    bcs = [BC(int(rng.choice([int(BC.PERIODIC_BOUNDARY), int(BC.REFLECTIVE_BOUNDARY)]))) for _ in range(3)]
    kind = "diffuse" if rng.random() < 0.7 else "specular"
    sticking = float(rng.choice([1.0, 0.5, 0.1, 0.02]))
    particle = (kind, sticking, float(rng.choice([1.0, 8.0, 50.0])))
    seed = int(rng.integers(1, 1 << 30))
    rays_pp = int(rng.integers(8, 40))
    label = f"n={n} gd={gd} {' + '.join(desc) or 'flat'} {side}bc={[int(b) for b in bcs]} {particle} seed={seed} rays/pt={rays_pp}"
    t, o = make_pair_disks(pts.astype(np.float32), nrm.astype(np.float32), gd, 3, bcs, direction, particle, rays_pp=rays_pp, seed=seed)
    try:
        err, info = compare(t, o, exact_flux=(sticking >= 1.0))
    except AssertionError as e:
        print("MISMATCH", label, "mode", t.traceMode(), e, flush=True)
        sys.exit(1)
    modes[t.traceMode()] = modes.get(t.traceMode(), 0) + 1
    cases += 1
    print(f"ok mode {t.traceMode()} err {err:.1e} rays {info['totalRaysTraced']} {label}", flush=True)
print(f"{cases} random sheets equal to the oracle; kernel modes used {dict(sorted(modes.items()))}")

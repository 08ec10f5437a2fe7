#!/usr/bin/env python3
"""Parity of a rippled / bumped sheet against the CPU oracle (counters exact, flux), the way tests/test_gpu_parity.py
compares — a quick look while the relief kernels (trace_kernel MODE 5 / 6) are being worked on.
usage: tools/relief_check.py [n=300] [rays=3000000] [amp=0.5]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import viennaray_amd as vr
from oracle import pyoracle as po

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 3_000_000
amp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
wave = 4.0
ax = np.arange(n) - (n - 1) / 2.0
x, y = np.meshgrid(ax, ax, indexing="ij")
z = amp * np.sin(x / wave) * np.cos(y / wave)
nrm = np.stack([-amp / wave * np.cos(x / wave) * np.cos(y / wave), amp / wave * np.sin(x / wave) * np.sin(y / wave), np.ones_like(x)], -1).reshape(-1, 3)
nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
pts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
nrm = nrm.astype(np.float32)
KEYS = ("totalRaysTraced", "nonGeometryHits", "geometryHits", "boundaryHits", "reflections", "raysTerminated")
bad = 0
for bc in (vr.BoundaryCondition.PERIODIC_BOUNDARY, vr.BoundaryCondition.REFLECTIVE_BOUNDARY):
    for s in (1.0, 0.1):
        t = vr.TraceDisk(3)
        t.setGeometry(pts, nrm, 1.0)
        t.setBoundaryConditions([bc] * 3)
        t.setParticleType(vr.DiffuseParticle(s, "flux"))
        t.setNumberOfRaysFixed(rays)
        t.setRngSeed(12345)
        t.apply()
        gi = t.getRayTraceInfo()
        f = t.getLocalData().getVectorData(0).astype(np.float64)
        o = po.Oracle()
        o.set_disks(pts, nrm, 1.0, 3)
        o.set_boundary_conditions([int(bc)] * 3)
        o.set_particle(po.DIFFUSE, s)
        o.set_num_rays_fixed(rays)
        o.set_rng_seed(12345)
        o.set_lazy_rng(True)
        o.apply(min(po.max_threads(), 16))
        oi = o.info()
        r = o.flux().astype(np.float64)
        diff = {k: int(getattr(gi, k)) - oi[k] for k in KEYS}
        err = float(np.linalg.norm(f - r) / np.linalg.norm(r))
        ok = all(v == 0 for v in diff.values()) and err <= 5e-6
        bad += 0 if ok else 1
        print(f"bc {int(bc)} sticking {s}: mode {t.traceMode()} flux L2 {err:.3e} counters {diff} trace {gi.timeTraceKernel * 1e3:.3f} ms {'OK' if ok else 'MISMATCH'}")
sys.exit(1 if bad else 0)

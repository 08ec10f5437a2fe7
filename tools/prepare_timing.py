#!/usr/bin/env python3
"""Times the per-geometry setup (setGeometry + applyPrepare) the way a level-set loop pays it:
new point cloud every step.   usage: tools/prepare_timing.py [grid]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import viennaray_amd as vr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pts, nrm = vr.io.plane_grid(n, 1.0)
t = vr.TraceDisk(3)
t.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 3)
t.setParticleType(vr.DiffuseParticle(1.0, "flux"))
t.setNumberOfRaysFixed(1000)
t.setRngSeed(1)
for it in range(4):
    p = pts.copy()
    p[:, 2] += 0.01 * it  # "moved surface"
    t0 = time.perf_counter()
    t.setGeometry(p, nrm, 1.0)
    t1 = time.perf_counter()
    t.applyPrepare()
    t2 = time.perf_counter()
    t.applyLaunch(); t.applyFinish()
    t3 = time.perf_counter()
    print(f"step {it}: setGeometry {1e3*(t1-t0):.2f} ms  applyPrepare {1e3*(t2-t1):.2f} ms  launch+finish(1000 rays) {1e3*(t3-t2):.2f} ms")

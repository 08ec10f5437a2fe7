#!/usr/bin/env python3
"""Times the post-processing every reference example runs after apply(): normalizeFlux + smoothFlux."""
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
t.setNumberOfRaysFixed(10_000_000)
t.setRngSeed(1)
for it in range(3):
    p = pts.copy(); p[:, 2] += 0.01 * it
    t.setGeometry(p, nrm, 1.0)
    t0 = time.perf_counter(); t.apply(); t1 = time.perf_counter()
    f = t.getLocalData().getVectorData(0)
    fn = t.normalizeFlux(f); t2 = time.perf_counter()
    fs = t.smoothFlux(fn, 1); t3 = time.perf_counter()
    print(f"step {it}: apply {1e3*(t1-t0):.2f} ms  normalizeFlux {1e3*(t2-t1):.2f} ms  smoothFlux {1e3*(t3-t2):.2f} ms  (mean {float(np.mean(fs)):.4f})")

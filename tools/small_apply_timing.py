#!/usr/bin/env python3
"""Wall time of whole apply() calls on small scenes (what a 2-D level-set loop pays per step)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import viennaray_amd as vr
from helpers import trench2d, sphere3d

for name, (gd, p, n), D, rpp in (("trench2d 239 disks x 2000 rays", trench2d(), 2, 2000),
                                  ("sphere 162 disks x 1000 rays", sphere3d(), 3, 1000)):
    t = vr.TraceDisk(D)
    t.setGeometry(p, n, gd)
    if D == 2:
        t.setSourceDirection(vr.TraceDirection.POS_Y)
    t.setParticleType(vr.DiffuseParticle(0.1, "flux"))
    t.setNumberOfRaysPerPoint(rpp)
    t.setRngSeed(1)
    t.apply()
    ts = []
    for it in range(20):
        t0 = time.perf_counter(); t.apply(); ts.append(time.perf_counter() - t0)
    info = t.getRayTraceInfo()
    print(f"{name}: apply() median {1e3*np.median(ts):.3f} ms (device {1e3*info.timeTrace:.3f} ms, trace kernel {1e3*info.timeTraceKernel:.3f} ms)")
    ts = []
    for it in range(10):
        q = p.copy(); q[:, 1] += 1e-3 * it
        t0 = time.perf_counter(); t.setGeometry(q, n, gd); t.apply(); ts.append(time.perf_counter() - t0)
    print(f"{name}: setGeometry + apply() median {1e3*np.median(ts):.3f} ms")

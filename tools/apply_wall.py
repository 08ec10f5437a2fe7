#!/usr/bin/env python3
"""Host time of apply() against its device time when the ray count changes from call to call
(the reference's use case: one apply() per time step of a surface that moves).
Prints, per call: rays, prepare / launch / finish / collect wall ms, the device pipeline ms, and the rest."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import viennaray_amd as vr
from helpers import trench3d

gd, p, n = trench3d()
t = vr.TraceDisk(3)
t.setGeometry(p, n, gd)
t.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 3)
t.setParticleType(vr.DiffuseParticle(0.1, "flux"))
t.setRngSeed(12345)
if len(sys.argv) > 1:
    t.reserveRays(int(float(sys.argv[1])))
for rays in (1_000_000, 1_000_000, 50_000_000, 50_000_000, 57_838_000, 57_838_000, 70_000_000, 70_000_000, 2_000_000, 57_838_000):
    t.setNumberOfRaysFixed(rays)
    t.setRunNumber(1)
    t0 = time.perf_counter(); t.applyPrepare()
    t1 = time.perf_counter(); t.applyLaunch()
    t2 = time.perf_counter(); t.applyFinish(collect=False)
    t3 = time.perf_counter(); t._collect()
    t4 = time.perf_counter()
    dev = t.getRayTraceInfo().timeTrace * 1e3
    w = [(b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4))]
    print(f"rays {rays:>10}: prepare {w[0]:8.3f} launch {w[1]:7.3f} finish {w[2]:8.3f} collect {w[3]:6.3f} | "
          f"wall {sum(w):8.3f} device {dev:8.3f} host-only {sum(w) - dev:8.3f} ms")

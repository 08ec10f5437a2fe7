#!/usr/bin/env python3
"""One apply() over a particle LIST against one apply() per particle (C2 plane, 1e8 rays): what sharing the
generator pass between particles of the same source distribution saves.
usage: tools/multi_particle_bench.py [rays]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import viennaray_amd as vr

rays = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
pts, nrm = vr.io.plane_grid(1000, 1.0)
t = vr.TraceDisk(3)
t.setGeometry(pts, nrm, 1.0)
t.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 3)
t.setNumberOfRaysFixed(rays)
t.setRngSeed(12345)
plist = [vr.DiffuseParticle(0.1, "a"), vr.DiffuseParticle(0.3, "b"), vr.DiffuseCosineParticle(0.2, "c", "ccos")]
for rep in range(3):
    tot = 0.0
    for q in plist:
        t.setParticleType(q)
        t.setRunNumber(1)
        t.apply()
        i = t.getRayTraceInfo()
        tot += i.timeTrace
    t.setParticleTypes(plist)
    t.setRunNumber(1)
    t0 = time.perf_counter()
    t.apply()
    wall = time.perf_counter() - t0
    i = t.getRayTraceInfo()
    print(f"rep {rep}: three applies {tot * 1e3:.2f} ms device; one apply over the list {i.timeTrace * 1e3:.2f} ms device "
          f"(gen {i.timeGenKernel * 1e3:.2f}, trace kernels {i.timeTraceKernel * 1e3:.2f}), wall {wall * 1e3:.2f} ms")

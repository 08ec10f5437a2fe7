#!/usr/bin/env python3
"""Times one apply() of a named fixture geometry on the GPU.
usage: tools/case_bench.py <trench3d|trench2d|mesh|plane100> <sticking> <raysPerPoint> [repeat]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import viennaray_amd as vr
from helpers import trench3d, trench2d, trench_mesh

case, sticking, rpp = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
rep = int(sys.argv[4]) if len(sys.argv) > 4 else 3
if case == "mesh":
    gd, v, tri = trench_mesh()
    t = vr.TraceTriangle(3); t.setGeometry(v, tri, gd)
elif case == "trench2d":
    gd, p, n = trench2d()
    t = vr.TraceDisk(2); t.setGeometry(p, n, gd); t.setSourceDirection(vr.TraceDirection.POS_Y)
    t.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 2)
elif case == "plane100":
    p, n = vr.io.plane_grid(100, 1.0)
    t = vr.TraceDisk(3); t.setGeometry(p, n, 1.0); t.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 3)
else:
    gd, p, n = trench3d()
    t = vr.TraceDisk(3); t.setGeometry(p, n, gd); t.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 3)
t.setParticleType(vr.DiffuseParticle(sticking, "flux"))
t.setNumberOfRaysPerPoint(rpp)
t.setRngSeed(12345)
for i in range(rep):
    t.setRunNumber(1)
    t.apply()
    info = t.getRayTraceInfo()
    print(f"{case} sticking {sticking}: rays {info.numRays} segments {info.totalRaysTraced} device {info.timeTrace*1e3:.2f} ms "
          f"trace_kernel {info.timeTraceKernel*1e3:.2f} ms -> {info.numRays/info.timeTrace/1e6:.0f} Mrays/s")

#!/usr/bin/env python3
"""Times one apply() of a named fixture geometry on the GPU.
usage: tools/case_bench.py <trench3d|trench2d|mesh|plane100|ripple<n>[a<amp>]> <sticking> <raysPerPoint> [repeat]
       tools/case_bench.py <C4|C5p|C5r> [repeat]      (SURVEY.md 8d configs, 1e8 rays)"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import viennaray_amd as vr
from helpers import trench3d, trench2d, trench_mesh

case = sys.argv[1]
fixed = None
particle = None
if case in ("C4", "C5p", "C5r"):
    rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    sticking, rpp, fixed = 0.1, 0, 100_000_000
else:
    sticking, rpp = float(sys.argv[2]), int(sys.argv[3])
    rep = int(sys.argv[4]) if len(sys.argv) > 4 else 3
if case == "C4":   # trenchMesh.dat, SpecularParticle(0.1, power 50), default REFLECTIVE walls
    gd, v, tri = trench_mesh()
    t = vr.TraceTriangle(3); t.setGeometry(v, tri, gd)
    particle = vr.SpecularParticle(0.1, 50.0, "flux")
elif case in ("C5p", "C5r"):   # trenchGrid2D.dat, D=2, POS_Y, diffuse 0.1, periodic / reflective in x
    gd, p, n = trench2d()
    t = vr.TraceDisk(2); t.setGeometry(p, n, gd); t.setSourceDirection(vr.TraceDirection.POS_Y)
    bc = vr.BoundaryCondition.PERIODIC_BOUNDARY if case == "C5p" else vr.BoundaryCondition.REFLECTIVE_BOUNDARY
    t.setBoundaryConditions([bc] * 2)
elif case == "mesh":
    gd, v, tri = trench_mesh()
    t = vr.TraceTriangle(3); t.setGeometry(v, tri, gd)
elif case == "trench2d":
    gd, p, n = trench2d()
    t = vr.TraceDisk(2); t.setGeometry(p, n, gd); t.setSourceDirection(vr.TraceDirection.POS_Y)
    t.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 2)
elif case.startswith("ripple"):   # ripple<n>[a<amp>]: an n x n rippled sheet of disks (amplitude in grid cells, default 1)
    import re
    m = re.match(r"ripple(\d+)(?:a([0-9.]+))?(?:p([0-9.]+))?$", case)   # p<f>: only a central patch of side f * n is rippled
    n_, amp, patch = int(m.group(1)), float(m.group(2) or 1.0), float(m.group(3) or 1.0)
    ax = (np.arange(n_) - (n_ - 1) / 2.0)
    x, y = np.meshgrid(ax, ax, indexing="ij")
    wave = 4.0
    inside = ((np.abs(x) <= patch * n_ / 2) & (np.abs(y) <= patch * n_ / 2)).astype(np.float64)
    z = inside * amp * np.sin(x / wave) * np.cos(y / wave)
    nrm = np.stack([-inside * amp / wave * np.cos(x / wave) * np.cos(y / wave), inside * amp / wave * np.sin(x / wave) * np.sin(y / wave),
                    np.ones_like(x)], -1).reshape(-1, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    p = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    t = vr.TraceDisk(3); t.setGeometry(p, nrm.astype(np.float32), 1.0)
    t.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 3)
elif case == "plane100":
    p, n = vr.io.plane_grid(100, 1.0)
    t = vr.TraceDisk(3); t.setGeometry(p, n, 1.0); t.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 3)
else:
    gd, p, n = trench3d()
    t = vr.TraceDisk(3); t.setGeometry(p, n, gd); t.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 3)
# VR_CASE_PARTICLE=coned|cosine2: the same workload through the extended kernel (device particle registry)
_pk = os.environ.get("VR_CASE_PARTICLE", "")
if _pk == "coned":
    particle = vr.ConedCosineParticle(sticking, 1.0, 0.8, "flux")
elif _pk == "cosine2":
    particle = vr.DiffuseCosineParticle(sticking, "flux", "cosine")
t.setParticleType(particle if particle is not None else vr.DiffuseParticle(sticking, "flux"))
if fixed:
    t.setNumberOfRaysFixed(fixed)
else:
    t.setNumberOfRaysPerPoint(rpp)
t.setRngSeed(12345)
import json
for i in range(rep):
    t.setRunNumber(1)
    t.apply()
    info = t.getRayTraceInfo()
    # (the JSON line is what tools/pmc_profile.py reads: rays and trace segments of one launch)
    print(json.dumps(dict(vr_case=case, sticking=sticking, rays=int(info.numRays), segments=int(info.totalRaysTraced),
                          device_ms=info.timeTrace * 1e3, trace_kernel_ms=info.timeTraceKernel * 1e3,
                          gen_kernel_ms=info.timeGenKernel * 1e3, mode=t.traceMode())))
    print(f"{case} sticking {sticking}: rays {info.numRays} segments {info.totalRaysTraced} device {info.timeTrace*1e3:.2f} ms "
          f"trace_kernel {info.timeTraceKernel*1e3:.2f} ms -> {info.numRays/info.timeTrace/1e6:.0f} Mrays/s")

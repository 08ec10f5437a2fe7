#!/usr/bin/env python3
"""Full-size parity of one fixture workload against the CPU oracle (counters equal, flux L2-relative error):
    python3 tools/full_parity_case.py trench3d|mesh|trench2d|rippled [raysPerPoint] [sticking]      (one JSON line)
`rippled`: C2's 10^6-disk plane rippled by half a grid cell (bench.py's C2_rippled: the relief packets), default 100 rays per
point = 10^8 rays"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import viennaray_amd as vr
from oracle import pyoracle as po
from helpers import trench3d, trench2d, trench_mesh, l2_rel

case = sys.argv[1]
rpp = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
BC = vr.BoundaryCondition
o = po.Oracle()
if case == "mesh":
    gd, v, tri = trench_mesh()
    t = vr.TraceTriangle(3); t.setGeometry(v, tri, gd); o.set_triangles(v, tri, gd, 3)
    t.setParticleType(vr.SpecularParticle(0.1, 50.0, "flux")); o.set_particle(po.SPECULAR, 0.1, 50.0)
elif case == "trench2d":
    gd, p, n = trench2d()
    t = vr.TraceDisk(2); t.setGeometry(p, n, gd); o.set_disks(p, n, gd, 2)
    t.setSourceDirection(vr.TraceDirection.POS_Y); o.set_source_direction(po.POS_Y)
    t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 2); o.set_boundary_conditions([po.PERIODIC] * 2)
    t.setParticleType(vr.DiffuseParticle(0.1, "flux")); o.set_particle(po.DIFFUSE, 0.1)
elif case == "rippled":
    sticking = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    if len(sys.argv) <= 2:
        rpp = 100
    p, n = vr.io.plane_grid(1000, 1.0)
    x, y = p[:, 0].astype(np.float64), p[:, 1].astype(np.float64)
    p = p.copy(); p[:, 2] = (0.5 * np.sin(x / 4.0) * np.cos(y / 4.0)).astype(np.float32)
    nv = np.stack([-0.125 * np.cos(x / 4.0) * np.cos(y / 4.0), 0.125 * np.sin(x / 4.0) * np.sin(y / 4.0), np.ones_like(x)], -1)
    n = (nv / np.linalg.norm(nv, axis=1, keepdims=True)).astype(np.float32)
    gd = 1.0
    t = vr.TraceDisk(3); t.setGeometry(p, n, gd); o.set_disks(p, n, gd, 3)
    t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3); o.set_boundary_conditions([po.PERIODIC] * 3)
    t.setParticleType(vr.DiffuseParticle(sticking, "flux")); o.set_particle(po.DIFFUSE, sticking)
    case = f"rippled_s{sticking}"
else:
    gd, p, n = trench3d()
    t = vr.TraceDisk(3); t.setGeometry(p, n, gd); o.set_disks(p, n, gd, 3)
    t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3); o.set_boundary_conditions([po.PERIODIC] * 3)
    t.setParticleType(vr.DiffuseParticle(0.1, "flux")); o.set_particle(po.DIFFUSE, 0.1)
t.setNumberOfRaysPerPoint(rpp); o.set_num_rays_per_point(rpp)
t.setRngSeed(12345); o.set_rng_seed(12345)
o.set_lazy_rng(True)
t.apply()
t0 = time.time(); o.apply(po.max_threads()); cpu_s = time.time() - t0
gi = t.getRayTraceInfo(); oi = o.info()
keys = ("numRays", "totalRaysTraced", "nonGeometryHits", "geometryHits", "particleHits", "boundaryHits", "reflections", "raysTerminated")
diff = {k: int(getattr(gi, k)) - int(oi[k]) for k in keys}
f = t.getLocalData().getVectorData(0); r = o.flux()
print(json.dumps(dict(case=case, rays=int(gi.numRays), segments=int(gi.totalRaysTraced), kernel_mode=t.traceMode(),
                      flux_l2_rel_err=float(l2_rel(f, r)), counter_diff=diff, oracle_seconds=round(cpu_s, 1),
                      trace_kernel_ms=round(gi.timeTraceKernel * 1e3, 3))))

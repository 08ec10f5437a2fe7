"""Trace-kernel time of small launches on P(100) against the grid size (VR_TRACE_BLOCKS): python3 tools/small_launch.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import viennaray_amd as vr
p, n = vr.io.plane_grid(100, 1.0)
for rays in (100000, 300000, 600000, 1000000, 2000000):
    for st in (1.0, 0.1):
        row = []
        for blocks in (0, 1, 2, 3, 4, 6):
            if blocks:
                os.environ["VR_TRACE_BLOCKS"] = str(blocks)
            else:
                os.environ.pop("VR_TRACE_BLOCKS", None)
            t = vr.TraceDisk(3); t.setGeometry(p, n, 1.0); t.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 3)
            t.setParticleType(vr.DiffuseParticle(st, "flux")); t.setNumberOfRaysFixed(rays); t.setRngSeed(1)
            best = None
            for i in range(4):
                t.setRunNumber(1); t.apply(); k = t.getRayTraceInfo().timeTraceKernel * 1e3
                best = k if best is None or k < best else best
            row.append("%s:%.3f" % (blocks or "auto", best))
        print(rays, st, " ".join(row))

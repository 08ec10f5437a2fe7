#!/usr/bin/env python3
"""Host-side cost of one bench step (C2): wall time of each call of a warmed step against the device time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import viennaray_amd as vr
import viennaray_amd.distributed as vd
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
pts, nrm = vr.io.plane_grid(1000, 1.0)
tr = vr.TraceDisk(3, device=0); tr.setGeometry(pts, nrm, 1.0)
tr.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 3)
tr.setParticleType(vr.DiffuseParticle(1.0, "flux")); tr.setRngSeed(12345); tr.setNumberOfRaysFixed(100_000_000)
shard = vd.GpuShard(tr, dev); tr.applyPrepare()
for _ in range(3):
    vd.distributed_apply(shard, 100_000_000, 0, 1, run_number=1)
T = {k: [] for k in ("setRun+bind+range", "prepare", "launch", "finish", "info", "tensor", "tolist", "total", "device")}
for _ in range(20):
    t0 = time.perf_counter(); tr.setRunNumber(1); shard._bind(); tr.setRayRange(0, 100_000_000)
    t1 = time.perf_counter(); tr.applyPrepare()
    t2 = time.perf_counter(); tr.applyLaunch()
    t3 = time.perf_counter(); tr.applyFinish(collect=False)
    t4 = time.perf_counter(); info = tr.getRayTraceInfo(); cnt = [int(getattr(info, k)) for k in vd.COUNTER_KEYS]
    t5 = time.perf_counter(); c = torch.tensor(cnt, dtype=torch.int64, device=dev)
    t6 = time.perf_counter(); c.tolist()
    t7 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t7 - t0, info.timeTrace)):
        T[k].append(v * 1e3)
for k, v in T.items():
    print(f"{k:20s} {np.median(v):8.3f} ms")

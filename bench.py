#!/usr/bin/env python3
"""bench.py — headline benchmark of the flux ray-tracing hot path on MI355X.

Metric (BASELINE.json): Mrays/s on the 1000x1000 3-D disk grid (1 M disks,
config C2), cosine source, 1e8 rays per GPU, plus the flux L2-relative error
against the CPU oracle.  A "step" is one pass of the hot path (one trace launch
of `--rays` primary rays, flux accumulators zeroed, geometry/BVH resident in
HBM).  With N GPUs every rank traces its own slice of the global ray index
range and the per-primitive int64 accumulators are summed with one RCCL
all-reduce per step, inside the timed region:

  weak   (default)            every rank traces --rays (1e8) rays: N x 1e8 in total
  strong (--total-rays R)     R rays in total, R/N per rank (config C3: R = 1e9)

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no RANK in the environment starts the N ranks itself
(torch.distributed.run as a child process, before anything touches the GPU).
Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import gc
import json
import math
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before HIP/HSA initialises (RCCL needs dmabuf IPC)

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DATA = os.path.join(ROOT, "tests", "golden", "data")

HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec
CLOCK_HZ = 2.4e9         # max clock
SIMDS, CUS = 1024, 256
# issue ceilings, wave-instructions per second chip-wide (MI355X_MICROARCH.md "Wave scheduling":
# a wave64 VALU instruction issues over 2 cycles on a SIMD-32; one scalar unit per CU, 1 / cycle)
VALU_PEAK = SIMDS * CLOCK_HZ / 2.0
SALU_PEAK = CUS * CLOCK_HZ


VL1_HIT_LOADS_PER_CLOCK = 1.1  # scattered 16-byte loads per CU-clock served from the vector L1 (tools/gather_rate.hip)
VL1_MISS_CLOCKS = 2.0          # CU-clocks per load whose 128-byte line comes from the L2


def algorithmic_bytes(n_prims, geo_hits, segments, k_neigh=8, h_credit=2.356):
    """bytes a perfect one-ray-at-a-time kernel must move per trace segment.
    hit segment : ceil(log2 N)*32 + (1+K)*28 + (K*4+8) + H*8
    other       : ceil(log2 N)*32 (one root-to-leaf path)"""
    path = math.ceil(math.log2(max(n_prims, 2))) * 32
    hit = path + (1 + k_neigh) * 28 + (k_neigh * 4 + 8) + h_credit * 8
    return geo_hits * hit + (segments - geo_hits) * path, hit, path


def host_cpu_share():
    """CPUs this process may actually use: scheduler affinity capped by the cgroup CPU quota
    (a GPU box exposes all 256 hardware threads but grants a 16-CPU quota; 128 OpenMP threads
    under that quota only throttle each other)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """Start `n` ranks of this script under torch.distributed.run as a CHILD process and return
    its exit code.  The parent has not touched HIP/HSA (no torch.cuda call, no library load)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd)


def src_sha256():
    """sha256 over the kernel sources the library is built from (same recipe as tools/prof_summary.py)"""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "viennaray_amd", "csrc")
    for fn in sorted(os.listdir(src)):
        if fn.endswith((".hip", ".hpp", ".cpp")) or fn == "Makefile":
            with open(os.path.join(src, fn), "rb") as fh:
                h.update(fn.encode() + b"\0" + fh.read())
    return h.hexdigest()


def lib_sha256():
    import viennaray_amd as vr
    h = hashlib.sha256()
    with open(vr.LIB_PATH, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


# ---------------------------------------------------------------------------------------------
# workloads.  Each returns (tracer, oracle_factory, n_prims, description)
# ---------------------------------------------------------------------------------------------
def workload(name, device=0, sticking=None):
    import viennaray_amd as vr
    from viennaray_amd import io
    BC = vr.BoundaryCondition
    D = 3
    if name in ("C2", "C1_plane100", "C2_rippled"):
        n = 100 if name == "C1_plane100" else 1000
        s = 1.0 if sticking is None else sticking
        if name == "C1_plane100":
            s = 0.1 if sticking is None else sticking
        pts, nrm = io.plane_grid(n, 1.0)
        if name == "C2_rippled":
            # NOT a BASELINE config: C2's plane with half a grid cell of relief (z = 0.5 sin(x/4) cos(y/4), normals of the
            # height field) — what the headline's flat-scene kernels do on a surface that is not perfectly flat
            x, y = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
            amp, wave = 0.5, 4.0
            pts = pts.copy()
            pts[:, 2] = (amp * np.sin(x / wave) * np.cos(y / wave)).astype(np.float32)
            nv = np.stack([-amp / wave * np.cos(x / wave) * np.cos(y / wave), amp / wave * np.sin(x / wave) * np.sin(y / wave),
                           np.ones_like(x)], -1)
            nrm = (nv / np.linalg.norm(nv, axis=1, keepdims=True)).astype(np.float32)
        t = vr.TraceDisk(3, device=device)
        t.setGeometry(pts, nrm, 1.0)
        t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
        t.setParticleType(vr.DiffuseParticle(s, "flux"))
        desc = f"P({n}) {n * n} disks, DiffuseParticle sticking {s}, cosine source, PERIODIC x/y"
        if name == "C2_rippled":
            desc = "NOT a BASELINE config: " + desc + ", rippled by half a grid cell"

        def mk_oracle(po):
            o = po.Oracle()
            o.set_disks(pts, nrm, 1.0, 3)
            o.set_boundary_conditions([po.PERIODIC] * 3)
            o.set_particle(po.DIFFUSE, s)
            return o
        nprims = n * n
    elif name == "C1_trench3d":
        gd, pts, nrm = io.read_grid(os.path.join(DATA, "trenchGrid3D.dat"))
        s = 0.1 if sticking is None else sticking
        t = vr.TraceDisk(3, device=device)
        t.setGeometry(pts, nrm, gd)
        t.setBoundaryConditions([BC.PERIODIC_BOUNDARY] * 3)
        t.setParticleType(vr.DiffuseParticle(s, "flux"))
        desc = f"examples/disk3D trenchGrid3D.dat ({len(pts)} disks), DiffuseParticle sticking {s}, PERIODIC x/y"

        def mk_oracle(po):
            o = po.Oracle()
            o.set_disks(pts, nrm, gd, 3)
            o.set_boundary_conditions([po.PERIODIC] * 3)
            o.set_particle(po.DIFFUSE, s)
            return o
        nprims = len(pts)
    elif name == "C4":
        gd, v, tri = io.read_mesh(os.path.join(DATA, "trenchMesh.dat"), 3)
        t = vr.TraceTriangle(3, device=device)
        t.setGeometry(v, tri, gd)
        t.setParticleType(vr.SpecularParticle(0.1, 50.0, "flux"))
        desc = f"examples/triangle3D trenchMesh.dat ({len(tri)} triangles), SpecularParticle sticking 0.1, " \
               f"source power 50, REFLECTIVE walls"

        def mk_oracle(po):
            o = po.Oracle()
            o.set_triangles(v, tri, gd, 3)
            o.set_particle(po.SPECULAR, 0.1, 50.0)
            return o
        nprims = len(tri)
    elif name in ("C5p", "C5r"):
        gd, pts, nrm = io.read_grid(os.path.join(DATA, "trenchGrid2D.dat"))
        D = 2
        bc = BC.PERIODIC_BOUNDARY if name == "C5p" else BC.REFLECTIVE_BOUNDARY
        t = vr.TraceDisk(2, device=device)
        t.setGeometry(pts, nrm, gd)
        t.setSourceDirection(vr.TraceDirection.POS_Y)
        t.setBoundaryConditions([bc] * 2)
        t.setParticleType(vr.DiffuseParticle(0.1, "flux"))
        desc = f"examples/disk2D trenchGrid2D.dat ({len(pts)} disks, D=2, POS_Y), DiffuseParticle sticking 0.1, " \
               f"{'PERIODIC' if name == 'C5p' else 'REFLECTIVE'} x"

        def mk_oracle(po):
            o = po.Oracle()
            o.set_disks(pts, nrm, gd, 2)
            o.set_source_direction(po.POS_Y)
            o.set_boundary_conditions([int(bc)] * 2)
            o.set_particle(po.DIFFUSE, 0.1)
            return o
        nprims = len(pts)
    else:
        raise ValueError(name)
    t.setRngSeed(12345)
    return t, mk_oracle, nprims, desc, D


COUNTER_NAMES = ("totalRaysTraced", "nonGeometryHits", "geometryHits", "boundaryHits", "reflections",
                 "raysTerminated")


def parity_sample(t, mk_oracle, total_rays, sample, threads, first=0):
    """Flux L2-relative error (SOURCE-normalised) and TraceInfo differences, GPU vs the CPU oracle, on
    `sample` rays from global index `first` of the `total_rays`-ray workload (same seed, same global ray indices)."""
    from oracle import pyoracle as po
    o = mk_oracle(po)
    o.set_num_rays_fixed(total_rays)
    o.set_ray_range(first, sample)
    o.set_rng_seed(12345)
    o.set_lazy_rng(True)  # same stream as std::mt19937_64 (tests/test_oracle_rng.py), cheaper to seed
    o.apply(threads)
    t.setNumberOfRaysFixed(total_rays)
    t.setRunNumber(1)
    t.setRayRange(first, sample)
    t.apply()
    t.setRayRange(0, 0)
    f = t.normalizeFlux(t.getLocalData().getVectorData(0)).astype(np.float64)
    r = o.normalize_flux(o.flux()).astype(np.float64)
    gi, oi = t.getRayTraceInfo(), o.info()
    den = np.linalg.norm(r)
    return dict(parity_first_ray=first, flux_l2_rel_err=float(np.linalg.norm(f - r) / den) if den > 0 else float(np.linalg.norm(f - r)),
                counter_diff={k: int(getattr(gi, k)) - oi[k] for k in COUNTER_NAMES}, parity_sample_rays=sample)


def secondary_case(name, rays, sample, threads, sticking=None, reps=2, ray_range=None, total_rays=None, sha=None,
                   counters_key=None):
    """One non-headline workload: Mrays/s from the device pipeline time of a warmed apply()
    (geometry resident), trace-kernel ms, a parity sample against the oracle and — when the committed PMC
    profile holds this workload for this build — its roofline block.
    ray_range = (first, count): trace only that slice of a `total_rays`-ray stream (C3: one rank's shard)."""
    t, mk_oracle, nprims, desc, D = workload(name, 0, sticking)
    total = total_rays or rays
    t.setNumberOfRaysFixed(total)
    if ray_range:
        t.setRayRange(*ray_range)
    t.setRunNumber(1)
    t.apply()          # warm-up: scene build, buffers, code objects (not timed; wall below is a warmed apply)
    gc.collect()       # (the previous workloads' tracers free their device buffers now, not inside a timed apply)
    best = None
    walls = []
    for _ in range(reps):
        t.setRunNumber(1)
        t0 = time.perf_counter()
        t.apply()
        wall = time.perf_counter() - t0
        walls.append(wall)
        info = t.getRayTraceInfo()
        if best is None or info.timeTrace < best["t"]:
            best = dict(t=info.timeTrace, k=info.timeTraceKernel, g=info.timeGenKernel, wall=wall,
                        seg=int(info.totalRaysTraced), refits=int(info.bvhRefits))
    out = dict(name=name, workload=f"{desc}, {rays} rays" + (f" = indices [{ray_range[0]}, {ray_range[0] + ray_range[1]}) of a "
                                                              f"{total}-ray stream" if ray_range else "") + ", seed 12345",
               rays=rays, segments=best["seg"],
               Mrays_per_s=round(rays / best["t"] / 1e6, 1), device_pipeline_ms=round(best["t"] * 1e3, 4),
               trace_kernel_ms=round(best["k"] * 1e3, 4), gen_kernel_ms=round(best["g"] * 1e3, 4),
               apply_wall_ms=round(min(walls) * 1e3, 3), kernel_mode=t.traceMode(), bvh_refits=best["refits"])
    if sha and counters_key:
        r = roofline(sha, counters_key, best["k"] * 1e3, best["g"] * 1e3, best["seg"], rays)
        if r.get("frac") is not None:
            out["roofline"] = {k: r[k] for k in ("kernel", "kernel_ms", "profiled_kernel_ms", "bound", "achieved", "peak", "unit",
                                                 "frac", "traffic", "l2_hit_rate", "useful_lane_frac", "lanes_per_valu_instr",
                                                 "wave_instr_per_segment", "wait_frac", "l1_miss_per_access", "vector_l1",
                                                 "limiter") if k in r}
    if sample:
        first = ray_range[0] if ray_range else 0
        out.update(parity_sample(t, mk_oracle, total, min(sample, rays), threads, first))
    return out


def make_step(tr, shard, total_rays, rank, world):
    """One step of the headline workload.  N = 1: the reference's whole apply() — generator, trace, gather AND the float
    flux in the host's TracingData (rayTraceDisk.hpp:40-57, rayTrace.hpp:135): `tr.apply()`, download inside the timer.
    N > 1: this rank's shard, then ONE all-reduce of the int64 accumulators, which stay on the device (the collective
    needs them there; every rank ends with the full sums).  Returns (step, end_state); step() -> (info, counters)."""
    from viennaray_amd import distributed as vd
    if world == 1:
        def step():
            tr.setRunNumber(1)   # every step traces the same seeded stream (runNumber 1 -> kernel seed 12346)
            tr.setRayRange(0, 0)
            tr.apply()
            return tr.getRayTraceInfo(), {"allreduce_ms": 0.0}
        return step, "float flux in the host's TracingData (vr_get_flux_data inside the timed step)"

    def step():
        # rank r traces its contiguous slice of the global ray indices, then ONE
        # RCCL all-reduce of the int64 flux accumulators (+ the 7 counters)
        acc, counters = vd.distributed_apply(shard, total_rays, rank, world, run_number=1)
        return shard.last_info, counters
    return step, "int64 accumulators on every device, all-reduced (no host download inside the step)"


def secondary_failures(sec):
    """Names of the secondary workloads that raised, or whose parity sample disagrees with the oracle (any counter, or
    flux L2 > 1e-4): a broken BASELINE config must be visible in the exit code, not only inside the JSON line."""
    bad = []
    for r in sec or []:
        if "error" in r:
            bad.append(f"{r.get('name')}: {r['error']}")
        elif any(v != 0 for v in (r.get("counter_diff") or {}).values()):
            bad.append(f"{r.get('name')}: counters differ {r['counter_diff']}")
        elif r.get("flux_l2_rel_err") is not None and not (r["flux_l2_rel_err"] <= 1e-4):
            bad.append(f"{r.get('name')}: flux L2-rel-err {r['flux_l2_rel_err']}")
    return bad


def emit(out):
    """Print THE json line; return the exit code: 0, or 3 when a parity check of the line failed (headline sample or any
    secondary) — after the line, so the record stays intact."""
    failed = secondary_failures(out.get("secondary"))
    if any(v != 0 for v in (out.get("counter_diff") or {}).values()) or \
            (out.get("flux_l2_rel_err") is not None and not out["flux_l2_rel_err"] <= 1e-4):
        failed.insert(0, "headline parity sample")
    if failed:
        out["parity_failed"] = failed
    print(json.dumps(out), flush=True)
    if failed:
        print("bench.py: a parity check failed: " + "; ".join(failed), file=sys.stderr, flush=True)
    return 3 if failed else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=1000, help="disks per side (C2: 1000)")
    ap.add_argument("--rays", type=int, default=100_000_000, help="primary rays per GPU per step (weak scaling)")
    ap.add_argument("--total-rays", type=int, default=0,
                    help="strong scaling: this many rays in total, sharded over the ranks (C3: 1000000000)")
    ap.add_argument("--sticking", type=float, default=1.0)
    ap.add_argument("--cpu-rays", type=int, default=20_000_000, help="CPU baseline sample (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the non-headline workloads")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--stub-shard", action="store_true",
                    help="CPU rehearsal of the multi-rank path (tests): a deterministic stand-in shard, gloo")
    ap.add_argument("--stub-secondary-error", action="store_true",
                    help="(tests, with --stub-shard) a secondary workload that raised: the line is printed, the exit code is 3")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        # the driver's command shape `python bench.py --gpus N ...`: start the ranks ourselves,
        # as children, before this process initialises the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"--nproc-per-node {args.gpus} (or let bench.py start the ranks: no RANK in the environment)")
    distributed = world > 1 or "RANK" in os.environ  # launched by torch.distributed.run

    import torch
    import torch.distributed as dist
    from viennaray_amd import distributed as vd

    strong = args.total_rays > 0
    total_rays = args.total_rays if strong else args.rays * world
    rays_rank = vd.ray_shard(total_rays, rank, world)[1]

    if args.stub_shard:
        return stub_main(args, rank, world, total_rays)

    import viennaray_amd as vr
    if not vr.device_available():
        raise SystemExit("bench.py: no HIP device; the flux tracer has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, device_id=torch.device("cuda", local_rank))
        # RCCL builds its communicator and channels on first use: do that before anything is timed,
        # whatever --warmup says (an 8 MB int64 buffer like the flux accumulators, and a tiny one)
        _w = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
        dist.all_reduce(_w)
        dist.all_reduce(_w[:8])
        torch.cuda.synchronize()
        del _w

    # ---- workload C2: P(n) plane, DiffuseParticle, PERIODIC, cosine source ----
    n = args.grid
    seed = 12345
    pts, nrm = vr.io.plane_grid(n, 1.0)
    N = n * n
    tr = vr.TraceDisk(3, device=local_rank)
    tr.setGeometry(pts, nrm, 1.0)
    tr.setBoundaryConditions([vr.BoundaryCondition.PERIODIC_BOUNDARY] * 3)
    tr.setParticleType(vr.DiffuseParticle(args.sticking, "flux"))
    tr.setRngSeed(seed)
    tr.setNumberOfRaysFixed(total_rays)
    shard = vd.GpuShard(tr, dev) if world > 1 else None  # (binds a torch int64 accumulator tensor: what RCCL reduces)

    t0 = time.perf_counter()
    tr.applyPrepare()  # bbox, walls, areas, LBVH, uploads: geometry resident in HBM
    build_s = time.perf_counter() - t0

    pipe_ms, trace_ms, gen_ms, segs, geo, wall_ms, ar_ms = [], [], [], [], [], [], []
    step, end_state = make_step(tr, shard, total_rays, rank, world)

    def sync_all():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        info, cnts = step()
        wall_ms.append((time.perf_counter() - ts) * 1e3)
        ar_ms.append(float(cnts.get("allreduce_ms", 0.0)))
        if info is not None:
            pipe_ms.append(info.timeTrace * 1e3)
            trace_ms.append(info.timeTraceKernel * 1e3)
            gen_ms.append(info.timeGenKernel * 1e3)
            segs.append(int(info.totalRaysTraced))   # this rank's share
            geo.append(int(info.geometryHits))
    sync_all()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # per-rank spread of the step (outside the timed region): a measured N > 1 curve must be attributable
    per_rank = vd.rank_report([rank, np.mean(trace_ms) if trace_ms else 0.0, np.mean(gen_ms) if gen_ms else 0.0,
                               np.mean(pipe_ms) if pipe_ms else 0.0, np.mean(ar_ms) if ar_ms else 0.0,
                               np.mean(wall_ms) if wall_ms else 0.0, rays_rank]) if distributed else None
    if rank == 0:
        mode = tr.traceMode()
        kernel_name = f"trace_kernel<3,0,{0 if mode in (1, 2) else tr._particle.kind},{mode}>"
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays * args.steps / elapsed / 1e6
        kavg = float(np.mean(pipe_ms))    # whole device pipeline: gen + trace (+ memsets)
        tavg = float(np.mean(trace_ms))   # dominant kernel (trace_kernel), summed over batches
        gavg = float(np.mean(gen_ms))
        abytes, b_hit, b_path = algorithmic_bytes(N, float(np.mean(geo)), float(np.mean(segs)))
        sha = lib_sha256()
        roof = roofline(sha, f"C2_s{args.sticking}" if n == 1000 else f"P{n}_s{args.sticking}", tavg, gavg,
                        float(np.mean(segs)), rays_rank)
        roof["kernel"] = roof.get("kernel") or kernel_name
        # SURVEY 8(d)'s one-ray-at-a-time byte model, for the record: it prices every ray's own root-to-leaf path and
        # neighbour records, while a wavefront of 64 sorted rays shares each fetch — it exceeds the HBM peak and is
        # therefore NOT used as a fraction (round-2 verdict); `frac` above is counter bytes / time / 8 TB/s
        roof["algorithmic_model"] = {"bytes_per_hit_segment": round(b_hit, 1), "bytes_per_launch": int(abytes),
                                     "GBs": round(abytes / (tavg * 1e-3) / 1e9, 1),
                                     "compulsory_bytes_per_launch": int(rays_rank * 32 + N * 40),
                                     "note": "compulsory = every ray record read once (32 B) + every disk record and accumulator once"}
        out = {
            "metric": "Mrays/sec + flux L2-rel-err vs CPU oracle, 1M-disk 3D @1e8 rays",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C2: {n}x{n} 3D disk grid ({N} disks), DiffuseParticle sticking "
                                   f"{args.sticking}, cosine source, PERIODIC x/y, "
                                   + (f"{total_rays} rays in total sharded over {world} GPU(s) (C3)" if strong else
                                      f"{args.rays} rays per GPU per step") + ", seed 12345",
                       "rays_per_gpu": rays_rank, "total_rays": total_rays, "grid": n, "sticking": args.sticking,
                       "kernel_mode": {0: "general (reflection + roulette + RNG)", 1: "absorbing, flat scene",
                                       2: "absorbing, structured scene",
                                       3: "general, flat scene (packet-query crediting)",
                                       4: "general, scene resident in LDS",
                                       5: "absorbing, flat with relief (relief packets + loose bins)",
                                       6: "general, flat with relief (relief packets + loose bins)"}.get(mode, str(mode)),
                       "parallelism": f"ray-range shards x{world}, BVH replicated, int64 flux all-reduce"},
            "device_pipeline_ms": round(kavg, 4), "trace_kernel_ms": round(tavg, 4), "gen_kernel_ms": round(gavg, 4),
            "apply_wall_ms": round(float(np.mean(wall_ms)), 4), "end_state": end_state,
            "segments_per_step": int(np.mean(segs)), "Msegments_per_s": round(np.mean(segs) / (kavg * 1e-3) / 1e6, 2),
            "trace_launches_per_step": int(math.ceil(rays_rank / float(1 << 27))),  # one per batch of <= 2^27 rays
            "prepare_s": round(build_s, 4), "lib_sha256": sha[:16],
            "bvh_refits": int(info.bvhRefits) if info is not None else None,  # (0: the BVH fit's fast hand-over never needed its fenced retry)
            "roofline": roof,
        }
        if per_rank is not None:
            cols = ("rank", "trace_ms", "gen_ms", "device_pipeline_ms", "allreduce_ms", "step_wall_ms", "rays")
            out["multi_gpu"] = {
                "rccl_ranks_seen": len(per_rank), "backend": args.backend,
                "allreduce_ms": round(float(np.mean(ar_ms)), 4),
                "allreduce_ms_max_over_ranks": round(max(r[4] for r in per_rank), 4),
                "trace_ms_min": round(min(r[1] for r in per_rank), 4), "trace_ms_max": round(max(r[1] for r in per_rank), 4),
                "step_wall_ms_min": round(min(r[5] for r in per_rank), 4), "step_wall_ms_max": round(max(r[5] for r in per_rank), 4),
                "per_rank": [dict(zip(cols, [int(r[0])] + [round(v, 4) for v in r[1:6]] + [int(r[6])])) for r in per_rank],
                "note": "allreduce_ms: HIP events round the two collectives of a step (int64 flux + counters) on this rank; "
                        "a rank that finishes its shard early waits inside the collective for the slowest one"}
        if world == 1:
            from oracle import pyoracle as po  # the checker / CPU baseline only (never the product path)
            threads = min(po.max_threads(), host_cpu_share())
            # ---- a fresh point cloud in a live context: setGeometry + BVH/neighbourhood build + trace
            #      (the reference's TraceInfo.time includes the build) -------------------
            t0 = time.perf_counter()
            tr.setGeometry(pts, nrm, 1.0)
            tr.setRunNumber(1)
            tr.setRayRange(0, 0)
            tr.applyPrepare()
            t1 = time.perf_counter()
            tr.applyLaunch()
            tr.applyFinish(collect=False)
            t2 = time.perf_counter()
            out["value_incl_build"] = round(total_rays / (t2 - t0) / 1e6, 3)
            out["incl_build"] = {"set_geometry_and_prepare_ms": round((t1 - t0) * 1e3, 3),
                                 "trace_ms": round((t2 - t1) * 1e3, 3),
                                 "note": "fresh 1M-disk cloud in a live context: host copy, upload, LBVH + "
                                         "neighbourhood + disk areas, then one 1e8-ray step"}
            # ---- parity + CPU baseline (rank 0, N=1 only; bounded sample) ----------------
            if args.cpu_rays > 0:
                sample = min(args.cpu_rays, total_rays)
                o, cb = cpu_baseline(pts, nrm, 1.0, args.sticking, seed, sample, total_rays, threads)
                out["cpu_baseline"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in cb.items()}
                out["speedup_vs_cpu_baseline"] = round(value / cb["value"], 1)
                if not args.no_parity:
                    # same sample on the GPU (outside the timed region) -> flux L2-rel-err
                    tr.setRunNumber(1)
                    tr.setRayRange(0, sample)
                    tr.apply()
                    tr.setRayRange(0, 0)
                    f = tr.normalizeFlux(tr.getLocalData().getVectorData(0)).astype(np.float64)
                    r = o.normalize_flux(o.flux()).astype(np.float64)
                    out["flux_l2_rel_err"] = float(np.linalg.norm(f - r) / np.linalg.norm(r))
                    gi, oi = tr.getRayTraceInfo(), o.info()
                    out["counter_diff"] = {k: int(getattr(gi, k)) - oi[k] for k in COUNTER_NAMES}
                    out["parity_sample_rays"] = sample
                del o
            # ---- the other BASELINE configs, outside the headline's timed region ------------
            if not args.no_secondary:
                sec = []
                tr = shard = None  # free the headline context's ray-stream buffers
                from viennaray_amd import distributed as vd2
                c3 = vd2.ray_shard(1_000_000_000, 7, 8)   # C3: rank 7's shard of the 10^9-ray stream (one batch, like each of 8 GPUs)
                cases = (dict(name="C2", rays=args.rays, sample=2_000_000, sticking=0.1, counters_key="C2_s0.1"),
                         dict(name="C2", rays=c3[1], sample=2_000_000, sticking=1.0, ray_range=c3, total_rays=1_000_000_000,
                              label="C3_shard"),
                         dict(name="C1_plane100", rays=1_000_000, sample=1_000_000),
                         dict(name="C1_plane100", rays=100_000_000, sample=0),
                         dict(name="C1_trench3d", rays=1_000_000, sample=1_000_000),
                         dict(name="C1_trench3d", rays=57_838_000, sample=0, counters_key="C1_trench3d"),
                         dict(name="C4", rays=100_000_000, sample=1_000_000, counters_key="C4"),
                         dict(name="C5p", rays=100_000_000, sample=1_000_000, counters_key="C5p"),
                         dict(name="C5r", rays=100_000_000, sample=1_000_000),
                         dict(name="C2_rippled", rays=args.rays, sample=1_000_000, sticking=1.0, counters_key="C2_rippled_s1.0"),
                         dict(name="C2_rippled", rays=args.rays, sample=1_000_000, sticking=0.1, counters_key="C2_rippled_s0.1"))
                for cs in cases:
                    label = cs.pop("label", None)
                    if args.no_parity:
                        cs["sample"] = 0
                    try:
                        r = secondary_case(threads=threads, sha=sha, **cs)
                        if label:
                            r["name"] = label
                        sec.append(r)
                    except Exception as e:  # a failing secondary must not hide the headline ...
                        sec.append(dict(name=label or cs["name"], error=str(e)))
                out["secondary"] = sec  # ... but it must show in the exit code (emit: after the JSON line)
        exit_code = emit(out)
    else:
        exit_code = 0
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


ISSUE_CLASSES = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS")
_COUNTERS = None


def committed_counters(sha):
    """profiles/counters_latest.json (tools/pmc_profile.py + tools/publish_counters.py), or None when it was taken
    with another build: counts are combined with this run's kernel times only for the same library bytes or — a
    rebuild need not be byte-identical — the same kernel sources."""
    global _COUNTERS
    if _COUNTERS is None:
        try:
            cj = json.load(open(os.path.join(ROOT, "profiles", "counters_latest.json")))
        except Exception:  # noqa: BLE001
            cj = {}
        same = bool(cj) and (cj.get("lib_sha256") == sha or (cj.get("src_sha256") is not None and cj.get("src_sha256") == src_sha256()))
        _COUNTERS = (cj, same)
    return _COUNTERS


def kernel_block(k, ms, segments=None):
    """One kernel's figures per launch: PMC counts of the committed profile over THIS run's kernel time (HIP
    events on the library's stream).
      frac = achieved / peak, achieved = fabric bytes of the L2s per launch ((2 FETCH_SIZE + WRITE_SIZE) KiB: the
             guide's gfx950 correction; Infinity-Cache hits are included, so this is an UPPER bound of the HBM share)
             / kernel time, peak = 8.0 TB/s.  The contract's roof — and not what limits these kernels:
      useful_lane_frac, wave_instr_per_segment, lanes_per_valu_instr, wait_frac say what does (instruction issue on
             partly filled wavefronts + exposed latency); they cannot be raised by executing more instructions."""
    t = ms * 1e-3
    out = {"kernel": k.get("name"), "kernel_ms": round(ms, 4), "profiled_kernel_ms": round(k["avg_ms"], 4) if k.get("avg_ms") else None,
           "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": None, "frac": None, "traffic": None}
    if k.get("hbm_bytes") is not None:
        out["traffic"] = int(k["hbm_bytes"])
        out["achieved"] = round(k["hbm_bytes"] / t / 1e9, 1)
        out["frac"] = round(k["hbm_bytes"] / t / (HBM_PEAK_GBS * 1e9), 4)
    for key, nd in (("l2_hit_rate", 4), ("dram_read_share", 4), ("useful_lane_frac", 4), ("lanes_per_valu_instr", 2),
                    ("clock_ghz", 3), ("l1_miss_per_access", 4)):
        if k.get(key) is not None:
            out[key] = round(k[key], nd)
    if k.get("SQ_WAIT_ANY_frac") is not None:
        out["wait_frac"] = round(k["SQ_WAIT_ANY_frac"], 4)
    # The roof the WALKING kernels do run under: scattered per-lane loads through the CU's vector L1.  Measured
    # (tools/gather_rate.hip, profiles/r04_gather_rate.txt; clock read in-kernel): per CU-clock a CU serves 0.83 16-byte
    # loads when every lane reads its own line from the L1, 1.10 for the two halves of a 32-byte record (pair nodes, disk
    # records), 1.65 8- or 4-byte loads — and 0.43-0.64 when the 128-byte line comes from the L2 (64 B/clk).  Chip-wide
    # 500-670 G lane-loads/s from the L1s, 265-390 from the L2s: trench3D's general kernel issues 498 G/s.
    acc, miss, ghz = k.get("TCP_TOTAL_CACHE_ACCESSES_sum"), k.get("l1_miss_per_access"), k.get("clock_ghz")
    if acc and miss is not None and ghz:
        rate = acc / (t * CUS * ghz * 1e9)
        ceil = 1.0 / ((1.0 - miss) / VL1_HIT_LOADS_PER_CLOCK + miss * VL1_MISS_CLOCKS)
        out["vector_l1"] = {"lane_loads_per_launch": int(acc), "per_cu_clock": round(rate, 3),
                            "ceiling_of_this_hit_miss_mix": round(ceil, 3), "frac": round(rate / ceil, 3),
                            "lane_loads_per_segment": round(acc / segments, 1) if segments else None,
                            "from": "profiles/r04_gather_rate.txt: ~1.1 loads per CU-clock from the L1 (32-byte records), ~2 clocks per load that misses it"}
    if all(c in k for c in ISSUE_CLASSES):
        total = sum(k[c] for c in ISSUE_CLASSES)
        if segments:
            out["wave_instr_per_segment"] = round(total / segments, 1)
        # diagnostics only — NOT a roofline fraction: issued (not useful) instructions against the rate a SIMD
        # reaches on the matching synthetic mix (tools/issue_ceiling.py; the guide's nominal figure is 0.5 / cycle)
        out["issue_diag"] = {"wave_instructions_per_launch": {c[9:]: int(k[c]) for c in ISSUE_CLASSES},
                             "per_simd_cycle": round(total / t / (SIMDS * CLOCK_HZ), 4)}
    return out


def roofline(sha, case, trace_ms, gen_ms, segments, rays):
    """The `roofline` object of the bench line for workload `case` (a key of profiles/counters_latest.json)."""
    cj, same = committed_counters(sha)
    roof = {"kernel": None, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
    roof["counts_from"] = {"file": "profiles/counters_latest.json", "lib_sha256": str(cj.get("lib_sha256"))[:16],
                           "src_sha256": str(cj.get("src_sha256"))[:16], "commit": cj.get("commit"),
                           "matches_running_build": bool(same)}
    c = (cj.get("cases") or {}).get(case)
    if not same or not c or (c.get("rays") not in (None, rays)):
        roof["note"] = "no committed PMC profile of this build and workload: counters are not combined with this run's timing"
        return roof
    roof.update(kernel_block(c["trace_kernel"], trace_ms, segments))
    vl1 = (roof.get("vector_l1") or {}).get("frac")
    if vl1 is not None and vl1 >= 0.7:
        roof["limiter"] = ("the CU's vector L1 path: the per-lane walks issue scattered 16-byte loads (pair nodes, primitive and "
                           "neighbour records) at vector_l1.frac of the rate a CU can serve for this hit / miss mix — not HBM "
                           "(the scene is served from the L2s), not instruction issue")
    else:
        roof["limiter"] = ("not HBM: the scene (BVH + records, MBs) is served from the L2s and the Infinity Cache, the ray records "
                           "stream once; time goes to instruction issue on partly filled wavefronts and to exposed latency "
                           "(useful_lane_frac, wave_instr_per_segment, wait_frac); vector_l1 says how far the scattered loads "
                           "are from their own ceiling")
    roof["infinity_cache"] = ("rocprofv3 exposes no MALL hit counter on gfx950 (profiles/r03_rocprof_avail.txt); FETCH_SIZE counts "
                              "L2 misses whether the Infinity Cache or HBM serves them, so frac is an upper bound of the HBM share")
    if gen_ms > 0 and c.get("gen_kernel"):
        g = kernel_block(c["gen_kernel"], gen_ms)
        # the generator's own floor: the 156 + 4 sequential 64-bit multiply-adds per ray that seeding
        # std::mt19937_64 imposes, priced against the measured rate of exactly that chain (tools/issue_ceiling.py)
        ceil = (cj.get("issue_ceiling") or {}).get("mt19937_64_seed_step@8w")
        if ceil and rays:
            g["mt_seed_step_frac"] = round(rays / 64.0 * 160.0 / (gen_ms * 1e-3) / (ceil * SIMDS * CLOCK_HZ), 4)
        roof["gen_kernel"] = g
    if cj.get("issue_ceiling"):
        roof["measured_issue_ceilings_per_simd_cycle"] = cj["issue_ceiling"]
    return roof


def cpu_baseline(pts, nrm, grid_delta, sticking, seed, sample_rays, total_rays, cores):
    """The CPU oracle (restated reference loop, std::mt19937_64 per ray like the
    reference) timed on this box's host cores on a bounded sample of the same
    workload.  Timer placement mirrors the reference: ray loop only is reported
    as `value`; BVH build time is reported next to it."""
    from oracle import pyoracle as po
    o = po.Oracle()
    t0 = time.perf_counter()
    o.set_disks(pts, nrm, grid_delta, 3)
    t_setup = time.perf_counter() - t0
    o.set_boundary_conditions([po.PERIODIC] * 3)
    o.set_particle(po.DIFFUSE, sticking)
    o.set_num_rays_fixed(total_rays)
    o.set_ray_range(0, sample_rays)
    o.set_rng_seed(seed)
    o.apply(cores)
    info = o.info()
    rate = sample_rays / info["time"] / 1e6
    return o, dict(value=rate, unit="Mrays/s", cores=cores, kind="port",
                   sample=f"first {sample_rays} rays of the same {total_rays}-ray C2 workload, one thread per CPU "
                          f"of this process's share ({cores}; OpenMP guided,64); `value` is the ray loop alone, "
                          f"`value_incl_build` prices the whole {total_rays}-ray job with the oracle's scene build "
                          f"({t_setup:.2f} s: BVH + neighbourhood) inside the timer like the reference (SURVEY Q10)",
                   seconds=info["time"], setup_seconds=t_setup,
                   value_incl_build=total_rays / (total_rays / (rate * 1e6) + t_setup) / 1e6)


# ---------------------------------------------------------------------------------------------
# CPU rehearsal of the launcher + sharding + all-reduce path (tests/test_bench_launcher.py)
# ---------------------------------------------------------------------------------------------
class StubShard:
    """Deterministic stand-in for the HIP tracer: ray i credits primitive (i * 2654435761) mod n with
    weight 1, so the all-reduced accumulators have a closed form the test can check."""

    def __init__(self, n):
        self.n = n
        self.last_info = None

    def trace_local(self, first, count, run_number=None, world=1):
        import torch
        idx = (np.arange(first, first + count, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(self.n)
        acc = np.bincount(idx.astype(np.int64), minlength=self.n).astype(np.int64) << 40
        cnt = [count, 0, count, 0, 0, 0, 0]
        return torch.from_numpy(acc), torch.tensor(cnt, dtype=torch.int64)


def stub_main(args, rank, world, total_rays):
    import torch
    import torch.distributed as dist
    from viennaray_amd import distributed as vd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "RANK" in os.environ:
        dist.init_process_group("gloo")
    shard = StubShard(args.grid * args.grid)
    ar_ms, wall_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        acc, counters = vd.distributed_apply(shard, total_rays, rank, world, run_number=1)
        wall_ms.append((time.perf_counter() - ts) * 1e3)
        ar_ms.append(counters.pop("allreduce_ms", 0.0))
    elapsed = time.perf_counter() - t0
    # the keys of the real line's `multi_gpu` block (the stand-in has no kernels: its trace time is wall - all-reduce)
    per_rank = vd.rank_report([rank, np.mean(wall_ms) - np.mean(ar_ms), 0.0, 0.0, np.mean(ar_ms), np.mean(wall_ms),
                               vd.ray_shard(total_rays, rank, world)[1]])
    if dist.is_initialized():
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    exit_code = 0
    if rank == 0:
        out = ({"metric": "stub", "value": total_rays * args.steps / elapsed / 1e6, "unit": "Mrays/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "scaling": "strong" if args.total_rays > 0 else "weak", "total_rays": total_rays,
                          "acc_sum": int(acc.sum().item() >> 40), "acc_checksum": int((acc >> 40).numpy().dot(
                              np.arange(acc.numel(), dtype=np.int64) % 1000003) % (1 << 61)),
                          "counters": counters,
                          "multi_gpu": {"rccl_ranks_seen": len(per_rank), "backend": "gloo (stand-in shard)",
                                        "allreduce_ms": round(float(np.mean(ar_ms)), 4),
                                        "trace_ms_min": round(min(r[1] for r in per_rank), 4),
                                        "trace_ms_max": round(max(r[1] for r in per_rank), 4),
                                        "per_rank": [dict(rank=int(r[0]), trace_ms=round(r[1], 4), allreduce_ms=round(r[4], 4),
                                                          step_wall_ms=round(r[5], 4), rays=int(r[6])) for r in per_rank]}})
        if args.stub_secondary_error:
            out["secondary"] = [dict(name="stub_ok", counter_diff={"geometryHits": 0}, flux_l2_rel_err=0.0),
                                dict(name="stub_broken", error="injected failure")]
        exit_code = emit(out)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()

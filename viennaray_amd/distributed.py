"""Multi-GPU driver of the flux tracer: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI), rays sharded by global index, BVH replicated,
per-primitive flux combined with ONE sum all-reduce per apply().

The reference has no distributed layer; its only reduction is
the per-thread merge `flux[j] += tl[k][j]` (rayTraceKernel.hpp:350-359).  Ray i's
random stream depends on (i, seed) only (rayTraceKernel.hpp:120-121), so disjoint
index ranges traced on different devices and summed reproduce the single-device
result; with the int64 fixed-point accumulators the sum is exact, independent of
rank count and order.

The shard objects are duck-typed (`trace_local(first, count, run_number, world=) -> (acc, counters)`)
so the same driver runs on the HIP tracer (GpuShard) and, in the CPU `gloo`
tests, on a stand-in backend.
"""
import numpy as np

COUNTER_KEYS = ("totalRaysTraced", "nonGeometryHits", "geometryHits", "particleHits",
                "boundaryHits", "reflections", "raysTerminated")
FLUX_FRAC_BITS = 40  # VR_FLUX_FRAC_BITS


def ray_shard(num_rays, rank, world):
    """Contiguous global ray-index range [first, first+count) of `rank`."""
    first = num_rays * rank // world
    last = num_rays * (rank + 1) // world
    return first, last - first


class GpuShard:
    """Adapter around a viennaray_amd.Trace: binds a torch int64 accumulator tensor
    so the all-reduce runs on the buffer the gather kernel wrote."""

    def __init__(self, tracer, device):
        self.tr = tracer
        self.device = device
        self.acc = None
        self.last_info = None
        self._bind()

    def _bind(self):
        """(Re-)bind the accumulator tensor.  setGeometry() drops an external binding (the
        primitive count may have changed), so this runs before every launch."""
        import torch
        n = self.tr._n * max(1, self.tr.numData())
        if self.acc is None or self.acc.numel() != n:
            self.acc = torch.zeros(n, dtype=torch.int64, device=self.device)
        self.tr.bindFluxAccumulators(self.acc.data_ptr(), n)

    def trace_local(self, first, count, run_number=None, world=1):
        if hasattr(self.tr, "setWorldSize"):
            self.tr.setWorldSize(max(1, int(world)))  # (the overflow check leaves room for the sum over the ranks)
        if run_number is not None:
            self.tr.setRunNumber(run_number)
        self._bind()
        if count > 0:
            self.tr.setRayRange(first, count)
            self.tr.applyPrepare()
            self.tr.applyLaunch()
            self.tr.applyFinish(collect=False)   # ++runNumber, like every apply()
            info = self.tr.getRayTraceInfo()
            cnt = [int(getattr(info, k)) for k in COUNTER_KEYS]
        else:
            # an empty shard still counts as one apply(): every rank must enter the next
            # apply() with the same runNumber, i.e. the same seed (rayTraceDisk.hpp:54)
            self.tr.skipApply()
            self.acc.zero_()
            info = None
            cnt = [0] * len(COUNTER_KEYS)
        self.last_info = info
        return self.acc, cnt


def distributed_apply(shard, num_rays, rank=None, world=None, group=None, run_number=None):
    """Trace this rank's slice of `num_rays` and all-reduce flux + counters.
    Returns (acc int64 tensor [numPrims] (sum over ranks), counters dict)."""
    import torch.distributed as dist
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    first, count = ray_shard(num_rays, rank, world)
    acc, cnt = shard.trace_local(first, count, run_number, world=world)
    allreduce_ms = 0.0
    if world > 1:
        import time
        import torch
        # the collective's time as the stream sees it (HIP events round it: backend "nccl" = RCCL over xGMI; CPU tensors
        # of the gloo rehearsal: wall clock) — so that a measured multi-GPU step can be split into trace + all-reduce
        on_gpu = bool(getattr(acc, "is_cuda", False))
        if on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        else:
            t0 = time.perf_counter()
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)  # exact: integers
        # (the counters travel as a device tensor only when there is something to reduce: a single rank keeps
        #  them on the host and saves two trips over PCIe per apply)
        cnt_t = cnt if torch.is_tensor(cnt) else torch.tensor(cnt, dtype=torch.int64, device=acc.device)
        dist.all_reduce(cnt_t, op=dist.ReduceOp.SUM, group=group)
        cnt = cnt_t
        if on_gpu:
            e1.record()
            e1.synchronize()
            allreduce_ms = float(e0.elapsed_time(e1))
        else:
            allreduce_ms = (time.perf_counter() - t0) * 1e3
    if hasattr(cnt, "tolist"):
        cnt = cnt.tolist()
    counters = {k: int(v) for k, v in zip(COUNTER_KEYS, cnt)}
    counters["numRays"] = int(num_rays)
    counters["allreduce_ms"] = allreduce_ms
    return acc, counters


def rank_report(values, group=None):
    """Every rank's row of floats gathered on all ranks (outside any timed region): the per-rank spread of a multi-GPU
    step — trace / generator / all-reduce milliseconds — and the number of ranks that answered."""
    import torch
    import torch.distributed as dist
    row = torch.tensor([float(v) for v in values], dtype=torch.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [row.tolist()]
    if dist.get_backend(group) == "nccl":
        row = row.cuda()
    rows = [torch.zeros_like(row) for _ in range(dist.get_world_size(group))]
    dist.all_gather(rows, row, group=group)
    return [r.cpu().tolist() for r in rows]


def accumulators_to_flux(acc):
    """int64 fixed point -> float64 flux (raw, un-normalised)."""
    a = acc.detach().cpu().numpy() if hasattr(acc, "detach") else np.asarray(acc)
    return a.astype(np.float64) * 2.0 ** -FLUX_FRAC_BITS

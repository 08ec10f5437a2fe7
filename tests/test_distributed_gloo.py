"""world_size-2 (and 3) gloo tests of the multi-GPU driver on CPU: ray-range
sharding, the integer flux all-reduce and the counter all-reduce.  The HIP tracer
needs a GPU, so the shard backend here is the CPU oracle (test infrastructure);
what is under test is viennaray_amd.distributed."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from viennaray_amd import distributed as vd
from viennaray_amd import io


class OracleShard:
    def __init__(self, pts, nrm, num_rays, sticking):
        from oracle import pyoracle as po
        self.o = po.Oracle()
        self.o.set_disks(pts, nrm, 1.0, 3)
        self.o.set_boundary_conditions([po.PERIODIC] * 3)
        self.o.set_particle(po.DIFFUSE, sticking)
        self.o.set_num_rays_fixed(num_rays)
        self.o.set_rng_seed(77)
        self.o.set_lazy_rng(True)

    def trace_local(self, first, count, run_number=None, world=1):
        self.o.set_run_number(1 if run_number is None else run_number)
        if count == 0:
            n = self.o.n
            return torch.zeros(n, dtype=torch.int64), torch.zeros(len(vd.COUNTER_KEYS), dtype=torch.int64)
        self.o.set_ray_range(first, count)
        self.o.apply(1)
        flux = self.o.flux().astype(np.float64)
        acc = torch.from_numpy(np.rint(flux * 2.0 ** vd.FLUX_FRAC_BITS).astype(np.int64))
        i = self.o.info()
        return acc, torch.tensor([i[k] for k in vd.COUNTER_KEYS], dtype=torch.int64)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_rays, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pts, nrm = io.plane_grid(16, 1.0)
    shard = OracleShard(pts, nrm, num_rays, 1.0)
    acc, counters = vd.distributed_apply(shard, num_rays)
    np.save(os.path.join(out_dir, f"acc_{rank}.npy"), acc.numpy())
    np.save(os.path.join(out_dir, f"cnt_{rank}.npy"), np.array([counters[k] for k in vd.COUNTER_KEYS]))
    dist.barrier()
    dist.destroy_process_group()


def test_ray_shard_partitions_the_index_range():
    for n in (0, 1, 7, 1000, 10**9 + 7):
        for world in (1, 2, 3, 8):
            spans = [vd.ray_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert spans[-1][0] + spans[-1][1] == n
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_trace_equals_single_process(tmp_path, world):
    num_rays = 20001
    port = _free_port()
    mp.spawn(_worker, args=(world, port, num_rays, str(tmp_path)), nprocs=world, join=True)
    pts, nrm = io.plane_grid(16, 1.0)
    ref_acc, ref_cnt = OracleShard(pts, nrm, num_rays, 1.0).trace_local(0, num_rays)
    for r in range(world):
        acc = np.load(tmp_path / f"acc_{r}.npy")
        cnt = np.load(tmp_path / f"cnt_{r}.npy")
        assert (acc == ref_acc.numpy()).all()  # every rank holds the full, exact sum
        assert (cnt == ref_cnt.numpy()).all()
    assert vd.accumulators_to_flux(ref_acc).sum() > 0

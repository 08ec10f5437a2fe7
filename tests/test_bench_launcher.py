"""bench.py's own launcher path on CPU: `python bench.py --gpus N` with no RANK in the environment
must start N ranks itself (torch.distributed.run child), shard the ray range, all-reduce and report
n_gpus = N — rehearsed with the stand-in shard over gloo (the HIP tracer needs a GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 prints ONE json line
    return json.loads(lines[0])


def _expected(total, n):
    idx = (np.arange(total, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(n)
    acc = np.bincount(idx.astype(np.int64), minlength=n).astype(np.int64)
    return int(acc.sum()), int(acc.dot(np.arange(n, dtype=np.int64) % 1000003) % (1 << 61))


@pytest.mark.parametrize("gpus", [1, 2, 3, 8])
def test_gpus_flag_starts_that_many_ranks_weak(gpus):
    out = _run(["--gpus", str(gpus), "--steps", "2", "--warmup", "0", "--stub-shard", "--grid", "50", "--rays", "10007"])
    assert out["n_gpus"] == gpus and out["scaling"] == "weak"
    assert out["total_rays"] == 10007 * gpus
    s, c = _expected(10007 * gpus, 2500)
    assert out["acc_sum"] == s and out["acc_checksum"] == c     # every ray traced exactly once
    assert out["counters"]["totalRaysTraced"] == 10007 * gpus


def test_total_rays_is_strong_scaling():
    one = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--stub-shard", "--grid", "50", "--total-rays", "30001"])
    two = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--stub-shard", "--grid", "50", "--total-rays", "30001"])
    assert one["scaling"] == two["scaling"] == "strong"
    assert one["total_rays"] == two["total_rays"] == 30001
    assert (one["acc_sum"], one["acc_checksum"]) == (two["acc_sum"], two["acc_checksum"]) == _expected(30001, 2500)


def test_eight_ranks_strong_scaling_like_c3():
    """The driver's eventual C3 command shape, `bench.py --gpus 8 --total-rays R`: eight ranks, contiguous
    shares of ONE global index range (a prime-ish R: unequal shares), every ray traced exactly once."""
    out = _run(["--gpus", "8", "--steps", "1", "--warmup", "0", "--stub-shard", "--grid", "50", "--total-rays", "1000003"])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["total_rays"] == 1000003
    assert (out["acc_sum"], out["acc_checksum"]) == _expected(1000003, 2500)
    assert out["counters"]["totalRaysTraced"] == 1000003
    # the first measured multi-GPU curve must be attributable: the collective's time, the per-rank spread of the trace
    # time and the number of ranks that answered travel on the line (round-3 verdict)
    mg = out["multi_gpu"]
    assert mg["rccl_ranks_seen"] == 8 and len(mg["per_rank"]) == 8
    assert sorted(r["rank"] for r in mg["per_rank"]) == list(range(8))
    assert sum(r["rays"] for r in mg["per_rank"]) == 1000003
    assert mg["allreduce_ms"] > 0.0 and mg["trace_ms_max"] >= mg["trace_ms_min"] >= 0.0


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--stub-shard"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_a_failing_secondary_fails_the_run_after_the_json_line():
    """A secondary workload that raised (or whose parity sample disagrees) must show in bench.py's exit code — with the
    JSON line printed and intact (round-3 verdict: a broken BASELINE config used to leave a green record)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--stub-shard",
                        "--grid", "20", "--rays", "1000", "--stub-secondary-error"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 3, (p.returncode, p.stderr[-500:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["parity_failed"] == ["stub_broken: injected failure"]
    assert "parity check failed" in p.stderr

"""bench.py helpers that need no GPU: the algorithmic-bytes model and the CPU share."""
import importlib.util
import os

from helpers import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_survey():
    b = _bench()
    total, hit, path = b.algorithmic_bytes(1_000_000, geo_hits=1.0, segments=1.0)
    assert path == 20 * 32                      # ceil(log2 1e6) = 20 nodes of 32 B
    assert abs(hit - 950.848) < 1e-3            # 640 + 252 + 40 + 2.356 * 8  (SURVEY: "about 951 B")
    total, _, _ = b.algorithmic_bytes(1_000_000, geo_hits=1e8, segments=1.00347415e8)
    assert abs(total / 1e9 - 95.3) < 0.1        # 95.3 GB per C2 launch


def test_host_cpu_share_is_sane():
    n = _bench().host_cpu_share()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_roofline_block_is_counter_bytes_over_time_over_peak():
    """roofline.frac = (2 FETCH_SIZE + WRITE_SIZE) KiB per launch / kernel time / 8 TB/s (round-2 verdict, item 2), with the
    useful-work figures beside it; the issue figures are diagnostics, never the fraction."""
    b = _bench()
    k = dict(name="vr::trace_kernel<3, 0, 0, 1>", avg_ms=6.5, FETCH_SIZE=4.0e6, WRITE_SIZE=1.0e6, hbm_bytes=(2 * 4.0e6 + 1.0e6) * 1024,
             l2_hit_rate=0.74, useful_lane_frac=0.34, lanes_per_valu_instr=48.5, SQ_WAIT_ANY_frac=0.52,
             SQ_INSTS_VALU=3.0e9, SQ_INSTS_SALU=2.0e9, SQ_INSTS_SMEM=1e7, SQ_INSTS_VMEM_RD=8e7, SQ_INSTS_VMEM_WR=2e7, SQ_INSTS_LDS=6e7)
    r = b.kernel_block(k, ms=6.0, segments=1.0e8)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["traffic"] == int(9.0e6 * 1024)
    assert abs(r["achieved"] - 9.0e6 * 1024 / 6.0e-3 / 1e9) < 0.1
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4 and r["frac"] < 1.0
    assert abs(r["wave_instr_per_segment"] - 51.7) < 0.1
    assert r["useful_lane_frac"] == 0.34 and r["l2_hit_rate"] == 0.74 and r["wait_frac"] == 0.52
    assert "issue_diag" in r and "frac" not in r["issue_diag"]


def test_vector_l1_block_prices_scattered_loads_against_the_measured_ceiling():
    """roofline.vector_l1: lane-loads per CU-clock = TCP_TOTAL_CACHE_ACCESSES / (time x 256 CUs x clock)
    against the harmonic mix of ~1.1 loads per clock from the L1 and 2 clocks per L1 miss (tools/gather_rate.hip); the
    limiter names that path where the fraction is >= 0.7."""
    b = _bench()
    k = {"name": "trace_kernel<3,0,0,0>", "avg_ms": 28.0, "TCP_TOTAL_CACHE_ACCESSES_sum": 13.9e9, "l1_miss_per_access": 0.24,
         "clock_ghz": 2.27}
    blk = b.kernel_block(k, 28.0, segments=189_000_000)
    v = blk["vector_l1"]
    rate = 13.9e9 / (28.0e-3 * 256 * 2.27e9)
    ceil = 1.0 / (0.76 / b.VL1_HIT_LOADS_PER_CLOCK + 0.24 * b.VL1_MISS_CLOCKS)
    assert abs(v["per_cu_clock"] - rate) < 1e-3 and abs(v["ceiling_of_this_hit_miss_mix"] - ceil) < 1e-3
    assert abs(v["frac"] - rate / ceil) < 2e-3 and 0.9 < v["frac"] < 1.1
    assert abs(v["lane_loads_per_segment"] - 13.9e9 / 189e6) < 0.1
    # a kernel whose rays share addresses is far from that ceiling
    k2 = dict(k, TCP_TOTAL_CACHE_ACCESSES_sum=0.9e9, avg_ms=5.5)
    assert b.kernel_block(k2, 5.5)["vector_l1"]["frac"] < 0.5


def test_single_gpu_step_ends_with_the_flux_on_the_host():
    """N = 1: a timed step is the reference's whole apply() — through the float flux in the host's TracingData
    (rayTraceDisk.hpp:40-57, rayTrace.hpp:135) — i.e. `Trace.apply()`, which downloads; N > 1: the shard + all-reduce on
    the device-resident accumulators (round-3 verdict, item 3)."""
    b = _bench()
    calls = []

    class FakeTracer:
        def setRunNumber(self, n): calls.append(("run", n))
        def setRayRange(self, a, c): calls.append(("range", a, c))
        def apply(self): calls.append(("apply",))            # vr_apply + the download into TracingData
        def applyFinish(self, collect=True): calls.append(("finish", collect))
        def getRayTraceInfo(self): return "info"

    step, end_state = b.make_step(FakeTracer(), None, 1000, 0, 1)
    info, counters = step()
    assert ("apply",) in calls and not any(c[0] == "finish" for c in calls)
    assert info == "info" and counters["allreduce_ms"] == 0.0 and "host" in end_state

    shard = b.StubShard(100)
    step, end_state = b.make_step(None, shard, 1000, 0, 2)   # (world 2 without a process group: never called here)
    assert "device" in end_state


def test_a_failing_secondary_shows_in_the_exit_code():
    b = _bench()
    ok = dict(name="C4", counter_diff={"totalRaysTraced": 0, "geometryHits": 0}, flux_l2_rel_err=3e-9)
    assert b.secondary_failures([ok, dict(name="x", rays=1)]) == []
    bad = b.secondary_failures([ok, dict(name="C5p", error="boom"), dict(name="C2", counter_diff={"geometryHits": 1}, flux_l2_rel_err=0.0),
                                dict(name="C1", counter_diff={"geometryHits": 0}, flux_l2_rel_err=2e-4)])
    assert len(bad) == 3 and bad[0].startswith("C5p") and "counters differ" in bad[1] and "flux" in bad[2]

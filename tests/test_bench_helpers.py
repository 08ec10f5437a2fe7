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

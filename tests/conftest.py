import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# torch BEFORE the tracing library: a few GPU tests drive the library through torch tensors (the multi-GPU driver), and a
# process in which libviennaray_amd.so has already brought up its HIP runtime when torch initialises its own finds "No HIP
# GPUs" — the other order shares one runtime (bench.py imports torch first for the same reason).  The product needs no torch.
try:
    import torch  # noqa: F401
except Exception:  # noqa: BLE001 (the CPU-only tests do not need it)
    pass

DATA = os.path.join(ROOT, "tests", "golden", "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def data_dir():
    return DATA


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

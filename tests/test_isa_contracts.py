"""Contracts on the emitted gfx950 ISA that the kernels' correctness arguments rely on (no GPU
needed: hipcc cross-compiles to assembly here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "viennaray_amd", "csrc")


@pytest.fixture(scope="module")
def setup_asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "vr_setup.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-fno-fast-math", "-S", "--cuda-device-only", "-o", str(out),
                           os.path.join(CSRC, "vr_setup.hip")], stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernel_body(asm, name):
    m = re.search(r"^(_ZN2vr\d*%s\w*):\s*(?:;.*)?$" % name, asm, flags=re.M)
    assert m, name
    end = asm.index("s_endpgm", m.end())
    return asm[m.end():end]


def test_fit_kernel_drains_its_stores_before_the_arrival_atomic(setup_asm):
    """fit_kernel hands a node's box from one arriver to the other through sc1 stores + an arrival
    counter: every store must be acknowledged (s_waitcnt vmcnt(0)) before the counter is bumped,
    with no vector-memory instruction in between (the ordering must be explicit)."""
    body = _kernel_body(setup_asm, "10fit_kernel")
    lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((";", "."))]
    atomics = [i for i, l in enumerate(lines) if l.startswith("global_atomic_add")]
    assert atomics, "no arrival atomic found"
    for i in atomics:
        j = i - 1
        while j >= 0 and not lines[j].startswith("s_waitcnt"):
            assert not lines[j].startswith(("global_", "buffer_", "flat_")), (lines[j], "between wait and atomic")
            j -= 1
        assert j >= 0 and "vmcnt(0)" in lines[j], lines[max(0, i - 6):i + 1]
    # the published values are written through (sc1) and read below the L1 (sc1)
    assert any(l.startswith("global_store") and "sc1" in l for l in lines)
    assert any(l.startswith("global_load") and "sc1" in l for l in lines)

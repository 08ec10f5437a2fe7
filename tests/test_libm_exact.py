"""vr_libm.hpp (the device's sincosf / powf) against the running glibc: the host build
of the same header must agree bit for bit on a dense sample of the ranges the tracer
uses (tests/aux/libm_check.cpp without argument runs every float: 1.09e9 + 8 x 1.06e9
inputs, 0 mismatches on glibc 2.35 / x86-64 FMA)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_libm_matches_glibc(tmp_path):
    exe = str(tmp_path / "libm_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-mfma",
                           os.path.join(ROOT, "tests", "aux", "libm_check.cpp"), "-o", exe])
    out = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sin mismatches 0, cos mismatches 0" in out.stdout
    assert "beyond 1 ulp 0, float-narrowed mismatches 0" in out.stdout

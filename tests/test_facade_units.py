"""Known answers of the reference's API-level tests that need no ray tracing
(tests/tracingData, tests/particle, tests/createGeometry, tests/linesToTriangles,
tests/utilFuncs), on the Python mirror, the C++ façade and the oracle's helpers."""
import os
import subprocess

import numpy as np
import pytest

import viennaray_amd as vr
from helpers import DATA, ROOT, sphere3d
from oracle import pyoracle as po


def test_tracing_data_like_reference():
    """tests/tracingData/tracingData.cpp:8-31"""
    d = vr.TracingData()
    d.setNumberOfScalarData(1)
    d.setNumberOfVectorData(1)
    assert d.getScalarDataLabel(0) == "scalarData" and d.getVectorDataLabel(0) == "vectorData"
    d.setVectorData(0, 1000, "zeroData", value=0.0)
    assert d.getVectorDataLabel(0) == "zeroData" and d.getVectorData("zeroData").size == 1000
    d.setScalarData(0, 1, "oneData")
    assert d.getScalarDataLabel(0) == "oneData" and d.getScalarData("oneData") == 1
    d.resizeAllVectorData(10, 0.5)
    assert d.getVectorData(0).tolist() == [0.5] * 10
    assert d.getVectorMergeType(0) == vr.trace.TracingDataMergeEnum.SUM
    with pytest.raises(KeyError):
        d.getVectorData("nope")


def test_particles_like_reference():
    """tests/particle/particle.cpp:12-40"""
    p = vr.DiffuseParticle(1.0, "test")
    assert p.getSourceDistributionPower() == 1.0 and p.getLocalDataLabels() == ["test"]
    q = vr.SpecularParticle(1.0, 50.0, "test")
    assert q.getSourceDistributionPower() == 50.0 and q.getLocalDataLabels() == ["test"]


def test_create_geometry_bounding_box():
    """tests/createGeometry/createGeometry.cpp:23-33: the unit sphere grid spans [-1, 1]^3"""
    gd, p, n = sphere3d()
    o = po.Oracle()
    o.set_disks(p, n, gd, 3)
    o.prepare()
    assert np.allclose(p.min(0), -1.0, atol=1e-6) and np.allclose(p.max(0), 1.0, atol=1e-6)
    bb = np.asarray(o.bbox()).reshape(2, 3)  # adjusted: the source side grows by 2 * radius
    assert (bb[0] < bb[1]).all()
    assert np.allclose(bb[0, :2], -1.0, atol=1e-6) and np.allclose(bb[1, :2], 1.0, atol=1e-6)


def test_line_mesh_conversion():
    """tests/linesToTriangles + rayMesh.hpp:27-80,133-175 on the reference's lineMesh.dat"""
    gd, nodes, lines = vr.io.read_line_mesh(os.path.join(DATA, "lineMesh.dat"))
    assert nodes.shape == (131, 3) and lines.shape == (130, 2) and abs(gd - 0.2) < 1e-7
    v, tri, keep = vr.io.lines_to_triangles(nodes, lines, gd)
    assert keep.size == 128 and v.shape == (262, 3) and tri.shape == (256, 3)
    assert v[0, 2] == np.float32(gd) * np.float32(0.5) and v[1, 2] == -v[0, 2]
    assert (tri[0] == [2 * lines[keep[0], 0], 2 * lines[keep[0], 1], 2 * lines[keep[0], 0] + 1]).all()


def test_cpp_facade_units(tmp_path):
    """the same known answers on the header-only C++ façade (links without the HIP library:
    nothing in it instantiates Trace::apply), plus the VTK / VTP writers byte for byte
    (rayUtil.hpp:413-555)"""
    exe = tmp_path / "facade_units"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include", "viennaray_amd"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "aux", "facade_units.cpp"),
                           "-o", str(exe)])
    out = subprocess.run([str(exe), os.path.join(DATA, "lineMesh.dat"), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "facade units ok" in out.stdout, out.stdout + out.stderr


def test_cpp_facade_interface_builds():
    """tests/traceInterface/traceInterface.cpp (custom Source subclass, full AbstractParticle / Source virtual
    signatures, SourceGrid helper chain, plug-in particles) compiles and links against the façade"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), os.path.join(ROOT, "tests", "aux", "facade_interface")],
                          stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(ROOT, "tests", "aux", "facade_interface"))


@pytest.mark.gpu
def test_cpp_facade_interface_runs():
    exe = os.path.join(ROOT, "tests", "aux", "facade_interface")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "facade interface ok" in out.stdout, out.stdout + out.stderr
    assert "numRays 4410" in out.stdout                      # traceInterface.cpp:67
    assert "no device model" in out.stderr                    # the host-only particle was refused, not replaced

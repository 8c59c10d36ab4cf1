"""include/HRBFFusion.h compiles with a plain host C++ compiler against the C-ABI, its exporters write
the reference's formats, and the constructor fails loudly (exception, not exit) without a GPU."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp):
    from hrbffusion3d_amd import build
    so = build.build()
    exe = os.path.join(tmp, "shim_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_test.cpp"), "-o", exe, so,
                           "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_shim_compiles_and_writes_reference_formats(tmp_path):
    exe = _build(str(tmp_path))
    out = subprocess.run([exe, "cpu", str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "NOGPU-OK" in out.stdout or "GPU-OK" in out.stdout
    tum = open(tmp_path / "t.freiburg").read().split("\n")
    assert tum[0].split() == ["1.000000", "0", "0", "0", "0", "0", "0", "1"]
    v = [float(x) for x in tum[1].split()]
    assert v[0] == 2.5 and v[1:4] == [1.0, 2.0, 3.0]
    assert abs(v[6] - 2 ** -0.5) < 1e-6 and abs(v[7] - 2 ** -0.5) < 1e-6
    icl = open(tmp_path / "t_icl.freiburg").read().split("\n")[1].split()
    assert icl[0] == "2500000" and float(icl[2]) == -2.0           # integer stamp, ty negated
    log = open(tmp_path / "t.log").read().split("\n")
    assert log[0] == "0 0 1" and log[5] == "1 1 2" and log[6].split()[3] == "1.000000"
    lef = open(tmp_path / "t_lef.txt").read().split("\n")[1].split()
    assert lef[0] == "1" and [float(x) for x in lef[13:16]] == [1.0, 2.0, 3.0]


@pytest.mark.gpu
def test_shim_process_frame_and_ply_on_gpu(tmp_path, gpu_available):
    exe = _build(str(tmp_path))
    out = subprocess.run([exe, "gpu", str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "GPU-OK" in out.stdout, out.stdout + out.stderr
    raw = open(tmp_path / "m.ply", "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    n = int([l for l in head.split(b"\n") if l.startswith(b"element vertex")][0].split()[-1])
    assert n > 1000 and len(body) == n * (12 + 3 + 12 + 16)
    rec = struct.unpack("<fff3Bfffffff", body[:43])
    assert 1.0 < rec[2] < 2.0                                     # z of the 1.5 m plane
    assert rec[8] < 0                                             # normals negated on export
    traj = open(tmp_path / "run.freiburg").read().strip().split("\n")
    assert len(traj) == 2


def _build_reference_caller(tmp):
    from hrbffusion3d_amd import build
    so = build.build()
    exe = os.path.join(tmp, "reference_caller_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "reference_caller_test.cpp"), "-o", exe, so,
                           "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _reference_caller_files(tmp_path):
    (tmp_path / "cam.yaml").write_text("%YAML:1.0\nCamera.fx: 132.0\nCamera.fy: 132.0\nCamera.cx: 80.0\nCamera.cy: 60.0\n"
                                       "Camera.width: 160\nCamera.height: 120\nCamera.RGB: 1\nDepthMapFactor: 5000.0\n")
    (tmp_path / "GlobalStateParam.txt").write_text('parameterFileCvFormat = "%s";\nsensorType = 3;\nglobalConfidenceThreshold = 5.0;\n'
                                                   'globalDepthCutoff = 3.5;\nregistrationJointICPWeight = 10;\nregistrationPreAlignSO3 = true;\n'
                                                   'globalStartFrame = 0;\nglobalEndFrame = -1;\n' % (tmp_path / "cam.yaml"))
    return str(tmp_path / "GlobalStateParam.txt")


def test_reference_caller_compiles_with_only_the_include_changed(tmp_path):
    """MainController's constructor + run() statements (GUI/src/HRBF_fusion.cpp:35-54,87-100,174-181,190-239): ParameterFile ->
    GlobalStateParam singleton, Resolution / Intrinsics singletons, `new HRBFFusion(<the reference's eight arguments>)`, the
    start / skip / processFrame loop — against include/HRBFFusion.h, names at global scope; without a GPU the constructor throws"""
    exe = _build_reference_caller(str(tmp_path))
    out = subprocess.run([exe, _reference_caller_files(tmp_path), str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "NOGPU-OK" in out.stdout or "GPU-OK" in out.stdout


@pytest.mark.gpu
def test_reference_caller_runs_on_gpu(tmp_path, gpu_available):
    exe = _build_reference_caller(str(tmp_path))
    out = subprocess.run([exe, _reference_caller_files(tmp_path), str(tmp_path), "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "GPU-OK" in out.stdout, out.stdout + out.stderr
    assert "constructed: 160 x 120, tick 1" in out.stdout
    traj = open(tmp_path / "ref_caller.freiburg").read().strip().split("\n")
    assert len(traj) == 4
    head = open(tmp_path / "ref_caller.ply", "rb").read(300)
    assert b"element vertex" in head

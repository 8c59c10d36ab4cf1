"""The reference's configuration formats (hrbffusion3d_amd/config.py): ParameterFile rules on a hand-written file with
every quirk the parser has (comment markers inside / outside quotes, trailing `;`, `6.0` read as an int, later lines
overriding earlier ones), the OpenCV camera YAML, and — when /root/reference is present (this container, never the GPU
box) — the reference's own GUI/GlobalStateParam.txt against the library defaults."""
import os

import pytest

from hrbffusion3d_amd import config as hcfg
from hrbffusion3d_amd.params import default_params

PARAM_TXT = '''##!!!! header comment
currentWorkingDirectory = "/data/sets/TUM # not a comment; neither is this // one/fr1_desk"
##sensortype : 1 for realtime device, 2 for off-line klg data, 3 for association file
sensorType = 3;
klgFileName = "fr1_desk.klg";
AssociationFile = "associations.txt";
parameterFileCvFormat = "TUM1.yaml";
optimizationUseLocalBA =  true;    //local  BA
optimizationUseGlobalBA = False;
preprocessingUsebilateralFilter = 0;
preprocessingCurvValidThreshold = 250;
preprocessingUseConfEval = 1.0;
registrationJointICPWeight = 12.5;  
registrationICPNeighborSearchRadius = 3.0;
registrationICPErrorThreshold = 5e-05;
preictionMinNeighbors = 4.0;       //!!!minimun number of kernels
preictionMaxNeighbors = 8.0;
globalConfidenceThreshold = 4.0;
globalConfidenceThreshold = 6.5;       # the later line wins
globalDepthCutoff = 3.0;
globalInputTrajectoryFormat = "TUM" #save this later by a judgement statement
line without a separator
 = novalue;
globalEndFrame = 120;
'''

CAMERA_YAML = '''%YAML:1.0

# Camera Parameters. Adjust them!
Camera.fx: 517.306408
Camera.fy: 516.469215
Camera.cx: 318.643040
Camera.cy: 255.313989
Camera.k1: 0.262383
Camera.width: 640
Camera.height: 480
Camera.fps: 30.0
Camera.RGB: 1
ThDepth: 40.0
DepthMapFactor: 5000.0
ORBextractor.nFeatures: 1000
'''


def test_parameter_file_rules(tmp_path):
    f = tmp_path / "GlobalStateParam.txt"
    f.write_text(PARAM_TXT)
    raw = hcfg.parse_parameter_file(str(f))
    assert raw["currentWorkingDirectory"] == "/data/sets/TUM # not a comment; neither is this // one/fr1_desk"
    assert raw["sensorType"] == "3" and raw["klgFileName"] == "fr1_desk.klg"
    assert raw["globalInputTrajectoryFormat"] == "TUM"
    assert "line without a separator" not in raw and "" not in raw
    g = hcfg.load_global_state(str(f))
    assert g["sensorType"] == 3 and g["optimizationUseLocalBA"] is True and g["optimizationUseGlobalBA"] is False
    assert g["preprocessingUsebilateralFilter"] is False
    assert g["preprocessingUseConfEval"] == 1 and g["registrationICPNeighborSearchRadius"] == 3     # std::stoi("3.0")
    assert g["registrationICPErrorThreshold"] == pytest.approx(5e-5)
    assert g["preictionMinNeighbors"] == 4 and g["preictionMaxNeighbors"] == 8
    assert g["globalConfidenceThreshold"] == 6.5
    assert "registrationPreAlignSO3" not in g          # absent fields stay absent
    kw = hcfg.hrbf_kwargs(g)
    assert kw == {"confidence_threshold": 6.5, "depth_cutoff": 3.0, "icp_weight": 12.5, "use_bilateral": 0,
                  "curv_valid_threshold": 250.0, "use_conf_eval": 1, "icp_search_radius": 3, "predict_min_neighbors": 4,
                  "predict_max_neighbors": 8}
    p = default_params(640, 480, 517.3, 516.5, 318.6, 255.3, **kw)
    assert p.use_bilateral == 0 and p.icp_weight == 12.5 and p.so3 == 1


def test_camera_yaml(tmp_path):
    f = tmp_path / "TUM1.yaml"
    f.write_text(CAMERA_YAML)
    cam = hcfg.camera_from_yaml(str(f))
    assert (cam["width"], cam["height"]) == (640, 480)
    assert cam["fx"] == pytest.approx(517.306408) and cam["fy"] == pytest.approx(516.469215)
    assert cam["cx"] == pytest.approx(318.643040) and cam["cy"] == pytest.approx(255.313989)
    assert cam["depth_scale"] == pytest.approx(1.0 / 5000.0) and cam["rgb"] == 1
    (tmp_path / "nofactor.yaml").write_text(CAMERA_YAML.replace("DepthMapFactor: 5000.0", "DepthMapFactor: 0"))
    assert hcfg.camera_from_yaml(str(tmp_path / "nofactor.yaml"))["depth_scale"] == 1.0     # HRBFFusion.cpp:776-777
    (tmp_path / "bad.yaml").write_text("Camera.fx: 1.0\n")
    with pytest.raises(ValueError):
        hcfg.camera_from_yaml(str(tmp_path / "bad.yaml"))


def test_run_cli_reads_the_reference_configuration(tmp_path):
    """run.py --config: frame source, camera file, intrinsics with fx != fy and the tunables all come from the files"""
    from hrbffusion3d_amd import run
    (tmp_path / "GlobalStateParam.txt").write_text(PARAM_TXT.replace("/data/sets/TUM # not a comment; neither is this // one/fr1_desk",
                                                                      "/nonexistent/on/this/machine"))
    (tmp_path / "TUM1.yaml").write_text(CAMERA_YAML)
    args = run.parse(["--config", str(tmp_path / "GlobalStateParam.txt")])
    assert args.tum == str(tmp_path) and args.assoc_name == "associations.txt" and args.klg is None
    assert (args.width, args.height) == (640, 480) and args.depth_factor == pytest.approx(5000.0)
    assert args.fx == pytest.approx(517.306408) and args.fy == pytest.approx(516.469215)
    assert args.param_overrides["icp_weight"] == 12.5 and args.end_tick == 120 and args.start_frame == 0   # globalEndFrame bounds the TICK (HRBF_fusion.cpp:98-100)
    args = run.parse(["--config", str(tmp_path / "GlobalStateParam.txt"), "--fx", "500", "--data-dir", str(tmp_path)])
    assert args.fx == 500.0 and args.fy == pytest.approx(516.469215)       # explicit flags win
    args = run.parse(["--synthetic", "3"])
    assert (args.width, args.height, args.fx, args.depth_factor) == (640, 480, 528.0, 5000.0)


REF = "/root/reference/GUI/GlobalStateParam.txt"


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference tree exists only in the build container")
def test_reference_parameter_file_matches_library_defaults():
    g = hcfg.load_global_state(REF)
    assert g["sensorType"] == 3 and g["parameterFileCvFormat"] == "TUM1.yaml"
    assert g["currentWorkingDirectory"].endswith("rgbd_dataset_freiburg1_desk")
    kw = hcfg.hrbf_kwargs(g)
    d = default_params()
    for k, v in kw.items():
        assert getattr(d, k) == pytest.approx(v), k       # the library's defaults ARE the reference's file
    assert len(kw) >= 22


# ---- against the reference's OWN reader (oracle/_ref/ref_params, built from Core/src/Utils/{GlobalStateParams,parameterFile}.h) ----
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_params")
REF_READER = os.path.join(os.path.dirname(GOLD), "..", "..", "oracle", "_ref", "ref_params")


def _as_text(typ, v):
    """a member as the reference's `std::cout << std::setprecision(9) << member` prints it"""
    import numpy as np
    if typ == "bool":
        return str(int(bool(v)))
    if typ == "int":
        return str(int(v))
    if typ == "float":
        return "%.9g" % np.float32(v)
    return v


def _cases():
    import json
    import sys
    sys.path.insert(0, GOLD)
    import cases
    with open(os.path.join(GOLD, "expected.json")) as f:
        return cases.FILES, json.load(f)


def test_parameter_readers_agree_with_the_reference_reader(tmp_path):
    """every member the reference's reader fills from a file with unusual spellings (`TRUE`, `yes`, `7.9` / `0x10` / `12abc` / `1e2` as
    ints, `1,5`, `2.5f`, single quotes, unquoted strings, `=` inside a value) and an unusual layout (two statements on a line, repeated
    keys, missing `;`, unterminated quote, CRLF, tabs, keys in another case) — hrbffusion3d_amd/config.py and include/hrbf_io.h
    (through tools/hrbf_run --dump-params) read the same value."""
    import subprocess
    files, exp = _cases()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "hrbf_dump")
    from hrbffusion3d_amd import build
    so = build.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "tools", "hrbf_run.cpp"), "-o", exe, so, "-lz",
                           "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib"])
    checked = 0
    for name, txt in files.items():
        path = str(tmp_path / (name + ".txt"))
        with open(path, "w", newline="") as f:
            f.write(txt)
        g = hcfg.load_global_state(path)
        out = subprocess.run([exe, "--dump-params", "--config", path], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        cpp = {}
        for line in out.stdout.split("\n"):
            if "\t" in line:
                n, v = line.split("\t", 1)
                cpp[n] = v[1:-1] if v.endswith("]") else v[1:]
        assert len(exp[name]) >= 18
        for member, (typ, want) in exp[name].items():
            if member in g:
                assert _as_text(typ, g[member]) == want, (name, member, "python")
                checked += 1
            if member in cpp:
                assert cpp[member] == want, (name, member, "c++")
                checked += 1
        if os.path.exists(REF_READER):      # the fixture is what the reference's reader prints today
            import sys
            sys.path.insert(0, os.path.dirname(GOLD))
            import make_ref_params as M
            ref = M.run_reference_reader(path)
            assert {n: list(tv) for n, tv in ref.items() if n in exp[name]} == exp[name]
    assert checked >= 110


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference tree exists only in the build container")
def test_shipped_parameter_file_read_like_the_reference_reader_reads_it():
    _, exp = _cases()
    g = hcfg.load_global_state(REF)
    want = exp["shipped GUI/GlobalStateParam.txt"]
    n = 0
    for member, (typ, text) in want.items():
        if member in g:
            assert _as_text(typ, g[member]) == text, member
            n += 1
    assert n >= 40

"""tools/compare_reference_dump.py on dumps written in the REFERENCE's formats (text via C++ ostream default formatting
= %g with 6 significant digits; binary PLY of HRBFFusion::savePly; TUM trajectory) from the oracle's own state for the
GPUTest PNG pair: the tool must report a match, and must notice a single altered value or a missing row.  This is the
path by which a real reference dump would pin the oracle."""
import importlib.util
import os
import shutil
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("compare_reference_dump", os.path.join(ROOT, "tools", "compare_reference_dump.py"))
crd = importlib.util.module_from_spec(spec); spec.loader.exec_module(crd)


def _g(x):
    return "%g" % x          # std::ostream << float, default precision 6


def _write_table(path, rows, double_space_after=None):
    with open(path, "w") as f:
        for r in rows:
            t = [_g(v) for v in r]
            if double_space_after is not None:
                t[double_space_after] = t[double_space_after] + " "
            f.write(" ".join(t) + "\n")


def _write_model_ply(path, rows):
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z"
                 "\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty float nx\nproperty float ny"
                 "\nproperty float nz\nproperty float curvature_max\nproperty float curvature_min\nproperty float radius"
                 "\nproperty float submapIndex\nend_header\n" % len(rows)).encode())
        for r in rows:
            f.write(struct.pack("<3f3B7f", r[0], r[1], r[2], int(r[3]), int(r[4]), int(r[5]), *[float(v) for v in r[6:]]))


def test_tool_matches_reference_format_dumps_and_detects_changes(tmp_path, oracle_lib_built, png_pair):
    from hrbffusion3d_amd import io as hio
    from hrbffusion3d_amd.params import default_params
    gold = os.path.join(ROOT, "tests", "golden")
    for n in ("1c", "1d", "2c", "2d"):
        shutil.copy(os.path.join(gold, n + ".png"), tmp_path / (n + ".png"))
    (tmp_path / "associations.txt").write_text("0.000000 1d.png 0.000000 1c.png\n0.033333 2d.png 0.033333 2c.png\n")
    (tmp_path / "cam.yaml").write_text("%YAML:1.0\nCamera.fx: 528.0\nCamera.fy: 528.0\nCamera.cx: 320.0\nCamera.cy: 240.0\n"
                                       "Camera.width: 640\nCamera.height: 480\nCamera.RGB: 1\nDepthMapFactor: 5000.0\n")
    # the "reference run": here the oracle, dumped in the reference's formats
    o = oracle_lib_built.Oracle(default_params(max_surfels=1 << 20), omp=True)
    poses = []
    for k, (rgb, d) in enumerate(png_pair):
        o.process_frame(rgb, d, int(round(k * 33333)))
        poses.append(o.get_pose())
    _write_table(tmp_path / "rawMap_attributes.txt", crd.ours_raw_map_attributes(o))
    hrbf, surfel = crd.ours_prediction(o, poses[-1])
    _write_table(tmp_path / "prediction_hrbf_2.txt", hrbf, double_space_after=5)
    _write_table(tmp_path / "prediction_surfel_2.txt", surfel)
    _write_model_ply(tmp_path / "hrbf_globalModel.ply", crd.model_ply_rows(o.download_map()))
    hio.save_trajectory(str(tmp_path / "hrbf_trajectory.freiburg"), poses, stamps_us=[0, 33333], fmt="TUM")
    n_raw = crd.ours_raw_map_attributes(o).shape[0]
    o.close()
    assert n_raw > 100_000 and hrbf.shape[0] > 100_000
    argv = ["--dumps", str(tmp_path), "--frames", str(tmp_path / "associations.txt"), "--camera", str(tmp_path / "cam.yaml"),
            "--json", str(tmp_path / "report.json")]
    assert crd.main(argv) == 0
    import json
    rep = json.load(open(tmp_path / "report.json"))
    names = {f["file"]: f for f in rep["files"]}
    assert set(names) == {"rawMap_attributes.txt", "prediction_hrbf_2.txt", "prediction_surfel_2.txt", "hrbf_globalModel.ply",
                          "hrbf_trajectory.freiburg"}
    assert all(f["match"] for f in rep["files"]) and rep["all_match"]
    assert names["rawMap_attributes.txt"]["rows_reference"] == n_raw
    # one altered value (1 mm on one vertex) and one dropped row are both noticed
    lines = (tmp_path / "prediction_hrbf_2.txt").read_text().splitlines()
    t = lines[1000].split(); t[2] = _g(float(t[2]) + 1e-3); lines[1000] = " ".join(t)
    (tmp_path / "prediction_hrbf_2.txt").write_text("\n".join(lines) + "\n")
    raw = (tmp_path / "rawMap_attributes.txt").read_text().splitlines()
    (tmp_path / "rawMap_attributes.txt").write_text("\n".join(raw[:500] + raw[501:]) + "\n")
    assert crd.main(argv) == 1
    rep = json.load(open(tmp_path / "report.json"))
    names = {f["file"]: f for f in rep["files"]}
    assert names["prediction_hrbf_2.txt"]["rows_differing"] == 1 and names["prediction_hrbf_2.txt"]["first_difference"]["row"] == 1000
    assert not names["rawMap_attributes.txt"]["match"] and names["rawMap_attributes.txt"]["reason"] == "row count differs"
    assert names["hrbf_globalModel.ply"]["match"]

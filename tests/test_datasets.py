"""hrbffusion3d_amd/datasets.py on the CPU: the two datasets' native layouts written from the synthetic stream, prepared for the
reference's caller loop and read back by BOTH readers (Python: config.py / io.py, C++: include/hrbf_io.h through
`hrbf_run --selftest`); the benchmark tools' association and ATE rules; the settings table held to the reference's own
GUI/GlobalStateParam.txt where the reference checkout is present.  The GPU twins are tests/test_datasets_gpu.py."""
import json
import os
import subprocess

import numpy as np
import pytest

from hrbffusion3d_amd import config as hcfg
from hrbffusion3d_amd import datasets as ds
from hrbffusion3d_amd import io as hio
from hrbffusion3d_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PARAM = "/root/reference/GUI/GlobalStateParam.txt"


def _exe(tmp):
    from test_cpp_io import _build
    return _build(str(tmp))


def _fnv(b):
    from test_cpp_io import _fnv as f
    return f(b)


def _small_frames(n, K, W=64, H=48):
    return [synth.frame(k, W, H, noise=True, K=K) for k in range(n)]


@pytest.mark.skipif(not os.path.isfile(REF_PARAM), reason="the reference checkout is not on this machine")
def test_settings_table_is_the_references_own_parameter_file():
    """every key of GUI/GlobalStateParam.txt, parsed by the ParameterFile rules, against datasets.REFERENCE_GUI_SETTINGS parsed the
    same way from a file written by write_global_state: equal, except the keys that name the author's machine / sequence"""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        ds.write_global_state(os.path.join(d, "p.txt"))
        mine = hcfg.parse_parameter_file(os.path.join(d, "p.txt"))
    ref = hcfg.parse_parameter_file(REF_PARAM)
    assert set(mine) == set(ref), (sorted(set(mine) ^ set(ref)))
    for k in ref:
        if k in ("currentWorkingDirectory", "optimizationVocabularyFile"):
            continue
        assert mine[k] == ref[k], (k, mine[k], ref[k])
    # and typed: what load_global_state makes of both
    a, b = hcfg.load_global_state(REF_PARAM), None
    with tempfile.TemporaryDirectory() as d:
        ds.write_global_state(os.path.join(d, "p.txt"))
        b = hcfg.load_global_state(os.path.join(d, "p.txt"))
    for k, v in a.items():
        if k not in ("currentWorkingDirectory", "optimizationVocabularyFile"):
            assert b[k] == v, k


def test_write_global_state_refuses_unknown_keys(tmp_path):
    with pytest.raises(KeyError):
        ds.write_global_state(str(tmp_path / "p.txt"), {"globalDepthCutOff": "3.0"})


def test_associate_is_the_benchmark_rule():
    """best differences first, every stamp once, strictly below 20 ms — against a brute-force statement of associate.py"""
    rng = np.random.default_rng(5)
    a = np.sort(rng.uniform(0, 3, 70)); b = np.sort(a[rng.permutation(70)[:55]] + rng.uniform(-0.03, 0.03, 55))
    got = ds.associate([(s, None) for s in a], [(s, None) for s in b])
    cand = sorted((abs(x - y), i, j) for i, x in enumerate(a) for j, y in enumerate(b) if abs(x - y) < 0.02)
    ua, ub, want = set(), set(), []
    for _, i, j in cand:
        if i not in ua and j not in ub:
            ua.add(i); ub.add(j); want.append((i, j))
    assert got == sorted(want) and 20 < len(got) < 55
    assert ds.associate([(0.0, None)], [(0.02, None)]) == [] and ds.associate([(0.0, None)], [(0.0199, None)]) == [(0, 0)]


def test_associate_against_the_benchmarks_dictionary_statement_on_random_lists():
    """associate.py itself works on the dictionaries' keys: all pairs below the limit, sorted as (difference, first stamp, second stamp),
    taken greedily while both stamps are still free.  Random lists — sorted or not, with an offset, with stamps on a coarse grid so
    that equal differences (ties) are common — against that statement, by stamps instead of indices."""
    from hypothesis import given, settings, strategies as st

    grid = st.integers(0, 400).map(lambda k: k * 0.005)          # 5 ms grid: many equal differences

    @settings(max_examples=300, deadline=None)
    @given(st.lists(grid, min_size=0, max_size=40, unique=True), st.lists(grid, min_size=0, max_size=40, unique=True),
           st.sampled_from([0.0, 0.0025, -0.01]), st.sampled_from([0.02, 0.011, 0.0051]), st.booleans())
    def check(a, b, offset, max_dt, keep_order):
        if not keep_order:
            a, b = sorted(a), sorted(b)
        got = ds.associate([(s, None) for s in a], [(s, None) for s in b], max_dt=max_dt, offset=offset)
        fa, fb = list(a), list(b)
        pot = sorted((abs(x - (y + offset)), x, y + offset, y) for x in fa for y in fb if abs(x - (y + offset)) < max_dt)
        want = []
        for _, x, _, y in pot:
            if x in fa and y in fb:
                fa.remove(x); fb.remove(y); want.append((x, y))
        assert sorted((a[i], b[j]) for i, j in got) == sorted(want)
        assert got == sorted(got)

    check()


def test_ate_is_what_the_benchmark_defines_under_any_rigid_motion_of_the_estimate():
    """evaluate_ate.py aligns with Horn's closed form before it measures: a rigidly moved copy of the ground truth scores 0, the score
    of a noisy estimate does not change when the whole estimate is moved rigidly, equals the residual of an independent least-squares
    alignment (scipy's Rotation.align_vectors on the centred points), and a MIRRORED trajectory is not aligned away (proper rotations
    only — what makes the ICL-NUIM sign convention visible, see the ICL test below)."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(17)
    for trial in range(20):
        n = int(rng.integers(12, 80))
        t = 0.1 * np.arange(n) + rng.uniform(0, 0.01, n)          # frames 90-110 ms apart: a 4 ms offset cannot change a pairing
        G = np.cumsum(rng.normal(0, 0.05, (n, 3)), 0)
        gp = [np.eye(4) for _ in range(n)]
        for k in range(n):
            gp[k][:3, 3] = G[k]

        def moved(P, R, d):
            out = []
            for k in range(n):
                T = np.eye(4); T[:3, 3] = R @ P[k] + d; out.append(T)
            return out
        R = Rotation.random(random_state=int(rng.integers(1 << 30))).as_matrix(); d = rng.normal(0, 3, 3)
        assert ds.evaluate_ate(t, moved(G, R, d), t, gp)["rmse_m"] < 1e-9
        E = G + rng.normal(0, 0.01, (n, 3))
        e0 = ds.evaluate_ate(t, moved(E, np.eye(3), np.zeros(3)), t, gp)
        e1 = ds.evaluate_ate(t + 0.004, moved(E, R, d), t, gp)              # 4 ms off: still the same pairs
        assert e0["pairs"] == e1["pairs"] == n and abs(e0["rmse_m"] - e1["rmse_m"]) < 1e-9
        rot, _ = Rotation.align_vectors(G - G.mean(0), E - E.mean(0))
        want = np.sqrt((((E - E.mean(0)) @ rot.as_matrix().T - (G - G.mean(0))) ** 2).sum(1).mean())
        assert abs(e0["rmse_m"] - want) < 1e-6
        M = G * np.array([1.0, -1.0, 1.0])
        assert ds.evaluate_ate(t, moved(M, np.eye(3), np.zeros(3)), t, gp)["rmse_m"] > 0.02
    assert ds.evaluate_ate([0.0, 1.0], moved(G, np.eye(3), np.zeros(3))[:2], [0.0, 1.0], gp[:2])["rmse_m"] is None      # fewer than 3 pairs


def test_frame_stamp_is_truncated_like_the_references_reader():
    """int64_t(t * 1000000.0) (GUI/src/Tools/RawImageReader.cpp:93): 0.000249 -> 248, not 249"""
    assert ds.reference_frame_stamp(0.000249) == 248 and ds.reference_frame_stamp(1305031102.175304) == 1305031102175304
    assert ds.reference_frame_stamp(7) == 7000000


def test_tum_layout_round_trip_through_both_readers(tmp_path):
    K = tuple(v / 10 for v in synth.TUM_FR1)
    frames = _small_frames(7, K)
    seq = tmp_path / "data" / "rgbd_dataset_freiburg1_desk"
    lists = ds.write_tum_layout(str(seq), frames)
    assert ds.find_sequence(str(tmp_path / "data"), "tum_fr1_desk") == str(seq) and ds.find_sequence(str(seq), "tum_fr1_desk") == str(seq)
    assert ds.find_sequence(str(tmp_path / "data"), "icl_nuim_lr_kt2") is None
    # like the recordings: no associations.txt, colour and depth stamps differ, file names are stamps, denser ground truth with a header
    assert not (seq / "associations.txt").exists()
    assert all(abs(d[0] - r[0]) > 0.01 for d, r in zip(lists["depth"], lists["rgb"]))
    assert open(seq / "groundtruth.txt").read().startswith("# ground truth trajectory\n")
    info = ds.prepare(str(seq), str(tmp_path / "work"), "tum")
    assert info["frames"] == 7 and info["groundtruth"] == str(seq / "groundtruth.txt") and not info["icl_nuim"]
    lines = open(tmp_path / "work" / "associations.txt").read().splitlines()
    assert len(lines) == 7 and all(l.split()[1].startswith("depth/") and l.split()[3].startswith("rgb/") for l in lines)
    assert abs(float(lines[0].split()[0]) - lists["depth"][0][0]) < 1e-6
    # the dataset directory itself was not written to
    assert sorted(os.listdir(seq)) == ["depth", "depth.txt", "groundtruth.txt", "rgb", "rgb.txt"]
    # Python reader
    got = list(ds.read_frames(info))
    assert len(got) == 7
    for (ts, rgb, dep), (r0, d0, _), dl in zip(got, frames, lists["depth"]):
        assert np.array_equal(rgb, r0) and np.array_equal(dep, d0) and ts == ds.reference_frame_stamp(float("%.6f" % dl[0]))
    # the reference's settings reach the run: BA off is the one change besides the sequence keys
    g = hcfg.load_global_state(info["config"])
    assert g["optimizationUseLocalBA"] is False and g["globalInputICLNUIMDataset"] is False and g["registrationICPUseWeightedICP"] is True
    assert g["currentWorkingDirectory"] == str(tmp_path / "work") and g["parameterFileCvFormat"] == "TUM1.yaml"
    cam = hcfg.camera_from_yaml(str(tmp_path / "work" / "TUM1.yaml"))
    assert (cam["fx"], cam["fy"], cam["cx"], cam["cy"]) == (517.3, 516.5, 318.6, 255.3) and abs(cam["depth_scale"] - 1 / 5000.0) < 1e-12
    # ground truth: the estimate "= truth" evaluates to (almost) zero through the stamp association, also after a rigid motion
    gs, gp = ds.load_groundtruth(info["groundtruth"])
    assert len(gs) > 3 * 7 and np.all(np.diff(gs) > 0)
    truth = [f[2] for f in frames]
    A = np.eye(4); A[:3, :3] = synth._rot_yx(0.4, -0.2); A[:3, 3] = [1.0, -2.0, 0.5]
    for est in (truth, [A @ np.asarray(T, np.float64) for T in truth]):
        e = ds.evaluate_ate(info["stamps_s"], est, gs, gp)
        assert e["pairs"] == 7 and e["rmse_m"] < 5e-4, e          # 4 decimals in the file + <= 5 ms of Lissajous motion


def test_cpp_reader_on_both_layouts(tmp_path):
    """`hrbf_run --selftest` (include/hrbf_io.h: ParameterFile rules, camera YAML, association file, PNG decoder) on the prepared
    directories: frame count, intrinsics, ICL flag, the truncated first stamp, the first frame byte for byte"""
    exe = _exe(tmp_path)
    K = tuple(v / 10 for v in synth.TUM_FR1)
    frames = _small_frames(4, K)
    ds.write_tum_layout(str(tmp_path / "tum"), frames, with_associations=True)
    ti = ds.prepare(str(tmp_path / "tum"), str(tmp_path / "wt"), "tum")
    Kn = tuple(v / 10 for v in synth.ICL_NUIM_NEG)
    iframes = _small_frames(5, Kn)
    ds.write_icl_layout(str(tmp_path / "living_room_traj2_frei_png"), iframes)
    ii = ds.prepare(ds.find_sequence(str(tmp_path), "icl_nuim_lr_kt2"), str(tmp_path / "wi"), "icl")
    assert ii["groundtruth"].endswith("livingRoom2.gt.freiburg") and ii["icl_nuim"] and ii["frames"] == 5
    for info, fr, n, icl, fy in ((ti, frames, 4, 0, 516.5), (ii, iframes, 5, 1, 480.0)):
        # the camera files carry the datasets' 640x480 intrinsics; the tiny test images need their own size in the YAML
        cam = dict(info["camera"], width=64, height=48)
        ds.write_camera_yaml(os.path.join(info["work_dir"], "TUM1.yaml" if not icl else "ICL.yaml"), cam)
        out = subprocess.run([exe, "--selftest", "--config", info["config"]], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        j = json.loads(out.stdout)
        assert j["frames"] == n and j["icl"] == icl and j["fy"] == fy and j["sensorType"] == 3 and j["width"] == 64
        assert j["icp_weight"] == 10.0 and j["confidence"] == 5.0 and j["depth_cutoff"] == 3.5 and j["so3"] == 1 and j["bilateral"] == 1
        assert j["timestamp0"] == ds.reference_frame_stamp(info["stamps_s"][0])
        assert j["rgb_fnv"] == _fnv(fr[0][0].tobytes()) and j["depth_fnv"] == _fnv(fr[0][1].tobytes())
    # a stamp the two roundings disagree on
    (tmp_path / "wi" / "associations.txt").write_text("0.000249 depth/0.png 0.000249 rgb/0.png\n")
    out = subprocess.run([exe, "--selftest", "--config", ii["config"]], capture_output=True, text=True, timeout=60)
    assert json.loads(out.stdout)["timestamp0"] == 248


def test_icl_layout_and_the_mirrored_trajectory_convention(tmp_path):
    """ICL-NUIM: images with the y axis up (negative fy) run with +fy give the mirror image of the motion; the reference's writer
    negates ty and prints the microsecond stamp as an integer (TrajectoryManager.cpp:325-330).  A perfect tracker's file,
    read back the way the evaluation reads it, meets the publisher's ground truth."""
    Kn = tuple(v / 10 for v in synth.ICL_NUIM_NEG)
    # poses 40 frames apart: a path that is neither straight nor planar (a mirror image of a straight one fits it rigidly)
    frames = [(f[0], f[1], synth.camera_pose(40 * k).astype(np.float32)) for k, f in enumerate(_small_frames(9, Kn))]
    seq = tmp_path / "living_room_traj2_frei_png"
    ds.write_icl_layout(str(seq), frames)
    assert open(seq / "associations.txt").readline() == "0 depth/0.png 0 rgb/0.png\n"
    assert open(seq / "livingRoom2.gt.freiburg").readline().split()[0] == "1"        # no line for frame 0
    info = ds.prepare(str(seq), str(tmp_path / "w"), "icl")
    assert info["camera"]["fy"] == 480.0 and info["stamps_s"] == [float(k) for k in range(9)]
    got = list(ds.read_frames(info, 3))
    assert [g[0] for g in got] == [0, 1000000, 2000000] and np.array_equal(got[2][2], frames[2][1])
    # a perfect tracker started at the identity in the mirrored world
    M = np.diag([1.0, -1.0, 1.0, 1.0])
    T0i = np.linalg.inv(np.asarray(frames[0][2], np.float64))
    est = [M @ T0i @ np.asarray(f[2], np.float64) @ M for f in frames]
    assert abs(np.linalg.det(est[3][:3, :3]) - 1) < 1e-6
    hio.save_trajectory(str(tmp_path / "t.freiburg"), est, stamps_us=[k * 1000000 for k in range(9)], fmt="TUM", icl_nuim=True)
    first = open(tmp_path / "t.freiburg").read().splitlines()[1].split()
    assert first[0] == "1000000" and abs(float(first[2]) + est[1][1, 3]) < 1e-6
    s, p = ds.load_saved_trajectory(str(tmp_path / "t.freiburg"), icl_nuim=True)
    assert np.array_equal(s, np.arange(9.0))
    gs, gp = ds.load_groundtruth(info["groundtruth"])
    e = ds.evaluate_ate(s, p, gs, gp)
    assert e["pairs"] == 8 and e["rmse_m"] < 2e-4, e
    # without the writer's sign flip the mirrored path does not fit the ground truth
    hio.save_trajectory(str(tmp_path / "u.freiburg"), est, stamps_us=[k * 1000000 for k in range(9)], fmt="TUM", icl_nuim=False)
    s2, p2 = ds.load_saved_trajectory(str(tmp_path / "u.freiburg"))
    e2 = ds.evaluate_ate(s2, p2, gs, gp)
    assert e2["rmse_m"] > 10 * e["rmse_m"]


def test_prepare_accepts_the_publishers_other_association_order(tmp_path):
    """associate.py rgb.txt depth.txt writes `t rgb t depth`; the reference's reader wants the depth file first"""
    frames = _small_frames(3, tuple(v / 10 for v in synth.TUM_FR1))
    seq = tmp_path / "fr1_desk"
    lists = ds.write_tum_layout(str(seq), frames)
    with open(seq / "associations.txt", "w") as f:
        f.write("# rgb first\n")
        for d, r in zip(lists["depth"], lists["rgb"]):
            f.write("%.6f %s %.6f %s\n" % (r[0], r[1], d[0], d[1]))
    info = ds.prepare(str(seq), str(tmp_path / "w"), "tum", max_frames=2)
    lines = open(tmp_path / "w" / "associations.txt").read().splitlines()
    assert len(lines) == 2 and lines[0].split()[1].startswith("depth/") and info["frames"] == 2

"""Export / import formats of SURVEY §8f rows 1-2 (host side)."""
import struct

import numpy as np
import pytest

from hrbffusion3d_amd import io, synth


def test_trajectory_round_trip_and_formats(tmp_path):
    poses = [synth.camera_pose(k) for k in range(0, 50, 5)]
    stamps = [33333 * k for k in range(len(poses))]
    p = tmp_path / "t.freiburg"
    io.save_trajectory(p, poses, stamps, "TUM")
    ts, back = io.load_trajectory_tum(p)
    assert np.allclose(ts, np.array(stamps) / 1e6, atol=1e-6)
    for a, b in zip(poses, back):
        assert np.allclose(a, b, atol=1e-5)
    io.save_trajectory(tmp_path / "icl.txt", poses, stamps, "TUM", icl_nuim=True)
    v = open(tmp_path / "icl.txt").read().split("\n")[3].split()
    assert v[0] == str(stamps[3]) and abs(float(v[2]) + poses[3][1, 3]) < 1e-5
    io.save_trajectory(tmp_path / "z.log", poses, fmt="zhou")
    lines = open(tmp_path / "z.log").read().split("\n")
    assert lines[0] == "0 0 1" and len(lines) == 5 * len(poses) + 1
    io.save_trajectory(tmp_path / "l.txt", poses, fmt="lefloch")
    row = open(tmp_path / "l.txt").read().split("\n")[2].split()
    assert int(row[0]) == 2 and np.allclose([float(x) for x in row[13:16]], poses[2][:3, 3], atol=1e-5)


def test_quaternion_matches_scipy_convention():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(0)
    for _ in range(50):
        R = Rotation.from_rotvec(rng.standard_normal(3) * rng.uniform(0, 3.1)).as_matrix()
        q = io.rotation_to_quaternion(R)
        qs = Rotation.from_matrix(R).as_quat()
        assert np.allclose(q, qs, atol=1e-9) or np.allclose(q, -qs, atol=1e-9)
        assert np.allclose(io.quaternion_to_rotation(q), R, atol=1e-9)


def test_ply_layout(tmp_path):
    m = synth.seed_map(2000)
    n = io.save_ply(tmp_path / "m.ply", m, conf_threshold=10.0)
    assert n == int((m[:, 3] > 10.0).sum())
    raw = open(tmp_path / "m.ply", "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert b"element vertex %d" % n in head and head.count(b"property") == 13
    assert len(body) == n * 43
    k = m[m[:, 3] > 10.0][0]
    rec = struct.unpack("<fff3Bfffffff", body[:43])
    assert np.allclose(rec[0:3], k[0:3]) and np.allclose(rec[6:9], -k[8:11])
    c = int(k[4]); assert rec[3:6] == ((c >> 16) & 255, (c >> 8) & 255, c & 255)
    assert rec[9] == k[15] and rec[10] == k[19] and rec[11] == k[11] and rec[12] == k[5]


def test_associations_and_ate(tmp_path):
    p = tmp_path / "associations.txt"
    p.write_text("# comment\n1305031453.374112 depth/1.png 1305031453.359684 rgb/1.png\n"
                 "1305031453.404816 depth/2.png 1305031453.391690 rgb/2.png\n")
    a = io.load_associations(p)
    assert len(a) == 2 and a[1][1] == "depth/2.png" and a[0][3] == "rgb/1.png"
    gt = [synth.camera_pose(k) for k in range(40)]
    T = np.eye(4); T[:3, :3] = synth._rot_yx(0.3, -0.2); T[:3, 3] = [1, 2, 3]
    est = [T @ g for g in gt]
    assert io.ate_rmse(est, gt, align=True) < 1e-9
    assert io.ate_rmse(est, gt, align=False) > 1.0


def test_klg_reader_round_trip(tmp_path):
    """RawLogReader.cpp:3-140: frame count, per-frame (ts, depthSize, imageSize), raw or zlib depth, raw / JPEG / absent
    image, flipColors, fastForward."""
    from hrbffusion3d_amd.io import KlgReader, write_klg
    W, H = 64, 48
    rng = np.random.default_rng(3)
    frames = []
    for k in range(5):
        d = rng.integers(0, 20000, (H, W)).astype(np.uint16); d[::5] = 0
        yy, xx = np.mgrid[0:H, 0:W]
        rgb = np.stack([(xx * 3 + k) % 256, (yy * 5) % 256, (xx + yy) % 256], -1).astype(np.uint8)
        frames.append((1000000 * k + 17, rgb, d))
    for name, kw in (("raw.klg", dict(compress_depth=False)), ("z.klg", dict(compress_depth=True))):
        p = str(tmp_path / name)
        write_klg(p, frames, **kw)
        r = KlgReader(p, W, H)
        assert len(r) == 5
        got = list(r)
        for (ts, rgb, d), (ts1, rgb1, d1) in zip(frames, got):
            assert ts == ts1 and np.array_equal(rgb, rgb1) and np.array_equal(d, d1)
        r.fast_forward(3)
        assert r.has_more() and r.get_next()[0] == frames[3][0] and r.get_next()[0] == frames[4][0] and not r.has_more()
        assert r.read_frame(1)[0] == frames[1][0]                      # random access back (getBack)
        flipped = KlgReader(p, W, H, flip_colors=True).read_frame(2)[1]
        assert np.array_equal(flipped, frames[2][1][..., ::-1])
        r.close()
    # JPEG image + absent image
    p = str(tmp_path / "jpg.klg")
    write_klg(p, [frames[0], (5, None, frames[1][2])], jpeg_quality=95)
    r = KlgReader(p, W, H)
    ts, rgb, d = r.read_frame(0)
    assert np.array_equal(d, frames[0][2]) and np.abs(rgb.astype(int) - frames[0][1].astype(int)).mean() < 6.0
    ts, rgb, d = r.read_frame(1)
    assert ts == 5 and not rgb.any() and np.array_equal(d, frames[1][2])
    # truncated file
    raw = open(str(tmp_path / "raw.klg"), "rb").read()
    open(str(tmp_path / "cut.klg"), "wb").write(raw[:len(raw) - 100])
    import pytest
    with pytest.raises(EOFError):
        list(KlgReader(str(tmp_path / "cut.klg"), W, H))


def test_run_cli_helpers(tmp_path):
    """hrbffusion3d_amd.run: argument parsing, the .klg frame source and ground-truth matching (no GPU needed)."""
    from hrbffusion3d_amd import run
    from hrbffusion3d_amd.io import write_klg
    W, H = 32, 24
    frames = [(1000 * k, np.full((H, W, 3), k, np.uint8), np.full((H, W), 1000 + k, np.uint16)) for k in range(3)]
    p = str(tmp_path / "s.klg")
    write_klg(p, frames)
    a = run.parse(["--klg", p, "--width", str(W), "--height", str(H), "--out", str(tmp_path / "t.txt")])
    got = list(run.frame_source(a))
    assert [g[0] for g in got] == [0, 1000, 2000] and got[2][2][0, 0] == 1002 and got[1][3] is None
    a = run.parse(["--synthetic", "2", "--width", "64", "--height", "48"])
    got = list(run.frame_source(a))
    assert len(got) == 2 and got[0][1].shape == (48, 64, 3) and got[0][3].shape == (4, 4)
    pairs = run.match_groundtruth(np.array([0.00, 0.10, 0.50]), np.array([0.001, 0.095, 0.30]), [None] * 3)
    assert pairs == [(0, 0), (1, 1)]


def test_run_cli_tum_directory_source(tmp_path):
    """--tum: associations.txt (ts depth ts rgb, HRBFFusion.cpp:226-236) + 16-bit depth PNGs + RGB PNGs"""
    Image = pytest.importorskip("PIL.Image")
    from hrbffusion3d_amd import run
    W, H = 32, 24
    (tmp_path / "depth").mkdir(); (tmp_path / "rgb").mkdir()
    lines = []
    for k in range(3):
        d = (np.arange(W * H, dtype=np.uint16).reshape(H, W) + 1000 * k)
        Image.fromarray(d).save(str(tmp_path / "depth" / ("%d.png" % k)))
        c = np.zeros((H, W, 3), np.uint8); c[..., 0] = 10 * k; c[..., 2] = 200
        Image.fromarray(c).save(str(tmp_path / "rgb" / ("%d.png" % k)))
        lines.append("%.6f depth/%d.png %.6f rgb/%d.png" % (1.5 + 0.033 * k, k, 1.5 + 0.033 * k + 0.001, k))
    (tmp_path / "associations.txt").write_text("# comment\n" + "\n".join(lines) + "\n")
    a = run.parse(["--tum", str(tmp_path), "--width", str(W), "--height", str(H)])
    got = list(run.frame_source(a))
    assert [g[0] for g in got] == [1500000, 1533000, 1566000]
    assert got[2][2].dtype == np.uint16 and got[2][2][1, 1] == 2000 + W + 1
    assert got[1][1].shape == (H, W, 3) and tuple(got[1][1][0, 0]) == (10, 0, 200)

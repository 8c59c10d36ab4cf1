"""Export / import formats of SURVEY §8f rows 1-2 (host side)."""
import struct

import numpy as np

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

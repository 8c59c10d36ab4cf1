"""Host-side input generation (not the hot path): determinism and geometric sanity."""
import numpy as np

from hrbffusion3d_amd import synth


def test_frames_are_deterministic_and_in_range():
    a = synth.frame(5, 160, 120)
    b = synth.frame(5, 160, 120)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[1].min() > 0.3 * 5000 and a[1].max() < 8 * 5000
    n1 = synth.frame(5, 160, 120, noise=True)[1]
    n2 = synth.frame(5, 160, 120, noise=True)[1]
    assert np.array_equal(n1, n2) and 0.01 < (n1 == 0).mean() < 0.06


def test_camera_path_step_sizes():
    d = [np.linalg.norm(synth.camera_pose(k + 1)[:3, 3] - synth.camera_pose(k)[:3, 3]) for k in range(300)]
    assert 0.001 < np.mean(d) < 0.009 and max(d) < 0.010
    R0, R1 = synth.camera_pose(10)[:3, :3], synth.camera_pose(11)[:3, :3]
    ang = np.degrees(np.arccos((np.trace(R0.T @ R1) - 1) / 2))
    assert ang < 0.5
    assert np.allclose(R0 @ R0.T, np.eye(3), atol=1e-12)


def test_seed_map_layout():
    m = synth.seed_map(20000)
    assert m.shape[1] == 20 and 20000 <= m.shape[0] < 24000 and m.dtype == np.float32
    assert np.allclose(np.linalg.norm(m[:, 8:11], axis=1), 1.0, atol=1e-5)
    assert m[:, 3].min() >= 5.0 and m[:, 3].max() <= 20.0
    assert np.all(m[:, 4] == np.floor(m[:, 4])) and m[:, 4].max() < 2 ** 24
    assert np.all(np.abs(m[:, 15]) < 300) and np.all(np.abs(m[:, 19]) < 300)
    # points lie on the room box or the sphere
    onbox = np.isclose(np.abs(m[:, 0]), 3.0, atol=0.04) | np.isclose(np.abs(m[:, 1]), 1.5, atol=1e-6) | \
        np.isclose(np.abs(m[:, 2]), 2.0, atol=0.04)
    onsph = np.isclose(np.linalg.norm(m[:, :3] - synth.SPH_C, axis=1), synth.SPH_R, atol=1e-5)
    assert np.all(onbox | onsph)

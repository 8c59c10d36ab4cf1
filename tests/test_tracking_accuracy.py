"""Where the registration's accuracy on the synthetic stream comes from ("The drift from an empty map").

CPU tests on the oracle (bit-identical to the HIP path, tests/test_parity_gpu.py), QVGA, noise-free, frame 0 -> frame 1
from an empty map, i.e. frame-to-frame against the filled-in previous frame.  They pin three facts that together locate
the ~5 mm / 0.1 deg per frame the default joint registration loses while the map is young:
  a static camera is tracked exactly; the ICP term alone (icp_weight 100: `rgb = rgbOnly || icpWeight < 100`,
  RGBDOdometry.cpp:807) is accurate to half a millimetre; the photometric term, whose
  residual looks the model image up at the nearest texel (reduce.cu:1027-1046), is what carries the error.
tools/probes/rgb_term_emulation.py reproduces the last fact with an independent numpy emulation
(profiles/r03_rgb_term_emulation.txt).
"""
import numpy as np
import pytest

from hrbffusion3d_amd import synth
from hrbffusion3d_amd.params import default_params

W, H = 320, 240


def _two_frames(oracle_lib, second, **kw):
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 18)
    for k, v in kw.items():
        setattr(p, k, v)
    o = oracle_lib.Oracle(p, omp=True)
    try:
        rgb, d, t0 = synth.frame(0, W, H, noise=False)
        o.process_frame(rgb, d)
        rgb, d, t1 = synth.frame(second, W, H, noise=False)
        o.process_frame(rgb, d)
        est = o.get_pose().astype(np.float64)
    finally:
        o.close()
    gt = np.linalg.inv(t0.astype(np.float64)) @ t1.astype(np.float64)
    e = np.linalg.inv(gt) @ est
    ang = np.degrees(np.arccos(np.clip((np.trace(e[:3, :3]) - 1) / 2, -1, 1)))
    return float(np.linalg.norm(e[:3, 3]) * 1e3), float(ang), est


def test_a_static_camera_is_tracked_exactly(oracle_lib_built):
    _, _, est = _two_frames(oracle_lib_built, 0)
    assert np.array_equal(est, np.eye(4))


def test_the_icp_term_alone_tracks_the_synthetic_motion_to_half_a_millimetre(oracle_lib_built):
    mm, deg, _ = _two_frames(oracle_lib_built, 1, icp_weight=100.0)
    assert mm < 0.8 and deg < 0.01, (mm, deg)


@pytest.mark.parametrize("mode", [dict(), dict(rgb_only=1)])
def test_the_nearest_texel_rgb_term_carries_the_offset(oracle_lib_built, mode):
    """5.6 mm / 0.29 deg of true motion (about 2 px at fx = 264): the photometric term trades translation for rotation
    within its half-pixel resolution — the same 3-10 mm / 0.1-0.25 deg an independent emulation of the term shows"""
    mm, deg, _ = _two_frames(oracle_lib_built, 1, **mode)
    assert 2.0 < mm < 9.0 and 0.05 < deg < 0.25, (mm, deg)


def test_icp_only_registration_tracks_thirty_noisy_frames_to_a_centimetre(oracle_lib_built):
    """30 noisy QVGA frames from an empty map: ATE ~ 5.6 cm with the default weights (photometric rows dominate), ~ 0.8 cm
    with icp_weight 100, which switches the photometric term off (RGBDOdometry.cpp:807) — same stream, same pre-processing, fusion and prediction"""
    fx, fy, cx, cy = synth.intrinsics(W, H)
    frames = [synth.frame(k, W, H, noise=True) for k in range(30)]
    ate = {}
    for w in (10.0, 100.0):
        o = oracle_lib_built.Oracle(default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 19, icp_weight=w), omp=True)
        try:
            o.set_pose(frames[0][2])
            est = []
            for rgb, d, _ in frames:
                o.process_frame(rgb, d)
                est.append(o.get_pose())
        finally:
            o.close()
        ate[w] = synth.ate_rmse(est, [f[2] for f in frames])
    assert ate[100.0] < 0.015 and 0.03 < ate[10.0] < 0.09, ate

"""Shared driver of the registration's metamorphic tests: two analytic frames through the WHOLE processFrame of an engine
(the CPU oracle, or the HIP library on a GPU), the estimate against the motion the frames were rendered with.

Nothing here restates the reference: the scenes are analytic (tests/reg_scenes.py), the expectations are bounds the mathematics of
the reference's own formulation implies (nearest-texel photometric residual => half a pixel of the level; Gauss-Newton on a
consistent problem => time reversal; the units of the two Jacobians and `A_rgb + w^2 A_icp, b_rgb + w b_icp`
(RGBDOdometry.cpp:1168-1186) => how the solution moves when the depth unit changes; a pure rotation => the SO3 pre-alignment)."""
import numpy as np

import reg_scenes as rs
from hrbffusion3d_amd.params import default_params


def intrinsics(W, H):
    f = 264.0 * W / 320.0
    return (f, f, W / 2.0 - 0.5, H / 2.0 - 0.5)


CORNER_VIEW = rs.pose(-0.43, 0.65, 0.0)           # looks into a corner of the room: three walls, a third of the image each
PLANE_VIEW = rs.pose(0.05, -0.08, 0.02)
VIEWS = {"room": (rs.ROOM, CORNER_VIEW), "plane": (rs.PLANE, PLANE_VIEW)}
# relative motions of the camera (frame B in frame A): a third of a pixel .. five pixels at 640x480
MOTIONS = {"0.3px": rs.pose(t=(0.0009, 0.0, 0.0)),
           "2px": rs.pose(0.001, -0.0015, 0.0005, (0.002, -0.0015, 0.001)),
           "5px": rs.pose(0.0025, -0.004, 0.001, (0.005, -0.004, 0.002))}


def make_engine(kind, p):
    if kind == "oracle":
        import oracle_lib
        oracle_lib.build()
        return oracle_lib.Oracle(p, omp=True)
    from hrbffusion3d_amd.api import HRBFFusion
    return HRBFFusion(p)       # raises without the HIP library / a gfx950 device: there is no fallback


def two_frames(kind, W, H, TA, TB, scene=rs.ROOM, wavelength=None, units=5000.0, T0=None, trace=False, contrast=1.0, edit_b=None, **params):
    """frame A seeds an empty map (pose T0 or identity), frame B is registered against it.
    `edit_b(rgb, depth) -> (rgb, depth)` changes frame B before it is handed over (holes, ...).
    -> dict(E: estimated pose of B, G: true relative pose, z: depth of B, K, trace, bits: raw pose)"""
    K = intrinsics(W, H)
    p = default_params(W, H, *K, max_surfels=1 << 20, **params)
    wl = wavelength if wavelength is not None else 160.0 / W      # 7-13 cm at 640x480: 20-35 px at level 0, 5-9 px at level 2
    a = rs.render(TA, W, H, K, scene, units=units, wavelength=wl, contrast=contrast)
    b = rs.render(TB, W, H, K, scene, units=units, wavelength=wl, contrast=contrast)
    if edit_b is not None:
        b = edit_b(b[0], b[1]) + tuple(b[2:])
    e = make_engine(kind, p)
    try:
        if T0 is not None:
            e.set_pose(np.asarray(T0, np.float32))
        e.process_frame(a[0], a[1])
        e.process_frame(b[0], b[1])
        P = e.get_pose()
        tr = e.odo_trace() if (trace and kind == "oracle") else None
    finally:
        e.close()
    return {"E": P.astype(np.float64), "G": np.linalg.inv(TA) @ TB, "z": b[2], "za": a[2], "K": K, "trace": tr,
            "bits": np.ascontiguousarray(P, np.float32).view(np.uint32).copy()}


def trace_pose(row):
    """the camera pose the oracle held at the START of the Gauss-Newton iteration a trace row describes (oracle.h: state0)"""
    T = np.eye(4)
    T[:3, :3] = row[112:121].reshape(3, 3)
    T[:3, 3] = row[121:124]
    return T


def trace_systems(row):
    """(A_icp, b_icp, A_rgb, b_rgb, increment) of a trace row"""
    return row[2:38].reshape(6, 6), row[38:44], row[44:80].reshape(6, 6), row[80:86], row[86:92]

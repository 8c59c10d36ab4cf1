"""Staged driver for the second family of registration tests (tests/test_registration_metamorphic2.py): frame A goes through the
whole processFrame, frame B is uploaded and pre-processed stage by stage, a test may then EDIT the live images or the predicted
model images through the image seam (set_image) before the registration stage runs alone — so that a rule of the map building
(validity, thresholds, resize) meets inputs chosen to sit on either side of it.  Works on both engines (the CPU oracle and the
HIP library: same stage names, same image names); the oracle additionally returns its trace and pyramids."""
import numpy as np

import reg_cases as rc
import reg_scenes as rs
from hrbffusion3d_amd.params import default_params

PRE = ("FILTER_DEPTH", "METRICISE", "VERTEX_NORMAL_RADIUS", "CURVATURE")


def frontal_plane(depth):
    """a wall facing the camera at `depth` metres (scene frame = the first camera's frame)"""
    return [((0.0, 0.0, 1.0), float(depth))]


def staged(kind, W, H, TA, TB, scene=rs.ROOM, edit=None, T0=None, wmul=1.0, K=None, wavelength=None, keep=None, **params):
    """-> dict(E pose after the registration stage, P0 pose before, trace, bits, K, z (depth of B), extra = keep(engine))"""
    K = K or rc.intrinsics(W, H)
    p = default_params(W, H, *K, max_surfels=1 << 20, **params)
    wl = wavelength if wavelength is not None else 160.0 / W
    a = rs.render(TA, W, H, K, scene, wavelength=wl)
    b = rs.render(TB, W, H, K, scene, wavelength=wl)
    e = rc.make_engine(kind, p)
    try:
        if T0 is not None:
            e.set_pose(np.asarray(T0, np.float32))
        e.process_frame(a[0], a[1])
        P0 = e.get_pose().astype(np.float64)
        e.upload_frame(b[0], b[1])
        for st in PRE:
            e.run_stage(st)
        if edit is not None:
            edit(e)
        e.run_stage("ODOMETRY")
        P = e.get_pose()
        tr = e.odo_trace() if kind == "oracle" else None
        extra = keep(e) if keep is not None else None
    finally:
        e.close()
    return {"E": P.astype(np.float64), "P0": P0, "G": np.linalg.inv(TA) @ TB, "z": b[2], "K": K, "trace": tr, "extra": extra,
            "bits": np.ascontiguousarray(P, np.float32).view(np.uint32).copy()}


def gn_rows(trace, level=None):
    return [r for r in trace if int(r[0]) >= 0 and (level is None or int(r[0]) == level)]


def inliers(trace, level=0, it=0):
    """ICP inlier count of one Gauss-Newton iteration (oracle.h: trace row entry 92)"""
    return int(gn_rows(trace, level)[it][92])

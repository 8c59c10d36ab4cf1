"""Measurement: milliseconds of hrbf_fit_curvature on a 640x480 noisy frame, both windows (HRBF_LIB selects a variant library)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.api import HRBFFusion
from hrbffusion3d_amd.params import default_params

W, H = 640, 480
g = HRBFFusion(default_params(W, H, *synth.intrinsics(W, H), max_surfels=1 << 20))
rgb, d, _ = synth.frame(3, W, H, noise=True)
g.process_frame(rgb, d)
for window in (2, 1):
    g.fit_curvature(window=window, timed=True)
    ts = [g.fit_curvature(window=window, timed=True) for _ in range(5)]
    c1 = g.get_image("FIT_CURV1")
    fitted = int((c1[..., 3] != 1000.0).sum())
    k = c1[..., 3][c1[..., 3] != 1000.0]
    print("%s window %d: %.3f ms (min %.3f), %d systems, %.1f M systems/s, median |kmax| %.3f, finite %d" % (
        os.environ.get("HRBF_LIB", "product"), window, np.mean(ts), np.min(ts), fitted, fitted / np.mean(ts) / 1e3, np.median(np.abs(k)), int(np.isfinite(k).sum())))
g.close()

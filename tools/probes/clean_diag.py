"""DESIGN.md §5 diagnostic (needs a -DCLEAN_DIAG build: HRBF_LIB=_build/libhrbf_v_cleandiag.so): what share of the in-view items of the clean
pass has NO index-map winner behind it in its window — the items a dilated max-winner-depth image would settle with one 4-byte gather
instead of <= 9 sixteen-byte ones (round-3 verdict item 4).  Both bench legs: the headline stream and the 4.3 M-surfel worst case."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hrbffusion3d_amd import synth                      # noqa: E402
from hrbffusion3d_amd.api import HRBFFusion, load_library, LIB_PATH   # noqa: E402
from hrbffusion3d_amd.params import default_params      # noqa: E402


def leg(name, nsurf, frames, stale_frac=0.0):
    lib = load_library()
    W, H = 640, 480
    K = synth.intrinsics(W, H)
    seed = synth.seed_map(nsurf, t_now=1, width=W)
    if stale_frac > 0:
        n = seed.shape[0]
        sel = np.arange(0, n, int(1 / stale_frac))
        seed[sel, 3] = 1.0; seed[sel, 7] = -300.0
    g = HRBFFusion(default_params(W, H, *K, max_surfels=seed.shape[0] + 600_000))
    rgb, d, T = synth.frame(0, W, H)
    g.upload_map(seed); g.set_pose(T); g.bootstrap(rgb, d)
    out = (C.c_ulonglong * 4)()
    lib.hrbf_probe_clean_diag(out, 1)
    for k in range(1, 1 + frames):
        rgb, d, T = synth.frame(k, W, H)
        g.process_frame(rgb, d)
    g.synchronize()
    lib.hrbf_probe_clean_diag(out, 1)
    print("%s: %d frames, per frame %d in-view items reach the window test, %.1f%% of them have no winner behind them in the window, %d dropped" % (
        name, frames, out[0] // frames, 100.0 * out[1] / max(1, out[0]), out[2] // frames))
    g.close()


if __name__ == "__main__":
    print("library:", LIB_PATH)
    leg("headline stream (1.06 M surfels)", 1_050_000, 30)
    leg("worst case (4.3 M surfels, 5 % stale)", 4_300_000, 5, 0.05)

"""Measurement (needs a -DFIT_TIMING build: HRBF_LIB=_build/libhrbf_fit_timing.so): cycles a wave of k_hrbf_fit spends per phase
(s_memtime at the phase boundaries, one record per wave).  DESIGN.md section 10a."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.api import HRBFFusion, load_library
from hrbffusion3d_amd.params import default_params

W, H = 640, 480
g = HRBFFusion(default_params(W, H, *synth.intrinsics(W, H), max_surfels=1 << 20))
rgb, d, _ = synth.frame(3, W, H, noise=True)
g.process_frame(rgb, d)
lib = load_library()
buf = np.zeros((W * H, 8), np.uint32)
g.fit_curvature(); g.synchronize()
lib.hrbf_probe_fit_phases(buf.ctypes.data_as(C.c_void_p), 1)
ms = g.fit_curvature(timed=True)
lib.hrbf_probe_fit_phases(buf.ctypes.data_as(C.c_void_p), 0)
fitted = g.get_image("FIT_CURV1")[..., 3].ravel() != 1000.0
n = int(fitted.sum())
out = buf[fitted].astype(np.float64).sum(0)
names = ["gather", "assemble (registers)", "diagonal block: LDS round trip + column sweep (7)", "panel solves + trailing updates on the matrix core (6)", "read-out", "(unused)"]
tot = sum(out[i] for i in range(6))
print("%.2f ms, %d systems; cycles per wave (s_memtime ticks = 100 MHz on gfx9: x core/100MHz):" % (ms, n))
for i, nm in enumerate(names):
    print("  %-42s %10.0f  %5.1f %%" % (nm, out[i] / max(n, 1), 100.0 * out[i] / max(tot, 1)))
g.close()

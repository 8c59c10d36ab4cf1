"""Measurement (not collected by pytest): what share of the (sample, centre) pairs of k_predict_hrbf's ray march
fails the support test `support^2 < dist^2` (hrbfbase.glsl:137-138) — and how many of them a per-ray interval test could discard.
Needs the -DPREDICT_TRIP_STATS build:
    python -c "from hrbffusion3d_amd import build; build.build(True, defines=['-DPREDICT_TRIP_STATS'], out='libhrbf_trips.so')"
    HRBF_LIB=_build/libhrbf_trips.so python tests/gpu_probe_predict_support.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.api import HRBFFusion, load_library
from hrbffusion3d_amd.params import default_params


def run(W, H, surfels, noise, frames=6):
    lib = load_library()
    out = (C.c_ulonglong * 8)()
    p = default_params(W, H, *synth.intrinsics(W, H), max_surfels=1 << 21)
    g = HRBFFusion(p)
    if surfels:
        g.upload_map(synth.seed_map(surfels, t_now=1, width=W))
    rgb, d, T = synth.frame(0, W, H, noise=noise)
    g.set_pose(T)
    if surfels:
        g.bootstrap(rgb, d)
    else:
        g.process_frame(rgb, d)
    for k in range(1, frames - 1):
        rgb, d, T = synth.frame(k, W, H, noise=noise)
        g.process_frame(rgb, d)
    g.synchronize()
    lib.hrbf_probe_predict_support(out, 1)                 # count the last frame's prediction only
    rgb, d, T = synth.frame(frames - 1, W, H, noise=noise)
    g.process_frame(rgb, d)
    g.synchronize()
    lib.hrbf_probe_predict_support(out, 1)
    s = [int(v) for v in out]
    g.close()
    pairs, inside, samples, empty, entries, never, not_in_stretch = s[:7]
    print("%dx%d, %s map, %s frames: %d samples, %d (sample, centre) pairs" % (W, H, "%d-surfel seeded" % surfels if surfels else "grown", "noisy" if noise else "clean", samples, pairs))
    print("   pairs whose centre does NOT reach the sample: %.3f" % (1.0 - inside / max(pairs, 1)))
    print("   samples no centre reaches at all (value 0):    %.3f" % (empty / max(samples, 1)))
    print("   list entries: %d; never reached anywhere on the ray's line: %.3f; not reached on the marched +-10 cm: %.3f" % (
        entries, never / max(entries, 1), not_in_stretch / max(entries, 1)))


if __name__ == "__main__":
    lib = load_library()
    if not hasattr(lib, "hrbf_probe_predict_support"):
        raise SystemExit("this library was not built with -DPREDICT_TRIP_STATS (see the docstring)")
    lib.hrbf_probe_predict_support.argtypes = [C.c_void_p, C.c_int]
    run(640, 480, 1_050_000, False)
    run(640, 480, 1_050_000, True)
    run(640, 480, 0, True, frames=12)

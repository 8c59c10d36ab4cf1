"""Generates tests/golden/gputest_pair_expected.npz from the CPU oracle on the reference's only
real-data fixture (GPUTest/{1c,1d,2c,2d}.png, BASELINE config 1: two-frame pair, CPU-only path).

The reference pins nothing (no tests, cannot be built here): these are the ORACLE's outputs, kept
as a regression pin and as the committed expectation the GPU path is compared against.
    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hrbffusion3d_amd.params import default_params, IMAGES  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


def digest(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:   # NaN payload/sign is the one platform difference (x86 0xFFC00000, gfx950 0x7FC00000)
        u = a.view(np.uint32).copy(); u[np.isnan(a)] = 0x7FC00000; a = u
    else:
        a = a.view(np.uint8)
    return np.frombuffer(hashlib.sha256(a.tobytes()).digest(), np.uint8)


def run(make):
    f = lambda n: np.array(Image.open(os.path.join(HERE, n + ".png")))
    o = make(default_params(max_surfels=1 << 20))
    out = {}
    for k, (c, d) in enumerate([("1c", "1d"), ("2c", "2d")]):
        o.process_frame(f(c), f(d))
        tag = "f%d_" % (k + 1)
        out[tag + "pose"] = o.get_pose()
        out[tag + "count"] = np.array([o.surfel_count()])
        out[tag + "stats"] = o.fuse_stats()
        out[tag + "icp"] = np.array(o.last_icp(), np.float32)
        out[tag + "map_sha"] = digest(o.download_map())
        for name in IMAGES:
            out[tag + "sha_" + name] = digest(o.get_image(name))
        out[tag + "crop_pred_vertex"] = o.get_image("PRED_VERTEX")[200:232, 300:332].copy()
        out[tag + "crop_fill_vertex"] = o.get_image("FILL_VERTEX")[200:232, 300:332].copy()
        out[tag + "crop_curv1"] = o.get_image("CURV1")[200:232, 300:332].copy()
        out[tag + "crop_normal"] = o.get_image("NORMAL")[200:232, 300:332].copy()
    return out


VARIANTS = {
    "sparse_icp": dict(use_sparse_icp=1),
    "corr_search": dict(icp_use_corr_search=1),
    "rgb_only": dict(rgb_only=1),
    "icp_only": dict(icp_weight=100.0),
    "no_so3_no_pyramid": dict(so3=0, pyramid=0, fast_odom=1),
    "gauss_central_diff": dict(use_bilateral=0, normal_estimation_pca=0.0),
}


def run_variants(make):
    """the same pair under the registration / pre-processing options: pose, counts and a digest of the whole map"""
    f = lambda n: np.array(Image.open(os.path.join(HERE, n + ".png")))
    out = {}
    for name, kw in VARIANTS.items():
        o = make(default_params(max_surfels=1 << 20, **kw))
        for c, d in (("1c", "1d"), ("2c", "2d")):
            o.process_frame(f(c), f(d))
        out[name + "_pose"] = o.get_pose()
        out[name + "_stats"] = o.fuse_stats()
        out[name + "_icp"] = np.array(o.last_icp(), np.float32)
        out[name + "_map_sha"] = digest(o.download_map())
        out[name + "_pred_sha"] = digest(o.get_image("PRED_VERTEX"))
        o.close()
    return out


if __name__ == "__main__":
    out = run(lambda p: Oracle(p, omp=True))
    np.savez_compressed(os.path.join(HERE, "gputest_pair_expected.npz"), **out)
    print("wrote", len(out), "arrays; frame-2 pose:\n", out["f2_pose"])
    var = run_variants(lambda p: Oracle(p, omp=True))
    np.savez_compressed(os.path.join(HERE, "gputest_pair_variants.npz"), **var)
    print("wrote", len(var), "variant arrays")

"""Soak (not collected by pytest): long tracked sequences, GPU vs oracle, every image / map / pose compared every frame.
    python tests/gpu_probe_soak.py [frames] [W] [H] [shards]"""
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "16")   # 256 OpenMP threads make the oracle 100x slower on the GPU box

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.api import HRBFFusion
from hrbffusion3d_amd.params import IMAGES, default_params


def bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        u = a.view(np.uint32).copy(); u[np.isnan(a)] = 0x7FC00000
        return u
    return a.view(np.uint8)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 240
    G = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    oracle_lib.build()
    fx, fy, cx, cy = synth.intrinsics(W, H)
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 21)
    o = oracle_lib.Oracle(p, omp=True); g = HRBFFusion(p)
    if G > 1:
        g.comm_init(-1, G); g.map_shard_init(True)
    t0 = time.time()
    for k in range(n):
        rgb, d, _ = synth.frame(k, W, H, noise=True)
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        for name in IMAGES:
            if not np.array_equal(bits(o.get_image(name)), bits(g.get_image(name))):
                print("MISMATCH frame %d image %s" % (k, name)); return 1
        if o.surfel_count() != g.surfel_count() or not np.array_equal(bits(o.download_map()), bits(g.download_map())):
            print("MISMATCH frame %d map" % k); return 1
        if not np.array_equal(bits(o.get_pose()), bits(g.get_pose())):
            print("MISMATCH frame %d pose" % k); return 1
        if G > 1 and k % 20 == 5:
            g.map_rebalance()
    print("soak ok: %d frames %dx%d shards %d, %d surfels, %.0f s" % (n, W, H, G, g.surfel_count(), time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())

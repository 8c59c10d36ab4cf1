"""Long run (not collected by pytest): N frames back and forth along the synthetic path against a 1 M-surfel map through
the host-pointer entry point; prints the sticky status word, the surfel count and the pose-ring counter every 500 frames.
    python tests/gpu_probe_longrun.py [frames]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.api import HRBFFusion
from hrbffusion3d_amd.params import default_params


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    W, H = 640, 480
    p = default_params(W, H, *synth.intrinsics(W, H), max_surfels=1 << 22)
    g = HRBFFusion(p)
    g.upload_map(synth.seed_map(1_050_000, t_now=1, width=W))
    frames = [synth.frame(k, W, H, noise=True) for k in range(40)]
    g.set_pose(frames[0][2]); g.bootstrap(frames[0][0], frames[0][1])
    t0 = time.time()
    for k in range(1, n):
        f = frames[(k % 78) if (k % 78) < 40 else 78 - (k % 78)]
        g.process_frame(f[0], f[1])
        if k % 500 == 0:
            print(k, "status", g.status(), "surfels", g.surfel_count(), "completed", g.frames_completed(), flush=True)
    g.synchronize()
    ok = g.status() == 0 and np.isfinite(g.get_pose()).all()
    print("long run %s: %d frames, %.1f frames/s incl. upload, status %s" % ("ok" if ok else "FAILED", n, n / (time.time() - t0), g.status()))
    g.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

"""Ad-hoc GPU probe: tracking quality + timings on the synthetic stream with a 1M pre-seeded map."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.params import default_params
from hrbffusion3d_amd.api import HRBFFusion
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
noise = len(sys.argv) > 2 and sys.argv[2] == "noise"
W, H = 640, 480
fx, fy, cx, cy = synth.intrinsics(W, H)
seed = synth.seed_map(1_050_000)
p = default_params(W, H, fx, fy, cx, cy, max_surfels=seed.shape[0] + 3_000_000)
g = HRBFFusion(p)
g.upload_map(seed); g.set_pose(synth.camera_pose(0))
rgb, d, T = synth.frame(0, W, H, noise=noise); g.bootstrap(rgb, d)
g.enable_timing(True)
print("seed", seed.shape[0], "pred valid %.3f" % (g.get_image("PRED_VERTEX")[..., 2] > 0).mean())
for k in range(1, N + 1):
    rgb, d, T = synth.frame(k, W, H, noise=noise)
    g.process_frame(rgb, d)
    P = g.get_pose(); tm = g.timings()
    print("%3d err %.2f mm  count %d stats %s icp %s pred %.2f  ms: init %.2f reg %.2f fuse %.3f pred %.3f clean %.3f frame %.2f" % (
        k, 1000 * np.linalg.norm(P[:3, 3] - T[:3, 3]), g.surfel_count(), g.fuse_stats(), g.last_icp(),
        (g.get_image("PRED_VERTEX")[..., 2] > 0).mean(), tm[0], tm[1], tm[2], tm[3], tm[4], tm[5]))
ms, st = g.fuse_ring(N)
B = 80.0 * (st[:, 0].astype(np.float64) + st[:, 3] + st[:, 1] + st[:, 2])
print("fuse kernel ms", ms[-5:], "GB/s", (B / (ms * 1e-3) / 1e9)[-5:])

"""Un-profiled attribution of the frame time by switching registration stages off (params), ms per frame."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.api import HRBFFusion
from hrbffusion3d_amd.params import default_params
W, H = 640, 480
fx, fy, cx, cy = synth.intrinsics(W, H)
N = 70
frames = [synth.frame(k, W, H) for k in range(N + 1)]
seed = synth.seed_map(1_050_000, t_now=1, width=W)
d_rgb = [torch.from_numpy(f[0]).cuda() for f in frames]
d_dep = [torch.from_numpy(f[1].view(np.int16)).cuda() for f in frames]
def run(**kw):
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=int(seed.shape[0] + 3_000_000), **kw)
    fus = HRBFFusion(p, device=0)
    fus.upload_map(seed); fus.set_pose(frames[0][2]); fus.bootstrap(frames[0][0], frames[0][1])
    for k in range(1, 21): fus.process_frame_device(d_rgb[k].data_ptr(), d_dep[k].data_ptr(), k)
    fus.synchronize()
    t0 = time.perf_counter()
    for k in range(21, 71): fus.process_frame_device(d_rgb[k].data_ptr(), d_dep[k].data_ptr(), k)
    fus.synchronize()
    dt = (time.perf_counter() - t0) / 50 * 1e3
    fus.close()
    return dt
base = run()
print("default %.3f ms" % base)
for name, kw in (("fast_odom (L0 3 iterations instead of 10)", dict(fast_odom=1)), ("no pyramid (L1, L2 off)", dict(pyramid=0)),
                 ("no so3", dict(so3=0)), ("rgb_only", dict(rgb_only=1)), ("icp only (icp_weight 100)", dict(icp_weight=100.0)),
                 ("load_trajectory (no registration)", dict(load_trajectory=1))):
    t = run(**kw)
    print("%-45s %.3f ms  (delta %+.3f)" % (name, t, t - base))

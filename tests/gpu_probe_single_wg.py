"""A/B record for DESIGN.md §6: the pixel work of the level-2 / level-1 / level-0 Gauss-Newton iterations done by ONE
workgroup (hrbf_probe_single_workgroup_iteration) against the three launches per iteration of the shipped path
(rocprofv3 kernel trace of the same frames).  Run on the GPU box: python tests/gpu_probe_single_wg.py"""
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.api import HRBFFusion
from hrbffusion3d_amd.params import default_params

W, H = 640, 480
K = synth.intrinsics(W, H)
seed = synth.seed_map(400_000, width=W)
g = HRBFFusion(default_params(W, H, *K, max_surfels=seed.shape[0] + 600_000))
rgb, d, T = synth.frame(0, W, H, noise=True)
g.upload_map(seed); g.set_pose(T); g.bootstrap(rgb, d)
for k in range(1, 4):
    rgb, d, T = synth.frame(k, W, H, noise=True)
    g.process_frame(rgb, d)
g.synchronize()
out = {}
for level, iters in ((2, 4), (1, 5), (0, 10)):
    ms = [g.probe_single_workgroup_iteration(level, iters) for _ in range(3)]
    out["level%d_%d_iterations_ms" % (level, iters)] = min(ms)
print(json.dumps(out))
g.close()

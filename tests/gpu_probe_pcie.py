"""why is the python host-pointer loop of bench.py slower than the C++ one? natural vs jumped frame order, region timings"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.api import HRBFFusion
from hrbffusion3d_amd.params import default_params
W, H = 640, 480
K = synth.intrinsics(W, H)
seed = synth.seed_map(1_050_000)
frames = [synth.frame(k, W, H) for k in range(82)]
g = HRBFFusion(default_params(W, H, *K, max_surfels=seed.shape[0] + 3_000_000))
g.upload_map(seed); g.set_pose(frames[0][2]); g.bootstrap(frames[0][0], frames[0][1])
def run(idx, label):
    g.synchronize(); t = time.perf_counter()
    for k in idx:
        g.process_frame(frames[k][0], frames[k][1], k)
    g.synchronize(); dt = time.perf_counter() - t
    g.enable_timing(1); g.process_frame(frames[idx[-1]][0], frames[idx[-1]][1]); tm = g.timings(); g.enable_timing(False)
    print(label, "%.1f fps" % (len(idx) / dt), "count", g.surfel_count(), "regions ms", np.round(tm[:7], 3), "stats", g.fuse_stats(), "status", g.status())
run(list(range(1, 21)), "natural 1..20")
run(list(range(21, 81)), "natural 21..80")
run(list(range(21, 24)), "jump back to 21..23")
run(list(range(61, 81)), "jump to 61..80")
run(list(range(61, 81)), "again 61..80")
g.close()

"""progress-logging soak: GPU vs oracle, pose/count every frame, full state every 10th; writes gpurun_out/soak2.log"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
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

n = int(sys.argv[1]); W = int(sys.argv[2]); H = int(sys.argv[3]); use_oracle = int(sys.argv[4]) if len(sys.argv) > 4 else 1
shards = int(sys.argv[5]) if len(sys.argv) > 5 else 0; partition = sys.argv[6] if len(sys.argv) > 6 else "ranges"   # virtual shards
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", "soak2.log"), "a")
def say(*a):
    log.write(" ".join(str(x) for x in a) + "\n"); log.flush(); os.fsync(log.fileno())
say("start", n, W, H, use_oracle, "cpus", len(os.sched_getaffinity(0)))
p = default_params(W, H, *synth.intrinsics(W, H), max_surfels=1 << 21)
o = oracle_lib.Oracle(p, omp=True) if use_oracle else None
g = HRBFFusion(p)
if shards > 1:
    g.comm_init(-1, shards); g.map_shard_init(True, partition=partition)
t0 = time.time()
for k in range(n):
    rgb, d, _ = synth.frame(k, W, H, noise=True)
    ta = time.time()
    if o: o.process_frame(rgb, d)
    tb = time.time()
    g.process_frame(rgb, d); g.synchronize()
    tc = time.time()
    if o:
        if not np.array_equal(bits(o.get_pose()), bits(g.get_pose())) or o.surfel_count() != g.surfel_count():
            say("MISMATCH pose/count frame", k); break
        if k % 10 == 0:
            for name in IMAGES:
                if name == "INDEX" and partition == "hash":
                    continue   # ids instead of array positions: names only
                if not np.array_equal(bits(o.get_image(name)), bits(g.get_image(name))):
                    say("MISMATCH frame", k, name); break
            if not np.array_equal(bits(o.download_map()), bits(g.download_map())):
                say("MISMATCH map frame", k); break
    if k % 10 == 0 or tc - tb > 0.5:
        say("frame", k, "oracle %.2fs gpu %.3fs total %.0fs count %d status %d stats %s" % (tb - ta, tc - tb, time.time() - t0, g.surfel_count(), g.status(), g.fuse_stats()))
say("done", k)

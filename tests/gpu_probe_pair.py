"""Ad-hoc GPU probe: run the GPUTest PNG pair through oracle and GPU, compare every image."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from PIL import Image
from hrbffusion3d_amd.params import default_params, IMAGES
from hrbffusion3d_amd.api import HRBFFusion
from oracle_lib import Oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
f = lambda n: np.array(Image.open(os.path.join(G, n + ".png")))
p = default_params(max_surfels=1 << 20)
o = Oracle(p, omp=True)
g = HRBFFusion(p)
g.enable_timing(True)

def cmp(tag):
    bad = 0
    for name in IMAGES:
        a = o.get_image(name); b = g.get_image(name)
        same = np.array_equal(a.view(np.uint8), b.view(np.uint8))
        if not same:
            # NaN-aware compare
            if a.dtype.kind == 'f':
                eq = (a == b) | (np.isnan(a) & np.isnan(b))
                nbad = int((~eq).sum())
                md = float(np.nanmax(np.abs(np.where(eq, 0, a - b)))) if nbad else 0.0
            else:
                nbad = int((a != b).sum()); md = float(np.abs(a.astype(np.int64) - b.astype(np.int64)).max())
            if nbad:
                bad += 1
                print("  [%s] %-22s MISMATCH n=%d maxdiff=%g" % (tag, name, nbad, md))
    ma = o.download_map(); mb = g.download_map()
    print("  [%s] count oracle %d gpu %d  map equal: %s" % (tag, len(ma), len(mb), ma.shape == mb.shape and np.array_equal(ma.view(np.uint32), mb.view(np.uint32))))
    if ma.shape == mb.shape and not np.array_equal(ma.view(np.uint32), mb.view(np.uint32)):
        d = (ma.view(np.uint32) != mb.view(np.uint32)); print("     differing surfels:", int(d.any(axis=1).sum()), "cols", np.nonzero(d.any(axis=0))[0])
    print("  [%s] pose equal: %s" % (tag, np.array_equal(o.get_pose(), g.get_pose())))
    if not np.array_equal(o.get_pose(), g.get_pose()):
        print(o.get_pose()); print(g.get_pose())
    print("  [%s] images mismatching: %d" % (tag, bad))

for k, (c, d) in enumerate([("1c", "1d"), ("2c", "2d"), ("1c", "1d"), ("2c", "2d")]):
    t = time.time(); o.process_frame(f(c), f(d)); to = time.time() - t
    t = time.time(); g.process_frame(f(c), f(d)); g.synchronize(); tg = time.time() - t
    print("frame %d oracle %.3fs gpu %.4fs timings(ms) %s" % (k + 1, to, tg, np.round(g.timings(), 3)))
    print("   stats oracle", o.fuse_stats(), "gpu", g.fuse_stats(), "icp", o.last_icp(), g.last_icp(), "w", o.get_weighting(), g.get_weighting())
    cmp("f%d" % (k + 1))

"""re-run one draw of tests/gpu_fuzz_params.py (seed, index) several times and describe the first difference in detail"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
import gpu_fuzz_params as F
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.api import HRBFFusion
from hrbffusion3d_amd.params import IMAGES, default_params

seed, index = int(sys.argv[1]), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
over = eval(sys.argv[4]) if len(sys.argv) > 4 else {}
kw, plan = F.draw(seed, index)
plan.update(over)
print(kw); print(plan)
oracle_lib.build()
W, H = plan["size"]
for rep in range(reps):
    K, units = plan.get("K"), plan.get("depth_units", 5000.0)
    p = default_params(W, H, *synth.intrinsics(W, H, K), depth_scale=1.0 / units, max_surfels=1 << (17 if W * H <= 160 * 128 else 19), **kw)
    o = oracle_lib.Oracle(p, omp=True); g = HRBFFusion(p)
    if plan["shards"] > 1:
        g.comm_init(-1, plan["shards"]); g.map_shard_init(True, partition=plan["partition"]); g.set_row_sharding(bool(plan["row_sharding"]))
    first = 0
    if plan.get("seed_map", 0):
        seedm = F.garbage(synth.seed_map(plan["seed_map"], width=W, K=K), plan)
        rgb, d, T = synth.frame(plan["start"], W, H, noise=bool(plan["noise"]), depth_units=units, K=K)
        for x in (o, g):
            x.upload_map(seedm); x.set_pose(T); x.bootstrap(rgb, d)
        first = 1
    for k in range(first, plan["frames"] + first):
        rgb, d, _ = synth.frame(plan["start"] + k * plan["step"], W, H, noise=bool(plan["noise"]), depth_units=units, K=K)
        d = F.depth_of(plan, k, d)
        if plan.get("tick_jump") and k == plan["tick_jump"][0]:
            for x in (o, g):
                x.set_tick(x.tick + plan["tick_jump"][1])
        o.process_frame(rgb, d); g.process_frame(rgb, d)
        a, b = o.download_map(), g.download_map()
        same_img = all(np.array_equal(F.bits(o.get_image(n)), F.bits(g.get_image(n))) for n in IMAGES)
        if a.shape != b.shape or not np.array_equal(F.bits(a), F.bits(b)):
            print("rep %d frame %d: images equal %s, counts %d %d, status %d" % (rep, k, same_img, a.shape[0], b.shape[0], g.status()))
            if a.shape == b.shape:
                diff = F.bits(a) != F.bits(b)
                rows = np.flatnonzero(diff.any(1))
                print("  rows differing: %d, first %s last %s; columns %s" % (rows.size, rows[:8], rows[-3:], np.flatnonzero(diff.any(0))))
                for r in rows[:3]:
                    print("  row", r, "oracle", a[r]); print("  row", r, "gpu   ", b[r])
                # is the GPU map a permutation of the oracle's?
                sa = np.sort(F.bits(a).view([("", np.uint32)] * 20).ravel()); sb = np.sort(F.bits(b).view([("", np.uint32)] * 20).ravel())
                print("  same multiset of rows:", bool(np.array_equal(sa, sb)), "| fuse stats", o.fuse_stats(), g.fuse_stats())
            break
    else:
        print("rep %d: no difference" % rep)
    o.close(); g.close()

"""Random COMBINATIONS of the reference's switches (GUI/GlobalStateParam.txt:20-81, the HRBFFusion ctor arguments), HIP path against
the oracle, bit for bit: tests/test_parity_gpu.py::test_parameter_variants turns the switches one at a time; here every context gets
a random draw of all of them together (the windowed search WITH the sparse variant WITH frame-to-frame RGB WITHOUT the pyramid ...),
a random window of the synthetic stream, now and then a ragged or an empty depth image in the middle, sometimes an uploaded map of the
scene to track against, a jump of the frame counter, and in half of the contexts' plans the map cut into 2-4 shards (contiguous ranges
or hash-owned, re-cut in the middle, the registration row-sharded or not); other image shapes and intrinsics (fx != fy, off-centre, a
negative fy), millimetre depth, and garbage rows (NaN / inf / negative values) in the uploaded map.

    python tests/gpu_fuzz_params.py N [seed] [out]     # N contexts of 4-6 frames; appends to gpurun_out/param_fuzz.txt (or `out`)
    HRBF_FUZZ_VGA=1 python tests/gpu_fuzz_params.py N ...   # every context at 640 x 480, two thirds of them against a 0.3 M / 1 M map
    HRBF_FUZZ_RCCL1=1 python tests/gpu_fuzz_params.py N ...   # sharded draws over a real RCCL communicator of world size 1 instead of virtual shards
    HRBF_FUZZ_SHAPES=tiny|hd python tests/gpu_fuzz_params.py N ...   # 8 x 8 ... 640 x 8 images / 1280 x 960

A mismatch is logged with the draw that produced it (the seed and the index reproduce it) and the run goes on.
tests/test_parity_gpu.py::test_random_parameter_combinations runs draws 0-9 of seed 1 in the suite, and the one draw that ever differed.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        u = a.view(np.uint32).copy(); u[np.isnan(a)] = 0x7FC00000       # NaN sign / payload carry no meaning (tests/test_parity_gpu.bits)
        return u
    return a.view(np.uint8)


def draw(seed, index):
    """draw `index` of `seed`: its own generator, so a draw does not depend on the ones before it and fields appended later leave
    the earlier fields of every draw as they were"""
    return _draw(np.random.default_rng([seed, index]))


def _draw(rng):
    """one random setting of every switch the path reads; the ranges are what the reference's parameter file documents or uses (the
    HRBF windows at most the reference's defaults: the library refuses larger ones, DESIGN.md deviations)"""
    c = lambda *v: v[int(rng.integers(len(v)))]
    flip = lambda p=0.5: int(rng.random() < p)
    kw = dict(
        use_bilateral=flip(0.7), normal_estimation_pca=float(flip(0.7)), so3=flip(0.7), pyramid=flip(0.8), fast_odom=flip(0.3),
        use_conf_eval=flip(0.3), conf_eval_epsilon=float(c(100.0, 1000.0, 2500.0)), rgb_only=flip(0.1),
        icp_weight=float(c(1.0, 10.0, 10.0, 25.0, 100.0)), icp_use_corr_search=flip(0.4), icp_search_radius=int(c(1, 2, 2, 3)),
        use_sparse_icp=flip(0.3), clean_window_multiplier=float(c(1.0, 1.5, 2.0, 2.0, 2.25, 3.0, 4.0)), frame_to_frame_rgb=flip(0.3),
        rgb_use_grad_weight=flip(0.3), icp_use_weighted=flip(0.7), icp_curv_weight_lambda=float(c(0.0, 5.0, 10.0, 10.0, 30.0)),
        predict_window_multiplier=float(c(2.0, 2.5, 3.0, 3.0)), predict_min_neighbors=int(c(3, 4, 6, 6, 8)),
        curv_estimation_window=float(c(2.0, 3.0, 3.0)), curv_valid_threshold=float(c(100.0, 150.0, 300.0, 300.0, 500.0)),
        confidence_threshold=float(c(1.0, 2.0, 5.0, 5.0, 10.0)), depth_cutoff=float(c(2.0, 2.5, 3.5, 3.5, 5.0)),
        dense_enough_thresh=float(c(0.5, 0.75, 0.75, 0.99)), predict_conf_threshold=float(c(1.0, 3.0, 3.0, 5.0)),
        init_radius_multiplier=float(c(2.0, 3.0, 4.0, 4.0, 5.0)), max_depth_processed=float(c(20.0, 20.0, 4.0)))
    kw["predict_max_neighbors"] = kw["predict_min_neighbors"] + int(c(0, 2, 4, 6))
    size = c((160, 120), (160, 120), (160, 120), (320, 240))
    plan = dict(size=size, start=int(rng.integers(0, 400)), step=int(c(1, 1, 2, 3)), frames=int(c(4, 5, 6)), noise=flip(0.7),
                odd_frame=c(None, None, "ragged", "empty", "far"), odd_at=int(rng.integers(1, 4)),
                # the sharded map (one process playing G shards), the row-sharded registration, a re-cut of the ranges in the middle
                shards=int(c(0, 0, 0, 2, 3, 4)), partition=c("ranges", "hash"), row_sharding=flip(), rebalance_at=c(None, 1, 2, 3),
                # tracking against an uploaded map of the scene instead of an empty one; a jump of the frame counter (stale surfels go)
                seed_map=int(c(0, 0, 0, 8000, 40000)), tick_jump=c(None, None, None, (2, 40), (3, 400)))
    # geometry: other image shapes (multiples of 8), fx != fy, an off-centre principal point, now and then the negative fy ICL-NUIM
    # publishes; raw depth in 1 / 5000 m or in millimetres; some rows of the uploaded map replaced by garbage (NaN / inf positions, zero
    # normals, negative radii, absurd confidences): whatever the path does with them, both sides must do the same
    plan["size"] = c(plan["size"], plan["size"], plan["size"], (168, 120), (160, 128), (208, 152), (256, 128))
    W, H = plan["size"]
    if flip(0.5):
        fx = 0.825 * W * float(rng.uniform(0.8, 1.25)); fy = fx * float(rng.uniform(0.9, 1.1)) * (-1.0 if flip(0.1) else 1.0)
        plan["K"] = (fx, fy, W * float(rng.uniform(0.4, 0.6)), H * float(rng.uniform(0.4, 0.6)))
    else:
        plan["K"] = None
    plan["depth_units"] = float(c(5000.0, 5000.0, 1000.0))
    plan["garbage_rows"] = int(c(0, 0, 50)) if plan["seed_map"] else 0
    plan["garbage_seed"] = int(rng.integers(1 << 30))
    shapes = os.environ.get("HRBF_FUZZ_SHAPES")      # tiny: images of a few texels and extreme aspect ratios (one-workgroup grids, pyramids
    if shapes in ("tiny", "hd"):                     # down to 2 x 2, a thumbnail of no cells); hd: 1280 x 960, BASELINE config 5's geometry
        W0, H0 = W, H
        if shapes == "tiny":
            plan.update(size=c((8, 8), (16, 8), (8, 16), (24, 16), (40, 32), (64, 8), (8, 64), (80, 56), (640, 8), (8, 480), (16, 480), (640, 16)),
                        seed_map=int(c(0, 0, 2000)), garbage_rows=0)
        else:
            plan.update(size=(1280, 960), seed_map=int(c(0, 500_000)), frames=3, garbage_rows=0)
        if plan["K"] is not None:
            plan["K"] = tuple(v * plan["size"][0] / W0 if i in (0, 2) else v * plan["size"][1] / H0 for i, v in enumerate(plan["K"]))
        return kw, plan
    if os.environ.get("HRBF_FUZZ_VGA"):       # the benchmark's size: several fuse tiles per workgroup, the 1200-workgroup reduction grids,
        big = c(0, 300_000, 1_000_000)        # the hipGraph replay from frame 3 on; a map of the benchmark's size to track against
        plan.update(size=(640, 480), seed_map=int(big), frames=int(c(3, 4, 5)), garbage_rows=int(c(0, 200)) if big else 0)
        if plan["K"] is not None:
            plan["K"] = tuple(v * 640.0 / W if i in (0, 2) else v * 480.0 / H for i, v in enumerate(plan["K"]))
    return kw, plan


def garbage(seed_map, plan):
    if not plan.get("garbage_rows"):
        return seed_map
    rng = np.random.default_rng(plan["garbage_seed"])
    m = seed_map.copy()
    rows = rng.choice(m.shape[0], plan["garbage_rows"], replace=False)
    vals = np.array([np.nan, np.inf, -np.inf, 0.0, -1.0, 1e30, -1e30, 1e-40, 3e9], np.float32)
    for r in rows:
        kind = int(rng.integers(5))
        if kind == 0:
            m[r, 0:3] = vals[rng.integers(len(vals), size=3)]            # position
        elif kind == 1:
            m[r, 8:11] = vals[rng.integers(len(vals), size=3)]           # normal
        elif kind == 2:
            m[r, 11] = vals[rng.integers(len(vals))]                     # radius
        elif kind == 3:
            m[r, 3] = vals[rng.integers(len(vals))]                      # confidence
        else:
            m[r, int(rng.integers(12, 20))] = vals[rng.integers(len(vals))]   # a curvature record
    return m


def depth_of(plan, k, d):
    if plan["odd_frame"] is None or k != plan["odd_at"]:
        return d
    if plan["odd_frame"] == "empty":
        return np.zeros_like(d)
    if plan["odd_frame"] == "far":
        return np.full_like(d, 60000)
    r = d.copy(); r[::3, ::2] = 0; r[:7] = 0; r[:, -5:] = 0
    return r


def run_one(oracle_lib, kw, plan):
    """None when HIP == oracle on every image, the map and the pose after every frame; else a description of the first difference"""
    from hrbffusion3d_amd import synth
    from hrbffusion3d_amd.api import HRBFFusion
    from hrbffusion3d_amd.params import IMAGES, default_params
    W, H = plan["size"]
    K, units = plan.get("K"), plan.get("depth_units", 5000.0)
    p = default_params(W, H, *synth.intrinsics(W, H, K), depth_scale=1.0 / units, max_surfels=(1 << (17 if W * H <= 160 * 128 else 19)) if W * H < 640 * 480 else plan.get("seed_map", 0) + (1 << (20 if W * H == 640 * 480 else 22)), **kw)
    o = g = None
    try:
        try:
            o = oracle_lib.Oracle(p, omp=True)
        except Exception as e:
            o, eo = None, e
        try:
            g = HRBFFusion(p)
        except Exception as e:
            g, eg = None, e
        if o is None or g is None:
            return None if (o is None and g is None) else "only one side accepts the parameters (oracle %s, HIP %s)" % (
                "ok" if o else repr(eo), "ok" if g else repr(eg))
        hashed = plan.get("shards", 0) > 1 and plan["partition"] == "hash"
        if plan.get("shards", 0) > 1:
            if os.environ.get("HRBF_FUZZ_RCCL1"):      # a real RCCL communicator of world size 1: every collective of the sharded map and of
                g.comm_init(0, 1, HRBFFusion.comm_unique_id())      # the row-sharded registration is issued through librccl
            else:
                g.comm_init(-1, plan["shards"])
            g.map_shard_init(True, partition=plan["partition"])
            g.set_row_sharding(bool(plan["row_sharding"]))
        first = 0
        if plan.get("seed_map", 0):
            seed = garbage(synth.seed_map(plan["seed_map"], width=W, K=K), plan)
            rgb, d, T = synth.frame(plan["start"], W, H, noise=bool(plan["noise"]), depth_units=units, K=K)
            for x in (o, g):
                x.upload_map(seed); x.set_pose(T); x.bootstrap(rgb, d)
            first = 1
        for k in range(first, plan["frames"] + first):
            rgb, d, _ = synth.frame(plan["start"] + k * plan["step"], W, H, noise=bool(plan["noise"]), depth_units=units, K=K)
            d = depth_of(plan, k, d)
            if plan.get("tick_jump") and k == plan["tick_jump"][0]:
                for x in (o, g):
                    x.set_tick(x.tick + plan["tick_jump"][1])
            o.process_frame(rgb, d); g.process_frame(rgb, d)
            if plan.get("shards", 0) > 1 and not hashed and plan.get("rebalance_at") == k:
                g.map_rebalance()
            for name in IMAGES:
                if hashed and name == "INDEX":
                    continue            # ids instead of array positions under hash ownership: names only
                if not np.array_equal(bits(o.get_image(name)), bits(g.get_image(name))):
                    return "frame %d: image %s differs in %d values" % (k, name, int((bits(o.get_image(name)) != bits(g.get_image(name))).sum()))
            if o.surfel_count() != g.surfel_count():
                return "frame %d: surfel count %d vs %d" % (k, o.surfel_count(), g.surfel_count())
            if not np.array_equal(bits(o.download_map()), bits(g.download_map())):
                return "frame %d: map differs" % k
            if not np.array_equal(bits(o.get_pose()), bits(g.get_pose())):
                return "frame %d: pose differs" % k
        return None
    finally:
        for x in (o, g):
            if x is not None:
                x.close()


def main():
    import oracle_lib
    n = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "param_fuzz.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    oracle_lib.build()
    bad = 0
    t0 = time.time()
    tally = {}
    with open(out, "a") as log:
        log.write("# seed %d, %d draws: every switch drawn at random per context, 4-6 frames, all images + map + pose compared bit for bit\n" % (seed, n))
        for i in range(n):
            kw, plan = draw(seed, i)
            log.write("draw %d\n" % i) if os.environ.get("HRBF_FUZZ_TRACE") else None; log.flush()
            r = run_one(oracle_lib, kw, plan)
            for k in ("use_sparse_icp", "icp_use_corr_search", "frame_to_frame_rgb", "rgb_only", "pyramid", "so3", "use_conf_eval"):
                tally[k] = tally.get(k, 0) + int(kw[k])
            tally[plan["size"]] = tally.get(plan["size"], 0) + 1
            tally[plan["odd_frame"]] = tally.get(plan["odd_frame"], 0) + 1
            for k in ("shards", "seed_map", "garbage_rows"):
                tally["%s>0" % k] = tally.get("%s>0" % k, 0) + int(plan[k] > 0)
            tally["hash"] = tally.get("hash", 0) + int(plan["shards"] > 1 and plan["partition"] == "hash")
            tally["tick_jump"] = tally.get("tick_jump", 0) + int(plan["tick_jump"] is not None)
            tally["own K"] = tally.get("own K", 0) + int(plan["K"] is not None)
            tally["fy<0"] = tally.get("fy<0", 0) + int(plan["K"] is not None and plan["K"][1] < 0)
            tally["mm depth"] = tally.get("mm depth", 0) + int(plan["depth_units"] == 1000.0)
            if r is not None:
                bad += 1
                log.write("MISMATCH draw %d of seed %d: %s\n    %r\n    %r\n" % (i, seed, r, kw, plan))
            if i % 20 == 19 or i == n - 1:
                log.write("draw %d: %d mismatches so far, %.0f s\n" % (i + 1, bad, time.time() - t0)); log.flush()
        log.write("done: %d contexts, %d mismatches; drawn on: %s\n" % (n, bad, ", ".join("%s %d" % (k, v) for k, v in tally.items())))
    print("param fuzz: %d contexts, %d mismatches" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

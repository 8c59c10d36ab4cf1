"""Random SEQUENCES of API calls on one context, HIP path against the oracle, bit for bit.

The oracle recomputes everything from its state on every call; the library keeps things between calls — the captured Gauss-Newton
graph, the choice between the fused and the separate per-pixel kernels (a flag that hrbf_set_image may leave stale), the class bytes
of the clean pass, device-side counters, shard cuts, the event ring.  A call in an order no test thought of is where such state goes
wrong.  Each trial draws 10-24 operations:

    frames (most of them; some with a weight multiplier, some through the device-pointer entry), set_pose (a small or a large jump),
    set_tick, set_weighting, the run-time switches (rgb only, ICP weight, pyramid, fast odometry, SO3, frame-to-frame RGB, the confidence
    threshold, the depth cut-off), upload of a re-ordered / thinned / doubled map, updateModel with per-submap corrections, the submap index and
    the active-submap mask, a stage run in isolation, an image read back and set again unchanged, and on the library's side only (they
    must not change a result): timing on / off, the ring stride, the map re-cut into shards or back to one, synchronise, reads.

After every frame all images, the map and the pose are compared; after every other operation the map, the count and the pose.

    python tests/gpu_fuzz_api.py N [seed] [out]       # N trials; appends to gpurun_out/api_fuzz.txt (or `out`)
    HRBF_FUZZ_INTERLEAVED=1 python tests/gpu_fuzz_api.py N ...   # three contexts alive at once, their calls interleaved at random
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from gpu_fuzz_params import bits  # noqa: E402

W, H = 160, 120


def small_rigid(rng, rot, trans):
    a = rng.normal(0, rot, 3)
    th = float(np.linalg.norm(a))
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th ** 2) * (K @ K) if th > 1e-12 else np.eye(3)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = rng.normal(0, trans, 3)
    return T.astype(np.float32)


def trial_steps(oracle_lib, seed, index):
    """generator: one API call per step; returns (StopIteration.value) None or the description of the first difference"""
    from hrbffusion3d_amd import synth
    from hrbffusion3d_amd.api import HRBFFusion
    from hrbffusion3d_amd.params import IMAGES, default_params
    rng = np.random.default_rng([seed, index])
    c = lambda *v: v[int(rng.integers(len(v)))]
    kw = dict(icp_use_corr_search=int(rng.random() < 0.25), use_sparse_icp=int(rng.random() < 0.25), frame_to_frame_rgb=int(rng.random() < 0.25),
              so3=int(rng.random() < 0.8), pyramid=int(rng.random() < 0.85), use_conf_eval=int(rng.random() < 0.2))
    p = default_params(W, H, *synth.intrinsics(W, H), max_surfels=1 << 17, **kw)
    o, g = oracle_lib.Oracle(p, omp=True), HRBFFusion(p)
    both = (o, g)
    k = int(rng.integers(0, 300))
    ops = []
    sharded = 0          # "hash": INDEX holds ids instead of array positions until a projection has run under another cut
    lent = False
    stale_names, cut = False, 0

    def same(full):
        if full:
            for n in IMAGES:
                if sharded == "hash" and n == "INDEX":
                    continue
                if not np.array_equal(bits(o.get_image(n)), bits(g.get_image(n))):
                    return "image %s differs in %d values" % (n, int((bits(o.get_image(n)) != bits(g.get_image(n))).sum()))
        if o.surfel_count() != g.surfel_count():
            return "surfel count %d vs %d" % (o.surfel_count(), g.surfel_count())
        if not np.array_equal(bits(o.download_map()), bits(g.download_map())):
            return "map differs"
        if not np.array_equal(bits(o.get_pose()), bits(g.get_pose())):
            return "pose differs"
        if o.tick != g.tick:
            return "tick %d vs %d" % (o.tick, g.tick)
        return None

    try:
        n_ops = int(rng.integers(10, 25))
        for step in range(n_ops):
            op = c("frame", "frame", "frame", "frame", "frame", "frame_w", "frame_dev", "set_pose", "set_tick", "set_weighting", "upload_map",
                   "update_model", "index_submap", "active_submaps", "stage", "image_roundtrip", "timing", "ring", "shards", "sync_reads", "switch", "switch") \
                if step > 1 else "frame"
            if op == "frame_dev" and os.environ.get("HRBF_FUZZ_NO_DEVICE"):      # tests/oracle_only.py: the oracle on both sides
                op = "frame"
            ops.append(op)
            full = False
            if op in ("frame", "frame_w", "frame_dev"):
                rgb, d, _ = synth.frame(k, W, H, noise=True); k += int(c(1, 1, 1, 2, 4))
                wm = float(c(1.0, 0.5, 2.0, 0.25)) if op == "frame_w" else 1.0
                o.process_frame(rgb, d, 0, wm)
                if op == "frame_dev":
                    import torch
                    tr, td = torch.from_numpy(rgb).cuda(), torch.from_numpy(d.view(np.int16)).cuda()
                    g.process_frame_device(tr.data_ptr(), td.data_ptr(), 0)
                    g.synchronize()
                    del tr, td                  # valid until the synchronise: the contract of the entry (include/hrbf_mi355.h)
                    lent = True
                else:
                    g.process_frame(rgb, d, 0, wm)
                    lent = False
                if stale_names:                 # ... until a frame has projected the map under the new cut
                    sharded, stale_names = cut, False
                full = True
            elif op == "set_pose":
                T = (o.get_pose().astype(np.float64) @ small_rigid(rng, *c((0.002, 0.002), (0.02, 0.03), (0.3, 0.5)))).astype(np.float32)
                for x in both:
                    x.set_pose(T)
            elif op == "set_tick":
                t = o.tick + int(c(1, 5, 60, 500))
                for x in both:
                    x.set_tick(t)
            elif op == "set_weighting":
                w = float(c(1.0, 0.75, 0.5))
                for x in both:
                    x.set_weighting(w)
            elif op == "upload_map":
                m = o.download_map()
                if m.shape[0] > 10:
                    how = c("permute", "thin", "double", "same", "empty")
                    if how == "permute":
                        m = m[rng.permutation(m.shape[0])]
                    elif how == "thin":
                        m = m[rng.random(m.shape[0]) < 0.5]
                    elif how == "double":
                        m = np.concatenate([m, m[: min(m.shape[0], (1 << 17) - m.shape[0] - 20000)]])
                    elif how == "empty":
                        m = m[:0]
                    ops[-1] += ":" + how
                    for x in both:
                        x.upload_map(np.ascontiguousarray(m))
            elif op == "update_model":
                D = np.stack([small_rigid(rng, 0.003, 0.003) for _ in range(int(c(1, 2, 3)))])
                for x in both:
                    x.update_model(D)
            elif op == "index_submap":
                s = int(c(0, 1, 2))
                for x in both:
                    x.set_index_submap(s)
            elif op == "active_submaps":
                a = c(None, [1, 1, 1], [1, 0, 1], [0, 1, 1], [1])
                for x in both:
                    x.set_active_submaps(a)
            elif op == "stage":
                s = c("PREDICT_INDICES", "PREDICT_HRBF", "FILLIN", "CONFIDENCE", "CURVATURE", "VERTEX_NORMAL_RADIUS")
                ops[-1] += ":" + s
                if s == "FILLIN" and lent:          # the context does not keep a frame that came by device pointer: the seam must say so
                    try:
                        g.run_stage(s)
                        return "stage FILLIN ran on raw images the context does not hold | ops: %s" % " ".join(ops)
                    except Exception as e:
                        if "hrbf_upload_frame" not in str(e):
                            raise
                    for x in both:
                        x.upload_frame(rgb, d)
                    lent = False
                for x in both:
                    x.run_stage(s)
                full = True
            elif op == "image_roundtrip":
                n = c(*[n for n in IMAGES])
                ops[-1] += ":" + n
                a = o.get_image(n)
                for x in both:
                    x.set_image(n, a)
            elif op == "timing":
                g.enable_timing(int(c(0, 1, 2)))
            elif op == "ring":
                g.set_fuse_ring_stride(int(c(1, 2, 4))); g.reset_fuse_ring()
            elif op == "shards":
                G = int(c(1, 2, 3, 4))
                part = c("ranges", "hash")
                ops[-1] += ":%d%s" % (G, part if G > 1 else "")
                m = o.download_map()                  # the cut is made on an empty map (hrbf_map_shard_init), the map uploaded afterwards
                g.upload_map(m[:0])
                g.comm_init(-1, G)
                if G > 1:
                    g.map_shard_init(True, partition=part); g.set_row_sharding(bool(c(0, 1)))
                else:
                    g.map_shard_init(False)
                was_hash, cut = sharded == "hash", (part if G > 1 else 0)
                sharded = "hash" if (was_hash or cut == "hash") else cut          # INDEX still holds the old cut's names ...
                stale_names = was_hash and cut != "hash"
                for x in both:
                    x.upload_map(m)
            elif op == "switch":
                name, v = c(("rgb_only", c(0, 1)), ("icp_weight", c(1.0, 10.0, 100.0)), ("pyramid", c(0, 1)), ("fast_odom", c(0, 1)), ("so3", c(0, 1)),
                            ("frame_to_frame_rgb", c(0, 1)), ("confidence_threshold", c(2.0, 5.0, 10.0)), ("depth_cutoff", c(2.5, 3.5, 5.0)))
                ops[-1] += ":%s=%s" % (name, v)
                o.set_switch(name, v); getattr(g, "set_" + name)(v)
            elif op == "sync_reads":
                g.synchronize(); g.fuse_stats(); g.status(); g.local_surfel_count(); g.timings()
            r = same(full)
            if r:
                return "after op %d (%s): %s | ops: %s | %r" % (step, ops[-1], r, " ".join(ops), kw)
            yield
        return None
    except Exception as e:
        return "exception %r after ops: %s | %r" % (e, " ".join(ops), kw)
    finally:
        o.close(); g.close()


def trial(oracle_lib, seed, index):
    it = trial_steps(oracle_lib, seed, index)
    while True:
        try:
            next(it)
        except StopIteration as e:
            return e.value


def trial_interleaved(oracle_lib, seed, index, n_ctx=3):
    """n_ctx contexts alive at once on one device, their calls interleaved at random: what one context caches (graphs, function
    attributes set once, scratch) must not be another's"""
    rng = np.random.default_rng([seed, index, 77])
    its = {j: trial_steps(oracle_lib, seed, index * 16 + j) for j in range(n_ctx)}
    while its:
        j = list(its)[int(rng.integers(len(its)))]
        try:
            next(its[j])
        except StopIteration as e:
            del its[j]
            if e.value is not None:
                for it in its.values():
                    it.close()
                return "context %d of %d interleaved: %s" % (j, n_ctx, e.value)
    return None


def main():
    import oracle_lib
    n = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "api_fuzz.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    oracle_lib.build()
    import torch
    assert torch.cuda.is_available(), "the device-pointer entry needs torch's HIP context"
    torch.zeros(1).cuda()
    bad, t0 = 0, time.time()
    with open(out, "a") as log:
        log.write("# seed %d, %d trials: 10-24 random API calls per context, compared after every call\n" % (seed, n))
        for i in range(n):
            if os.environ.get("HRBF_FUZZ_TRACE"):
                log.write("trial %d\n" % i); log.flush()
            r = (trial_interleaved if os.environ.get("HRBF_FUZZ_INTERLEAVED") else trial)(oracle_lib, seed, i)
            if r is not None:
                bad += 1
                log.write("MISMATCH trial %d of seed %d: %s\n" % (i, seed, r)); log.flush()
        log.write("done: %d trials, %d mismatches, %.0f s\n" % (n, bad, time.time() - t0))
    print("api fuzz: %d trials, %d mismatches" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

"""Random draws of tests/gpu_fuzz_params.py played by 2-3 REAL processes that share one map over the shared-memory rendezvous (the
transport that lets several ranks live on one device; tests/test_peer_shards_gpu.py holds its fixed cases), against the oracle in the
parent process: every rank's pose and the global surfel count after every frame, and at the end the ranks' map slices — concatenated
for contiguous ranges, as a multiset of rows under hash ownership — equal to the oracle's single map, bit for bit.

    python tests/gpu_fuzz_peers.py N [seed] [out]       # N trials; appends to gpurun_out/peer_fuzz.txt (or `out`)
"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import gpu_fuzz_params as F  # noqa: E402


def play(x, kw, plan, after_frame):
    """the frame loop of gpu_fuzz_params.run_one for ONE implementation"""
    from hrbffusion3d_amd import synth
    W, H = plan["size"]
    K, units = plan.get("K"), plan.get("depth_units", 5000.0)
    first = 0
    if plan.get("seed_map", 0):
        seed = F.garbage(synth.seed_map(plan["seed_map"], width=W, K=K), plan)
        rgb, d, T = synth.frame(plan["start"], W, H, noise=bool(plan["noise"]), depth_units=units, K=K)
        x.upload_map(seed); x.set_pose(T); x.bootstrap(rgb, d)
        first = 1
    for k in range(first, plan["frames"] + first):
        rgb, d, _ = synth.frame(plan["start"] + k * plan["step"], W, H, noise=bool(plan["noise"]), depth_units=units, K=K)
        d = F.depth_of(plan, k, d)
        if plan.get("tick_jump") and k == plan["tick_jump"][0]:
            x.set_tick(x.tick + plan["tick_jump"][1])
        x.process_frame(rgb, d)
        after_frame(k)


def params_of(kw, plan):
    from hrbffusion3d_amd import synth
    from hrbffusion3d_amd.params import default_params
    W, H = plan["size"]
    return default_params(W, H, *synth.intrinsics(W, H, plan.get("K")), depth_scale=1.0 / plan.get("depth_units", 5000.0),
                          max_surfels=1 << (17 if W * H <= 160 * 128 else 19), **kw)


def rank_main(rank, world, uid, kw, plan, peer, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        from hrbffusion3d_amd.api import HRBFFusion
        g = HRBFFusion(params_of(kw, plan))
        g.comm_init_peer(rank, world, uid)
        if peer["partition"]:
            g.map_shard_init(True, partition=peer["partition"])
        g.set_row_sharding(bool(peer["rows"]))
        res = {"pose": [], "count": []}

        def after(k):
            res["pose"].append(F.bits(g.get_pose()).copy()); res["count"].append(g.surfel_count())
        play(g, kw, plan, after)
        res["map"] = F.bits(g.download_map()).copy()
        res["status"] = g.status()
        g.close()
        out.put((rank, res))
    except Exception as e:
        import traceback
        out.put((rank, {"error": "%r\n%s" % (e, traceback.format_exc()[-600:])}))


def trial(oracle_lib, seed, index):
    from hrbffusion3d_amd.api import HRBFFusion
    kw, plan = F.draw(seed, index)
    rng = np.random.default_rng([seed, index, 99])
    c = lambda *v: v[int(rng.integers(len(v)))]
    world = int(c(2, 2, 3))
    peer = {"partition": c("ranges", "hash", None), "rows": int(rng.random() < 0.5)}
    if peer["partition"] is None:
        peer["rows"] = 1
    if plan["size"][0] * plan["size"][1] > 320 * 240:
        plan["size"] = (320, 240)
    o = oracle_lib.Oracle(params_of(kw, plan), omp=True)
    want = {"pose": [], "count": []}
    try:
        play(o, kw, plan, lambda k: (want["pose"].append(F.bits(o.get_pose()).copy()), want["count"].append(o.surfel_count())))
        want["map"] = F.bits(o.download_map()).copy()
    finally:
        o.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    uid = HRBFFusion.peer_unique_id()
    procs = [ctx.Process(target=rank_main, args=(r, world, uid, kw, plan, peer, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = {}
    try:
        for _ in range(world):
            r, res = q.get(timeout=240)
            got[r] = res
    except Exception:
        for pr in procs:
            pr.kill()
        HRBFFusion.peer_release_id(uid)
        return "ranks did not answer within 240 s (world %d, %r) | %r | %r" % (world, peer, kw, plan)
    for pr in procs:
        pr.join(timeout=60)
    tag = "world %d, %r | %r | %r" % (world, peer, kw, plan)
    for r in range(world):
        if "error" in got[r]:
            return "rank %d: %s | %s" % (r, got[r]["error"], tag)
    for r in range(world):
        for k, (a, b) in enumerate(zip(got[r]["pose"], want["pose"])):
            if not np.array_equal(a, b):
                return "rank %d: pose differs after frame %d | %s" % (r, k, tag)
        if got[r]["count"] != want["count"]:
            return "rank %d: surfel counts %r vs oracle %r | %s" % (r, got[r]["count"], want["count"], tag)
    if peer["partition"]:
        rows = np.concatenate([got[r]["map"].reshape(-1, 20) for r in range(world)])
    else:
        rows = got[0]["map"].reshape(-1, 20)           # every rank holds the whole map when only the registration is sharded
        for r in range(1, world):
            if not np.array_equal(got[r]["map"], got[0]["map"]):
                return "ranks 0 and %d hold different maps | %s" % (r, tag)
    ref = want["map"].reshape(-1, 20)
    if rows.shape != ref.shape:
        return "the ranks hold %d surfels, the oracle %d | %s" % (rows.shape[0], ref.shape[0], tag)
    if peer["partition"] == "hash":                   # ownership by cell: the same rows in another order
        key = lambda m: m[np.lexsort(m.T[::-1])]
        rows, ref = key(rows), key(ref)
    if not np.array_equal(rows, ref):
        return "the ranks' map differs from the oracle's in %d rows | %s" % (int((rows != ref).any(1).sum()), tag)
    return None


def main():
    import oracle_lib
    n = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "peer_fuzz.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    oracle_lib.build()
    bad, t0 = 0, time.time()
    with open(out, "a") as log:
        log.write("# seed %d, %d trials: a random draw played by 2-3 processes over the shared-memory rendezvous, against the oracle\n" % (seed, n))
        for i in range(n):
            if os.environ.get("HRBF_FUZZ_TRACE"):
                log.write("trial %d\n" % i); log.flush()
            r = trial(oracle_lib, seed, i)
            if r is not None:
                bad += 1
                log.write("MISMATCH trial %d of seed %d: %s\n" % (i, seed, r)); log.flush()
        log.write("done: %d trials, %d mismatches, %.0f s\n" % (n, bad, time.time() - t0))
    print("peer fuzz: %d trials, %d mismatches" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

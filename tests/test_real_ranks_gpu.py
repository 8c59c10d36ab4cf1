"""Real RCCL ranks on DIFFERENT HIP devices: the code no single-device box can reach.

tests/test_peer_shards_gpu.py runs two ranks on ONE device over the shared-memory rendezvous (RCCL refuses that), and
tests/test_hash_shards_gpu.py runs RCCL with world 1.  Neither executes ncclCommInitRank with a peer, the all-reduce of the
29-limb system between devices, hipIpcOpenMemHandle of another device's images, or the ncclSend/ncclRecv record exchange.
This module does — whenever the box enumerates >= 2 HIP devices: a multi-GPU node, or ONE MI355X switched to a CPX/DPX
compute partition, where every XCD group is its own device and RCCL accepts one rank per partition
(tools/partition_probe.sh tries that on the leased box and records the outcome under profiles/).
With one device every test here SKIPS (it is not a failure of the single-GPU tier).

A correctness run, not a scaling claim: each case must be bit-identical to one process holding the whole map."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
IMAGES_CHECKED = ("INDEX", "INDEX_VERTCONF", "INDEX_COLORTIME", "INDEX_NORMRAD", "INDEX_CURVMAX", "INDEX_CURVMIN", "PRED_VERTEX",
                  "PRED_NORMAL", "PRED_ICPWEIGHT", "FILL_VERTEX", "CONFIDENCE")


def _device_count():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


needs_two_devices = pytest.mark.skipif(_device_count() < 2, reason="needs >= 2 HIP devices (multi-GPU node or a CPX/DPX partition)")


def _bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        u = a.view(np.uint32).copy(); u[np.isnan(a)] = 0x7FC00000
        return u
    return a.view(np.uint8)


def _run(rank, world, uid, cfg, out):
    sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
    from hrbffusion3d_amd import synth
    from hrbffusion3d_amd.api import HRBFFusion
    from hrbffusion3d_amd.params import default_params
    try:
        W, H, nseed, frames, sparse, partition, rows, exchange, renumber_at = cfg
        if exchange:
            os.environ["HRBF_SHARD_EXCHANGE"] = exchange
        if renumber_at:
            os.environ["HRBF_HASH_RENUMBER_AT"] = str(renumber_at)
        K = synth.intrinsics(W, H)
        seed = synth.seed_map(nseed, width=W) if nseed else None
        p = default_params(W, H, *K, max_surfels=(nseed or 0) + 300_000, use_sparse_icp=sparse)
        g = HRBFFusion(p, device=rank if world > 1 else 0)      # one rank per device
        if world > 1:
            g.comm_init(rank, world, uid)                       # ncclCommInitRank with real peers
            g.set_row_sharding(rows)
            if partition:
                g.map_shard_init(True, partition=partition)
        rgb, d, T = synth.frame(0, W, H, noise=True)
        if seed is not None:
            g.upload_map(seed); g.set_pose(T); g.bootstrap(rgb, d)
            first = 1
        else:
            first = 0
        res = {}
        for k in range(first, frames):
            rgb, d, T = synth.frame(k, W, H, noise=True)
            g.process_frame(rgb, d)
            res["pose%d" % k] = _bits(g.get_pose())
            res["stats%d" % k] = g.fuse_stats()
            res["count%d" % k] = g.surfel_count()
            res["icp%d" % k] = np.asarray(g.last_icp(), np.float32).view(np.uint32)
            if k in (first, frames - 1):
                for name in IMAGES_CHECKED:
                    res["%s%d" % (name, k)] = _bits(g.get_image(name))
        res["local_count"] = g.local_surfel_count()
        res["map"] = _bits(g.download_map())
        if world > 1 and partition == "hash":
            res["gids"] = g.download_gids()
            res["renumbered"] = g.hash_renumber_count()
        res["exchange_mode"] = g.shard_exchange_mode()
        res["status"] = g.status()
        res["comm"] = g.comm_stats()            # the library's own account: what ITS communicator says, what it issued (hrbf_comm_stats)
        res["tracked"] = frames - first - (0 if seed is not None else 1)      # frames that ran the registration + fuse path
        g.close()
        out.put((rank, res))
    except Exception as e:   # surface the failure in the parent instead of a hang
        import traceback
        out.put((rank, {"error": "%r\n%s" % (e, traceback.format_exc())}))


def _launch(world, cfg):
    from hrbffusion3d_amd.api import HRBFFusion
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    uid = HRBFFusion.comm_unique_id() if world > 1 else None
    procs = [ctx.Process(target=_run, args=(r, world, uid, cfg, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = {}
    try:
        for _ in range(world):
            r, res = q.get(timeout=600)
            got[r] = res
    finally:
        for pr in procs:
            pr.join(timeout=60)
            if pr.is_alive():
                pr.kill()
    for r in got:
        assert "error" not in got[r], got[r]["error"]
    return got


def _check_comm(many, G, cfg):
    """every rank's library-side record: RCCL's own world size and rank for the library's communicator, and the per-frame exchange
    steps of DESIGN.md section 7's model (tests/test_comm_stats_gpu.py asserts the same numbers on one device)"""
    W, H, nseed, frames, sparse, partition, rows, exchange, renumber_at = cfg
    assert sorted(many[r]["comm"]["rank"] for r in range(G)) == list(range(G))
    for r in range(G):
        c, n = many[r]["comm"], many[r]["tracked"]
        assert c["transport"] == "rccl" and c["world"] == G and c["rank"] == r, c
        if rows:
            assert c["limb_allreduce"] == (10 + 2 * 19) * n, c
        else:
            assert c["limb_allreduce"] == 0, c
        if partition:
            assert c["key_min_reduce"] >= 3 * n and c["key_min_reduce_bytes"] == c["key_min_reduce"] * 8 * W * H, c
            assert c["allgather"] >= (2 if partition == "hash" else 1) * n, c
            if exchange == "records":
                assert c["send"] > 0 and c["recv"] > 0 and c["send_bytes"] > 0, c
            else:
                assert c["word_allreduce"] >= 3 * n and c["send"] == 0, c
        else:
            assert c["key_min_reduce"] == 0 and c["send"] == 0, c
        assert c["host_barriers"] == 0


def _world():
    return min(_device_count(), int(os.environ.get("HRBF_REAL_RANKS", "2")))


# (W, H, seed surfels, frames, sparse icp, map partition, row-sharded registration, record exchange, renumber threshold)
ROWS_ONLY = [(160, 120, 0, 6, 0, None, True, None, 0), (320, 240, 150_000, 5, 1, None, True, None, 0)]


@needs_two_devices
@pytest.mark.parametrize("cfg", ROWS_ONLY, ids=["from_empty_map", "uploaded_150k_sparse_icp"])
def test_row_sharded_registration_over_rccl_is_bit_identical(gpu_available, cfg):
    """every rank holds the whole map and frame, reduces its rows to the 29 exact-integer limb sums, ncclAllReduce per GN
    iteration (SURVEY §8e sharding 1): pose, images and map on every rank == the single process, bit for bit"""
    single = _launch(1, cfg)[0]
    many = _launch(_world(), cfg)
    for r, res in many.items():
        assert res["status"] == 0
        for k, v in single.items():
            if k in ("status", "exchange_mode", "comm", "tracked"):
                continue
            assert np.array_equal(res[k], v), "rank %d differs in %s" % (r, k)
    _check_comm(many, _world(), cfg)


RANGES = [(160, 120, 0, 6, 0, "ranges", True, None, 0), (320, 240, 150_000, 5, 1, "ranges", True, None, 0),
          (320, 240, 150_000, 5, 0, "ranges", False, "records", 0)]


@needs_two_devices
@pytest.mark.parametrize("cfg", RANGES, ids=["from_empty_map", "uploaded_150k_sparse_icp", "uploaded_150k_packed_records"])
def test_range_owned_map_over_rccl_is_bit_identical(gpu_available, cfg):
    single = _launch(1, cfg)[0]
    G = _world()
    many = _launch(G, cfg)
    for k, v in single.items():
        if k in ("map", "local_count", "status", "exchange_mode", "comm", "tracked"):
            continue
        if k.startswith("stats"):
            assert np.array_equal(sum(many[r][k].astype(np.int64) for r in range(G)), v), k
            continue
        for r in range(G):
            assert np.array_equal(many[r][k], v), "rank %d differs in %s" % (r, k)
    assert all(many[r]["status"] == 0 for r in range(G))
    if cfg[7] == "records":
        assert all(many[r]["exchange_mode"] == 2 for r in range(G))
    joined = np.concatenate([many[r]["map"].reshape(-1, 20) for r in range(G)])
    assert np.array_equal(joined, single["map"].reshape(-1, 20))
    _check_comm(many, G, cfg)


HASH = [(160, 120, 0, 7, 0, "hash", True, None, 0), (320, 240, 150_000, 5, 1, "hash", True, None, 0),
        (320, 240, 150_000, 5, 0, "hash", True, "records", 0), (160, 120, 0, 8, 0, "hash", True, None, 20000),
        (160, 120, 0, 8, 0, "hash", False, "records", 20000)]


@needs_two_devices
@pytest.mark.parametrize("cfg", HASH, ids=["from_empty_map", "uploaded_150k_sparse_icp", "uploaded_150k_packed_records",
                                           "ids_renumbered_peer_images", "ids_renumbered_packed_records"])
def test_hash_owned_map_over_rccl_is_bit_identical(gpu_available, cfg):
    """north_star's split: map owned by the spatial hash of the cell, u64 key min-reduce per projection, RCCL all-reduce of the
    6x6 system; with peer-mapped images between devices and with the ncclSend/ncclRecv record exchange; ids renumbered under both"""
    single = _launch(1, cfg)[0]
    G = _world()
    many = _launch(G, cfg)
    assert all(many[r]["status"] == 0 for r in range(G))
    for k, v in single.items():
        if k in ("map", "local_count", "status", "exchange_mode", "comm", "tracked"):
            continue
        if k.startswith("stats"):
            assert np.array_equal(sum(many[r][k].astype(np.int64) for r in range(G)), v), k
            continue
        for r in range(G):
            if k.startswith("INDEX") and not k.startswith("INDEX_"):
                assert np.array_equal(many[r][k].view(np.uint32) == 0, v.view(np.uint32) == 0), "rank %d differs in the zero pattern of %s" % (r, k)
                assert np.array_equal(many[0][k], many[r][k]), k
                continue
            assert np.array_equal(many[r][k], v), "rank %d differs in %s" % (r, k)
    n = single["local_count"]
    assert sum(many[r]["local_count"] for r in range(G)) == n
    if cfg[8]:
        assert len(set(many[r]["renumbered"] for r in range(G))) == 1 and many[0]["renumbered"] >= 2
    ids = np.concatenate([many[r]["gids"] for r in range(G)])
    assert len(np.unique(ids)) == n
    joined = np.concatenate([many[r]["map"].reshape(-1, 20) for r in range(G)])[np.argsort(ids, kind="stable")]
    assert np.array_equal(joined, single["map"].reshape(-1, 20))
    _check_comm(many, G, cfg)

"""Ownership by spatial hash (hrbf_map_shard_init(h, 2)), the parts that need no GPU.

1. The two-level z-test is the single map's z-test.  One map: per pixel the surfel with the smallest (depth, index) wins.
   Hash-owned shards: every shard finds its smallest (depth, LOCAL index), the shards' winners compete with (depth, global-order id).
   Local index order is id order inside a shard, so the two agree — also on exact depth ties, which is what makes the sharded
   index map bit-identical and not merely equivalent.  Restated in numpy on random maps with many ties.
2. hrbf_hash_owner: deterministic, spreads the synthetic room's surfels evenly over 2..8 shards, and splits the surfels IN VIEW of a
   frame too (contiguous ranges leave them with one or two shards: the imbalance DESIGN §7 describes).
"""
import numpy as np
import pytest

from hrbffusion3d_amd import api, synth


@pytest.mark.parametrize("G", [2, 3, 8])
def test_two_level_z_test_equals_the_single_map(G):
    rng = np.random.default_rng(G)
    n, P = 20000, 512
    pix = rng.integers(0, P, n)
    depth = rng.integers(1, 40, n).astype(np.uint64)            # few distinct depths: many exact ties per pixel
    gid = np.sort(rng.choice(10 * n, n, replace=False)).astype(np.uint64)   # ids with gaps, ascending = the global order
    owner = rng.integers(0, G, n)
    EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)
    single = np.full(P, EMPTY)
    np.minimum.at(single, pix, (depth << np.uint64(32)) | gid)
    reduced = np.full(P, EMPTY)
    own = np.full(P, -1)
    for g in range(G):
        mine = np.nonzero(owner == g)[0]                         # local index = position in `mine`: ascending in gid
        priv = np.full(P, EMPTY)
        np.minimum.at(priv, pix[mine], (depth[mine] << np.uint64(32)) | np.arange(len(mine), dtype=np.uint64))
        hit = priv != EMPTY
        keys = np.full(P, EMPTY)
        keys[hit] = (priv[hit] & np.uint64(0xFFFFFFFF00000000)) | gid[mine[(priv[hit] & np.uint64(0xFFFFFFFF)).astype(np.int64)]]
        better = keys < reduced
        reduced[better] = keys[better]; own[better] = g
    assert np.array_equal(reduced, single)
    hit = single != EMPTY
    winners = np.searchsorted(gid, single[hit] & np.uint64(0xFFFFFFFF))
    assert np.array_equal(owner[winners], own[hit])              # exactly one shard finds its private winner equal to the reduced key


def test_hash_owner_is_deterministic_and_balanced():
    lib = api.load_library()
    seed = synth.seed_map(200_000, width=640)
    pos = seed[:, :3]

    def owners(G, cell=0.25):
        return np.array([lib.hrbf_hash_owner(float(x), float(y), float(z), cell, G) for x, y, z in pos[::20]])
    a, b = owners(4), owners(4)
    assert np.array_equal(a, b)
    for G in (2, 4, 8):
        share = np.bincount(owners(G), minlength=G) / len(pos[::20])
        assert share.min() > 0.6 / G and share.max() < 1.5 / G, (G, share)
    # the surfels a camera sees: by hash every shard owns a fair part, by contiguous ranges of the array hardly more than one does
    _, _, T = synth.frame(0, 640, 480)
    fx, fy, cx, cy = synth.intrinsics(640, 480)
    cam = (np.linalg.inv(T)[:3, :3] @ pos.T + np.linalg.inv(T)[:3, 3:4]).T
    with np.errstate(divide="ignore", invalid="ignore"):
        u, v = cam[:, 0] / cam[:, 2] * fx + cx, cam[:, 1] / cam[:, 2] * fy + cy
    vis = np.nonzero((cam[:, 2] > 0.3) & (u >= 0) & (u < 640) & (v >= 0) & (v < 480))[0][::5]
    G = 4
    by_hash = np.bincount([lib.hrbf_hash_owner(*map(float, pos[i]), 0.25, G) for i in vis], minlength=G) / len(vis)
    by_range = np.bincount(vis * G // len(pos), minlength=G) / len(vis)
    assert by_hash.max() < 0.45 and by_hash.min() > 0.1, by_hash
    assert by_range.max() > by_hash.max(), (by_range, by_hash)
    assert lib.hrbf_hash_owner(float("nan"), 0.0, 0.0, 0.25, 4) == lib.hrbf_hash_owner(0.0, 0.0, 0.0, 0.25, 4) or True   # NaN is placed, never crashes

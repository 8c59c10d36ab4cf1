"""Measurement (not collected by pytest): how evenly the ray march of k_predict_hrbf loads a wave.  Needs the
-DPREDICT_TRIP_STATS build (PRED_TIME then carries samples | neighbours << 8 | found << 16):
    HRBF_LIB=_build/libhrbf_trips.so python tests/gpu_probe_predict_trips.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.api import HRBFFusion
from hrbffusion3d_amd.params import default_params


def main():
    W, H = 640, 480
    p = default_params(W, H, *synth.intrinsics(W, H), max_surfels=1 << 21)
    g = HRBFFusion(p)
    g.upload_map(synth.seed_map(1_050_000, t_now=1, width=W))
    rgb, d, T = synth.frame(0, W, H)
    g.set_pose(T); g.bootstrap(rgb, d)
    for k in range(1, 6):
        rgb, d, T = synth.frame(k, W, H)
        g.process_frame(rgb, d)
    t = g.get_image("PRED_TIME").reshape(H, W)
    trips = (t & 0xff).astype(np.int64); n = ((t >> 8) & 0xff).astype(np.int64); found = (t >> 16) & 1
    pairs = (n + 1) // 2
    work = trips * pairs
    def wave_stats(rows, cols):
        tw = trips.reshape(H // rows, rows, W // cols, cols).transpose(0, 2, 1, 3).reshape(-1, 64)
        pw = pairs.reshape(H // rows, rows, W // cols, cols).transpose(0, 2, 1, 3).reshape(-1, 64)
        return tw, pw, (tw * pw).sum(), (tw.max(1) * pw.max(1) * 64).sum()
    for rows, cols in ((4, 16), (8, 8), (16, 4), (2, 32), (1, 64)):
        _, _, lw, ww = wave_stats(rows, cols)
        print("wave of %2d rows x %2d columns: SIMD efficiency %.3f" % (rows, cols, lw / ww))
    tw, pw, lane_work, wave_work = wave_stats(4, 16)
    print("pixels %d found %d  mean samples %.1f  mean neighbours %.1f" % (t.size, found.sum(), trips.mean(), n.mean()))
    print("histogram of samples per ray:", np.bincount(trips.ravel(), minlength=47)[:47].tolist())
    print("lane work (samples x pairs) %d, wave work (max samples x max pairs x 64) %d: SIMD efficiency %.2f" % (lane_work, wave_work, lane_work / wave_work))
    print("max-samples per wave: mean %.1f; mean of lane samples %.1f" % (tw.max(1).mean(), tw.mean()))
    # how much a wave would do if its lanes were re-packed after s samples: lanes still active after s
    for s in (2, 12, 22, 26):
        print("rays still marching after %d samples: %.3f" % (s, (trips > s).mean()))
    # ---- round 5: would ranking the 256 rays of a tile by a predicted trip count and giving wave w the w-th quartile pay?
    # wave cost = max samples x max pairs of its 64 rays.  Candidates for the key, all available after the first TWO samples
    # (v0 at `closest`, v1 at the first 4 mm step; state of a ray is then still {sign, flipped or not}):
    #   perfect  = the true remaining work (upper bound of any predictor)
    #   secant   = phase-2 steps predicted from where the chord v0 -> v1 crosses zero, + 6 bisections; rays that have not
    #              flipped after one coarse step: as many coarse steps as a chord through (v0, v1) suggests, <= 23
    #   absv0    = |v0| alone (the key the verdict proposed; available one sample earlier)
    k1 = ((t >> 17) & 31).astype(np.int64); k2 = ((t >> 22) & 15).astype(np.int64); k3 = ((t >> 26) & 15).astype(np.int64)
    c1 = g.get_image("PRED_CURV1").reshape(H, W, 4).astype(np.float64)
    v0, v1 = c1[..., 0], c1[..., 1]
    flipped = (k1 == 1) & (k2 > 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        frac = np.where(np.abs(v0) + np.abs(v1) > 0, np.abs(v1) / (np.abs(v0) + np.abs(v1)), 0.5)     # zero crossing, measured back from the coarse sample
        more = np.where(np.abs(v0) > np.abs(v1), np.abs(v1) / np.maximum(np.abs(v0) - np.abs(v1), 1e-30), 23.0)   # further coarse steps to the crossing
    pred = np.where(trips <= 1, 0, np.where(flipped, np.ceil(frac * 10) + 6, np.minimum(np.ceil(more), 23) + 5 + 6))
    rest = np.maximum(trips - 2, 0)
    print("rank correlation of the remaining samples with: secant %.3f, |v0| %.3f" % (
        np.corrcoef(np.argsort(np.argsort(pred.ravel())), np.argsort(np.argsort(rest.ravel())))[0, 1],
        np.corrcoef(np.argsort(np.argsort(np.abs(v0).ravel())), np.argsort(np.argsort(trips.ravel())))[0, 1]))
    def tiles(a):
        return a.reshape(H // 16, 16, W // 16, 16).transpose(0, 2, 1, 3).reshape(-1, 256)
    T, Pp, R = tiles(trips), tiles(pairs), tiles(rest)
    def cost(order, head):   # head: samples every ray takes in the original 8x8 mapping before the re-ranking
        tt = np.take_along_axis(R if head else T, order, 1).reshape(-1, 4, 64); pp = np.take_along_axis(Pp, order, 1).reshape(-1, 4, 64)
        c = (tt.max(2) * pp.max(2) * 64).sum()
        if head:
            t8, p8, _, _ = wave_stats(8, 8)
            c += (np.minimum(t8, 2).max(1) * p8.max(1) * 64).sum()
        return c
    ident = np.tile(np.arange(256), (T.shape[0], 1))
    # identity order of a tile = the kernel's 8x8 blocks
    t8, p8, lw8, ww8 = wave_stats(8, 8)
    print("wave work now (8x8 blocks): %d = 1.000; lane work %d = %.3f" % (ww8, lw8, lw8 / ww8))
    for name, key, head in (("perfect, ranked after 2 samples", tiles(rest * pairs), True), ("secant, ranked after 2 samples", tiles(pred * pairs), True),
                            ("secant x 1 (pairs ignored)", tiles(pred), True), ("|v0| x pairs, ranked after 1 sample", None, False),
                            ("perfect, ranked after 0 samples", tiles(work), False)):
        if key is None:
            key = tiles(np.abs(v0) * pairs)
        order = np.argsort(key, axis=1, kind="stable")
        print("%-40s wave work %.3f of now" % (name, cost(order, head) / ww8))
    g.close()


if __name__ == "__main__":
    main()

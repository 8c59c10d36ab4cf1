"""Measurement (not collected by pytest): how evenly the ray march of k_predict_hrbf loads a wave.  Needs the
-DPREDICT_TRIP_STATS build (PRED_TIME then carries samples | neighbours << 8 | found << 16):
    HRBF_LIB=libhrbf_trips.so python tests/gpu_probe_predict_trips.py"""
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
    g.close()


if __name__ == "__main__":
    main()

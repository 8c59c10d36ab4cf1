"""Adversarial pixels at the stage seams, HIP path against the oracle, bit for bit.

The pipeline makes NaNs of its own (a flat patch has a 0 / 0 curvature direction, a merge at total confidence 0 a 0 / 0 position), and
what a stage does with a NaN, an infinity, a negative or a denormal operand is decided by comparisons and conversions that C leaves
open more often than arithmetic does (x < y ? x : y against a min instruction, (int) of a NaN).  The streams of the parity tests only
meet the NaNs the synthetic scene happens to produce; here every float image of a context that has processed two frames gets random
pixels replaced by {NaN, +-inf, +-0, -1, +-1e30, 1e-40, 3e9, 0.5} in the images ONE stage reads (STAGE_IO; surfel ids and time words stay
what they are: an id beyond the map is the caller's error), both sides get the same images, the stage runs on both, and the images it
writes, the map and the pose are compared.

    python tests/gpu_fuzz_stages.py N [seed] [out]      # N trials; appends to gpurun_out/stage_fuzz.txt (or `out`)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from gpu_fuzz_params import bits  # noqa: E402

VALS = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, -1.0, 1e30, -1e30, 1e-40, 3e9, 0.5], np.float32)
W, H = 160, 120


def corrupt(rng, a, frac):
    a = a.copy()
    flat = a.reshape(-1)
    n = max(1, int(frac * flat.size))
    at = rng.integers(flat.size, size=n)
    if a.dtype == np.float32:
        flat[at] = VALS[rng.integers(len(VALS), size=n)]
    else:
        flat[at] = rng.integers(256, size=n).astype(a.dtype)
    return a


# stage -> (float / byte images it reads that are corrupted, images it writes that are compared).  The map, the surfel count and the pose
# are compared after every stage.  FILTER_DEPTH + METRICISE are one seam (the HIP path computes the metric images inside the filter
# kernel): only the raw depth is corrupted there.
FRAME = ["VERTEX_RAW", "VERTEX_FILTERED", "NORMAL", "NORMAL_PCA", "RADIUS", "CURV1", "CURV2", "GRADIENT_MAG", "CONFIDENCE", "DEPTH_METRIC",
         "DEPTH_METRIC_FILTERED"]
INDEX_ATTR = ["INDEX_VERTCONF", "INDEX_COLORTIME", "INDEX_NORMRAD", "INDEX_CURVMAX", "INDEX_CURVMIN"]
PRED = ["PRED_IMAGE", "PRED_VERTEX", "PRED_NORMAL", "PRED_CURV1", "PRED_CURV2", "PRED_ICPWEIGHT"]
FILL = ["FILL_IMAGE", "FILL_VERTEX", "FILL_NORMAL", "FILL_CURV1", "FILL_CURV2", "FILL_ICPWEIGHT"]
STAGE_IO = {
    "FILTER_DEPTH": ([], ["DEPTH_FILTERED", "DEPTH_METRIC", "DEPTH_METRIC_FILTERED"]),
    "VERTEX_NORMAL_RADIUS": (["DEPTH_METRIC", "DEPTH_METRIC_FILTERED"], ["VERTEX_RAW", "VERTEX_FILTERED", "NORMAL", "NORMAL_PCA", "RADIUS"]),
    "CURVATURE": (["NORMAL", "VERTEX_FILTERED", "RADIUS"], ["CURV1", "CURV2", "GRADIENT_MAG", "NORMAL"]),
    "CONFIDENCE": (["CURV1", "CURV2", "GRADIENT_MAG", "NORMAL", "DEPTH_METRIC"], ["CONFIDENCE"]),
    "PREDICT_INDICES": ([], ["INDEX"] + INDEX_ATTR),
    "FUSE": (FRAME + INDEX_ATTR, []),
    "CLEAN": (FRAME + INDEX_ATTR, []),
    "PREDICT_HRBF": (INDEX_ATTR, PRED + ["PRED_TIME"]),
    "FILLIN": (PRED + FRAME, FILL),
    "ODOMETRY": (FILL + FRAME, []),
}


def first_difference(name, a, b):
    ba, bb = bits(a), bits(b)
    if np.array_equal(ba, bb):
        return None
    if a.dtype != np.float32:
        ba, bb = a, b
    w = tuple(np.argwhere(ba != bb)[0])
    return "%s differs in %d values, first at %s: oracle %r, kernel %r" % (name, int((ba != bb).sum()), list(map(int, w)), a[w], b[w])


def trial(oracle_lib, seed, index):
    """None, or a description of the first difference"""
    from hrbffusion3d_amd import synth
    from hrbffusion3d_amd.api import HRBFFusion
    from hrbffusion3d_amd.params import default_params
    rng = np.random.default_rng([seed, index])
    c = lambda *v: v[int(rng.integers(len(v)))]
    kw = dict(use_conf_eval=int(rng.random() < 0.3), icp_use_corr_search=int(rng.random() < 0.3), use_sparse_icp=int(rng.random() < 0.3),
              frame_to_frame_rgb=int(rng.random() < 0.3), normal_estimation_pca=float(rng.random() < 0.7), use_bilateral=int(rng.random() < 0.7))
    p = default_params(W, H, *synth.intrinsics(W, H), max_surfels=1 << 17, **kw)
    stage = c(*STAGE_IO)
    reads, writes = STAGE_IO[stage]
    start = int(rng.integers(0, 300))
    o, g = oracle_lib.Oracle(p, omp=True), HRBFFusion(p)
    both = (o, g)
    try:
        for k in range(2):
            rgb, d, _ = synth.frame(start + k, W, H, noise=True)
            o.process_frame(rgb, d); g.process_frame(rgb, d)
        rgb, d, _ = synth.frame(start + 2, W, H, noise=True)
        if stage == "FILTER_DEPTH" or rng.random() < 0.2:          # raw depth: any 16-bit word
            d = corrupt(rng, d.view(np.uint8), float(c(0.002, 0.02, 0.2))).view(np.uint16)
        for x in both:
            x.upload_frame(rgb, d)
            for s in ("FILTER_DEPTH", "METRICISE", "VERTEX_NORMAL_RADIUS", "CURVATURE", "CONFIDENCE"):      # the new frame's images
                if s == stage and s != "FILTER_DEPTH":
                    break
                x.run_stage(s)
                if s == "METRICISE" and stage == "FILTER_DEPTH":
                    break
        touched = []
        if stage != "FILTER_DEPTH":
            frac = float(c(0.002, 0.02, 0.2))
            touched = [n for n in reads if rng.random() < 0.6]
            for n in touched:
                a = corrupt(rng, o.get_image(n), frac)
                o.set_image(n, a); g.set_image(n, a)
            if stage in ("PREDICT_INDICES", "FUSE", "CLEAN", "PREDICT_HRBF") and rng.random() < 0.5:
                m = o.download_map()
                if m.shape[0]:
                    m = corrupt(rng, m, 0.001)
                    ok = np.isfinite(m[:, 5]) & (m[:, 5] >= 0) & (m[:, 5] < 16)
                    m[:, 5] = np.where(ok, np.floor(np.where(ok, m[:, 5], 0)), 0)     # the submap id stays an id
                    for x in both:
                        x.upload_map(m)
                    touched.append("map")
            for x in both:
                if stage == "CLEAN":
                    x.run_stage("FUSE"); x.run_stage("PREDICT_INDICES")
                x.run_stage(stage)
        for n in writes:
            r = first_difference("image " + n, o.get_image(n), g.get_image(n))
            if r:
                return "stage %s: %s (inputs touched: %s)" % (stage, r, ",".join(touched))
        if o.surfel_count() != g.surfel_count():
            return "stage %s: surfel count %d vs %d (inputs touched: %s)" % (stage, o.surfel_count(), g.surfel_count(), ",".join(touched))
        r = first_difference("map", o.download_map(), g.download_map()) or first_difference("pose", o.get_pose(), g.get_pose())
        if r:
            return "stage %s: %s (inputs touched: %s)" % (stage, r, ",".join(touched))
        return None
    finally:
        o.close(); g.close()


def main():
    import oracle_lib
    n = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "stage_fuzz.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    oracle_lib.build()
    bad, t0, kinds = 0, time.time(), {}
    with open(out, "a") as log:
        log.write("# seed %d, %d trials: random pixels of random images replaced by NaN / inf / ... , one stage run on both sides\n" % (seed, n))
        for i in range(n):
            if os.environ.get("HRBF_FUZZ_TRACE"):
                log.write("trial %d\n" % i); log.flush()
            r = trial(oracle_lib, seed, i)
            if r is not None:
                bad += 1
                key = r.split(":")[0] + ":" + r.split(":")[1].split(" differs")[0]
                kinds[key] = kinds.get(key, 0) + 1
                log.write("MISMATCH trial %d of seed %d: %s\n" % (i, seed, r)); log.flush()
        log.write("done: %d trials, %d mismatches, %.0f s; by stage / output: %s\n" % (n, bad, time.time() - t0, kinds))
    print("stage fuzz: %d trials, %d mismatches" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

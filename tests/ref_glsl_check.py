"""Shared by tests/test_ref_glsl.py (CPU: the C oracle; GPU: the HIP library): runs one implementation through the passes
recorded in tests/golden/ref_glsl/<scene>.npz — outputs of the reference's own GLSL shaders executed on llvmpipe
(tests/golden/make_ref_glsl.py) — on the inputs the shaders had bound, and compares.

`impl` is an object with the stage API both implementations share (tests/oracle_lib.Oracle, hrbffusion3d_amd.api.HRBFFusion):
upload_frame, set_image / get_image, run_stage, upload_map / download_map, set_pose, set_tick, set_weighting, update_model.

Tolerances are in units in the last place of the fp32 result ("ulp"), per output, and stated where they are used.  Why
they are not zero: the shaders' exp / sqrt / inversesqrt / acos / division are llvmpipe's (polynomial exp2, rsqrt-based
normalize, a * rcp(b)), the oracle's and the kernels' are hrbf_detmath.h's; everything else is IEEE and agrees bit for bit.
Discrete results — which pixels are valid, which surfel wins a pixel, which surfels merge / are created / are removed, the
order of the map — are compared exactly, except at the few stated tie pixels.
"""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_glsl")


def load(scene):
    """a scene's fixture arrays: `<scene>.fxz` (tests/fixture_codec.py: the same bits as the .npz make_ref_glsl.py wrote, CRC-checked,
    at half the size) or the plain .npz where that is what the tree holds"""
    import fixture_codec
    return fixture_codec.load(scene)


def ulp_diff(a, b):
    """distance in representable fp32 values between a and b (elementwise); NaN vs NaN = 0, NaN vs number = 2^31"""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai); bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    d = np.abs(ai - bi)
    na, nb = np.isnan(a), np.isnan(b)
    d[na & nb] = 0
    d[na ^ nb] = 1 << 31
    return d


class Report:
    """collects (pass, output, statistic) rows; `strict` raises on the first violated bound"""

    def __init__(self, strict=True, verbose=False):
        self.rows, self.strict, self.verbose = [], strict, verbose

    def add(self, what, ok, detail):
        self.rows.append((what, bool(ok), detail))
        if self.verbose:
            print("%-44s %s  %s" % (what, "ok  " if ok else "FAIL", detail))
        if self.strict and not ok:
            raise AssertionError("%s: %s" % (what, detail))

    def close_ulp(self, what, got, ref, max_ulp, frac_within=1.0, abs_floor=0.0, mask=None):
        """every element within max_ulp of the reference (or |difference| <= abs_floor, for results of cancelling sums
        whose magnitude is far below their terms'); frac_within < 1 allows that share of outliers (stated by the caller)"""
        got = np.asarray(got, np.float32); ref = np.asarray(ref, np.float32)
        d = ulp_diff(got, ref)
        with np.errstate(invalid="ignore"):
            small = np.abs(got.astype(np.float64) - ref.astype(np.float64)) <= abs_floor
        good = (d <= max_ulp) | small
        if mask is not None:
            good = good | ~mask
        share = float(good.mean()) if good.size else 1.0
        self.add(what, share >= frac_within, "within %d ulp: %.5f%% (need %.3f%%), worst %d ulp" % (
            max_ulp, 100 * share, 100 * frac_within, int(d[mask].max() if mask is not None and mask.any() else d.max() if d.size else 0)))

    def exact(self, what, got, ref):
        got = np.ascontiguousarray(got); ref = np.ascontiguousarray(ref)
        if got.dtype == np.float32:
            n = int((ulp_diff(got, ref) != 0).sum())
        else:
            n = int((got != ref).sum())
        self.add(what, n == 0, "%d of %d differ" % (n, got.size))


def part_filter(rep, g, fx, P="f2_"):
    """P1 filterDepth (depth_bilateral.frag / depth_guass.frag): up to 169 exp per pixel, normalised sum; P2 metriciseDepth"""
    g.upload_frame(fx[P + "rgb"], fx[P + "depth"])
    g.run_stage("FILTER_DEPTH")
    ref = fx[P + "DEPTH_FILTERED"]
    got = g.get_image("DEPTH_FILTERED")
    rep.exact("P1 which pixels are filtered", got == 0, ref == 0)
    rep.close_ulp("P1 DEPTH_FILTERED", got, ref, 16)
    g.set_image("DEPTH_FILTERED", ref)
    g.run_stage("METRICISE")
    rep.exact("P2 DEPTH_METRIC", g.get_image("DEPTH_METRIC"), fx[P + "DEPTH_METRIC"])
    # the HIP path computes the metric image inside the filter kernel (P1 + P2 fused), so there it carries P1's own rounding;
    # the oracle runs P2 on the DEPTH_FILTERED just set and is exact
    rep.close_ulp("P2 DEPTH_METRIC_FILTERED", g.get_image("DEPTH_METRIC_FILTERED"), fx[P + "DEPTH_METRIC_FILTERED"], 16)


def part_vertex_normal_radius(rep, g, fx, P="f2_", pca=True):
    """P3 computeVertexNormalRadius on the reference's metric images"""
    g.set_image("DEPTH_METRIC", fx[P + "DEPTH_METRIC"])
    g.set_image("DEPTH_METRIC_FILTERED", fx[P + "DEPTH_METRIC_FILTERED"])
    g.run_stage("VERTEX_NORMAL_RADIUS")
    vr, vf = g.get_image("VERTEX_RAW"), g.get_image("VERTEX_FILTERED")
    rep.exact("P3 VERTEX_RAW xyz", vr[..., :3], fx[P + "VERTEX_RAW"][..., :3])
    rep.close_ulp("P3 VERTEX_RAW w (radial confidence, exp)", vr[..., 3], fx[P + "VERTEX_RAW"][..., 3], 16)
    rep.exact("P3 VERTEX_FILTERED", vf, fx[P + "VERTEX_FILTERED"])
    n3 = g.get_image("NORMAL")
    rep.exact("P3 which pixels have a normal", (n3[..., :3] == 0).all(-1), (fx[P + "NORMAL_P3"][..., :3] == 0).all(-1))
    # PCA normal: eigen-solve of a covariance formed by cancellation (geometry.glsl:176-193), atan2/cos/sin inside;
    # central differences (geometry.glsl:152-174): a cross product of differences, normalised
    # (vertex differences of ~5 mm, their cross product cancels to ~1e-5 of products of ~2.5e-5: the components of the normal
    # carry ~1e-5 absolute of rounding, and llvmpipe contracts the cross product's a*b - c*d where the oracle may not)
    rep.close_ulp("P3 NORMAL (%s) xyz" % ("PCA" if pca else "central differences"), n3[..., :3], fx[P + "NORMAL_P3"][..., :3], 64,
                  abs_floor=2e-6 if pca else 2e-5)
    # (the radius divides by |n_z|: it inherits the central-difference normal's cancellation noise — 1e-4 relative of ~1 cm)
    rep.close_ulp("P3 NORMAL w = RADIUS", n3[..., 3], fx[P + "NORMAL_P3"][..., 3], 64, abs_floor=0.0 if pca else 4e-6)
    rep.close_ulp("P3 RADIUS", g.get_image("RADIUS"), fx[P + "RADIUS"], 64, abs_floor=0.0 if pca else 4e-6)


def part_curvature(rep, g, fx, P="f2_"):
    """P4 + P5 computeCurvatureGradient, updateNormalRad on the reference's vertex / normal images"""
    g.set_image("NORMAL", fx[P + "NORMAL_P3"])
    g.set_image("VERTEX_FILTERED", fx[P + "VERTEX_FILTERED"])
    g.run_stage("CURVATURE")
    gm, gmr = g.get_image("GRADIENT_MAG"), fx[P + "GRADIENT_MAG"]
    rep.exact("P4 which pixels have > 15 neighbours", gm == 0, gmr == 0)
    # sums of <= 49 Hessian terms with mixed signs: a few hundred ulp of the result where terms cancel
    rep.close_ulp("P4 GRADIENT_MAG", gm, gmr, 1024, abs_floor=1e-3)
    no, nor = g.get_image("NORMAL"), fx[P + "NORMAL"]
    rep.close_ulp("P5 NORMAL (HRBF gradient direction)", no[..., :3], nor[..., :3], 64, abs_floor=4e-6)
    rep.exact("P5 NORMAL w (radius carried over)", no[..., 3], nor[..., 3])
    for name in ("CURV1", "CURV2"):
        c, cr = g.get_image(name), fx[P + name]
        rep.exact("P4 %s which pixels are the 1000-sentinel" % name, c[..., 3] == 1000.0, cr[..., 3] == 1000.0)
        curvature_checks(rep, name, c, cr)


def part_confidence(rep, g, fx, P="f2_"):
    """P5 VertexConfidence on the reference's curvature images"""
    for name in ("CURV1", "CURV2", "GRADIENT_MAG", "NORMAL"):
        g.set_image(name, fx[P + name])
    g.set_image("DEPTH_METRIC", fx[P + "DEPTH_METRIC"])
    g.set_weighting(float(fx[P + "weighting"]))
    g.run_stage("CONFIDENCE")
    rep.close_ulp("P5 CONFIDENCE", g.get_image("CONFIDENCE"), fx[P + "CONFIDENCE"], 16)


def bind_frame(g, fx, P="f2_"):
    """frame 2's images as the reference's shaders left them: what the map passes read"""
    g.upload_frame(fx[P + "rgb"], fx[P + "depth"])
    for name, key in (("DEPTH_FILTERED", "DEPTH_FILTERED"), ("DEPTH_METRIC", "DEPTH_METRIC"), ("DEPTH_METRIC_FILTERED", "DEPTH_METRIC_FILTERED"),
                      ("VERTEX_RAW", "VERTEX_RAW"), ("VERTEX_FILTERED", "VERTEX_FILTERED"), ("RADIUS", "RADIUS"), ("CURV1", "CURV1"),
                      ("CURV2", "CURV2"), ("GRADIENT_MAG", "GRADIENT_MAG"), ("NORMAL", "NORMAL"), ("CONFIDENCE", "CONFIDENCE"),
                      ("NORMAL_PCA", "NORMAL_P3")):
        g.set_image(name, fx[P + key])
    g.set_weighting(float(fx[P + "weighting"]))


def stable_map_with_outliers(fx):
    m = fx["f1_map"].copy(); m[:, 3] += 6.0
    old = fx["x_old"]; m[old, 3] = 1.0; m[old, 7] = -250.0
    return np.concatenate([m, fx["x_extra"]])


def reference_final(fx, pre, map_in):
    """the map the reference's clean pass left: survivors of its fused map in order, then its new surfels"""
    ref_fused = map_in.copy(); ref_fused[fx[pre + "fused_rows"]] = fx[pre + "fused_vals"]
    keep = np.unpackbits(fx[pre + "keep"])[:map_in.shape[0]].astype(bool)
    rec = fx[pre + "records"]
    new = rec[fx[pre + "new_picks"]].copy(); new[:, 7] = 2.0
    out = np.concatenate([ref_fused[keep], new])
    assert out.shape[0] == int(fx[pre + "map_count"][0])
    return out


def part_prediction(rep, g, fx, ref_final, P="f2_"):
    """M1 + H2 predictHRBF, H3 fill-in on the reference's map"""
    T2 = fx[P + "pose"]
    g.upload_map(ref_final)
    g.set_pose(T2); g.set_tick(2)
    g.run_stage("PREDICT_INDICES")
    index_checks(rep, "M1 (prediction)", g, fx, "x_p_", full=True)
    for k in ("INDEX", "INDEX_VERTCONF", "INDEX_COLORTIME", "INDEX_NORMRAD", "INDEX_CURVMAX", "INDEX_CURVMIN"):
        g.set_image(k, fx["x_p_" + k])
    g.run_stage("PREDICT_HRBF")
    prediction_checks(rep, g, fx, "x_")
    for k in ("PRED_IMAGE", "PRED_VERTEX", "PRED_NORMAL", "PRED_CURV1", "PRED_CURV2", "PRED_ICPWEIGHT"):
        g.set_image(k, fx["x_" + k])
    g.set_image("VERTEX_FILTERED", fx[P + "VERTEX_FILTERED"])
    g.run_stage("FILLIN")
    fill_checks(rep, g, fx, "x_")


def run(impl, fx, rep, has_records=False, own_pca_normals=False):
    g, P = impl, "f2_"
    T2, w2 = fx[P + "pose"], float(fx[P + "weighting"])
    part_filter(rep, g, fx, P)
    part_vertex_normal_radius(rep, g, fx, P)
    part_curvature(rep, g, fx, P)
    part_confidence(rep, g, fx, P)
    g.set_image("CONFIDENCE", fx[P + "CONFIDENCE"])
    # data.vert recomputes the PCA normal of a new point from the filtered depth (data.vert:83-95) with the texture coordinate
    # of a VERTEX ATTRIBUTE; the normal image holds the fragment shader's, computed with an INTERPOLATED coordinate.  At
    # power-of-two sizes the two are the same number and the image can stand in; at other sizes the implementation's own P3
    # normals (own_pca_normals: correctly rounded coordinates, like the attribute) are the faithful input of the map passes
    if not own_pca_normals:
        g.set_image("NORMAL_PCA", fx[P + "NORMAL_P3"])
    else:
        g.set_image("DEPTH_METRIC", fx[P + "DEPTH_METRIC"]); g.set_image("DEPTH_METRIC_FILTERED", fx[P + "DEPTH_METRIC_FILTERED"])
        g.run_stage("VERTEX_NORMAL_RADIUS")
        for name in ("VERTEX_RAW", "VERTEX_FILTERED", "RADIUS", "NORMAL", "CURV1", "CURV2", "GRADIENT_MAG", "CONFIDENCE"):
            g.set_image(name, fx[P + name])

    # ---- F4 initialise (init_unstableTex.*) from frame 2's images at pose T2 -----------------------------------------------
    g.set_image("VERTEX_RAW", fx[P + "VERTEX_RAW"])
    g.set_pose(T2)
    g.run_stage("INITIALISE")
    im = g.download_map()
    rep.add("F4 surfel count", im.shape[0] == int(fx[P + "init_count"][0]), "%d vs reference %d" % (im.shape[0], int(fx[P + "init_count"][0])))
    ih = fx[P + "init_head"]
    rep.close_ulp("F4 position", im[:4096, 0:3], ih[:, 0:3], 4, abs_floor=1e-7)
    rep.close_ulp("F4 confidence (exp)", im[:4096, 3], ih[:, 3], 16)
    rep.exact("F4 colour / submap / init time / time", im[:4096, 4:8], ih[:, 4:8])
    rep.close_ulp("F4 normal + radius", im[:4096, 8:12], ih[:, 8:12], 4, abs_floor=1e-7)
    rep.exact("F4 curvature records", im[:4096, 12:20], ih[:, 12:20])

    # ---- (1) the plain second frame on the young map of frame 1 --------------------------------------------------------------
    map_flow(rep, g, fx, "f2_", "young map: ", fx["f1_map"], T2)
    # ---- (2) a stable map + surfels that must be removed -------------------------------------------------------------------------
    ref_final = map_flow(rep, g, fx, "x_", "stable map + outliers: ", stable_map_with_outliers(fx), T2)
    part_prediction(rep, g, fx, ref_final, P)
    # ---- f-3 updateModel ----------------------------------------------------------------------------------------------------
    g.upload_map(ref_final)
    g.update_model([fx["x_delta"]])
    um, umr = g.download_map()[:4096], fx["x_map_updated_head"]
    rep.close_ulp("f-3 updateModel positions", um[:, 0:3], umr[:, 0:3], 4)
    rep.close_ulp("f-3 updateModel normals", um[:, 8:11], umr[:, 8:11], 4, abs_floor=1e-7)
    rep.exact("f-3 updateModel everything else", np.delete(um, [0, 1, 2, 8, 9, 10], 1), np.delete(umr, [0, 1, 2, 8, 9, 10], 1))
    return rep


# ---- the reference's parameter variants of the GLSL rows, executed (tests/golden/ref_glsl/sphere_variants.npz) -------------
# name -> (parameter overrides, the part of the pipeline they reach).  Every variant runs ONE part on the inputs the default
# sphere fixture holds for it; only that part's outputs are stored, as "<name>__<key of the default fixture>".
VARIANTS = {
    "gauss_filter": (dict(use_bilateral=0), "filter"),                               # depth_guass.frag instead of depth_bilateral.frag
    "depth_cutoff": (dict(depth_cutoff=1.5), "filter"),                               # maxD cuts the plane behind the sphere (P1, P2)
    "central_diff_normals": (dict(normal_estimation_pca=0.0), "vnr"),                # geometry.glsl getNormal instead of getNormalPCA
    "radius_multiplier_3": (dict(init_radius_multiplier=3.0), "vnr"),
    "curv_window_2": (dict(curv_estimation_window=2.0), "curv"),                     # 5 x 5 HRBF window
    "conf_eval": (dict(use_conf_eval=1), "conf"),                                    # exp(-epsilon / sqrt(gradient_mag)) factor
    "clean_window_1": (dict(clean_window_multiplier=1.0), "clean"),
    "clean_window_2_25": (dict(clean_window_multiplier=2.25), "clean"),              # ceil(4.5) = 5 samples per axis
    "clean_window_4": (dict(clean_window_multiplier=4.0), "clean"),
    "clean_thresholds": (dict(confidence_threshold=9.0, curv_valid_threshold=40.0), "clean"),
    "fuse_central_diff": (dict(normal_estimation_pca=0.0), "fuse"),                # data.vert:91-94: central differences on HALF-PIXEL coordinates
    "fuse_radius_multiplier_3": (dict(init_radius_multiplier=3.0), "fuse"),        # data.vert:96: the record's radius (merge-or-keep branch of update.vert)
    "predict_small": (dict(predict_window_multiplier=2.0, predict_min_neighbors=4, predict_max_neighbors=6), "predict"),
    "fill_frame_to_frame_rgb": (dict(frame_to_frame_rgb=1), "predict"),             # fill_rgb.frag with passthrough: the live image everywhere
    "predict_conf_6_6": (dict(predict_conf_threshold=6.6), "predict"),            # about half of the stable map qualifies
}


def variant_fixture(base, var, name):
    """the default fixture with the variant's outputs in place of the default ones"""
    fx = dict(base)
    pre = name + "__"
    for k, v in var.items():
        if k.startswith(pre):
            fx[k[len(pre):]] = v
    return fx


def run_variant(impl, base, var, name, rep):
    kw, part = VARIANTS[name]
    fx, g = variant_fixture(base, var, name), impl
    if part == "filter":
        part_filter(rep, g, fx)
    elif part == "vnr":
        part_vertex_normal_radius(rep, g, fx, pca=kw.get("normal_estimation_pca", 1.0) != 0.0)
    elif part == "curv":
        part_curvature(rep, g, fx)
    elif part == "conf":
        part_confidence(rep, g, fx)
    elif part in ("clean", "fuse"):
        if kw.get("normal_estimation_pca", 1.0) == 0.0:
            fx["_normal_abs_floor"] = 2e-5
        bind_frame(g, base)
        map_flow(rep, g, fx, "x_", name + ": ", stable_map_with_outliers(base), base["f2_pose"])
    elif part == "predict":
        bind_frame(g, base)
        part_prediction(rep, g, fx, reference_final(base, "x_", stable_map_with_outliers(base)))
    return rep


def map_flow(rep, g, fx, pre, tag, map_in, T2):
    """predictIndices -> fuse -> predictIndices -> clean on `map_in` with frame 2's images set; returns the reference's final map"""
    g.upload_map(map_in)
    g.set_pose(T2); g.set_tick(2)
    g.run_stage("PREDICT_INDICES")
    index_checks(rep, tag + "M1", g, fx, pre + "a_", full=False)
    for k in ("INDEX", "INDEX_VERTCONF", "INDEX_NORMRAD"):
        g.set_image(k, fx[pre + "a_" + k])
    g.run_stage("FUSE")
    rec = fx[pre + "records"]
    st = g.fuse_stats()
    rep.add(tag + "F1 records", True, "stats %s; reference: %d merge marks, %d new" % (
        st.tolist(), int((rec[:, 7] == -1).sum()), int((rec[:, 7] == -2).sum())))
    fused = g.download_map()
    rows, vals = fx[pre + "fused_rows"], fx[pre + "fused_vals"]
    ch = np.nonzero((ulp_diff(fused, map_in) != 0).any(1))[0]
    # Association ties: a candidate whose normal is parallel to the new point's to within ~2e-4 rad has cos = 1 to within an ulp;
    # acos of 1 + ulp is NaN and rejects it (data.vert:150-152 through utils.glsl angleBetween) — which of 1 and 1 + ulp comes out
    # depends on the last bit of the division, so two executions can associate such a pixel with different neighbours (or none).
    # (at sizes that are no power of two the half-pixel walk's samples on exact texel boundaries are implementation-defined as
    # well — llvmpipe's fp32 floor lands some of them one texel low, DESIGN.md §8 — : the caller widens the allowance there)
    ties = int(fx.get("_tie_factor", 1.0) * max(2, rows.size // 400))
    odd = np.setxor1d(ch, rows)
    rep.add(tag + "F1/F2 which surfels were merged into (%d)" % rows.size, odd.size <= 2 * ties,
            "%d surfels merged in one execution only (acos-domain ties; allowed %d)" % (odd.size, 2 * ties))
    common = np.isin(rows, ch)
    gv, rv = fused[rows[common]], vals[common]
    same_rec = (ulp_diff(gv[:, 4:8], rv[:, 4:8]) == 0).all(1) & (ulp_diff(gv[:, 3], rv[:, 3]) <= 8)   # merged with the same record
    rep.add(tag + "F2 surfels merged with another record", int((~same_rec).sum()) <= 2 * ties, "%d" % int((~same_rec).sum()))
    gv, rv = gv[same_rec], rv[same_rec]
    rep.close_ulp(tag + "F2 merged position + confidence", gv[:, 0:4], rv[:, 0:4], 8)
    # (central differences, data.vert:91-94: the cross product of two ~5 mm differences cancels — see part_vertex_normal_radius)
    rep.close_ulp(tag + "F2 merged normal + radius", gv[:, 8:12], rv[:, 8:12], 16, abs_floor=float(fx.get("_normal_abs_floor", 1e-6)))
    rep.close_ulp(tag + "F2 merged curvature records", gv[:, 12:20], rv[:, 12:20], 16, abs_floor=1e-5)
    g.run_stage("PREDICT_INDICES")
    ic = g.get_image("INDEX")
    bad = int((ic != fx[pre + "c_INDEX"]).sum())
    rep.add(tag + "M1 (after fuse) INDEX", bad <= max(2, ic.size // 5000) + 2 * odd.size, "%d of %d pixels differ (sub-pixel snap ties)" % (bad, ic.size))
    g.run_stage("CLEAN")
    keep = np.unpackbits(fx[pre + "keep"])[:map_in.shape[0]].astype(bool)
    new = rec[fx[pre + "new_picks"]]
    ref_final = reference_final(fx, pre, map_in)
    final = g.download_map()
    # align the two maps row by row: key = position to 10 um + init time; rows present in both must come in the same order
    def keys(m):
        q = np.rint(m[:, 0:3].astype(np.float64) * 1e5).astype(np.int64)
        return [(int(a), int(b), int(c), float(t)) for (a, b, c), t in zip(q, m[:, 6])]
    kr = {k: i for i, k in enumerate(keys(ref_final))}
    pos = np.array([kr.get(k, -1) for k in keys(final)])
    found = pos >= 0
    unmatched = int((~found).sum()) + (ref_final.shape[0] - int(found.sum()))
    # a tie can turn a merge into a new surfel (+1 row) and moves a merge from one surfel to a neighbour (2 rows whose position differs)
    rep.add(tag + "F3 map after clean: rows", unmatched <= 4 * ties + 2 * odd.size, "%d rows vs reference %d (removed %d, appended %d); %d rows without partner" % (
        final.shape[0], ref_final.shape[0], int((~keep).sum()), new.shape[0], unmatched))
    rep.add(tag + "F3 map order", bool((np.diff(pos[found]) > 0).all()), "common rows in the same order")
    a, b = final[found], ref_final[pos[found]]
    rep.exact(tag + "F3 colour / submap / init time / time of every common row", a[:, 4:8], b[:, 4:8])
    rep.close_ulp(tag + "F3 map positions + confidence", a[:, 0:4], b[:, 0:4], 16, frac_within=1.0 - (4.0 * ties + 1) / max(1, a.shape[0]))
    rep.close_ulp(tag + "F3 map normals + radii", a[:, 8:12], b[:, 8:12], 64, abs_floor=max(2e-6, float(fx.get("_normal_abs_floor", 0.0))),
                  frac_within=1.0 - (4.0 * ties + 1) / max(1, a.shape[0]))
    rep.close_ulp(tag + "F3 map curvature records", a[:, 12:20], b[:, 12:20], 64, abs_floor=1e-5, frac_within=1.0 - (8.0 * ties + 1) / max(1, a.shape[0]))
    removed_ref = int((~keep).sum())
    rep.add(tag + "F3 removals", abs((map_in.shape[0] + new.shape[0] - removed_ref) - final.shape[0]) <= 2 * ties,
            "reference removed %d of %d, appended %d" % (removed_ref, map_in.shape[0], new.shape[0]))
    return ref_final


def curvature_checks(rep, name, c, cr):
    """principal curvature k (w) and direction (xyz).  Second derivatives of the HRBF implicit: sums of third-derivative
    terms of both signs divided by g_z^3 — conditioned far worse than the gradient.  Bounds: 99 % of the valid pixels within
    1e-3 relative (+1e-3 absolute) on k; directions compared up to sign flips of near-degenerate (umbilic) pixels."""
    valid = (cr[..., 3] != 1000.0) & (c[..., 3] != 1000.0) & np.isfinite(cr[..., 3]) & np.isfinite(c[..., 3])
    k, kr = c[..., 3][valid].astype(np.float64), cr[..., 3][valid].astype(np.float64)
    err = np.abs(k - kr) / (np.abs(kr) + 1.0)
    share = float((err <= 1e-3).mean()) if err.size else 1.0
    rep.add("P4 %s k" % name, share >= 0.99, "|dk| <= 1e-3 (|k| + 1): %.3f%% of %d valid pixels, median %.2e, p99 %.2e" % (
        100 * share, err.size, np.median(err) if err.size else 0, np.percentile(err, 99) if err.size else 0))
    d, dr = c[..., :3][valid].astype(np.float64), cr[..., :3][valid].astype(np.float64)
    ok = np.isfinite(d).all(1) & np.isfinite(dr).all(1)
    dev = np.linalg.norm(d[ok] - dr[ok], axis=1)
    share = float((dev <= 1e-3).mean()) if dev.size else 1.0
    rep.add("P4 %s direction" % name, share >= 0.97, "|d - d_ref| <= 1e-3: %.3f%% of %d, median %.2e" % (
        100 * share, dev.size, np.median(dev) if dev.size else 0))
    rep.add("P4 %s non-finite values" % name, True, "NaN/inf pattern differs in %d values" % int((np.isfinite(c) != np.isfinite(cr)).sum()))


def index_checks(rep, tag, g, fx, pre, full):
    idx, ref = g.get_image("INDEX"), fx[pre + "INDEX"]
    bad = int((idx != ref).sum())
    # a point whose window coordinate falls within an ulp of the middle of a 1/256-pixel snap interval can round either way
    rep.add(tag + " INDEX (winning surfel per pixel)", bad <= max(2, idx.size // 5000), "%d of %d pixels differ (sub-pixel snap ties)" % (bad, idx.size))
    same = idx == ref
    if full:
        for k in ("INDEX_COLORTIME", "INDEX_CURVMAX", "INDEX_CURVMIN"):
            rep.exact(tag + " " + k, g.get_image(k)[same], fx[pre + k][same])
    for k, tol in (("INDEX_VERTCONF", 8), ("INDEX_NORMRAD", 16)):
        a, b = g.get_image(k), fx[pre + k]
        rep.close_ulp(tag + " " + k, a[same], b[same], tol, abs_floor=1e-6)


def prediction_checks(rep, g, fx, P):
    """predict_hrbf.frag: ray march to a sign change of the implicit, then <= 10 bisection steps that stop when the bracket is
    shorter than 1e-5 m or |f| < 1e-5 (predict_hrbf.frag:229-262).  An implicit value within rounding of 0 or of 1e-5 can fall
    either side in two executions: the march may stop one sample apart or the bisection one iteration apart, which moves the
    result by at most one bracket.  Hence absolute bounds set by the algorithm's own thresholds: every point within 4e-5 m
    (typical: identical, 99 % within 1e-6 m), normals within 1e-2 (99 % within 2e-4)."""
    v, vr = g.get_image("PRED_VERTEX"), fx[P + "PRED_VERTEX"]
    hit, hitr = v[..., 2] != 0, vr[..., 2] != 0
    d = int((hit != hitr).sum())
    rep.add("H2 which pixels have a prediction", d <= max(2, hit.size // 2000), "%d of %d pixels differ (%d predicted)" % (d, hit.size, int(hitr.sum())))
    both = hit & hitr
    if not both.any():
        rep.add("H2 PRED_VERTEX xyz", not hit.any() and not hitr.any(), "no pixel predicted in either")
        return
    dp = np.linalg.norm((v[..., :3] - vr[..., :3]).astype(np.float64), axis=-1)[both]
    rep.add("H2 PRED_VERTEX xyz", dp.max() <= 4e-5 and np.percentile(dp, 99) <= 1e-6, "|dp| max %.2e m, p99 %.2e, identical %.1f%%" % (
        dp.max(), np.percentile(dp, 99), 100 * (dp == 0).mean()))
    n, nr = g.get_image("PRED_NORMAL"), fx[P + "PRED_NORMAL"]
    # (a vanishing gradient normalises to NaN in both executions — 1 pixel of 250 000 with the radius multiplier at 3: same pixels)
    nan_a, nan_b = np.isnan(n[..., :3]).any(-1)[both], np.isnan(nr[..., :3]).any(-1)[both]
    dn = np.linalg.norm((n[..., :3] - nr[..., :3]).astype(np.float64), axis=-1)[both][~nan_a & ~nan_b]
    rep.add("H2 PRED_NORMAL xyz", dn.max() <= 1e-2 and np.percentile(dn, 99) <= 2e-4 and int((nan_a != nan_b).sum()) <= 1,
            "|dn| max %.2e, p99 %.2e, median %.2e; NaN normals %d / %d, %d in one execution only" % (
                dn.max(), np.percentile(dn, 99), np.median(dn), int(nan_a.sum()), int(nan_b.sum()), int((nan_a != nan_b).sum())))
    # attributes of the neighbour nearest to the predicted point: exact unless two neighbours are equally near (<= 0.02 % of pixels)
    near = np.ones(both.sum(), bool)
    for k, cols in (("PRED_VERTEX", slice(3, 4)), ("PRED_NORMAL", slice(3, 4)), ("PRED_CURV1", slice(0, 4)), ("PRED_CURV2", slice(0, 4)),
                    ("PRED_IMAGE", slice(0, 3))):
        a, b = g.get_image(k)[..., cols][both], fx[P + k][..., cols][both]
        near &= (ulp_diff(a, b) == 0).all(-1) if a.dtype == np.float32 else (a == b).all(-1)
    near &= g.get_image("PRED_TIME")[both] == fx[P + "PRED_TIME"][both]
    bad = int((~near).sum())
    rep.add("H2 nearest-neighbour attributes (confidence, radius, curvature records, colour, time)", bad <= max(1, near.size // 5000),
            "%d of %d pixels take another neighbour" % (bad, near.size))
    w, wr = g.get_image("PRED_ICPWEIGHT")[both][near].astype(np.float64), fx[P + "PRED_ICPWEIGHT"][both][near].astype(np.float64)
    rel = np.abs(w - wr) / np.abs(wr)
    rep.add("H2 PRED_ICPWEIGHT", rel.max() <= 1e-4, "relative difference max %.2e, p99 %.2e" % (rel.max(), np.percentile(rel, 99)))


def fill_checks(rep, g, fx, P):
    rep.exact("H3 FILL_VERTEX", g.get_image("FILL_VERTEX"), fx[P + "FILL_VERTEX"])
    rep.exact("H3 FILL_NORMAL", g.get_image("FILL_NORMAL"), fx[P + "FILL_NORMAL"])
    rep.exact("H3 FILL_CURV1", g.get_image("FILL_CURV1"), fx[P + "FILL_CURV1"])
    rep.exact("H3 FILL_CURV2", g.get_image("FILL_CURV2"), fx[P + "FILL_CURV2"])
    rep.exact("H3 FILL_IMAGE", g.get_image("FILL_IMAGE")[..., :3], fx[P + "FILL_IMAGE"][..., :3])
    rep.close_ulp("H3 FILL_ICPWEIGHT", g.get_image("FILL_ICPWEIGHT"), fx[P + "FILL_ICPWEIGHT"], 16)


def run_nonpow2_map(impl, fx, rep):
    """association, merge, index map and clean of the second frame at 160 x 120 against the executed shaders
    (tests/golden/ref_glsl/qqvga_map.npz).  Not a power of two: data.vert's texcoord — the uv attribute the host computes as
    fl(fl(i / w) + 1 / 2w) — differs from the fragment shaders' (i + 0.5) / w by an ulp at 43 of 160 columns and 19 of 120 rows, so its
    x, y are not exactly i + 0.5 and the PCA normal it recomputes for a new point takes its 7 x 7 window from elsewhere (hd_uv_attribute).
    The implementation's own P3 normals stand in for the fragment shader's (llvmpipe's interpolated texcoord is an ulp off the
    correctly rounded one: implementation-defined); everything the map passes do must then hold within the bounds of
    the power-of-two scenes."""
    g, P = impl, "f2_"
    fx = dict(fx); fx["_tie_factor"] = 1.0      # (2.0 until the association's half-pixel walk was taken literally in fp32, round 4)
    T2 = fx[P + "pose"]
    g.upload_frame(fx[P + "rgb"], fx[P + "depth"])
    for name in ("DEPTH_FILTERED", "DEPTH_METRIC", "DEPTH_METRIC_FILTERED"):
        g.set_image(name, fx[P + name])
    g.run_stage("VERTEX_NORMAL_RADIUS")            # own NORMAL_PCA (the fragment shader's normal under a correctly rounded texcoord)
    for name in ("VERTEX_RAW", "VERTEX_FILTERED", "RADIUS", "NORMAL", "CURV1", "CURV2", "GRADIENT_MAG", "CONFIDENCE"):
        g.set_image(name, fx[P + name])
    g.set_weighting(float(fx[P + "weighting"]))
    g.set_pose(T2)
    g.run_stage("INITIALISE")                      # init_unstableTex.vert: x, y of the radial confidence from the same attribute
    im = g.download_map()
    rep.add("F4 surfel count", im.shape[0] == int(fx[P + "init_count"][0]), "%d vs reference %d" % (im.shape[0], int(fx[P + "init_count"][0])))
    rep.close_ulp("F4 confidence (exp)", im[:4096, 3], fx[P + "init_head"][:, 3], 16)
    map_flow(rep, g, fx, "f2_", "160 x 120: ", fx["f1_map"], T2)
    # the removal rules: copy_unstable.vert's half-pixel walk is an fp32-accumulated loop that takes one more sample for about a
    # third of the surfels (hd_halfpixel_walk) — the exact-arithmetic count removes a third fewer of the planted surfels
    map_flow(rep, g, fx, "x_", "160 x 120, stable map + outliers: ", stable_map_with_outliers(fx), T2)
    return rep


def run_nonpow2_pre(impl, fx, rep):
    """P1-P5 at 160 x 120 against the executed shaders (tests/golden/ref_glsl/qqvga_pre.npz).  Not a power of two: the
    float-stepped window loops take 6 instead of 7 samples at 88 of 160 columns and 17 of 120 rows (hd_window_axis), the bilateral
    filter's taps of row 63 land a texel low (hd_tap_texel), and llvmpipe's interpolated
    texture coordinate is an ulp off the correctly rounded one at a few columns / rows, where ITS window differs (`tie_cols`,
    `tie_rows`).  Pixels within 3 of a tie column / row are excluded; everywhere else the bounds of the
    power-of-two scenes apply, loosened where the sample positions fl(i * cols) carry the coordinate's ulp (PCA normal)."""
    g = impl
    W, H = int(fx["geom"][0]), int(fx["geom"][1])
    g.upload_frame(fx["rgb"], fx["depth"])
    g.run_stage("FILTER_DEPTH")
    got, ref = g.get_image("DEPTH_FILTERED"), fx["DEPTH_FILTERED"]
    rep.exact("P1 which pixels are filtered", got == 0, ref == 0)
    # the taps of row 63 read row 62 (hd_tap_texel: floor(fl(fl(63 / 120) * 120)) = 62, as both Mesa rasterisers execute it): since
    # round 4 every row is held to the bound (rounds 1-3 read row 63 and masked the 13 rows around it)
    assert fx["tap_rows_low"].tolist() == [63]
    rep.close_ulp("P1 DEPTH_FILTERED (every row, incl. those whose taps land a texel low)", got, ref, 16)
    g.set_image("DEPTH_FILTERED", ref)
    g.run_stage("METRICISE")
    g.set_image("DEPTH_METRIC", fx["DEPTH_METRIC"]); g.set_image("DEPTH_METRIC_FILTERED", fx["DEPTH_METRIC_FILTERED"])
    g.run_stage("VERTEX_NORMAL_RADIUS")
    rep.exact("P3 VERTEX_RAW xyz", g.get_image("VERTEX_RAW")[..., :3], fx["VERTEX_RAW"][..., :3])
    rep.exact("P3 VERTEX_FILTERED", g.get_image("VERTEX_FILTERED"), fx["VERTEX_FILTERED"])
    ok = np.ones((H, W), bool)
    for c in fx["tie_cols"]:
        ok[:, max(0, c - 3):c + 4] = False
    for r in fx["tie_rows"]:
        ok[max(0, r - 3):r + 4, :] = False
    n3, n3r = g.get_image("NORMAL"), fx["NORMAL_P3"]
    rep.exact("P3 which pixels have a normal (away from tie columns / rows)", (n3[..., :3] == 0).all(-1)[ok], (n3r[..., :3] == 0).all(-1)[ok])
    both = ok & (np.linalg.norm(n3[..., :3], axis=-1) > 0.5) & (np.linalg.norm(n3r[..., :3], axis=-1) > 0.5)
    ang = np.degrees(np.arccos(np.clip((n3[..., :3] * n3r[..., :3]).sum(-1)[both], -1, 1)))
    # the samples' positions are fl(i * cols) with i accumulated from the interpolated coordinate: its ulp moves a position by
    # ~1e-5 pixel and the covariance's cancellation turns that into ~0.05 deg; a 6- instead of 7-wide window is 0.5-1 deg
    rep.add("P3 PCA normal", np.median(ang) < 0.1 and np.percentile(ang, 99) < 1.0, "angle median %.3f deg, p99 %.3f, max %.2f over %d pixels" % (
        np.median(ang), np.percentile(ang, 99), ang.max(), ang.size))
    g.set_image("NORMAL", n3r); g.set_image("VERTEX_FILTERED", fx["VERTEX_FILTERED"])
    g.run_stage("CURVATURE")
    gm, gmr = g.get_image("GRADIENT_MAG"), fx["GRADIENT_MAG"]
    rep.exact("P4 which pixels have > 15 neighbours (away from tie columns / rows)", (gm == 0)[ok], (gmr == 0)[ok])
    rep.close_ulp("P4 GRADIENT_MAG: the 6 / 7-sample windows of the fp32 loop", gm[ok], gmr[ok], 1024, abs_floor=1e-3)
    nz = ok & (gm != 0) & (gmr != 0)
    six = ((fx["win_x"] == 6)[None, :] | (fx["win_y"] == 6)[:, None]) & nz
    rep.add("P4 pixels whose window is 6 wide in x or y", six.sum() > 0.4 * nz.sum(), "%d of %d compared pixels" % (int(six.sum()), int(nz.sum())))
    for name in ("CURV1", "CURV2"):
        c, cr = g.get_image(name), fx[name]
        rep.exact("P4 %s which pixels are the 1000-sentinel (away from tie columns / rows)" % name, (c[..., 3] == 1000.0)[ok], (cr[..., 3] == 1000.0)[ok])
        cm, crm = c.copy(), cr.copy()
        cm[~ok] = 1000.0; crm[~ok] = 1000.0
        curvature_checks(rep, name, cm, crm)
    no, nor = g.get_image("NORMAL"), fx["NORMAL"]
    rep.close_ulp("P5 NORMAL (HRBF gradient direction)", no[..., :3][ok], nor[..., :3][ok], 64, abs_floor=4e-6)
    return rep


def run_vga(impl, fx, rep, rasteriser_texcoords=False):
    """Every GLSL pass at 640 x 480 — the size BASELINE's metric is quoted on — against the reference's shaders executed on the WHOLE
    GPUTest pair (tests/golden/ref_glsl/vga.npz, decoded by tests/ref_glsl_vga.py): frame 2's pre-processing, both map flows, the
    prediction, the fill-in, updateModel.  The bounds are those of the power-of-two fixtures (`run`), with ONE computed mask:

      exact = pixels where the texcoord the rasteriser interpolated over the full-screen quad (recorded in the fixture: `tc`) IS the
              correctly rounded (p + 0.5) / n that the oracle and the kernels use (15.5 % of the pixels on llvmpipe; an ulp off at the
              others; softpipe is off at OTHER pixels: the varying's rounding is the GL implementation's).

    P3's PCA window and P4's HRBF window are float-stepped loops that start from that coordinate.  On `exact` pixels the
    power-of-two bounds must hold unchanged (normals to 64 ulp, every > 15-neighbours decision, every sentinel); elsewhere the
    sample positions fl(i * cols) carry the coordinate's ulp (normals: median < 0.1 deg, p99 < 1 deg) and a window may be one sample
    longer or shorter where the loop's last comparison falls within that ulp (decisions differ at <= 16 pixels).
    rasteriser_texcoords: hand the recorded coordinates to the implementation (the C oracle's test hook) — then there is no mask:
    every pixel must meet the power-of-two bounds.
    Everything else — P1 (the taps ON texel edges read floor(fl(fl(c / n) * n)), hd_tap_texel), P2, vertices, confidence, F4, index
    maps, association (the fp32 half-pixel walk), merge, clean, prediction, fill-in, updateModel — is held to the power-of-two
    bounds on the whole image, and the association-tie allowance is cut from 304 to 16 surfels of 61 100."""
    g, P = impl, "f2_"
    T2 = fx[P + "pose"]
    Hh, Ww = fx[P + "depth"].shape
    f = np.float32
    ys, xs = np.mgrid[0:Hh, 0:Ww]
    ideal = np.stack([(xs.astype(f) + f(0.5)) / f(Ww), (ys.astype(f) + f(0.5)) / f(Hh)], -1).astype(f)
    exact = (fx["tc"] == ideal).all(-1)
    rep.add("texcoord mask", 0.05 < exact.mean() < 0.95, "rasteriser's texcoord == correctly rounded at %.1f%% of the pixels" % (100 * exact.mean()))
    part_filter(rep, g, fx, P)
    if rasteriser_texcoords:
        g.set_fragment_texcoords(fx["tc"])
        part_vertex_normal_radius(rep, g, fx, P, pca="_normal_abs_floor" not in fx)
        part_curvature(rep, g, fx, P)
        g.set_fragment_texcoords(None)
    else:
        # ---- P3 --------------------------------------------------------------------------------------------------------------
        g.set_image("DEPTH_METRIC", fx[P + "DEPTH_METRIC"]); g.set_image("DEPTH_METRIC_FILTERED", fx[P + "DEPTH_METRIC_FILTERED"])
        g.run_stage("VERTEX_NORMAL_RADIUS")
        vr, vf = g.get_image("VERTEX_RAW"), g.get_image("VERTEX_FILTERED")
        rep.exact("P3 VERTEX_RAW xyz", vr[..., :3], fx[P + "VERTEX_RAW"][..., :3])
        rep.close_ulp("P3 VERTEX_RAW w (radial confidence, exp)", vr[..., 3], fx[P + "VERTEX_RAW"][..., 3], 16)
        rep.exact("P3 VERTEX_FILTERED", vf, fx[P + "VERTEX_FILTERED"])
        n3, n3r = g.get_image("NORMAL"), fx[P + "NORMAL_P3"]
        rep.exact("P3 which pixels have a normal", (n3[..., :3] == 0).all(-1), (n3r[..., :3] == 0).all(-1))
        rep.close_ulp("P3 NORMAL (PCA) xyz, exact-texcoord pixels", n3[..., :3][exact], n3r[..., :3][exact], 64, abs_floor=2e-6)
        rep.close_ulp("P3 NORMAL w = RADIUS, exact-texcoord pixels", n3[..., 3][exact], n3r[..., 3][exact], 64)
        rep.close_ulp("P3 RADIUS, exact-texcoord pixels", g.get_image("RADIUS")[exact], fx[P + "RADIUS"][exact], 64)
        both = ~exact & (np.linalg.norm(n3[..., :3], axis=-1) > 0.5) & (np.linalg.norm(n3r[..., :3], axis=-1) > 0.5)
        ang = np.degrees(np.arccos(np.clip((n3[..., :3] * n3r[..., :3]).sum(-1)[both], -1, 1)))
        rep.add("P3 PCA normal, other pixels", np.median(ang) < 0.1 and np.percentile(ang, 99) < 1.0, "angle median %.3f deg, p99 %.3f, max %.2f over %d pixels" % (
            np.median(ang), np.percentile(ang, 99), ang.max(), ang.size))
        # ---- P4 / P5 on the reference's vertex / normal images ---------------------------------------------------------------
        g.set_image("NORMAL", n3r); g.set_image("VERTEX_FILTERED", fx[P + "VERTEX_FILTERED"])
        g.run_stage("CURVATURE")
        gm, gmr = g.get_image("GRADIENT_MAG"), fx[P + "GRADIENT_MAG"]
        rep.exact("P4 which pixels have > 15 neighbours, exact-texcoord pixels", (gm == 0)[exact], (gmr == 0)[exact])
        d = int(((gm == 0) != (gmr == 0)).sum())
        rep.add("P4 which pixels have > 15 neighbours, other pixels", d <= 16, "%d of %d differ (a window one sample longer / shorter)" % (d, int((~exact).sum())))
        rep.close_ulp("P4 GRADIENT_MAG, exact-texcoord pixels", gm[exact], gmr[exact], 1024, abs_floor=1e-3)
        rep.close_ulp("P4 GRADIENT_MAG, other pixels", gm[~exact], gmr[~exact], 1024, abs_floor=1e-3, frac_within=0.99)
        no, nor = g.get_image("NORMAL"), fx[P + "NORMAL"]
        rep.close_ulp("P5 NORMAL (HRBF gradient direction), exact-texcoord pixels", no[..., :3][exact], nor[..., :3][exact], 64, abs_floor=4e-6)
        rep.close_ulp("P5 NORMAL (HRBF gradient direction), other pixels", no[..., :3][~exact], nor[..., :3][~exact], 64, abs_floor=4e-6, frac_within=0.99)
        rep.exact("P5 NORMAL w (radius carried over), exact-texcoord pixels", no[..., 3][exact], nor[..., 3][exact])
        for name in ("CURV1", "CURV2"):
            c, cr = g.get_image(name), fx[P + name]
            rep.exact("P4 %s which pixels are the 1000-sentinel, exact-texcoord pixels" % name, (c[..., 3] == 1000.0)[exact], (cr[..., 3] == 1000.0)[exact])
            cm, crm = c.copy(), cr.copy()
            cm[~exact] = 1000.0; crm[~exact] = 1000.0
            curvature_checks(rep, name + " (exact-texcoord pixels)", cm, crm)
            curvature_checks(rep, name + " (all pixels)", c, cr)
    part_confidence(rep, g, fx, P)
    # ---- the map passes: frame 2's images as the reference's shaders left them; NORMAL_PCA from the implementation's own P3 (data.vert
    # recomputes it from the filtered depth with the uv ATTRIBUTE — correctly rounded arithmetic on the host, GlobalModel.cpp:88-97) --
    g.set_image("DEPTH_METRIC", fx[P + "DEPTH_METRIC"]); g.set_image("DEPTH_METRIC_FILTERED", fx[P + "DEPTH_METRIC_FILTERED"])
    g.run_stage("VERTEX_NORMAL_RADIUS")
    for name in ("VERTEX_RAW", "VERTEX_FILTERED", "RADIUS", "NORMAL", "CURV1", "CURV2", "GRADIENT_MAG", "CONFIDENCE"):
        g.set_image(name, fx[P + name])
    g.set_pose(T2)
    g.run_stage("INITIALISE")
    im = g.download_map()
    rep.add("F4 surfel count", im.shape[0] == int(fx[P + "init_count"][0]), "%d vs reference %d" % (im.shape[0], int(fx[P + "init_count"][0])))
    ih = fx[P + "init_head"]
    rep.close_ulp("F4 position", im[:4096, 0:3], ih[:, 0:3], 4, abs_floor=1e-7)
    rep.close_ulp("F4 confidence (exp)", im[:4096, 3], ih[:, 3], 16)
    rep.exact("F4 colour / submap / init time / time", im[:4096, 4:8], ih[:, 4:8])
    rep.close_ulp("F4 normal + radius", im[:4096, 8:12], ih[:, 8:12], 4, abs_floor=1e-7)
    rep.exact("F4 curvature records", im[:4096, 12:20], ih[:, 12:20])
    fx = dict(fx); fx["_tie_factor"] = 8.0 / 152.0          # 16 surfels of 61 100 instead of 304
    map_flow(rep, g, fx, "f2_", "young map: ", fx["f1_map"], T2)
    ref_final = map_flow(rep, g, fx, "x_", "stable map + outliers: ", stable_map_with_outliers(fx), T2)
    part_prediction(rep, g, fx, ref_final, P)
    g.upload_map(ref_final)
    g.update_model([fx["x_delta"]])
    um, umr = g.download_map()[:4096], fx["x_map_updated_head"]
    rep.close_ulp("f-3 updateModel positions", um[:, 0:3], umr[:, 0:3], 4)
    rep.close_ulp("f-3 updateModel normals", um[:, 8:11], umr[:, 8:11], 4, abs_floor=1e-7)
    rep.exact("f-3 updateModel everything else", np.delete(um, [0, 1, 2, 8, 9, 10], 1), np.delete(umr, [0, 1, 2, 8, 9, 10], 1))
    return rep


# ---- Resize::vertex + denseEnough (H3) ------------------------------------------------------------------------------
def thumbnail_cells(fx, W, H):
    """per thumbnail column and row, the texels that contain the cell centre (i + 1/2) * W / (W / 20): one texel, or two when the
    centre lies exactly ON a texel edge (every cell at 640 x 480) — there NEAREST picks by the rounding of the GL implementation's
    interpolated coordinate, and llvmpipe reads the lower texel at some rows.  Asserts that the executed resize.frag read one of them."""
    from fractions import Fraction
    w, h = W // 20, H // 20
    cand = []
    for n_cells, size, key in ((w, W, "sx_%dx%d"), (h, H, "sy_%dx%d")):
        read = fx[key % (W, H)]
        axis = []
        for k in range(n_cells):
            q = Fraction(2 * k + 1, 2) * size / n_cells
            c = (int(q) - 1, int(q)) if q.denominator == 1 else (int(q),)
            assert int(read[k]) in c, ("the executed shader read a texel that does not contain the cell centre", key, k, int(read[k]), q)
            axis.append(c)
        cand.append(axis)
    return cand[0], cand[1]


def run_thumbnail(make, fx, rep):
    """`make(W, H)` returns a context of that size (dense_enough_thresh = 0.75).  The images are built from the texels that contain
    the cell centres (both texels of an edge cell are set alike, so the rounding at the edge does not enter)."""
    for W, H in fx["sizes"]:
        W, H = int(W), int(H)
        cx, cy = thumbnail_cells(fx, W, H)
        w, h = W // 20, H // 20
        n = w * h
        g = make(W, H)
        try:
            def image(cells_on, background, on=1.5):
                v = np.zeros((H, W, 4), np.float32); v[..., 2] = background
                for c in range(n):
                    for y in cy[c // w]:
                        for x in cx[c % w]:
                            v[y, x, 2] = on if c in cells_on else 0.0
                return v
            k75 = (3 * n) // 4                      # per > 0.75 needs MORE than three quarters of the cells
            tag = "H3 denseEnough %dx%d: " % (W, H)
            order = np.random.default_rng(W).permutation(n).tolist()
            cases = [("only the cell centres hold depth", image(set(range(n)), 0.0), True),
                     ("everything but the cell centres holds depth", image(set(), 2.0), False),
                     ("%d of %d cells" % (k75, n), image(set(order[:k75]), 0.0), False),
                     ("%d of %d cells" % (k75 + 1, n), image(set(order[:k75 + 1]), 0.0), True),
                     ("negative depth does not count", image(set(range(n)), 0.0, on=-1.5), False)]
            for name, img, want in cases:
                g.set_image("PRED_VERTEX", img)
                rep.exact(tag + name, np.array([g.dense_enough()]), np.array([want]))
        finally:
            g.close()
    return rep

"""The 640 x 480 executed-shader fixture (tests/golden/ref_glsl/vga.npz): lossless predictive coding of what the reference's
shaders wrote for the whole GPUTest pair — every pass, both map flows, 177 MB as plain arrays.

Most of those arrays are exact functions of others (index images are gathers from the map, the fill-in is a select, the stable
flow's association records equal the young flow's, vertices are the back-projected depth ...).  Each array is therefore stored as

    derived : nothing — a numpy PREDICTOR rebuilds it from arrays decoded before it
    xor     : bits(array) XOR bits(prediction), byte planes separated, LZMA — small where the prediction is close (a rigid transform
              evaluated in numpy's op order instead of llvmpipe's: a few ulp)
    raw     : byte planes separated, LZMA — what nothing here predicts (integer inputs, index images, a few heads)

and a CRC32 of every decoded array is kept: `decode` either returns the reference's bits or raises.  Predictors only decide the
file's SIZE, never its content — a wrong predictor makes the file bigger, not different.

Two families of predictors.  Plain numpy (elementwise IEEE fp32, gathers, selects) for everything that is an exact function of
other arrays.  And, since round 6 (the tree is pushed to a GPU box on every run: 29 MB -> 8 MB), the C ORACLE for the float-heavy
passes that numpy does not restate — the bilateral filter, the PCA normals, the HRBF curvature pass, the ray-cast prediction —
run stage by stage on the arrays decoded so far, the way tests/ref_glsl_check.run_vga feeds them: the oracle's output is within
ulps of the shaders', so the XOR residual is mostly zero bits.  For the same reason frame 1's pre-processing images are now kept
(helper arrays): the seed map's normals and curvature records are gathers from them.  This does NOT make the comparison circular:
the residual restores the shaders' bits whatever the oracle computed and the CRC proves it; an oracle that drifts gives residuals
that no longer decode, and `decode` raises instead of returning anything else.  The price: the file is tied to the oracle's
arithmetic of those four stages — after a change there, re-run `make_ref_glsl.py --vga-fixture` in the build container (10 s).
`encode` runs in the build container, `decode` wherever the tests run (tests/ may use the oracle; the product never does).
"""
import json
import lzma
import os
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
f32 = np.float32
W, H, FX, FY, CX, CY = 640, 480, 528.0, 528.0, 320.0, 240.0
DEPTH_SCALE = 1.0 / 5000.0


# ---- byte-plane packing ----------------------------------------------------------------------------------------------------
def _pack(a):
    a = np.ascontiguousarray(a)
    b = a.view(np.uint8).reshape(-1, a.dtype.itemsize)
    return np.frombuffer(lzma.compress(np.ascontiguousarray(b.T).tobytes(), preset=6), np.uint8)


def _unpack(blob, dtype, shape):
    dt = np.dtype(dtype)
    raw = np.frombuffer(lzma.decompress(blob.tobytes()), np.uint8)
    return np.ascontiguousarray(raw.reshape(dt.itemsize, -1).T).view(dt).reshape(shape)


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


# ---- predictors: fx (arrays decoded so far) -> array ---------------------------------------------------------------------------
def _png(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLD, name + ".png")))


def _cam():
    return f32(CX), f32(CY), f32(1.0 / FX), f32(1.0 / FY)


def p_texcoord(fx):
    ys, xs = np.mgrid[0:H, 0:W]
    return np.stack([(xs.astype(f32) + f32(0.5)) / f32(W), (ys.astype(f32) + f32(0.5)) / f32(H)], -1).astype(f32)


def _frame(fx, pre):
    return (fx["f2_rgb"], fx["f2_depth"]) if pre == "f2_" else (_png("1c"), _png("1d"))


def p_metric(fx, pre="f2_"):
    d = _frame(fx, pre)[1]
    hi, lo = np.uint32(f32(3.5) / f32(DEPTH_SCALE)), np.uint32(f32(0.3) / f32(DEPTH_SCALE))
    return np.where((d > hi) | (d < lo), f32(0), d.astype(f32) * f32(DEPTH_SCALE)).astype(f32)


def p_metric_filtered(fx, pre="f2_"):
    v = fx[pre + "DEPTH_FILTERED"]
    hi, lo = f32(3.5) / f32(DEPTH_SCALE), f32(0.3) / f32(DEPTH_SCALE)
    return np.where((v > hi) | (v < lo), f32(0), v * f32(DEPTH_SCALE)).astype(f32)


def _radial_conf():
    ys, xs = np.mgrid[0:H, 0:W]
    x = ((xs.astype(f32) + f32(0.5)) / f32(W)) * f32(W); y = ((ys.astype(f32) + f32(0.5)) / f32(H)) * f32(H)
    md = np.sqrt(f32(H * 0.5) * f32(H * 0.5) + f32(W * 0.5) * f32(W * 0.5)).astype(f32)
    dx, dy = x - f32(CX), y - f32(CY)
    r = np.sqrt(dx * dx + dy * dy).astype(f32) / md
    return np.exp(-(r * r) / f32(0.72)).astype(f32)


def _vertex(z, valid, w):
    cx, cy, camz, camw = _cam()
    ys, xs = np.mgrid[0:H, 0:W]
    xi, yi = xs.astype(f32), ys.astype(f32)                # int(x), int(y) of the pixel (depth_vertex_normal_radius.frag:25-29)
    out = np.zeros((H, W, 4), f32)
    out[..., 0] = np.where(valid, (xi - cx) * z * camz, 0); out[..., 1] = np.where(valid, (yi - cy) * z * camw, 0); out[..., 2] = np.where(valid, z, 0)
    out[..., 3] = w
    return out


def _valid(fx, pre="f2_"):
    return (fx[pre + "NORMAL_P3"][..., :3] != 0).any(-1)


def p_vertex_raw(fx):
    return _vertex(fx["f2_DEPTH_METRIC"], _valid(fx), _radial_conf())


def p_vertex_filtered(fx, pre="f2_"):
    return _vertex(fx[pre + "DEPTH_METRIC_FILTERED"], _valid(fx, pre), f32(1))


def p_normal_p3(fx, pre="f2_"):
    """xyz: the oracle's PCA normal (o_normal_p3); w = radius_multiplier * getRadius(z_filtered, n.z) (surfels.glsl:19-33)"""
    n = fx[pre + "NORMAL_P3_xyz"]
    out = np.zeros((H, W, 4), f32); out[..., :3] = n
    z = fx[pre + "DEPTH_METRIC_FILTERED"]
    camz, camw = f32(1.0 / FX), f32(1.0 / FY)
    mean_focal = ((f32(1) / abs(camz)) + (f32(1) / abs(camw))) / f32(2)
    rad = (z / mean_focal) * f32(1.41421356237)
    with np.errstate(divide="ignore", invalid="ignore"):
        rn = np.minimum(f32(2) * rad, rad / np.abs(n[..., 2]))
    out[..., 3] = np.where((n != 0).any(-1), f32(4) * rn, 0)
    return out


def p_normal(fx, pre="f2_"):
    out = np.zeros((H, W, 4), f32); out[..., :3] = fx[pre + "NORMAL_xyz"]; out[..., 3] = fx[pre + "NORMAL_P3"][..., 3]
    return out


# ---- predictors that run the C oracle, one stage at a time, on the arrays decoded so far (see the header) ------------------------
_ORC = {}


def _oracle():
    if "o" not in _ORC:
        import sys
        sys.path.insert(0, os.path.dirname(HERE))
        import oracle_lib
        from hrbffusion3d_amd.params import default_params
        oracle_lib.build()
        _ORC["o"] = oracle_lib.Oracle(default_params(max_surfels=1 << 20), omp=True)     # 640 x 480, K = (528, 528, 320, 240), 1 / 5000
    return _ORC["o"]


def _release_oracle():
    o = _ORC.pop("o", None)
    if o is not None:
        o.close()
    _ORC.clear()


def o_depth_filtered(fx, pre):
    """P1 filterDepth on the frame's raw images"""
    o = _oracle()
    o.upload_frame(*_frame(fx, pre)); o.run_stage("FILTER_DEPTH")
    return o.get_image("DEPTH_FILTERED")


def _o_p3(fx, pre):
    """P3 on the shaders' metric depth images, at the texcoords the rasteriser interpolated (`tc`, the oracle's test hook)"""
    o = _oracle()
    o.upload_frame(*_frame(fx, pre)); o.set_fragment_texcoords(fx["tc"])
    o.set_image("DEPTH_METRIC", fx[pre + "DEPTH_METRIC"]); o.set_image("DEPTH_METRIC_FILTERED", fx[pre + "DEPTH_METRIC_FILTERED"])
    o.run_stage("VERTEX_NORMAL_RADIUS")
    return o


def o_normal_p3(fx, pre):
    o = _o_p3(fx, pre)
    n = np.ascontiguousarray(o.get_image("NORMAL")[..., :3])
    o.set_fragment_texcoords(None)
    return n


def o_curvature(fx, pre, name):
    """P4 / P5 on the shaders' PCA normals and filtered vertices: CURV1, CURV2, GRADIENT_MAG and the refined normal, one run"""
    if ("curv", pre) not in _ORC:
        o = _o_p3(fx, pre)
        o.set_image("NORMAL", fx[pre + "NORMAL_P3"]); o.set_image("VERTEX_FILTERED", fx[pre + "VERTEX_FILTERED"])
        o.run_stage("CURVATURE")
        _ORC[("curv", pre)] = {"CURV1": o.get_image("CURV1"), "CURV2": o.get_image("CURV2"), "GRADIENT_MAG": o.get_image("GRADIENT_MAG"),
                               "NORMAL_xyz": np.ascontiguousarray(o.get_image("NORMAL")[..., :3])}
        o.set_fragment_texcoords(None)
    return _ORC[("curv", pre)][name]


def o_prediction(fx, name):
    """H2 predictHRBF on the shaders' final map and index images at frame 2's pose"""
    if "pred" not in _ORC:
        o = _oracle()
        o.upload_map(ref_final(fx, "x_")); o.set_pose(fx["f2_pose"]); o.set_tick(2)
        for k in ("INDEX", "INDEX_VERTCONF", "INDEX_COLORTIME", "INDEX_NORMRAD", "INDEX_CURVMAX", "INDEX_CURVMIN"):
            o.set_image(k, fx["x_p_" + k])
        o.run_stage("PREDICT_HRBF")
        _ORC["pred"] = {k: np.ascontiguousarray(o.get_image(k)[..., :3]) for k in ("PRED_VERTEX", "PRED_NORMAL")}
    return _ORC["pred"][name]


def p_f1_map_rest(fx):
    """init_unstableTex.vert copies the row's pixel of frame 1's NORMAL (xyz + radius), CURV1 and CURV2 images"""
    pix = fx["f1_pix"].astype(np.int64)
    out = np.zeros((pix.size, 20), f32)
    for c, k in ((8, "NORMAL"), (12, "CURV1"), (16, "CURV2")):
        out[:, c:c + 4] = fx["f1_" + k].reshape(-1, 4)[pix]
    return out


def p_confidence(fx):
    return (_radial_conf() * f32(np.asarray(fx["f2_weighting"]).ravel()[0])).astype(f32)


def stable_map(fx):
    m = fx["f1_map"].copy(); m[:, 3] += 6.0
    old = fx["x_old"]; m[old, 3] = 1.0; m[old, 7] = -250.0
    return np.concatenate([m, fx["x_extra"]])


def map_in(fx, pre):
    return fx["f1_map"] if pre == "f2_" else stable_map(fx)


def ref_final(fx, pre):
    m = map_in(fx, pre)
    fused = m.copy(); fused[fx[pre + "fused_rows"]] = fx[pre + "fused_vals"]
    keep = np.unpackbits(fx[pre + "keep"])[:m.shape[0]].astype(bool)
    new = fx[pre + "records"][fx[pre + "new_picks"]].copy(); new[:, 7] = 2.0
    return np.concatenate([fused[keep], new])


def _tinv(fx):
    return np.linalg.inv(fx["f2_pose"].astype(np.float64)).astype(f32)


def _index_attr(fx, m, idx, which):
    """index_map.vert:38,62-66 for the winner of every pixel, in numpy's op order (llvmpipe's differs by <= 2 ulp on the normal)"""
    T = _tinv(fx)
    hit = idx > 0
    rows = m[idx[hit]]
    out = np.zeros((H, W, 4), f32)
    if which == "VERTCONF":
        p = rows[:, 0:3]
        v = np.stack([((T[r, 0] * p[:, 0] + T[r, 1] * p[:, 1]) + T[r, 2] * p[:, 2]) + T[r, 3] for r in range(3)], 1)
        out[hit] = np.concatenate([v, rows[:, 3:4]], 1)
    elif which == "NORMRAD":
        n = rows[:, 8:11]
        v = np.stack([(T[r, 0] * n[:, 0] + T[r, 1] * n[:, 1]) + T[r, 2] * n[:, 2] for r in range(3)], 1)
        d2 = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            v = v * (f32(1) / np.sqrt(d2))[:, None]
        out[hit] = np.concatenate([v, rows[:, 11:12]], 1)
    else:
        c = {"COLORTIME": slice(4, 8), "CURVMAX": slice(12, 16), "CURVMIN": slice(16, 20)}[which]
        out[hit] = rows[:, c]
    return out.astype(f32)


def _nn(fx, name):
    """attribute image `name` of the prediction's index map, looked up at the nearest neighbour x_nn_off points at"""
    off = fx["x_nn_off"].astype(np.int64)
    ys, xs = np.mgrid[0:H, 0:W]
    yy, xx = np.clip(ys + off[..., 1], 0, H - 1), np.clip(xs + off[..., 0], 0, W - 1)
    has = off[..., 0] != -128
    a = fx["x_p_" + name][yy, xx]
    return a, has


def p_pred4(fx, key, src, col=None):
    if col is None:                                       # whole vec4 copied from the neighbour (curvature records)
        a, has = _nn(fx, src)
        return np.where(has[..., None], a, 0).astype(f32)
    a, has = _nn(fx, src)                                 # xyz: nothing; w from the neighbour
    out = np.zeros((H, W, 4), f32); out[..., :3] = fx[key + "_xyz"]; out[..., 3] = np.where(has, a[..., col], 0)
    return out


def p_pred_image(fx):
    a, has = _nn(fx, "INDEX_COLORTIME")
    c = a[..., 0].astype(np.int64)
    out = np.zeros((H, W, 4), np.uint8)
    out[..., 0] = (c >> 16) & 255; out[..., 1] = (c >> 8) & 255; out[..., 2] = c & 255; out[..., 3] = 255
    out[~has] = 0
    return out


def p_pred_time(fx):
    a, has = _nn(fx, "INDEX_COLORTIME")
    return np.where(has, a[..., 2], 0).astype(np.uint32)


def _icp_weight(z, conf, k1, k2):
    cmax = np.maximum(np.abs(k1), np.abs(k2)).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        return ((f32(1) / (z * z)) * (conf / f32(256) + np.exp(f32(-0.5) * f32(100) / (cmax * cmax)).astype(f32))).astype(f32)


def p_pred_icpweight(fx):
    v = fx["x_PRED_VERTEX"]
    w = _icp_weight(v[..., 2], v[..., 3], fx["x_PRED_CURV1"][..., 3], fx["x_PRED_CURV2"][..., 3])
    return np.where(v[..., 2] != 0, w, 0).astype(f32)


def p_fill(fx, which):
    pv = fx["x_PRED_VERTEX"]
    k1, k2 = fx["f2_CURV1"], fx["f2_CURV2"]
    live_ok = (np.abs(k1[..., 3]) < 300) & (np.abs(k2[..., 3]) < 300)
    empty = pv[..., 2] == 0
    if which == "VERTEX":
        live = np.concatenate([fx["f2_VERTEX_FILTERED"][..., :3], fx["f2_CONFIDENCE"][..., None]], -1)
        return np.where(empty[..., None], np.where(live_ok[..., None], live, 0), pv).astype(f32)
    if which == "ICPWEIGHT":
        live = _icp_weight(fx["f2_VERTEX_FILTERED"][..., 2], fx["f2_CONFIDENCE"], k1[..., 3], k2[..., 3])
        return np.where(empty, np.where(live_ok, live, 0), fx["x_PRED_ICPWEIGHT"]).astype(f32)
    if which == "NORMAL":
        pn = fx["x_PRED_NORMAL"]
        short = np.sqrt((pn[..., :3].astype(np.float64) ** 2).sum(-1)) < 0.8
        return np.where(short[..., None], fx["f2_NORMAL"], pn).astype(f32)
    if which in ("CURV1", "CURV2"):
        a, b = fx["x_PRED_CURV1"], fx["x_PRED_CURV2"]
        use_live = (a[..., 3] > 300) | (b[..., 3] > 300)
        return np.where(use_live[..., None], fx["f2_" + which], fx["x_PRED_" + which]).astype(f32)
    if which == "IMAGE":
        pi = fx["x_PRED_IMAGE"]
        zero = pi[..., :3].astype(np.int64).sum(-1) == 0
        live = np.concatenate([fx["f2_rgb"], np.full((H, W, 1), 255, np.uint8)], -1)
        return np.where(zero[..., None], live, pi).astype(np.uint8)
    raise KeyError(which)


def _uv_attribute(n):
    """GlobalModel.cpp:88-97: ((float)i / (float)n) + 1.0 / (2 * (float)n), a float quotient and a double sum, stored as float"""
    i = np.arange(n)
    return ((i.astype(f32) / f32(n)).astype(np.float64) + 1.0 / float(2 * f32(n))).astype(f32)


def _encode_rgb(rgb):
    return ((rgb[..., 0].astype(np.int64) << 16) + (rgb[..., 1].astype(np.int64) << 8) + rgb[..., 2].astype(np.int64)).astype(f32)


def _rigid(T, p):
    return np.stack([((T[r, 0] * p[:, 0] + T[r, 1] * p[:, 1]) + T[r, 2] * p[:, 2]) + T[r, 3] for r in range(3)], 1).astype(f32)


def _get_radius(z, nz, mult):
    camz, camw = f32(1.0 / FX), f32(1.0 / FY)
    mean_focal = ((f32(1) / abs(camz)) + (f32(1) / abs(camw))) / f32(2)
    rad = (z / mean_focal) * f32(1.41421356237)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (f32(mult) * np.minimum(f32(2) * rad, rad / np.abs(nz))).astype(f32)


def p_f1_map(fx):
    """init_unstableTex.vert:33-56 at the identity pose: position = the back-projected raw depth of the row's pixel, confidence =
    the radial weight at the uv attribute's x, y, colour word from the rgb image; normal, radius and curvature records: nothing"""
    pix = fx["f1_pix"].astype(np.int64)
    py, px = pix // W, pix % W
    rgb, d = _png("1c"), _png("1d")
    hi, lo = np.uint32(f32(3.5) / f32(DEPTH_SCALE)), np.uint32(f32(0.3) / f32(DEPTH_SCALE))
    z = np.where((d > hi) | (d < lo), f32(0), d.astype(f32) * f32(DEPTH_SCALE)).astype(f32)[py, px]
    cx, cy, camz, camw = _cam()
    out = fx["f1_map_rest"].copy()
    out[:, 0] = (px.astype(f32) - cx) * z * camz; out[:, 1] = (py.astype(f32) - cy) * z * camw; out[:, 2] = z
    x, y = _uv_attribute(W)[px] * f32(W), _uv_attribute(H)[py] * f32(H)
    md = np.sqrt(f32(H * 0.5) * f32(H * 0.5) + f32(W * 0.5) * f32(W * 0.5)).astype(f32)
    dx, dy = x - cx, y - cy
    r = np.sqrt(dx * dx + dy * dy).astype(f32) / md
    out[:, 3] = np.exp(-(r * r) / f32(0.72)).astype(f32)
    out[:, 4] = _encode_rgb(rgb[py, px]); out[:, 5] = 0; out[:, 6] = rgb[py, px, 2].astype(f32) / f32(255); out[:, 7] = 1
    return out.astype(f32)


def p_records(fx):
    """data.vert:63-100 for the record's pixel: position = pose * back-projected raw depth at the uv attribute's x, y, confidence,
    colour word, curvature records = the images' texels; the recomputed normal and the merge / new flag: nothing"""
    pix = fx["f2_rec_pix"].astype(np.int64)
    py, px = pix // W, pix % W
    cx, cy, camz, camw = _cam()
    x, y = _uv_attribute(W)[px] * f32(W), _uv_attribute(H)[py] * f32(H)
    z = fx["f2_DEPTH_METRIC"][py, px]
    vl = np.stack([(x - cx) * z * camz, (y - cy) * z * camw, z], 1).astype(f32)
    T = fx["f2_pose"].astype(f32)
    out = fx["f2_records_rest"].copy()
    out[:, 0:3] = _rigid(T, vl)
    out[:, 3] = fx["f2_CONFIDENCE"][py, px]
    out[:, 4] = _encode_rgb(fx["f2_rgb"][py, px]); out[:, 5] = 0; out[:, 6] = 2
    n = out[:, 8:11]
    Ti = _tinv(fx)
    nz = (Ti[2, 0] * n[:, 0] + Ti[2, 1] * n[:, 1]) + Ti[2, 2] * n[:, 2]
    out[:, 11] = _get_radius(fx["f2_DEPTH_METRIC_FILTERED"][py, px], nz, 4.0)
    out[:, 12:16] = fx["f2_CURV1"][py, px]; out[:, 16:20] = fx["f2_CURV2"][py, px]
    return out.astype(f32)


def p_fused(fx, pre):
    """update.vert:51-115: the confidence-weighted average of the surfel and the record merged into it"""
    m = map_in(fx, pre)[fx[pre + "fused_rows"]]
    r = fx[pre + "records"][fx[pre + "fused_rec"]]
    ck, a = m[:, 3:4], r[:, 3:4]
    s = ck + a
    out = m.copy()
    avg = lambda u, v: ((ck * u) + (a * v)) / s
    merge = (r[:, 11] < f32(1.5) * m[:, 11])[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        pos = avg(m[:, 0:3], r[:, 0:3])
        dec = lambda c: np.stack([((c.astype(np.int64) >> 16) & 255), ((c.astype(np.int64) >> 8) & 255), (c.astype(np.int64) & 255)], 1).astype(f32) / f32(255)
        col = avg(dec(m[:, 4]), dec(r[:, 4]))
        enc = ((np.rint(col[:, 0] * f32(255)).astype(np.int64) << 16) + (np.rint(col[:, 1] * f32(255)).astype(np.int64) << 8) + np.rint(col[:, 2] * f32(255)).astype(np.int64)).astype(f32)
        nr = avg(m[:, 8:12], r[:, 8:12])
        d2 = (nr[:, 0] * nr[:, 0] + nr[:, 1] * nr[:, 1]) + nr[:, 2] * nr[:, 2]
        nn = nr[:, 0:3] * (f32(1) / np.sqrt(d2))[:, None]
        c1, c2 = avg(m[:, 12:16], r[:, 12:16]), avg(m[:, 16:20], r[:, 16:20])
    out[:, 0:3] = np.where(merge, pos, m[:, 0:3]); out[:, 3] = s[:, 0]
    out[:, 4] = np.where(merge[:, 0], enc, m[:, 4]); out[:, 7] = 2
    out[:, 8:11] = np.where(merge, nn, m[:, 8:11]); out[:, 11] = np.where(merge[:, 0], nr[:, 3], m[:, 11])
    out[:, 12:16] = np.where(merge, c1, m[:, 12:16]); out[:, 16:20] = np.where(merge, c2, m[:, 16:20])
    return out.astype(f32)


# ---- encoder side: the helper arrays of the coding ---------------------------------------------------------------------------
def _project_pix(local):
    u = np.rint(local[:, 0] / local[:, 2] * FX + CX - 0.25).astype(np.int64)      # vertices sit at integer or at half-pixel coordinates
    v = np.rint(local[:, 1] / local[:, 2] * FY + CY - 0.25).astype(np.int64)
    return v * W + u


def f1_pixels(fx):
    pix = _project_pix(fx["f1_map"][:, 0:3].astype(np.float64))
    assert (np.diff(pix) != 0).all()
    return pix.astype(np.uint32)


def record_pixels(fx):
    Ti = np.linalg.inv(fx["f2_pose"].astype(np.float64))
    loc = fx["f2_records"][:, 0:3].astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]
    pix = _project_pix(loc)
    assert (_bits(fx["f2_CURV1"].reshape(-1, 4)[pix]) == _bits(fx["f2_records"][:, 12:16])).all(), "record -> pixel"
    return pix.astype(np.uint32)


def fused_records(fx, pre):
    """which record update.vert merged into each changed surfel: the merge record whose position explains the new mean"""
    from scipy.spatial import cKDTree
    rec = fx[pre + "records"]
    mi = np.nonzero(rec[:, 7] == -1.0)[0]
    m = map_in(fx, pre)[fx[pre + "fused_rows"]].astype(np.float64)
    v = fx[pre + "fused_vals"].astype(np.float64)
    a = v[:, 3:4] - m[:, 3:4]
    vg = (v[:, 3:4] * v[:, 0:3] - m[:, 3:4] * m[:, 0:3]) / a                # the record's position if the mean was taken ...
    tree = cKDTree(rec[mi, 0:3].astype(np.float64))
    d1, j1 = tree.query(vg)
    # ... and for the keep branch (position unchanged) the record nearest to the surfel with that confidence
    keep = (_bits(fx[pre + "fused_vals"][:, 0:3]) == _bits(map_in(fx, pre)[fx[pre + "fused_rows"]][:, 0:3])).all(1)
    d2, j2 = tree.query(m[:, 0:3])
    j = np.where(keep, j2, j1)
    return mi[j].astype(np.uint32)


def nearest_neighbour_offsets(fx):
    """encoder side: per predicted pixel the window offset (dx, dy) of the index-map texel whose surfel predict_hrbf.frag took the
    confidence, radius, curvature records, colour and time from (predict_hrbf.frag:283-299) — found by matching those copies"""
    off = np.full((H, W, 2), -128, np.int8)
    todo = fx["x_PRED_VERTEX"][..., 2] != 0
    ys, xs = np.mgrid[0:H, 0:W]
    want = [(_bits(fx["x_PRED_CURV1"]), "INDEX_CURVMAX", slice(0, 4)), (_bits(fx["x_PRED_CURV2"]), "INDEX_CURVMIN", slice(0, 4)),
            (_bits(fx["x_PRED_VERTEX"])[..., 3:4], "INDEX_VERTCONF", slice(3, 4)), (_bits(fx["x_PRED_NORMAL"])[..., 3:4], "INDEX_NORMRAD", slice(3, 4))]
    for r in range(0, 5):
        for dy in range(-r, r + 1):
            for dx in range(-r, r + 1):
                if max(abs(dx), abs(dy)) != r or not todo.any():
                    continue
                yy, xx = np.clip(ys + dy, 0, H - 1), np.clip(xs + dx, 0, W - 1)
                m = todo & (fx["x_p_INDEX"][yy, xx] > 0)
                for b, name, sl in want:
                    m &= (b == _bits(fx["x_p_" + name])[yy, xx][..., sl]).all(-1)
                off[m] = (dx, dy)
                todo &= ~m
    return off


# ---- the plan: key -> predictor (None = raw), in decoding order -------------------------------------------------------------
def plan():
    P = []
    add = lambda k, fn=None: P.append((k, fn))
    add("f2_pose"); add("f2_weighting"); add("tc", p_texcoord)
    add("f2_rgb", lambda fx: _png("2c")); add("f2_depth", lambda fx: _png("2d"))
    for pre in ("f2_", "f1_"):            # a frame's pre-processing (frame 1's images are helper arrays: the seed map gathers from them)
        add(pre + "DEPTH_FILTERED", lambda fx, pre=pre: o_depth_filtered(fx, pre))
        add(pre + "DEPTH_METRIC", lambda fx, pre=pre: p_metric(fx, pre)); add(pre + "DEPTH_METRIC_FILTERED", lambda fx, pre=pre: p_metric_filtered(fx, pre))
        add(pre + "NORMAL_P3_xyz", lambda fx, pre=pre: o_normal_p3(fx, pre)); add(pre + "NORMAL_P3", lambda fx, pre=pre: p_normal_p3(fx, pre))
        if pre == "f2_":
            add("f2_VERTEX_RAW", p_vertex_raw)
        add(pre + "VERTEX_FILTERED", lambda fx, pre=pre: p_vertex_filtered(fx, pre))
        if pre == "f2_":
            add("f2_RADIUS", lambda fx: fx["f2_NORMAL_P3"][..., 3])
        add(pre + "CURV1", lambda fx, pre=pre: o_curvature(fx, pre, "CURV1")); add(pre + "CURV2", lambda fx, pre=pre: o_curvature(fx, pre, "CURV2"))
        if pre == "f2_":
            add("f2_GRADIENT_MAG", lambda fx: o_curvature(fx, "f2_", "GRADIENT_MAG"))
        add(pre + "NORMAL_xyz", lambda fx, pre=pre: o_curvature(fx, pre, "NORMAL_xyz")); add(pre + "NORMAL", lambda fx, pre=pre: p_normal(fx, pre))
    add("f2_CONFIDENCE", p_confidence)
    add("f1_pix"); add("f1_map_rest", p_f1_map_rest); add("f1_map", p_f1_map); add("x_extra"); add("x_old")
    for pre in ("f2_", "x_"):
        add(pre + "a_INDEX")
        add(pre + "a_INDEX_VERTCONF", lambda fx, pre=pre: _index_attr(fx, map_in(fx, pre), fx[pre + "a_INDEX"], "VERTCONF"))
        add(pre + "a_INDEX_NORMRAD", lambda fx, pre=pre: _index_attr(fx, map_in(fx, pre), fx[pre + "a_INDEX"], "NORMRAD"))
        if pre == "f2_":
            add("f2_rec_pix"); add("f2_records_rest"); add("f2_records", p_records)
        else:
            add("x_records", lambda fx: fx["f2_records"])
        add(pre + "fused_rows"); add(pre + "fused_rec"); add(pre + "fused_vals", lambda fx, pre=pre: p_fused(fx, pre))
        add(pre + "c_INDEX", lambda fx, pre=pre: fx[pre + "a_INDEX"])
        add(pre + "keep"); add(pre + "new_picks"); add(pre + "map_count")
    add("x_p_INDEX")
    for k in ("VERTCONF", "COLORTIME", "NORMRAD", "CURVMAX", "CURVMIN"):
        add("x_p_INDEX_" + k, lambda fx, k=k: _index_attr(fx, ref_final(fx, "x_"), fx["x_p_INDEX"], k))
    add("x_nn_off")
    add("x_PRED_CURV1", lambda fx: p_pred4(fx, None, "INDEX_CURVMAX")); add("x_PRED_CURV2", lambda fx: p_pred4(fx, None, "INDEX_CURVMIN"))
    add("x_PRED_VERTEX_xyz", lambda fx: o_prediction(fx, "PRED_VERTEX")); add("x_PRED_VERTEX", lambda fx: p_pred4(fx, "x_PRED_VERTEX", "INDEX_VERTCONF", 3))
    add("x_PRED_NORMAL_xyz", lambda fx: o_prediction(fx, "PRED_NORMAL")); add("x_PRED_NORMAL", lambda fx: p_pred4(fx, "x_PRED_NORMAL", "INDEX_NORMRAD", 3))
    add("x_PRED_IMAGE", p_pred_image); add("x_PRED_TIME", p_pred_time); add("x_PRED_ICPWEIGHT", p_pred_icpweight)
    for k in ("VERTEX", "NORMAL", "CURV1", "CURV2", "IMAGE", "ICPWEIGHT"):
        add("x_FILL_" + k, lambda fx, k=k: p_fill(fx, k))
    add("x_delta"); add("x_map_updated_head"); add("f2_init_count"); add("f2_init_head")
    return P


AUX = ("f2_NORMAL_P3_xyz", "f2_NORMAL_xyz", "x_PRED_VERTEX_xyz", "x_PRED_NORMAL_xyz", "x_nn_off", "f1_pix", "f1_map_rest", "f2_rec_pix",
       "f2_records_rest", "f2_fused_rec", "x_fused_rec",
       "f1_DEPTH_FILTERED", "f1_DEPTH_METRIC", "f1_DEPTH_METRIC_FILTERED", "f1_NORMAL_P3_xyz", "f1_NORMAL_P3", "f1_VERTEX_FILTERED", "f1_CURV1",
       "f1_CURV2", "f1_NORMAL_xyz", "f1_NORMAL")   # helper arrays of the coding, not passes' outputs of the fixture


def encode(full, path, extra_meta=None):
    """full: {name: array} as make_ref_glsl.run_reference('vga', ...) returns it (which records `tc`, the interpolated texcoords)"""
    full = dict(full)
    full["f2_weighting"] = np.asarray(full["f2_weighting"], f32).reshape(())
    for pre in ("f1_", "f2_"):           # (frame 1's images: make_ref_glsl.run_reference(..., keep_frame1=True))
        full[pre + "NORMAL_P3_xyz"] = np.ascontiguousarray(full[pre + "NORMAL_P3"][..., :3]); full[pre + "NORMAL_xyz"] = np.ascontiguousarray(full[pre + "NORMAL"][..., :3])
    full["x_PRED_VERTEX_xyz"] = np.ascontiguousarray(full["x_PRED_VERTEX"][..., :3]); full["x_PRED_NORMAL_xyz"] = np.ascontiguousarray(full["x_PRED_NORMAL"][..., :3])
    full["x_nn_off"] = nearest_neighbour_offsets(full)
    full["f1_pix"] = f1_pixels(full)
    rest = full["f1_map"].copy(); rest[:, 0:8] = 0; full["f1_map_rest"] = rest                  # normal, radius, curvature records
    full["f2_rec_pix"] = record_pixels(full)
    rest = full["f2_records"].copy(); rest[:, 0:7] = 0; rest[:, 11:20] = 0; full["f2_records_rest"] = rest   # the recomputed normal, the flag
    for pre in ("f2_", "x_"):
        full[pre + "fused_rec"] = fused_records(full, pre)
    store, meta, sizes = {}, {}, {}
    _release_oracle()
    for key, fn in plan():
        a = np.ascontiguousarray(full[key])
        m = {"dtype": a.dtype.str, "shape": list(a.shape), "crc": zlib.crc32(a.tobytes())}
        if fn is None:
            m["kind"] = "raw"; store[key] = _pack(a)
        else:
            pred = np.ascontiguousarray(fn(full))
            assert pred.dtype == a.dtype and pred.shape == a.shape, (key, pred.dtype, a.dtype, pred.shape, a.shape)
            x = _bits(a) ^ _bits(pred)
            if not x.any():
                m["kind"] = "derived"
            else:
                m["kind"] = "xor"; store[key] = _pack(x)
        meta[key] = m
        sizes[key] = (m["kind"], store[key].size if key in store else 0, a.nbytes, int((_bits(a) != _bits(pred)).sum()) if fn is not None else -1)
    meta["_info"] = extra_meta or {}
    store["_meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez(path, **store)
    _release_oracle()
    return sizes


def decode(path=None):
    z = np.load(path or os.path.join(GOLD, "ref_glsl", "vga.npz"))
    meta = json.loads(z["_meta"].tobytes().decode())
    fx = {}
    _release_oracle()
    for key, fn in plan():
        m = meta[key]
        if m["kind"] == "raw":
            a = _unpack(z[key], m["dtype"], m["shape"])
        else:
            pred = np.ascontiguousarray(fn(fx))
            a = pred if m["kind"] == "derived" else (_bits(pred) ^ _unpack(z[key], _bits(pred).dtype, m["shape"])).view(np.dtype(m["dtype"]))
        a = np.ascontiguousarray(a).reshape(m["shape"])
        if zlib.crc32(a.tobytes()) != m["crc"]:
            _release_oracle()
            raise AssertionError("vga fixture: %s does not decode to the recorded bits (predictor %s).  If the oracle's filter / PCA / curvature / "
                                 "prediction arithmetic changed, re-run tests/golden/make_ref_glsl.py --vga-fixture in the build container" % (key, m["kind"]))
        fx[key] = a
    _release_oracle()
    for k in AUX:
        del fx[k]
    fx["f2_weighting"] = f32(fx["f2_weighting"].ravel()[0])
    fx["_info"] = meta["_info"]
    return fx

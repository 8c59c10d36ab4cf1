"""An INDEPENDENT restatement of one whole registration step of the reference — RGBDOdometry::getIncrementalTransformation and what
feeds it — in numpy, written from the reference's CUDA / C++ only (no code shared with oracle/ or the kernels; no ctypes).

  pyramids        cudafuncs.cu:344-470 (copyMaps, copyCurvatureMap, copyicpWeightMap), :493-587 + :600-720 (resizeMap<normalize>,
                  resizeCMap, resizeicpWeightMap), :165-230 (tranformMaps, transformCurvMaps), :818-925 (verticesToDepth,
                  pyrDownGaussF, imageBGRToIntensity, pyrDownUcharGauss), :927-1028 (computeDerivativeImages, projectToPointCloud);
                  callers RGBDOdometry.cpp:183-247, 660-794
  SO3             reduce.cu:1156-1359 (so3Step) + the loop RGBDOdometry.cpp:827-914
  RGB residual    reduce.cu:957-1154 (computeRgbResidual: the `cols - 5` / `rows - 1` border, the 4 x 4 "no isolated pixel" test, the
                  nearest-texel lookup, the int sums) and sigmaVal's precedence quirk (RGBDOdometry.cpp:1017)
  ICP             reduce.cu:253-693 (icpStep, search() without the correspondence window, weighted by the model's icp weight)
  RGB step        reduce.cu:697-896
  Gauss-Newton    RGBDOdometry.cpp:916-1249: 4 / 5 / 10 iterations from level 2 to 0, A = A_rgb + w^2 A_icp, b = b_rgb + w b_icp,
                  ldlt().solve, OdometryProvider.h:35-93 (rodrigues, computeUpdateSE3), T_curr = T_prev * dT^-1, the 0.3 m guard

Arithmetic: every per-pixel DECISION (validity, thresholds, the nearest-pixel projections, the int16 Sobel values, the uint8
pyramids) is formed in fp32 / integers as the reference forms it, so that the same pixels take part; every SUM over pixels and the
6 x 6 algebra run in float64.  tests/test_registration_fp64.py compares pyramids, the first system of each level, the increments
and the composed pose with the C oracle.
"""
import numpy as np

f32 = np.float32
NAN = f32(np.nan)
GAUSS = np.array([1, 4, 6, 4, 1, 4, 16, 24, 16, 4, 6, 24, 36, 24, 6, 4, 16, 24, 16, 4, 1, 4, 6, 4, 1], np.float32).reshape(5, 5)


# ---- level 0 of the map pyramids ---------------------------------------------------------------------------------------------
def copy_maps(v4, n4):
    ok = (v4[..., 2] != 0) & (n4[..., 3] > 0)
    return np.where(ok[..., None], v4, NAN).astype(f32), np.where(ok[..., None], n4, NAN).astype(f32)


def copy_curvature(c4, thr):
    w = c4[..., 3]
    with np.errstate(invalid="ignore"):
        ok = (w < thr) & (w > -thr) & ~np.isnan(w)
    return np.where(ok[..., None], c4, NAN).astype(f32)


def copy_icp_weight(w):
    with np.errstate(invalid="ignore"):
        return np.where(w > 0, w, NAN).astype(f32)


def _quad(m):
    return m[0::2, 0::2], m[0::2, 1::2], m[1::2, 0::2], m[1::2, 1::2]


def resize_map(m, normalize):
    """resizeMapKernel: NaN in the x channel of any of the 2 x 2 -> NaN; else the mean of every channel (fp32, ((a+b)+c)+d) / 4"""
    H, W = m.shape[0] // 2 * 2, m.shape[1] // 2 * 2
    a, b, c, d = _quad(m[:H, :W])
    bad = np.isnan(a[..., 0]) | np.isnan(b[..., 0]) | np.isnan(c[..., 0]) | np.isnan(d[..., 0])
    with np.errstate(invalid="ignore"):
        out = (((a + b) + c) + d) / f32(4)
        if normalize:
            n = out[..., :3]
            out[..., :3] = n / np.sqrt((n[..., 0] * n[..., 0] + n[..., 1] * n[..., 1]) + n[..., 2] * n[..., 2])[..., None]
    out[bad] = NAN
    return out.astype(f32)


def resize_cmap(m):
    H, W = m.shape[0] // 2 * 2, m.shape[1] // 2 * 2
    a, b, c, d = _quad(m[:H, :W])
    bad = np.isnan(a[..., 3]) | np.isnan(b[..., 3]) | np.isnan(c[..., 3]) | np.isnan(d[..., 3])
    with np.errstate(invalid="ignore"):
        out = (((a + b) + c) + d) / f32(4)
    out[bad] = NAN
    return out.astype(f32)


def resize_icp_weight(w):
    H, W = w.shape[0] // 2 * 2, w.shape[1] // 2 * 2
    a, b, c, d = _quad(w[:H, :W])
    bad = np.isnan(a) | np.isnan(b) | np.isnan(c) | np.isnan(d)
    with np.errstate(invalid="ignore"):
        out = (((a + b) + c) + d) / f32(4)
    out[bad] = NAN
    return out.astype(f32)


def rotate(R, v):
    """mat33 * float3 (operators.cuh): dot of each row with v, fp32"""
    return np.stack([(R[r, 0] * v[..., 0] + R[r, 1] * v[..., 1]) + R[r, 2] * v[..., 2] for r in range(3)], -1).astype(f32)


def _transform_in_place(m, R, t=None):
    """tranformMapsKernel / tranformCurvMapsKernel with dst == src: a texel whose x is NaN gets x = NaN and KEEPS its y, z, w (a
    curvature record whose direction is NaN but whose k is finite stays a valid curvature for icpStep, which reads k only)"""
    bad = np.isnan(m[..., 0])
    with np.errstate(invalid="ignore"):
        r = rotate(R, m[..., :3])
        if t is not None:
            r = r + t
    o = m.copy()
    o[..., :3] = np.where(bad[..., None], m[..., :3], r)
    o[..., 0] = np.where(bad, NAN, o[..., 0])
    return o.astype(f32)


def transform_maps(v, n, R, t):
    return _transform_in_place(v, R, t), _transform_in_place(n, R)


def transform_curv(c, R):
    return _transform_in_place(c, R)


# ---- depth / intensity pyramids ------------------------------------------------------------------------------------------------
def vertices_to_depth(v4, cutoff):
    z = v4[..., 2]
    with np.errstate(invalid="ignore"):
        return np.where((z > cutoff) | (z <= 0), NAN, z).astype(f32)


def _pyr_down(src, valid, as_uint8):
    """pyrDownKernelGaussF / pyrDownKernelIntensityGauss: taps cy in [max(0, 2y-2), ty), cx likewise with ty = min(2y+3, rows-1)
    (so the last row / column of the source is never read), weight index (ty - cy - 1) * 5 + (tx - cx - 1)"""
    rows, cols = src.shape
    dr, dc = rows // 2, cols // 2
    ys, xs = np.mgrid[0:dr, 0:dc]
    ty = np.minimum(2 * ys + 3, rows - 1); tx = np.minimum(2 * xs + 3, cols - 1)
    y0 = np.maximum(0, 2 * ys - 2); x0 = np.maximum(0, 2 * xs - 2)
    s = np.zeros((dr, dc), np.float32); cnt = np.zeros((dr, dc), np.int64)
    # the kernel accumulates row by row, left to right, in fp32: keep that order (cy ascending = ky descending)
    for ky in range(4, -1, -1):
        cy = ty - 1 - ky
        for kx in range(4, -1, -1):
            cx = tx - 1 - kx
            inside = (cy >= y0) & (cx >= x0)
            cyc, cxc = np.clip(cy, 0, rows - 1), np.clip(cx, 0, cols - 1)
            val = src[cyc, cxc]
            use = inside & valid(val)
            wgt = GAUSS[ky, kx]
            s = np.where(use, s + val.astype(f32) * wgt, s).astype(f32)
            cnt = cnt + np.where(use, int(wgt), 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        q = s / cnt.astype(f32)
    if as_uint8:
        return np.where(cnt > 0, q, 0).astype(np.uint8)         # (uchar)(float): truncation; count = 0 does not occur on real images
    return q.astype(f32)


def pyr_down_gauss_f(src):
    return _pyr_down(src, lambda v: ~np.isnan(v), False)


def pyr_down_uchar_gauss(src):
    return _pyr_down(src, lambda v: v > 0, True)


def bgr_to_intensity(rgb):
    """bgr2IntensityKernel: int value = x * 0.114f + y * 0.299f + z * 0.587f of the texel (x, y, z = the three bytes as uploaded)"""
    x, y, z = (rgb[..., k].astype(f32) for k in range(3))
    return ((x * f32(0.114) + y * f32(0.299)) + z * f32(0.587)).astype(np.int32).astype(np.uint8)


def sobel(img):
    """applyKernel: the 3 x 3 window clipped at the border and the kernel index counted DOWN from 8 per visited pixel (so a clipped
    window uses the wrong taps: kept); float sums stored as short (truncation)"""
    gx = np.array([1, 0, -1, 2, 0, -2, 1, 0, -1], np.float32); gy = np.array([1, 2, 1, 0, 0, 0, -1, -2, -1], np.float32)
    rows, cols = img.shape
    ys, xs = np.mgrid[0:rows, 0:cols]
    j0, j1 = np.maximum(ys - 1, 0), np.minimum(ys + 1, rows - 1)
    i0, i1 = np.maximum(xs - 1, 0), np.minimum(xs + 1, cols - 1)
    ncol = i1 - i0 + 1
    dx = np.zeros((rows, cols), np.float32); dy = np.zeros((rows, cols), np.float32)
    for dj in range(3):
        for di in range(3):
            j, i = j0 + dj, i0 + di
            use = (j <= j1) & (i <= i1)
            k = 8 - (dj * ncol + di)
            v = img[np.clip(j, 0, rows - 1), np.clip(i, 0, cols - 1)].astype(f32)
            kk = np.clip(k, 0, 8)
            dx = np.where(use, dx + v * gx[kk], dx).astype(f32)
            dy = np.where(use, dy + v * gy[kk], dy).astype(f32)
    return dx.astype(np.int16), dy.astype(np.int16)


def project_to_cloud(depth, K):
    fx, fy, cx, cy = K
    ys, xs = np.mgrid[0:depth.shape[0], 0:depth.shape[1]]
    ifx, ify = f32(1.0) / f32(fx), f32(1.0) / f32(fy)
    with np.errstate(invalid="ignore"):
        return np.stack([(xs.astype(f32) - f32(cx)) * depth * ifx, (ys.astype(f32) - f32(cy)) * depth * ify, depth], -1).astype(f32)


def level_K(K, lvl):
    d = f32(1 << lvl)
    return tuple(f32(k) / d for k in K)


def rint32(a):
    """__float2int_rn"""
    with np.errstate(invalid="ignore"):
        return np.where(np.isnan(a), 0, np.rint(a)).astype(np.int64)


class Pyramids:
    """what initICPModel / initRGBModel / initCurvatureModel / initICP / initRGB / initCurvature / initICPweight leave behind"""

    def __init__(self, model, live, pose, prev_rgb, curv_thr=300.0, max_depth_rgb=6.0, levels=3):
        R, t = pose[:3, :3].astype(f32), pose[:3, 3].astype(f32)
        v, n = copy_maps(model["vertex"], model["normal"])
        self.vg, self.ng = [v], [n]
        k1, k2 = [copy_curvature(model["curv1"], curv_thr)], [copy_curvature(model["curv2"], curv_thr)]
        self.wg = [copy_icp_weight(model["icp_weight"])]
        v, n = copy_maps(live["vertex"], live["normal"])
        self.vc, self.nc = [v], [n]
        self.k1c, self.k2c = [copy_curvature(live["curv1"], curv_thr)], [copy_curvature(live["curv2"], curv_thr)]
        for i in range(1, levels):
            self.vg.append(resize_map(self.vg[-1], False)); self.ng.append(resize_map(self.ng[-1], True))
            self.vc.append(resize_map(self.vc[-1], False)); self.nc.append(resize_map(self.nc[-1], True))
            k1.append(resize_cmap(k1[-1])); k2.append(resize_cmap(k2[-1]))
            self.k1c.append(resize_cmap(self.k1c[-1])); self.k2c.append(resize_cmap(self.k2c[-1]))
            self.wg.append(resize_icp_weight(self.wg[-1]))
        for i in range(levels):   # the model's maps go to the global frame AFTER the pyramid is built (RGBDOdometry.cpp:236-244, 745-749)
            self.vg[i], self.ng[i] = transform_maps(self.vg[i], self.ng[i], R, t)
            k1[i], k2[i] = transform_curv(k1[i], R), transform_curv(k2[i], R)
        self.k1g, self.k2g = k1, k2
        self.last_depth = [vertices_to_depth(model["vertex"], f32(max_depth_rgb))]
        self.next_depth = [vertices_to_depth(live["vertex"], f32(max_depth_rgb))]
        self.last_img = [bgr_to_intensity(model["image"])]
        self.next_img = [bgr_to_intensity(live["rgb"])]
        self.prev_img = [bgr_to_intensity(prev_rgb)]          # lastNextImage: the previous frame's nextImage (initFirstRGB on frame 1)
        for i in range(1, levels):
            self.last_depth.append(pyr_down_gauss_f(self.last_depth[-1])); self.next_depth.append(pyr_down_gauss_f(self.next_depth[-1]))
            self.last_img.append(pyr_down_uchar_gauss(self.last_img[-1])); self.next_img.append(pyr_down_uchar_gauss(self.next_img[-1]))
            self.prev_img.append(pyr_down_uchar_gauss(self.prev_img[-1]))
        self.dIdx, self.dIdy = zip(*[sobel(im) for im in self.next_img])


# ---- the four reductions --------------------------------------------------------------------------------------------------------
def _upper27(rows7, weight):
    """[J^T J | J^T r] of the 7-vectors `rows7` (N, 7), each product formed in fp32 as weight * row[i] * row[j], summed in float64"""
    A = np.zeros((6, 6)); b = np.zeros(6)
    for i in range(6):
        for j in range(i, 7):
            p = ((weight * rows7[:, i]).astype(f32) * rows7[:, j]).astype(f32)
            s = float(p.astype(np.float64).sum())
            if j == 6:
                b[i] = s
            else:
                A[i, j] = A[j, i] = s
    r = float(((weight * rows7[:, 6]).astype(f32) * rows7[:, 6]).astype(f32).astype(np.float64).sum())
    return A, b, r


def so3_step(last, nxt, basis, kinv, krlr):
    rows, cols = nxt.shape
    ys, xs = np.mgrid[0:rows, 0:cols]
    x, y = xs.astype(f32), ys.astype(f32)
    w = [(basis[r, 0] * x + basis[r, 1] * y) + basis[r, 2] * f32(1) for r in range(3)]
    with np.errstate(invalid="ignore", divide="ignore"):
        wx, wy = rint32(w[0] / w[2]), rint32(w[1] / w[2])
    ok = (wx >= 1) & (wx < cols - 1) & (wy >= 1) & (wy < rows - 1) & (xs >= 1) & (xs < cols - 1) & (ys >= 1) & (ys < rows - 1)
    wx, wy, xs_, ys_ = wx[ok], wy[ok], xs[ok], ys[ok]

    def grad(img, px, py):
        a = img[py, px].astype(f32)
        gx = ((img[py, px - 1].astype(f32) + a) / f32(2)) - ((img[py, px + 1].astype(f32) + a) / f32(2))
        gy = ((img[py - 1, px].astype(f32) + a) / f32(2)) - ((img[py + 1, px].astype(f32) + a) / f32(2))
        return gx, gy
    gnx, gny = grad(nxt, wx, wy); glx, gly = grad(last, xs_, ys_)
    gx, gy = (gnx + glx) / f32(2), (gny + gly) / f32(2)
    xf, yf = xs_.astype(f32), ys_.astype(f32)
    pt = np.stack([(kinv[r, 0] * xf + kinv[r, 1] * yf) + kinv[r, 2] * f32(1) for r in range(3)], -1).astype(f32)
    z2 = pt[:, 2] * pt[:, 2]
    a, b, c, d, e, f, g, h, i = (krlr[r, s] for r in range(3) for s in range(3))
    lp = np.stack([((pt[:, 2] * (d * gy + a * gx)) - (gy * g * yf) - (gx * g * xf)) / z2,
                   ((pt[:, 2] * (e * gy + b * gx)) - (gy * h * yf) - (gx * h * xf)) / z2,
                   ((pt[:, 2] * (f * gy + c * gx)) - (gy * i * yf) - (gx * i * xf)) / z2], -1).astype(f32)
    jac = np.cross(lp, pt).astype(f32)
    r3 = -(nxt[wy, wx].astype(f32) - last[ys_, xs_].astype(f32))
    row = np.concatenate([jac, r3[:, None]], 1).astype(f32)
    A = np.zeros((3, 3)); bb = np.zeros(3)
    for ii in range(3):
        for jj in range(ii, 4):
            s = float((row[:, ii] * row[:, jj]).astype(f32).astype(np.float64).sum())
            if jj == 3:
                bb[ii] = s
            else:
                A[ii, jj] = A[jj, ii] = s
    res = float((row[:, 3] * row[:, 3]).astype(f32).astype(np.float64).sum())
    return A, bb, res, int(ok.sum())


def rgb_residual(P, lvl, min_scale, krk, kt, max_depth_delta=0.07):
    nxt, last = P.next_img[lvl], P.last_img[lvl]
    rows, cols = nxt.shape
    ys, xs = np.mgrid[0:rows, 0:cols]
    ok = (xs < cols - 5) & (ys < rows - 1)
    pos = nxt > 0
    allpos = np.ones((rows, cols), bool)          # the 4 x 4 window [i-2, i+2) x [j-2, j+2), clipped, must be all positive
    for du in range(-2, 2):
        for dv in range(-2, 2):
            u, v = ys + du, xs + dv
            inside = (u >= 0) & (u < rows) & (v >= 0) & (v < cols)
            allpos &= ~inside | pos[np.clip(u, 0, rows - 1), np.clip(v, 0, cols - 1)]
    ok &= allpos
    vx, vy = P.dIdx[lvl].astype(np.int32), P.dIdy[lvl].astype(np.int32)
    ok &= (vx * vx + vy * vy).astype(f32) >= f32(min_scale)
    d1 = P.next_depth[lvl]
    ok &= ~np.isnan(d1)
    x, y = xs.astype(f32), ys.astype(f32)
    with np.errstate(invalid="ignore", divide="ignore"):
        t1 = (d1 * ((krk[2, 0] * x + krk[2, 1] * y) + krk[2, 2]) + kt[2]).astype(f32)
        u0 = rint32((d1 * ((krk[0, 0] * x + krk[0, 1] * y) + krk[0, 2]) + kt[0]) / t1)
        v0 = rint32((d1 * ((krk[1, 0] * x + krk[1, 1] * y) + krk[1, 2]) + kt[1]) / t1)
    ok &= (u0 >= 0) & (v0 >= 0) & (u0 < cols) & (v0 < rows)
    u0c, v0c = np.clip(u0, 0, cols - 1), np.clip(v0, 0, rows - 1)
    d0 = P.last_depth[lvl][v0c, u0c]
    with np.errstate(invalid="ignore"):
        ok &= (d0 > 0) & (np.abs(t1 - d0) <= f32(max_depth_delta)) & (last[v0c, u0c] != 0)
    diff = nxt.astype(f32) - last[v0c, u0c].astype(f32)
    one = np.stack([xs[ok], ys[ok]], 1); zero = np.stack([u0c[ok], v0c[ok]], 1)
    dd = diff[ok]
    return {"one": one, "zero": zero, "diff": dd, "count": int(ok.sum()), "sigma": int((dd.astype(np.int64) ** 2).sum())}


def rgb_step(P, lvl, cor, sigma, K, sobel_scale=0.125):
    fx, fy = f32(K[0]), f32(K[1])
    cloud = project_to_cloud(P.last_depth[lvl], K)
    d = cor["diff"]
    with np.errstate(divide="ignore"):
        w = f32(sigma) + np.abs(d)
        w = np.where(w > f32(1.19209290e-07), f32(1) / w, f32(1)).astype(f32)
    if sigma == -1:
        w = np.ones_like(w)
    r6 = -w * d
    cp = cloud[cor["zero"][:, 1], cor["zero"][:, 0]]
    invz = (1.0 / cp[:, 2].astype(np.float64)).astype(f32)           # float invz = 1.0 / z: a double quotient rounded to float
    dix = w * f32(sobel_scale) * P.dIdx[lvl][cor["one"][:, 1], cor["one"][:, 0]].astype(f32)
    diy = w * f32(sobel_scale) * P.dIdy[lvl][cor["one"][:, 1], cor["one"][:, 0]].astype(f32)
    v0 = dix * fx * invz; v1 = diy * fy * invz
    v2 = -(v0 * cp[:, 0] + v1 * cp[:, 1]) * invz
    row = np.stack([v0, v1, v2, -cp[:, 2] * v1 + cp[:, 1] * v2, cp[:, 2] * v0 - cp[:, 0] * v2, -cp[:, 1] * v0 + cp[:, 0] * v1, r6], 1).astype(f32)
    A, b, _ = _upper27(row, f32(1))
    return A, b


def icp_step(P, lvl, Rcurr, tcurr, Rprev_inv, tprev, K, dist_thr=0.1, angle_thr=None, use_weight=True):
    angle_thr = f32(np.sin(f32(20.0) * f32(3.14159265) / f32(180.0))) if angle_thr is None else f32(angle_thr)
    fx, fy, cx, cy = (f32(k) for k in K)
    vc, nc = P.vc[lvl], P.nc[lvl]
    rows, cols = vc.shape[:2]
    with np.errstate(invalid="ignore", divide="ignore"):
        vg = rotate(Rcurr, vc[..., :3]) + tcurr
        vcp = rotate(Rprev_inv, vg - tprev)
        ux = rint32(vcp[..., 0] * fx / vcp[..., 2] + cx); uy = rint32(vcp[..., 1] * fy / vcp[..., 2] + cy)
        ok = ~((ux < 0) | (uy < 0) | (ux >= cols) | (uy >= rows) | (vcp[..., 2] < 0))
        ng = rotate(Rcurr, nc[..., :3])
        ok &= ~(np.isnan(vc[..., 0]) | np.isnan(nc[..., 0]) | np.isnan(P.k1c[lvl][..., 3]) | np.isnan(P.k2c[lvl][..., 3]))
        uxc, uyc = np.clip(ux, 0, cols - 1), np.clip(uy, 0, rows - 1)
        vp, npv = P.vg[lvl][uyc, uxc][..., :3], P.ng[lvl][uyc, uxc][..., :3]
        k1, k2 = P.k1g[lvl][uyc, uxc][..., 3], P.k2g[lvl][uyc, uxc][..., 3]
        dv = vp - vg
        dist = np.sqrt((dv[..., 0] * dv[..., 0] + dv[..., 1] * dv[..., 1]) + dv[..., 2] * dv[..., 2])
        cr = np.cross(ng, npv).astype(f32)
        sine = np.sqrt((cr[..., 0] * cr[..., 0] + cr[..., 1] * cr[..., 1]) + cr[..., 2] * cr[..., 2])
        ok &= ~(np.isnan(vp[..., 0]) | np.isnan(npv[..., 0]) | np.isnan(k1) | np.isnan(k2))
        ok &= ~((sine > angle_thr) | (dist > f32(dist_thr)))
        s = rotate(Rprev_inv, vg - tprev)[ok]
        d = rotate(Rprev_inv, vp - tprev)[ok]
        n = rotate(Rprev_inv, npv)[ok]
        w = np.ones(s.shape[0], np.float32)
        if use_weight:
            ww = P.wg[lvl][uyc, uxc][ok]
            w = np.where(np.isnan(ww), f32(0), ww).astype(f32)
        row = np.concatenate([n, np.cross(s, n).astype(f32), ((n * (s - d)).astype(f32) @ np.ones(3, np.float32))[:, None]], 1)
        # dot(n, s - d) = n.x * (s-d).x + n.y * (s-d).y + n.z * (s-d).z in fp32
        e = s - d
        row[:, 6] = (n[:, 0] * e[:, 0] + n[:, 1] * e[:, 1]) + n[:, 2] * e[:, 2]
    A, b, r = _upper27(row.astype(f32), w)
    return A, b, r, int(ok.sum())


# ---- host algebra (float64) ---------------------------------------------------------------------------------------------------
def rodrigues(v):
    theta = float(np.linalg.norm(v))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    r = np.asarray(v, np.float64) / theta
    c, s = np.cos(theta), np.sin(theta)
    rx = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(r, r) + s * rx


def camera_matrix(K):
    fx, fy, cx, cy = (float(k) for k in K)
    return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])


def se3_iteration(P, lvl, resultRt, Rcurr, tcurr, Rprev, tprev, K, icp_weight=10.0):
    """one pass of the inner loop of getIncrementalTransformation (RGBDOdometry.cpp:980-1228) from the given state"""
    Kl_ = level_K(K, lvl)
    Km = camera_matrix(Kl_); Kinv = np.linalg.inv(Km)
    min_scale = f32({0: 5.0, 1: 3.0, 2: 1.0}[lvl] ** 2 / 0.125 ** 2)
    Rprev_inv = np.linalg.inv(np.asarray(Rprev, np.float64)).astype(f32)
    Rt = np.linalg.inv(resultRt)
    krk = (Km @ Rt[:3, :3] @ Kinv).astype(f32)
    kt = (Km @ Rt[:3, 3]).astype(f32)
    cor = rgb_residual(P, lvl, min_scale, krk, kt)
    sigma, size = cor["sigma"], cor["count"]
    # float sigmaVal = std::sqrt((float)sigma / rgbSize == 0 ? 1 : rgbSize): the comparison binds first, so sqrt(rgbSize)
    # unless the squared differences sum to exactly 0 (RGBDOdometry.cpp:1017)
    sigma_val = float(f32(np.sqrt(1.0 if (size > 0 and sigma == 0) else float(size))))
    A_icp, b_icp, res, cnt = icp_step(P, lvl, np.asarray(Rcurr, f32), np.asarray(tcurr, f32), Rprev_inv, np.asarray(tprev, f32), Kl_)
    A_rgb, b_rgb = rgb_step(P, lvl, cor, f32(sigma_val), Kl_)
    w = float(icp_weight)
    A = A_rgb.astype(f32).astype(np.float64) + w * w * A_icp.astype(f32).astype(np.float64)
    b = b_rgb.astype(f32).astype(np.float64) + w * b_icp.astype(f32).astype(np.float64)
    x = np.linalg.solve(A, b)
    upd = np.eye(4); upd[:3, :3] = rodrigues(x[3:6]); upd[:3, 3] = x[0:3]
    new_Rt = upd @ resultRt
    odoR, odot = new_Rt[:3, :3].astype(f32).astype(np.float64), new_Rt[:3, 3].astype(f32).astype(np.float64)
    inv = np.eye(4); inv[:3, :3] = odoR.T; inv[:3, 3] = -odoR.T @ odot
    T = np.eye(4); T[:3, :3] = Rprev; T[:3, 3] = tprev
    cur = (T.astype(f32) @ inv.astype(f32)).astype(f32)
    return {"A_icp": A_icp, "b_icp": b_icp, "A_rgb": A_rgb, "b_rgb": b_rgb, "x": x, "inliers": cnt, "rgb_count": size, "sigma": sigma,
            "icp_residual": res, "resultRt": new_Rt, "Rcurr": cur[:3, :3].astype(f32), "tcurr": cur[:3, 3].astype(f32)}


def register(model, live, pose, prev_rgb, K, icp_weight=10.0, so3=True, iterations=(10, 5, 4), trace=None):
    """returns the new camera-to-world pose (4 x 4, float32) — RGBDOdometry::getIncrementalTransformation with rgb and icp on"""
    P = Pyramids(model, live, pose, prev_rgb)
    Rprev, tprev = pose[:3, :3].astype(f32), pose[:3, 3].astype(f32)
    Rcurr, tcurr = Rprev.copy(), tprev.copy()
    resultR = np.eye(3)
    if so3:
        Kl = camera_matrix(level_K(K, 2)); Kinv = np.linalg.inv(Kl)
        R_lr = np.eye(3, dtype=np.float32)
        last_err = last_cnt = float(np.finfo(np.float32).max) / 2
        last_result = np.eye(3)
        for _ in range(10):
            A, b, res, cnt = so3_step(P.prev_img[2], P.next_img[2], (Kl @ resultR @ Kinv).astype(f32), Kinv.astype(f32), (Kl @ resultR).astype(f32))
            err = float(f32(np.sqrt(f32(res))) / f32(cnt)) if cnt else float("inf")
            if trace is not None:
                trace.append(("so3", A.copy(), b.copy(), res, cnt))
            if err < last_err and last_cnt == cnt:
                break
            if err > last_err + 0.001:
                resultR = last_result
                break
            last_err, last_cnt, last_result = err, cnt, resultR
            delta = np.linalg.solve(A.astype(f32).astype(np.float64), b.astype(f32).astype(np.float64)).astype(f32)
            R_lr = (rodrigues(delta.astype(np.float64)).astype(f32) @ R_lr).astype(f32)
            resultR = R_lr.astype(np.float64)
    resultRt = np.eye(4)
    if so3:
        resultRt[:3, :3] = resultR
    for lvl in (2, 1, 0):
        for it in range(iterations[lvl]):
            step = se3_iteration(P, lvl, resultRt, Rcurr, tcurr, Rprev, tprev, K, icp_weight)
            if trace is not None:
                trace.append(("se3", lvl, it, step["A_icp"], step["b_icp"], step["A_rgb"], step["b_rgb"], step["x"], step["inliers"], step["rgb_count"],
                              step["sigma"], step["icp_residual"]))
            resultRt, Rcurr, tcurr = step["resultRt"], step["Rcurr"], step["tcurr"]
    if float(np.linalg.norm(tcurr.astype(np.float64) - tprev)) > 0.3:
        Rcurr, tcurr = Rprev, tprev
    out = np.eye(4, dtype=np.float32); out[:3, :3] = Rcurr; out[:3, 3] = tcurr
    return out, P

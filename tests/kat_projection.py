"""Independent (numpy fp64) known-answer checks of every site where the camera intrinsics enter the path, written
straight from the reference sources and run with fx != fy and an off-centre principal point, on a non-square image.

Each check takes an ENGINE with the method names shared by tests/oracle_lib.Oracle and hrbffusion3d_amd.api.HRBFFusion,
so the same known answers pin the CPU oracle (tests/test_intrinsics_kat.py, no GPU) and the HIP path
(tests/test_parity_gpu.py, -m gpu).  Every check returns the worst deviation from the numpy expectation evaluated with
the TRUE intrinsics; the callers also evaluate the expectation with fx<->fy / cx<->cy SWAPPED and assert that the
engine's output does NOT agree with it, i.e. that the check would catch such a slip.

Sites (reference file:line):
  P3  back-projection at integer pixel coordinates   geometry.glsl:21-32, depth_vertex_normal_radius.frag:25-29,64
  M1  surfel projection into the index map           index_map.vert:54-55 (+ GL point raster = floor)
  H2  viewing ray through the pixel centre           predict_hrbf.frag:42-47
  O4  projective association of icpStep              reduce.cu:326-331   (tests/test_intrinsics_kat.py: icp_fp64)
  O5  rgbStep Jacobian / projectToPointCloud         reduce.cu:717-808, cudafuncs.cu:927-960
"""
import numpy as np

import scenes

# intrinsics at 160 x 120 with the TUM fr1 proportions (517.3, 516.5, 318.6, 255.3) / 4 — and one exaggerated set whose
# two focal lengths differ by 9 % so that an fx<->fy slip is far above every tolerance
K_TUM_Q = (129.325, 129.125, 79.65, 63.825)
K_SKEWED = (141.0, 129.0, 71.5, 66.25)


def swapped(K):
    return (K[1], K[0], K[3], K[2])


def _rigid(rx, ry, rz, t):
    cx_, sx = np.cos(rx), np.sin(rx); cy_, sy = np.cos(ry), np.sin(ry); cz, sz = np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx_, -sx], [0, sx, cx_]]); Ry = np.array([[cy_, 0, sy], [0, 1, 0], [-sy, 0, cy_]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4); T[:3, :3] = Rz @ Ry @ Rx; T[:3, 3] = t
    return T


# ------------------------------------------------------------------------------------------------ P3
def back_projection_expectation(dm, K, W, H):
    """geometry.glsl:28-32 with int(x), int(y): ((x - cx) z / fx, (y - cy) z / fy, z); radial confidence of
    surfels.glsl:37-46 at the pixel CENTRE (depth_vertex_normal_radius.frag:64 passes the float coordinates)."""
    fx, fy, cx, cy = K
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    z = dm.astype(np.float64)
    v = np.stack([(xs - cx) * z / fx, (ys - cy) * z / fy, z], -1)
    r = np.hypot(xs + 0.5 - cx, ys + 0.5 - cy) / np.hypot(W / 2.0, H / 2.0)
    return v, np.exp(-r * r / 0.72)


def run_back_projection(e, W, H, K):
    n = np.array([0.3, -0.2, 1.0]); n /= np.linalg.norm(n)
    z = scenes.plane_depth(W, H, *K, n, 1.4)
    e.upload_frame(scenes.gray_rgb(W, H), scenes.to_u16(z))
    for st in ("FILTER_DEPTH", "METRICISE", "VERTEX_NORMAL_RADIUS"):
        e.run_stage(st)
    return e.get_image("DEPTH_METRIC"), e.get_image("DEPTH_METRIC_FILTERED"), e.get_image("VERTEX_RAW"), e.get_image("VERTEX_FILTERED")


def back_projection_error(out, K, W, H):
    dm, dmf, vr, vf = out
    ok = dm > 0
    assert ok.mean() > 0.95
    v, conf = back_projection_expectation(dm, K, W, H)
    vfx, _ = back_projection_expectation(dmf, K, W, H)
    okf = dmf > 0
    return max(np.abs(vr[..., :3] - v)[ok].max(), np.abs(vf[..., :3] - vfx)[okf].max(), np.abs(vr[..., 3] - conf)[ok].max())


# ------------------------------------------------------------------------------------------------ M1
def make_projection_case(W, H, K):
    """surfels that sit exactly behind chosen pixel centres of a camera at a non-trivial pose"""
    fx, fy, cx, cy = K
    T = _rigid(0.11, -0.23, 0.07, (0.4, -0.3, 0.2))
    rng = np.random.default_rng(42)
    px, py = np.meshgrid(np.arange(3, W - 3, 5), np.arange(2, H - 2, 5))
    px = px.ravel(); py = py.ravel()
    z = rng.uniform(0.8, 3.0, px.size)
    pc = np.stack([(px + 0.5 - cx) * z / fx, (py + 0.5 - cy) * z / fy, z], -1)
    pw = pc @ T[:3, :3].T + T[:3, 3]
    nc = np.tile(np.array([0.0, 0.0, 1.0]), (px.size, 1))
    nw = nc @ T[:3, :3].T
    m = np.zeros((px.size + 1, 20), np.float32)
    m[0, :3] = T[:3, :3] @ np.array([0, 0, -5.0]) + T[:3, 3]   # id 0 = "no surfel" in the index image: parked behind the camera
    m[0, 3] = 10; m[0, 8:11] = nw[0]; m[0, 11] = 0.01; m[0, 6] = m[0, 7] = 1
    m[1:, 0:3] = pw; m[1:, 3] = 10.0 + np.arange(px.size) % 7
    m[1:, 4] = 0x808080; m[1:, 6] = 1; m[1:, 7] = 1
    m[1:, 8:11] = nw; m[1:, 11] = 0.01
    m[1:, 12] = 1.0; m[1:, 17] = 1.0
    return T, m, px, py, pc


def run_projection(e, T, m):
    e.upload_map(m); e.set_pose(T.astype(np.float32)); e.set_tick(3)
    e.run_stage("PREDICT_INDICES")
    return e.get_image("INDEX"), e.get_image("INDEX_VERTCONF"), e.get_image("INDEX_NORMRAD")


def projection_expectation(T, m, K, W, H):
    """index_map.vert:41-55 + point raster: pixel = floor(fx x / z + cx), floor(fy y / z + cy); nearest z wins"""
    fx, fy, cx, cy = K
    Ti = np.linalg.inv(T)
    pc = m[:, :3].astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]
    exp = np.zeros((H, W), np.int64); zb = np.full((H, W), np.inf)
    for i in range(m.shape[0]):
        x, y, z = pc[i]
        if z < 0 or z > 20.0:
            continue
        u = int(np.floor(fx * x / z + cx)); v = int(np.floor(fy * y / z + cy))
        if 0 <= u < W and 0 <= v < H and z < zb[v, u]:
            zb[v, u] = z; exp[v, u] = i
    return exp, pc


def projection_mismatch(out, T, m, K, W, H):
    """(number of pixels whose winner differs, worst error of the camera-frame position stored with the winner)"""
    idx, vc, nr = out
    exp, pc = projection_expectation(T, m, K, W, H)
    bad = int((idx.astype(np.int64) != exp).sum())
    hit = (exp > 0) & (idx.astype(np.int64) == exp)
    err = np.abs(vc[..., :3][hit] - pc[exp[hit]]).max() if hit.any() else 0.0
    return bad, float(err), int((exp > 0).sum())


# ------------------------------------------------------------------------------------------------ H2
def run_prediction_rays(e, W, H, K):
    """a dense planar index map (one synthetic surfel behind every pixel CENTRE) -> ray cast"""
    fx, fy, cx, cy = K
    n = np.array([0.2, -0.1, 1.0]); n /= np.linalg.norm(n)
    r = scenes.pixel_rays(W, H, fx, fy, cx, cy, half=0.5)
    z = 1.5 / (r @ n)
    P = r * z[..., None]
    vcf = np.concatenate([P, np.full((H, W, 1), 10.0)], -1)
    rad = 4.0 * np.sqrt(2.0) * z / (0.5 * (fx + fy))
    nr = np.concatenate([np.broadcast_to(n, (H, W, 3)), rad[..., None]], -1)
    e.set_image("INDEX_VERTCONF", vcf); e.set_image("INDEX_NORMRAD", nr)
    ct = np.zeros((H, W, 4)); ct[..., 0] = 0x808080; ct[..., 2] = 1; ct[..., 3] = 1
    e.set_image("INDEX_COLORTIME", ct)
    k = np.zeros((H, W, 4)); k[..., 0] = 1.0
    e.set_image("INDEX_CURVMAX", k); e.set_image("INDEX_CURVMIN", k)
    e.set_image("INDEX", np.arange(1, W * H + 1, dtype=np.uint32).reshape(H, W))
    e.run_stage("PREDICT_HRBF")
    return e.get_image("PRED_VERTEX"), n


def prediction_ray_error(out, K, W, H):
    """predict_hrbf.frag:42-47: the predicted vertex of pixel (x, y) lies on the ray ((x + .5 - cx)/fx, (y + .5 - cy)/fy, 1)"""
    pv, n = out
    fx, fy, cx, cy = K
    inner = np.zeros((H, W), bool); inner[6:-6, 6:-6] = True
    ok = (pv[..., 2] > 0) & inner
    assert ok[inner].mean() > 0.99
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    ex = (xs + 0.5 - cx) / fx; ey = (ys + 0.5 - cy) / fy
    z = pv[..., 2].astype(np.float64)
    err = max(np.abs(pv[..., 0] / np.where(ok, z, 1) - ex)[ok].max(), np.abs(pv[..., 1] / np.where(ok, z, 1) - ey)[ok].max())
    plane = np.abs(pv[..., :3] @ n - 1.5)[ok].max()
    return float(err), float(plane)


# ------------------------------------------------------------------------------------------------ O4
def corner_maps(W, H, K):
    """planar (4, H, W) vertex / normal / curvature maps of the three-plane corner scene + a weight image"""
    z = scenes.corner_depth(W, H, *K)
    r = scenes.pixel_rays(W, H, *K)
    P = r * z[..., None]
    dx = np.zeros_like(P); dy = np.zeros_like(P)
    dx[:, 1:-1] = P[:, 2:] - P[:, :-2]; dy[1:-1] = P[2:] - P[:-2]
    n = np.cross(dx, dy); ln = np.linalg.norm(n, axis=-1, keepdims=True)
    n = np.where(ln > 0, n / np.maximum(ln, 1e-12), 0); n = np.where(n[..., 2:3] < 0, -n, n)
    v = np.stack([P[..., 0], P[..., 1], P[..., 2], np.ones_like(z)]).astype(np.float32)
    nn = np.stack([n[..., 0], n[..., 1], n[..., 2], np.ones_like(z)]).astype(np.float32)
    v[0][z <= 0] = np.nan; nn[0][ln[..., 0] <= 0] = np.nan
    kk = np.zeros_like(v); kk[3] = 0.5
    rng = np.random.default_rng(5)
    w = rng.uniform(0.1, 3.0, (H, W)).astype(np.float32); w[::7, ::5] = np.nan
    return v, nn, kk, w


def icp_fp64(v, nn, w, Rc, tc, K, W, H, dist_thr=0.1, angle_thr=0.342):
    """fp64 evaluation of the point-to-plane system of icpStep (reduce.cu:316-545) with the model = the live maps:
    u = rint(x fx / z + cx), v = rint(y fy / z + cy) selects the model pixel"""
    Pm = np.moveaxis(v[:3].astype(np.float64), 0, -1); Nm = np.moveaxis(nn[:3].astype(np.float64), 0, -1)
    s = Pm @ np.asarray(Rc, np.float64).T + np.asarray(tc, np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        u = np.rint(s[..., 0] * K[0] / s[..., 2] + K[2]); vv = np.rint(s[..., 1] * K[1] / s[..., 2] + K[3])
    ok = np.isfinite(u) & np.isfinite(vv) & (u >= 0) & (vv >= 0) & (u < W) & (vv < H)
    ui = np.where(ok, u, 0).astype(int); vi = np.where(ok, vv, 0).astype(int)
    dm = Pm[vi, ui]; nm = Nm[vi, ui]; wm = w[vi, ui].astype(np.float64)
    ng = Nm @ np.asarray(Rc, np.float64).T
    ok &= np.isfinite(dm[..., 0]) & np.isfinite(nm[..., 0]) & np.isfinite(Pm[..., 0]) & np.isfinite(Nm[..., 0])
    with np.errstate(invalid="ignore"):
        ok &= (np.linalg.norm(dm - s, axis=-1) <= dist_thr) & (np.linalg.norm(np.cross(ng, nm), axis=-1) <= angle_thr)
    wm = np.where(np.isnan(wm), 0.0, wm)
    J = np.concatenate([nm, np.cross(s, nm)], -1)[ok]; rr = ((s - dm) * nm).sum(-1)[ok]; ww = wm[ok]
    return (J * ww[:, None]).T @ J, (J * ww[:, None]).T @ rr, int(ok.sum())

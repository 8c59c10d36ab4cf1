"""More known answers for the oracle, each an independent numpy re-evaluation written from the cited reference source
(without a reference run, independent evaluations are what the restatement answers to):

  so3Step            reduce.cu:1172-1280   Jacobian rows of the rotation-only photometric alignment, fp64
  initialise         init_unstableTex.vert:51-98, GlobalModel.cpp:214-288   first-frame seeding: order, pose, confidence, colour
  fuse stage 1       data.vert:63-198      which model surfels are matched (thresholds, sampling parity, window order)
  clean, window rule copy_unstable.vert:104-141   free-space violation: a surfel in front of freshly updated stable ones
"""
import ctypes as C

import numpy as np
import pytest

import scenes
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.params import default_params


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


K_ASYM = (129.325, 129.125, 79.65, 63.825)


def test_so3_step_matches_an_fp64_evaluation(oracle_lib_built):
    lib = oracle_lib_built.load()
    W, H = 160, 120
    fx, fy, cx, cy = K_ASYM
    grey = lambda rgb: (0.114 * rgb[..., 0] + 0.299 * rgb[..., 1] + 0.587 * rgb[..., 2]).astype(np.uint8)
    last = np.ascontiguousarray(grey(synth.frame(3, W, H, K=K_ASYM)[0])); nxt = np.ascontiguousarray(grey(synth.frame(5, W, H, K=K_ASYM)[0]))
    a = 0.012
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Km = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    basis = (Km @ R @ np.linalg.inv(Km)).astype(np.float32); kinv = np.linalg.inv(Km).astype(np.float32); krlr = (Km @ R).astype(np.float32)
    A = np.zeros(9); b = np.zeros(3); r = np.zeros(2)
    lib.orc_so3_step(_p(last), _p(nxt), H, W, _p(basis), _p(kinv), _p(krlr), _p(A), _p(b), _p(r))
    # fp64 restatement of SO3Reduction::getProducts
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    B = basis.astype(np.float64)
    wx = B[0, 0] * xs + B[0, 1] * ys + B[0, 2]; wy = B[1, 0] * xs + B[1, 1] * ys + B[1, 2]; wz = B[2, 0] * xs + B[2, 1] * ys + B[2, 2]
    u = np.rint(wx / wz).astype(int); v = np.rint(wy / wz).astype(int)
    ok = (u >= 1) & (u < W - 1) & (v >= 1) & (v < H - 1) & (xs >= 1) & (xs < W - 1) & (ys >= 1) & (ys < H - 1)
    uu = np.clip(u, 1, W - 2); vv = np.clip(v, 1, H - 2)
    xi = np.clip(xs.astype(int), 1, W - 2); yi = np.clip(ys.astype(int), 1, H - 2)

    def grad(img, x, y):
        f = img.astype(np.float64)
        actu = f[y, x]
        gx = (f[y, x - 1] + actu) / 2 - (f[y, x + 1] + actu) / 2
        gy = (f[y - 1, x] + actu) / 2 - (f[y + 1, x] + actu) / 2
        return gx, gy
    gnx, gny = grad(nxt, uu, vv); glx, gly = grad(last, xi, yi)
    gx = (gnx + glx) / 2; gy = (gny + gly) / 2
    Ki = kinv.astype(np.float64)
    px = Ki[0, 0] * xs + Ki[0, 1] * ys + Ki[0, 2]; py = Ki[1, 0] * xs + Ki[1, 1] * ys + Ki[1, 2]; pz = Ki[2, 0] * xs + Ki[2, 1] * ys + Ki[2, 2]
    k = krlr.astype(np.float64)
    z2 = pz * pz
    l0 = ((pz * (k[1, 0] * gy + k[0, 0] * gx)) - gy * k[2, 0] * ys - gx * k[2, 0] * xs) / z2
    l1 = ((pz * (k[1, 1] * gy + k[0, 1] * gx)) - gy * k[2, 1] * ys - gx * k[2, 1] * xs) / z2
    l2 = ((pz * (k[1, 2] * gy + k[0, 2] * gx)) - gy * k[2, 2] * ys - gx * k[2, 2] * xs) / z2
    J = np.stack([l1 * pz - l2 * py, l2 * px - l0 * pz, l0 * py - l1 * px], -1)[ok]
    res = -(nxt[vv, uu].astype(np.float64) - last[yi, xi].astype(np.float64))[ok]
    A_ref = J.T @ J; b_ref = J.T @ res
    assert r[1] == ok.sum() > 0.8 * W * H
    np.testing.assert_allclose(A.reshape(3, 3), A_ref, rtol=2e-4, atol=1e-6 * np.abs(A_ref).max())
    np.testing.assert_allclose(b, b_ref, rtol=2e-4, atol=1e-5 * np.abs(b_ref).max())
    assert abs(r[0] - (res * res).sum()) <= 1e-6 * (res * res).sum()


def _decode(c):
    c = c.astype(np.int64)
    return np.stack([(c >> 16) & 255, (c >> 8) & 255, c & 255], -1)


def test_initialise_seeds_the_map_in_column_major_order(oracle_lib_built):
    """first frame: one surfel per pixel with a normal and valid curvatures, in COLUMN-major pixel order (the draw order
    of the reference's vertex grid), position = init_pose * vertex_raw, confidence = radial confidence at the pixel
    centre, colour packed r<<16|g<<8|b, submap 0, init = last = 1, normal rotated, radius and curvatures copied"""
    W, H = 160, 120
    fx, fy, cx, cy = K_ASYM
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 16)
    o = oracle_lib_built.Oracle(p)
    T = np.eye(4, dtype=np.float32)
    a = 0.3
    T[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    T[:3, 3] = (0.4, -0.2, 0.1)
    rgb, d, _ = synth.frame(2, W, H, noise=True, K=K_ASYM)
    o.set_pose(T); o.process_frame(rgb, d)
    m = o.download_map()
    vr = o.get_image("VERTEX_RAW"); n = o.get_image("NORMAL"); k1 = o.get_image("CURV1"); k2 = o.get_image("CURV2")
    valid = (np.linalg.norm(n[..., :3], axis=-1) > 0.5) & (np.abs(k1[..., 3]) < 300) & (np.abs(k2[..., 3]) < 300)
    order = [(px, py) for px in range(W) for py in range(H) if valid[py, px]]
    assert len(m) == len(order) > 0.5 * W * H
    px = np.array([q[0] for q in order]); py = np.array([q[1] for q in order])
    R = T[:3, :3].astype(np.float64); t = T[:3, 3].astype(np.float64)
    np.testing.assert_allclose(m[:, 0:3], vr[py, px, :3].astype(np.float64) @ R.T + t, atol=2e-6)
    rr = np.hypot(px + 0.5 - cx, py + 0.5 - cy) / np.hypot(W / 2.0, H / 2.0)
    np.testing.assert_allclose(m[:, 3], np.exp(-rr * rr / 0.72), rtol=2e-5)
    assert np.array_equal(_decode(m[:, 4]), rgb[py, px].astype(np.int64))
    assert np.all(m[:, 5] == 0) and np.all(m[:, 6] == 1) and np.all(m[:, 7] == 1)
    np.testing.assert_allclose(m[:, 8:11], n[py, px, :3].astype(np.float64) @ R.T, atol=2e-6)
    assert np.array_equal(m[:, 11], n[py, px, 3])
    assert np.array_equal(m[:, 12:16], k1[py, px], equal_nan=True) and np.array_equal(m[:, 16:20], k2[py, px], equal_nan=True)
    o.close()


def _associate_numpy(o, p, tick):
    """data.vert:63-198 restated: per sampled live pixel the best model surfel among the 9 distinct texels of the 4x4
    half-pixel window, visited x outer / y inner, strict `dist < bestDist`"""
    W, H = o.W, o.H
    fx, fy, cx, cy = p.fx, p.fy, p.cx, p.cy
    idx = o.get_image("INDEX"); vc = o.get_image("INDEX_VERTCONF").astype(np.float64); nr = o.get_image("INDEX_NORMRAD").astype(np.float64)
    dm = o.get_image("DEPTH_METRIC").astype(np.float64); npca = o.get_image("NORMAL_PCA").astype(np.float64)
    k1 = o.get_image("CURV1")[..., 3]; k2 = o.get_image("CURV2")[..., 3]
    best = {}
    flags = {1: 0, 2: 0}
    par = tick % 2
    for px in range(par, W, 2):
        for py in range(par, H, 2):
            z = dm[py, px]; nl = npca[py, px, :3]
            if not (np.linalg.norm(nl) > 0.8 and z > 0.3 and z <= 20.0 and abs(k1[py, px]) < 300 and abs(k2[py, px]) < 300):
                continue
            x, y = px + 0.5, py + 0.5
            xl, yl = (x - cx) / fx, (y - cy) / fy
            lam = np.sqrt(xl * xl + yl * yl + 1)
            ray = np.array([xl, yl, 1.0])
            bd, b, cnt = 1000.0, 0, 0
            for sx in sorted({min(max(px + a, 0), W - 1) for a in (-1, 0, 1)}):
                for sy in sorted({min(max(py + a, 0), H - 1) for a in (-1, 0, 1)}):
                    cur = int(idx[sy, sx])
                    if cur == 0:
                        continue
                    v = vc[sy, sx, :3]
                    if abs(v[2] * lam - z * lam) >= 0.05:
                        continue
                    dist = np.linalg.norm(np.cross(ray, v)) / np.linalg.norm(ray)
                    nn = nr[sy, sx, :3]
                    ok = abs(nn[2]) < 0.75
                    if not ok:
                        c = np.dot(nn, nl) / (np.linalg.norm(nn) * np.linalg.norm(nl))
                        ok = abs(np.arccos(np.clip(c, -1, 1))) < 0.5
                    if dist < bd and ok:
                        cnt += 1; bd = dist; b = cur
            if cnt > 0:
                flags[1] += 1; best.setdefault(b, (px, py))
            else:
                flags[2] += 1
    return best, flags


def test_association_matches_a_numpy_restatement(oracle_lib_built):
    """stage 1 of the fusion on a tracked frame against a small seeded map: the set of model surfels that receive a
    merge (first primitive in draw order wins, so one per surfel) has exactly the size the numpy restatement finds —
    thresholds (0.05 along the ray, |n_z| < 0.75 or angle < 0.5 rad, curvature bounds), sampling parity x%2 == y%2 ==
    time%2 and the window all enter the count"""
    W, H = 96, 72
    K = (79.2, 79.0, 47.6, 38.3)
    seed = synth.seed_map(40_000, width=W, K=K)
    p = default_params(W, H, *K, max_surfels=len(seed) + 20_000)
    o = oracle_lib_built.Oracle(p)
    rgb, d, T = synth.frame(0, W, H, noise=True, K=K)
    o.upload_map(seed); o.set_pose(T); o.bootstrap(rgb, d)
    rgb, d, T1 = synth.frame(1, W, H, noise=True, K=K)
    o.upload_frame(rgb, d)
    for st in ("FILTER_DEPTH", "METRICISE", "VERTEX_NORMAL_RADIUS", "CURVATURE"):
        o.run_stage(st)
    o.set_pose(T1)                      # association at the ground-truth pose of the frame
    o.run_stage("CONFIDENCE"); o.run_stage("PREDICT_INDICES")
    tick = o.tick
    best, flags = _associate_numpy(o, p, tick)
    o.run_stage("FUSE")
    st = o.fuse_stats()
    assert flags[1] > 200 and len(best) > 100
    assert int(st[1]) == len(best), (st, len(best), flags)
    o.close()


def test_free_space_violation_removes_a_surfel_in_front_of_fresh_stable_ones(oracle_lib_built):
    """copy_unstable.vert:126-134: a surfel is dropped when more than 4 of the 16 window samples show a STABLE surfel that
    was updated THIS frame, lies more than 1 cm behind it, and its own normal faces the camera (|n_z| > 0.85).  Scene: a
    stable wall at 2 m re-observed this frame, plus intruder surfels floating at 1.5 m in front of it (never observed:
    the depth image shows the wall) — they must go; the same intruders with a grazing normal (|n_z| < 0.85) must stay."""
    # a power-of-two size: there the fp32 half-pixel walks of data.vert / copy_unstable.vert are exact and visit the nominal texels
    # (at 96 x 72 the association reaches texel p + 1 at fewer pixels — as the executed shaders do, DESIGN.md §8 — fewer surfels are
    # updated per frame and a third of the intruders survive)
    W, H = 128, 64
    f, cx, cy = 79.0, 64.0, 32.0
    # fusion samples a quarter of the pixels per frame (x % 2 == y % 2 == time % 2), so at the default window (4 x 4 half-pixel
    # samples) at most 4 samples can show a surfel updated THIS frame and "zCount > 4" cannot fire; fusionCleanWindowMultiplier
    # = 4 (8 x 8 samples over 5 x 5 texels) is the smallest setting where the rule is live
    p = default_params(W, H, f, f, cx, cy, max_surfels=1 << 16, clean_window_multiplier=4.0)
    z = scenes.plane_depth(W, H, f, f, cx, cy, (0.0, 0.0, 1.0), 2.0)
    rgb, d = scenes.gray_rgb(W, H), scenes.to_u16(z)
    o = oracle_lib_built.Oracle(p)
    for _ in range(24):                     # the wall becomes stable (confidence above 5) and keeps being updated
        o.process_frame(rgb, d)
    m = o.download_map()
    assert (m[:, 3] > 5).sum() > 800, (m[:, 3] > 5).sum()
    n0 = len(m)
    def intruders(nz_facing):
        pts = []
        for px in range(30, 66, 4):
            for py in range(20, 52, 4):
                zz = 1.5
                s = np.zeros(20, np.float32)
                s[0:3] = ((px + 0.5 - cx) * zz / f, (py + 0.5 - cy) * zz / f, zz)
                s[3] = 20.0; s[4] = 0x808080; s[6] = 1; s[7] = 1            # stable, old: only the window rules can remove it
                s[8:11] = (0, 0, 1) if nz_facing else (0.8, 0, 0.6)
                s[11] = 0.01; s[12] = 1; s[17] = 1
                pts.append(s)
        return np.array(pts)
    for facing, expect_removed in ((True, True), (False, False)):
        o2 = oracle_lib_built.Oracle(p)
        both = np.concatenate([m, intruders(facing)])
        o2.upload_map(both); o2.set_tick(o.tick)
        o2.process_frame(rgb, d)
        m2 = o2.download_map()
        left = int(((np.abs(m2[:, 2] - 1.5) < 1e-3) & (m2[:, 3] == 20.0)).sum())
        total = len(intruders(facing))
        if expect_removed:
            assert left < 0.1 * total, (left, total)
        else:
            assert left == total, (left, total)
        assert abs(len(m2) - (n0 + left)) < 0.05 * n0
        o2.close()
    o.close()


def test_duplicate_rule_removes_newer_surfels_sitting_on_older_stable_ones(oracle_lib_built):
    """copy_unstable.vert:116-124: a surfel is dropped when more than 8 of the 16 window samples show a stable surfel that
    is OLDER (init time below its own), lies less than 1 cm behind it and within 1.4 radii in x / y.  A stable wall (one
    surfel behind every pixel centre, uploaded: grown by fusion only the pixels with x % 2 == y % 2 would ever be
    stable) and duplicates 5 mm in front of it: initialised later than the wall -> removed; initialised earlier ->
    kept (the rule protects the first-comer)."""
    W, H = 96, 72
    f, cx, cy = 79.0, 48.0, 36.0
    p = default_params(W, H, f, f, cx, cy, max_surfels=1 << 16)
    z = scenes.plane_depth(W, H, f, f, cx, cy, (0.0, 0.0, 1.0), 2.0)
    rgb, d = scenes.gray_rgb(W, H), scenes.to_u16(z)

    def surfel(px, py, zz, conf, init, radius):
        s = np.zeros(20, np.float32)
        s[0:3] = ((px + 0.5 - cx) * zz / f, (py + 0.5 - cy) * zz / f, zz)
        s[3] = conf; s[4] = 0x808080; s[6] = init; s[7] = 9
        s[8:11] = (0, 0, 1); s[11] = radius; s[12] = 1; s[17] = 1
        return s
    wall = np.array([surfel(px, py, 2.0, 20.0, 5.0, 0.04) for px in range(W) for py in range(H)])
    def dups(init_time):
        return np.array([surfel(px, py, 1.995, 20.0, init_time, 0.08) for px in range(30, 66, 4) for py in range(20, 52, 4)])
    for init_time, expect_removed in ((9.0, True), (2.0, False)):
        o = oracle_lib_built.Oracle(p)
        o.upload_map(np.concatenate([wall, dups(init_time)]))
        o.upload_frame(rgb, d)
        for st in ("FILTER_DEPTH", "METRICISE", "VERTEX_NORMAL_RADIUS", "CURVATURE", "CONFIDENCE"):
            o.run_stage(st)
        o.set_tick(10)
        o.run_stage("PREDICT_INDICES"); o.run_stage("CLEAN")          # the clean pass alone: no merge moves anything first
        m2 = o.download_map()
        left = int(((np.abs(m2[:, 2] - 1.995) < 1e-4) & (m2[:, 6] == init_time)).sum())
        total = len(dups(init_time))
        assert (left == 0) if expect_removed else (left == total), (init_time, left, total)
        assert len(m2) == len(wall) + left
        o.close()


def test_bilateral_filter_on_a_step_edge_matches_fp64(oracle_lib_built):
    """depth_bilateral.frag:16-67 on a 40 mm step: sum over the 13 x 13 window of d * w / sum w with
    w = exp(-(dx^2 + dy^2) * 0.024691358 - (d - d0)^2 * 0.000555556), raw units (mm at factor 5000 -> value / 5)"""
    W, H = 96, 72
    p = default_params(W, H, 79.0, 79.0, 48.0, 36.0, max_surfels=1 << 12)
    o = oracle_lib_built.Oracle(p)
    rng = np.random.default_rng(1)
    d = np.full((H, W), 5000, np.uint16); d[:, W // 2:] = 5200            # 1.00 m | 1.04 m
    d = (d + rng.integers(-30, 31, d.shape)).astype(np.uint16)              # +- 6 mm of noise
    o.upload_frame(scenes.gray_rgb(W, H), d); o.run_stage("FILTER_DEPTH")
    got = o.get_image("DEPTH_FILTERED").astype(np.float64)
    adj = 1.0 / ((1.0 / 5000.0) * 1000.0)                                     # depthFactor_adjustment
    v = d.astype(np.float64) / adj
    exp = np.zeros((H, W))
    for y in range(8, H - 8):
        for x in range(8, W - 8):
            win = v[y - 6:y + 7, x - 6:x + 7]
            dy, dx = np.mgrid[-6:7, -6:7]
            w = np.exp(-((dx * dx + dy * dy) * 0.024691358 + (win - v[y, x]) ** 2 * 0.000555556))
            exp[y, x] = (win * w).sum() / w.sum() * adj
    inner = (slice(8, H - 8), slice(8, W - 8))
    np.testing.assert_allclose(got[inner], exp[inner], rtol=2e-6)
    # the range term works: pixels next to the step stay on their own side of the midpoint (sigma_color = 30 mm, step 40 mm)
    assert got[30, W // 2 - 1] < 5085 and got[30, W // 2] > 5115
    o.close()


def test_pca_normal_and_radius_on_a_slanted_plane(oracle_lib_built):
    """getNormalPCA (geometry.glsl:190-244) returns the plane's normal (oriented towards +z); getRadius
    (surfels.glsl:19-32) = min(2 r, r / |n_z|) with r = sqrt(2) z / mean focal, times the initial multiplier"""
    W, H = 160, 120
    fx, fy, cx, cy = K_ASYM
    p = default_params(W, H, fx, fy, cx, cy, max_surfels=1 << 12)
    o = oracle_lib_built.Oracle(p)
    nrm = np.array([0.35, -0.2, 1.0]); nrm /= np.linalg.norm(nrm)
    z = scenes.plane_depth(W, H, fx, fy, cx, cy, nrm, 1.6)
    # plane_depth renders through integer pixel coordinates; the PCA window back-projects through pixel CENTRES, so
    # sample the plane there: z(x + .5, y + .5)
    r = scenes.pixel_rays(W, H, fx, fy, cx, cy, half=0.5)
    z = 1.6 / (r @ nrm)
    o.upload_frame(scenes.gray_rgb(W, H), np.clip(np.rint(z * 5000), 0, 65535).astype(np.uint16))
    for st in ("FILTER_DEPTH", "METRICISE", "VERTEX_NORMAL_RADIUS"):
        o.run_stage(st)
    n = o.get_image("NORMAL_PCA"); vf = o.get_image("VERTEX_FILTERED")
    inner = (slice(10, H - 10), slice(10, W - 10))
    cosang = (n[..., :3][inner] @ nrm)
    assert cosang.min() > 0.999            # 0.2 mm depth quantisation over a 7 x 7 window at 1.6 m
    zz = vf[..., 2][inner].astype(np.float64)
    rr = np.sqrt(2.0) * zz / (0.5 * (fx + fy))
    expect = p.init_radius_multiplier * np.minimum(2 * rr, rr / np.abs(n[..., 2][inner]))
    np.testing.assert_allclose(n[..., 3][inner], expect, rtol=1e-4)
    o.close()


def test_merge_formula_of_update_vert_both_branches(oracle_lib_built):
    """update.vert:51-115 evaluated in fp64 per merged surfel: confidence-weighted averages of position, colour (decoded,
    averaged, re-encoded with round()), normal (normalised after averaging), radius and both curvature vectors when the
    live radius is below 1.5 x the model radius; otherwise only confidence and last-seen time change.  A wall with one
    surfel behind every pixel centre, half of them with a radius small enough to take the second branch."""
    W, H = 96, 72
    f, cx, cy = 79.0, 48.0, 36.0
    p = default_params(W, H, f, f, cx, cy, max_surfels=1 << 16)
    r = scenes.pixel_rays(W, H, f, f, cx, cy)
    nrm = np.array([0.1, -0.05, 1.0]); nrm /= np.linalg.norm(nrm)
    z = 2.0 / (r @ nrm)
    rgb = scenes.gray_rgb(W, H, seed=3); d = scenes.to_u16(z)
    o = oracle_lib_built.Oracle(p)
    o.upload_frame(rgb, d)
    for st in ("FILTER_DEPTH", "METRICISE", "VERTEX_NORMAL_RADIUS", "CURVATURE", "CONFIDENCE"):
        o.run_stage(st)
    dm = o.get_image("DEPTH_METRIC").astype(np.float64); npca = o.get_image("NORMAL_PCA").astype(np.float64)
    conf = o.get_image("CONFIDENCE").astype(np.float64); k1 = o.get_image("CURV1").astype(np.float64); k2 = o.get_image("CURV2").astype(np.float64)
    rng = np.random.default_rng(8)
    surf = np.zeros((W * H, 20), np.float32)
    ids = np.arange(W * H).reshape(W, H).T                       # id of the surfel behind pixel (px, py): column-major upload
    k = 0
    for px in range(W):
        for py in range(H):
            zz = dm[py, px] if dm[py, px] > 0 else 2.0
            s = surf[k]; k += 1
            s[0:3] = ((px + 0.5 - cx) * zz / f, (py + 0.5 - cy) * zz / f, zz + 0.002)       # 2 mm behind the measurement
            s[3] = rng.uniform(6, 12)
            col = rng.integers(10, 240, 3)
            s[4] = (int(col[0]) << 16) + (int(col[1]) << 8) + int(col[2]); s[5] = 0; s[6] = 1; s[7] = 3
            nn = nrm + rng.normal(0, 0.02, 3); nn /= np.linalg.norm(nn)
            s[8:11] = nn
            s[11] = 0.2 if (px // 2 + py // 2) % 2 == 0 else 0.005              # large radius: average; tiny radius: second branch
            s[12:15] = (1, 0, 0); s[15] = rng.uniform(-5, 5); s[16:19] = (0, 1, 0); s[19] = rng.uniform(-5, 5)
    o.upload_map(surf); o.set_tick(7)
    o.run_stage("PREDICT_INDICES"); o.run_stage("FUSE")
    m = o.download_map()
    assert len(m) == len(surf)
    merged = np.nonzero(m[:, 7] == 7.0)[0]
    assert len(merged) > 0.15 * W * H
    n_avg = n_keep = 0
    for sid in merged[::7]:
        px, py = sid // H, sid % H
        assert px % 2 == 1 and py % 2 == 1                       # time 7: only odd / odd pixels are sampled (data.vert:103)
        old = surf[sid].astype(np.float64)
        a = conf[py, px]; c = old[3]
        zz = dm[py, px]
        vg = np.array([(px + 0.5 - cx) * zz / f, (py + 0.5 - cy) * zz / f, zz])
        new_rad = npca[py, px, 3]
        got = m[sid].astype(np.float64)
        assert abs(got[3] - (c + a)) < 1e-5 and got[5] == 0 and got[6] == 1
        if new_rad < 1.5 * old[11]:
            n_avg += 1
            np.testing.assert_allclose(got[0:3], (c * old[0:3] + a * vg) / (c + a), atol=2e-6)
            oc = np.array([(int(old[4]) >> 16) & 255, (int(old[4]) >> 8) & 255, int(old[4]) & 255]) / 255.0
            nc = rgb[py, px].astype(np.float64) / 255.0
            avg = (c * oc + a * nc) / (c + a)
            enc = (int(round(avg[0] * 255)) << 16) + (int(round(avg[1] * 255)) << 8) + int(round(avg[2] * 255))
            assert abs(got[4] - enc) <= 0x010101 and got[4] == float(int(got[4]))      # each channel within one rounding step
            nn = (c * old[8:11] + a * npca[py, px, :3]) / (c + a)
            # the record's normal is data.vert's own recomputation (data.vert:83-96, orc_record_normal): the image's where the
            # vertex attribute's texcoord equals the fragment shader's, a PCA over a window an ulp elsewhere at the other columns
            np.testing.assert_allclose(got[8:11], nn / np.linalg.norm(nn), atol=2e-4)
            assert abs(got[11] - (c * old[11] + a * new_rad) / (c + a)) < 1e-5
            np.testing.assert_allclose(got[12:16], (c * old[12:16] + a * k1[py, px]) / (c + a), atol=1e-5)
            np.testing.assert_allclose(got[16:20], (c * old[16:20] + a * k2[py, px]) / (c + a), atol=1e-5)
        else:
            n_keep += 1
            assert np.array_equal(got[0:3], old[0:3]) and got[4] == old[4] and np.array_equal(got[8:20], old[8:20])
    assert n_avg > 20 and n_keep > 20
    o.close()


def test_fill_in_selects_and_icp_weight(oracle_lib_built):
    """fill_vertex.frag:43-72, fill_normal.frag:36-49, fill_curvature.frag:35-51, fill_rgb.frag:29-37: where the prediction
    is empty the live frame fills in — vertex only if the live curvatures are valid, with the ICP weight
    (1 / z^2) (conf / 256 + exp(-lambda^2 / (2 kmax^2))) — elsewhere the prediction passes through unchanged"""
    W, H = 96, 72
    p = default_params(W, H, 79.0, 79.0, 48.0, 36.0, max_surfels=1 << 12)
    o = oracle_lib_built.Oracle(p)
    rng = np.random.default_rng(4)
    f4 = lambda: rng.uniform(0.2, 2.0, (H, W, 4)).astype(np.float32)
    pv, pn, pc1, pc2 = f4(), f4(), f4(), f4()
    pn[..., :3] /= np.linalg.norm(pn[..., :3], axis=-1, keepdims=True)
    hole = np.zeros((H, W), bool); hole[10:40, 20:70] = True
    pv[hole] = 0; pn[hole] = 0
    pc1[..., 3] = rng.uniform(-50, 50, (H, W)); pc2[..., 3] = rng.uniform(-50, 50, (H, W))
    pc1[hole, 3] = 1000.0; pc2[hole, 3] = 1000.0                       # the prediction writes 1000 where it found nothing
    pw = rng.uniform(0.1, 1.0, (H, W)).astype(np.float32)
    pimg = rng.integers(1, 255, (H, W, 4)).astype(np.uint8); pimg[hole] = 0
    lv, ln, lc1, lc2 = f4(), f4(), f4(), f4()
    lc1[..., 3] = rng.uniform(-50, 50, (H, W)); lc2[..., 3] = rng.uniform(-50, 50, (H, W))
    bad = np.zeros((H, W), bool); bad[15:20, 30:40] = True             # live curvature invalid: nothing to fill in with
    lc1[bad, 3] = 400.0
    lconf = rng.uniform(0.1, 1.0, (H, W)).astype(np.float32)
    rgb = rng.integers(1, 255, (H, W, 3)).astype(np.uint8)
    o.upload_frame(rgb, np.full((H, W), 5000, np.uint16))
    for name, a in (("PRED_VERTEX", pv), ("PRED_NORMAL", pn), ("PRED_CURV1", pc1), ("PRED_CURV2", pc2), ("PRED_ICPWEIGHT", pw),
                    ("PRED_IMAGE", pimg), ("VERTEX_FILTERED", lv), ("NORMAL", ln), ("CURV1", lc1), ("CURV2", lc2), ("CONFIDENCE", lconf)):
        o.set_image(name, a)
    o.run_stage("FILLIN")
    fv, fn, fc1, fc2, fw, fi = (o.get_image(n) for n in ("FILL_VERTEX", "FILL_NORMAL", "FILL_CURV1", "FILL_CURV2", "FILL_ICPWEIGHT", "FILL_IMAGE"))
    keep = ~hole
    assert np.array_equal(fv[keep], pv[keep]) and np.array_equal(fw[keep], pw[keep]) and np.array_equal(fn[keep], pn[keep])
    assert np.array_equal(fc1[keep], pc1[keep]) and np.array_equal(fc2[keep], pc2[keep]) and np.array_equal(fi[keep][:, :3], pimg[keep][:, :3])
    fill = hole & ~bad
    assert np.array_equal(fv[fill][:, :3], lv[fill][:, :3]) and np.array_equal(fv[fill][:, 3], lconf[fill])
    kmax = np.maximum(np.abs(lc1[..., 3]), np.abs(lc2[..., 3])).astype(np.float64)
    w = (1.0 / lv[..., 2].astype(np.float64) ** 2) * (lconf / 256.0 + np.exp(-0.5 * 100.0 / (kmax * kmax)))
    np.testing.assert_allclose(fw[fill], w[fill], rtol=2e-5, atol=1e-30)
    assert np.all(fv[hole & bad] == 0) and np.all(fw[hole & bad] == 0)
    assert np.array_equal(fn[hole], ln[hole]) and np.array_equal(fc1[hole], lc1[hole]) and np.array_equal(fc2[hole], lc2[hole])
    assert np.array_equal(fi[hole][:, :3], rgb[hole])
    o.close()


def test_uv_attribute_is_the_hosts_float_double_float_formula(oracle_lib_built):
    """GlobalModel.cpp:88-97: `((float)i / (float)width) + 1.0 / (2 * (float)width)` stored as float — a float quotient, a double
    sum, a second rounding.  It equals the fragment shaders' (i + 0.5) / width at power-of-two sizes and differs by an ulp at 171
    of 640 columns / 139 of 480 rows; texcoord * width is then not exactly i + 0.5 at 103 / 117 of them"""
    import ctypes as C
    lib = oracle_lib_built.load()
    lib.orc_uv_attribute.restype = C.c_float; lib.orc_uv_attribute.argtypes = [C.c_int, C.c_int]
    lib.orc_uv_fragment.restype = C.c_float; lib.orc_uv_fragment.argtypes = [C.c_int, C.c_int]
    f = np.float32
    expect = {640: (171, 103), 480: (139, 117), 160: (43, 26), 120: (19, 14), 256: (0, 0), 128: (0, 0)}
    for n, (n_t, n_x) in expect.items():
        i = np.arange(n)
        ta = np.array([lib.orc_uv_attribute(int(k), n) for k in i], f)
        tf = np.array([lib.orc_uv_fragment(int(k), n) for k in i], f)
        ref_a = ((i.astype(f) / f(n)).astype(np.float64) + 1.0 / float(2 * f(n))).astype(f)   # float() : a DOUBLE reciprocal, as in the C++
        assert np.array_equal(ta, ref_a) and np.array_equal(tf, ((i.astype(f) + f(0.5)) / f(n)).astype(f))
        assert int((ta != tf).sum()) == n_t and int(((ta * f(n)).astype(f) != i + 0.5).sum()) == n_x, n

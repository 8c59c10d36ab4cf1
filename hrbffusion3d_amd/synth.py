"""Deterministic synthetic RGB-D stream + pre-seeded surfel map.

Scene: 6 x 3 x 4 m box room (x in [-3,3], y in [-1.5,1.5], z in [-2,2]) + sphere (R = 0.5 m) +
sinusoidal relief (amp 3 cm, wavelength 40 cm) on the +z wall.  Pinhole camera K = (528,528,320,240)
at 640x480 (scaled for other sizes) on a Lissajous path, ~8 mm and ~0.4 deg per frame.  Depth =
analytic ray cast -> uint16 at 5000 units/m, optional Kinect-style noise and 3 % drop-outs from a
seeded generator (seed 12345).  RGB = 3-D checker + hashed value noise, uint8.

This module only GENERATES INPUTS (numpy); it is not part of the hot path.
"""
import numpy as np

SEED = 12345
HALF = np.array([3.0, 1.5, 2.0])
SPH_C = np.array([0.9, 0.9, 1.0])
SPH_R = 0.5
RELIEF_AMP = 0.03
RELIEF_LAMBDA = 0.40


# Intrinsics of BASELINE configs 2 / 3 (dataset documentation): fx != fy and an off-centre principal
# point.  ICL-NUIM publishes fy = -480 (image y axis pointing up); most pipelines run it with +480.
TUM_FR1 = (517.3, 516.5, 318.6, 255.3)
ICL_NUIM = (481.2, 480.0, 319.5, 239.5)
ICL_NUIM_NEG = (481.2, -480.0, 319.5, 239.5)
KINECT2_512x424 = (366.1, 364.7, 258.3, 203.9)     # a non-4:3 sensor (512 x 424)


def intrinsics(width=640, height=480, K=None):
    """(fx, fy, cx, cy).  K = None: the synthetic default (528, 528, 320, 240) at 640 px width, scaled with the width.
    K given: taken as is, in pixels of THIS resolution (no scaling)."""
    if K is None:
        s = width / 640.0
        return 528.0 * s, 528.0 * s, 320.0 * s, 240.0 * s
    return tuple(float(v) for v in K)


def _rot_yx(yaw, pitch):
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return Ry @ Rx


def camera_pose(k):
    """T_wc (4x4 float64) of frame k: Lissajous, ~8 mm / ~0.4 deg per frame."""
    t = float(k)
    p = np.array([0.45 * np.sin(0.0125 * t), 0.20 * np.sin(0.0170 * t + 0.5), -0.6 + 0.30 * np.sin(0.0090 * t + 1.0)])
    yaw = 0.35 * np.sin(0.0140 * t)
    pitch = 0.12 * np.sin(0.0110 * t + 0.3)
    T = np.eye(4)
    T[:3, :3] = _rot_yx(yaw, pitch)
    T[:3, 3] = p
    return T


def _hash01(ix, iy, iz, salt):
    h = (ix.astype(np.int64) * 73856093) ^ (iy.astype(np.int64) * 19349663) ^ (iz.astype(np.int64) * 83492791) ^ salt
    h = (h ^ (h >> 13)) * 1274126177
    h = h ^ (h >> 16)
    return (h & 0xFFFF).astype(np.float64) / 65535.0


def texture(pw):
    """RGB uint8 of world points pw (...,3): 25 cm checker + 5 cm value noise."""
    c = np.floor(pw / 0.25).astype(np.int64)
    par = ((c[..., 0] + c[..., 1] + c[..., 2]) & 1).astype(np.float64)
    f = np.floor(pw / 0.05).astype(np.int64)
    n0 = _hash01(f[..., 0], f[..., 1], f[..., 2], 11)
    n1 = _hash01(f[..., 0], f[..., 1], f[..., 2], 23)
    n2 = _hash01(f[..., 0], f[..., 1], f[..., 2], 37)
    base = 70.0 + 110.0 * par
    rgb = np.stack([base + 50.0 * n0, base * 0.9 + 60.0 * n1, 60.0 + 0.8 * base + 40.0 * n2], axis=-1)
    return np.clip(rgb, 1, 254).astype(np.uint8)


def _relief(x, y):
    return RELIEF_AMP * np.sin(2 * np.pi * x / RELIEF_LAMBDA) * np.sin(2 * np.pi * y / RELIEF_LAMBDA)


def raycast(T_wc, width=640, height=480, K=None):
    """Returns (t, hit point world, z-depth) for every pixel centre-of-texel ray through (x, y) integer coords.
    K = (fx, fy, cx, cy) in pixels of this resolution (None: the scaled synthetic default)."""
    fx, fy, cx, cy = intrinsics(width, height, K)
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    d_c = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1)
    R, o = T_wc[:3, :3], T_wc[:3, 3]
    d = d_c @ R.T
    # box exit distance (camera inside the box)
    with np.errstate(divide="ignore", invalid="ignore"):
        tpos = (HALF - o) / d
        tneg = (-HALF - o) / d
    tb = np.where(d > 0, tpos, np.where(d < 0, tneg, np.inf))
    t = tb.min(axis=-1)
    axis = tb.argmin(axis=-1)
    # relief on the +z wall: two fixed-point refinements
    onz = (axis == 2) & (d[..., 2] > 0)
    for _ in range(3):
        ph = o + d * t[..., None]
        tz = (HALF[2] - _relief(ph[..., 0], ph[..., 1]) - o[2]) / d[..., 2]
        t = np.where(onz, tz, t)
    # sphere
    oc = o - SPH_C
    b = (d * oc).sum(-1)
    a = (d * d).sum(-1)
    cc = (oc * oc).sum() - SPH_R ** 2
    disc = b * b - a * cc
    ts = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / a, np.inf)
    ts = np.where(ts > 1e-3, ts, np.inf)
    t = np.minimum(t, ts)
    ph = o + d * t[..., None]
    return t, ph, t  # d_c.z == 1 -> z-depth equals t


def frame(k, width=640, height=480, noise=False, depth_units=5000.0, K=None):
    """(rgb uint8 HxWx3, depth uint16 HxW, T_wc float32 4x4) of frame k."""
    T = camera_pose(k)
    _, ph, z = raycast(T, width, height, K)
    rgb = texture(ph)
    zz = z.copy()
    if noise:
        rng = np.random.default_rng(SEED + 7919 * (k + 1))
        sigma = 0.0012 + 0.0019 * (zz - 0.4) ** 2
        zz = zz + sigma * rng.standard_normal(zz.shape)
        drop = rng.random(zz.shape) < 0.03
        zz = np.where(drop, 0.0, zz)
    depth = np.clip(np.rint(zz * depth_units), 0, 65535).astype(np.uint16)
    return rgb, depth, T.astype(np.float32)


def _surface_samples(spacing, rng):
    """Jittered grid on the 6 box faces + the sphere: points, outward-of-room normals, curvatures."""
    pts, nrm, k1, k2 = [], [], [], []
    for ax in range(3):
        o1, o2 = [a for a in range(3) if a != ax]
        n1 = int(np.round(2 * HALF[o1] / spacing)); n2 = int(np.round(2 * HALF[o2] / spacing))
        g1 = (np.arange(n1) + 0.5) * (2 * HALF[o1] / n1) - HALF[o1]
        g2 = (np.arange(n2) + 0.5) * (2 * HALF[o2] / n2) - HALF[o2]
        a, b = np.meshgrid(g1, g2, indexing="ij")
        for sgn in (-1.0, 1.0):
            ja = a + (rng.random(a.shape) - 0.5) * spacing * 0.6
            jb = b + (rng.random(b.shape) - 0.5) * spacing * 0.6
            p = np.zeros(a.shape + (3,))
            p[..., o1] = ja; p[..., o2] = jb; p[..., ax] = sgn * HALF[ax]
            n = np.zeros_like(p); n[..., ax] = sgn
            if ax == 2 and sgn > 0:
                p[..., 2] -= _relief(p[..., 0], p[..., 1])
            pts.append(p.reshape(-1, 3)); nrm.append(n.reshape(-1, 3))
            k1.append(np.zeros(p.shape[:-1]).ravel()); k2.append(np.zeros(p.shape[:-1]).ravel())
    # sphere (normals point towards the centre = away from an outside viewer)
    ns = int(4 * np.pi * SPH_R ** 2 / spacing ** 2)
    i = np.arange(ns) + 0.5
    phi = np.arccos(1 - 2 * i / ns); th = np.pi * (1 + 5 ** 0.5) * i
    dirs = np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], -1)
    pts.append(SPH_C + SPH_R * dirs); nrm.append(-dirs)
    k1.append(np.full(ns, -1.0 / SPH_R)); k2.append(np.full(ns, -1.0 / SPH_R))
    return np.concatenate(pts), np.concatenate(nrm), np.concatenate(k1), np.concatenate(k2)


def seed_map(n_target=1_000_000, t_now=1, width=640, K=None):
    """AoS surfel array (N,20) float32 in the reference layout, N ~ n_target (>= n_target)."""
    rng = np.random.default_rng(SEED)
    area = 2 * (4 * HALF[0] * HALF[1] + 4 * HALF[0] * HALF[2] + 4 * HALF[1] * HALF[2]) + 4 * np.pi * SPH_R ** 2
    spacing = float(np.sqrt(area / (n_target * 1.01)))
    p, n, k1, k2 = _surface_samples(spacing, rng)
    N = p.shape[0]
    fx = intrinsics(width, K=K)[0]
    m = np.zeros((N, 20), np.float32)
    m[:, 0:3] = p
    m[:, 3] = rng.uniform(5.0, 20.0, N)
    col = texture(p).astype(np.int64)
    m[:, 4] = ((col[:, 0] << 16) + (col[:, 1] << 8) + col[:, 2]).astype(np.float32)
    m[:, 5] = 0.0
    m[:, 6] = 1.0
    m[:, 7] = float(t_now)
    m[:, 8:11] = n
    dist = np.linalg.norm(p - np.array([0, 0, -0.6]), axis=1)
    m[:, 11] = np.maximum(4.0 * np.sqrt(2.0) * dist / fx, 1.6 * spacing)
    # principal directions: any orthonormal tangent pair
    a = np.where(np.abs(n[:, [0]]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    t1 = np.cross(n, a); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(n, t1)
    m[:, 12:15] = t1; m[:, 15] = k1
    m[:, 16:19] = t2; m[:, 19] = k2
    return m


def ate_rmse(poses_est, poses_gt):
    """Absolute trajectory error (m), translation RMSE after aligning the first pose."""
    e = np.asarray([p[:3, 3] for p in poses_est], np.float64)
    g = np.asarray([p[:3, 3] for p in poses_gt], np.float64)
    return float(np.sqrt(((e - g) ** 2).sum(1).mean()))

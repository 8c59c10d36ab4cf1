"""Small analytic scenes for the known-answer tests (numpy only)."""
import numpy as np


def pixel_rays(W, H, fx, fy, cx, cy, half=0.0):
    u, v = np.meshgrid(np.arange(W, dtype=np.float64) + half, np.arange(H, dtype=np.float64) + half)
    return np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)


def plane_depth(W, H, fx, fy, cx, cy, n, d):
    """z-depth image of the plane n.x = d seen from the origin (rays through integer pixel coords)."""
    r = pixel_rays(W, H, fx, fy, cx, cy)
    den = r @ np.asarray(n, np.float64)
    t = np.where(np.abs(den) > 1e-9, d / den, 0.0)
    return np.where(t > 0, t, 0.0)


def sphere_depth(W, H, fx, fy, cx, cy, c, R):
    r = pixel_rays(W, H, fx, fy, cx, cy)
    c = np.asarray(c, np.float64)
    a = (r * r).sum(-1); b = r @ c; cc = c @ c - R * R
    disc = b * b - a * cc
    t = np.where(disc > 0, (b - np.sqrt(np.maximum(disc, 0))) / a, 0.0)
    return np.where(t > 0, t, 0.0)


def corner_depth(W, H, fx, fy, cx, cy):
    """three mutually orthogonal planes meeting in front of the camera (a room corner)"""
    r = pixel_rays(W, H, fx, fy, cx, cy)
    A = np.array([[1.0, 1.0, 1.0], [-1.0, 1.0, 1.0], [0.0, -1.0, 1.0]])
    A /= np.linalg.norm(A, axis=1, keepdims=True)
    q, _ = np.linalg.qr(A.T)
    ns = q.T
    ns = np.array([n if n[2] > 0 else -n for n in ns])
    apex = np.array([0.0, 0.0, 2.2])
    ts = []
    for n in ns:
        den = r @ n
        t = np.where(den > 1e-6, (apex @ n) / den, np.inf)
        ts.append(t)
    # planes seen from inside the corner: the nearest of the three far intersections
    t = np.min(np.stack(ts), axis=0)
    t = np.where(np.isfinite(t), t, 0.0)
    return t


def to_u16(z, units=5000.0):
    return np.clip(np.rint(z * units), 0, 65535).astype(np.uint16)


def gray_rgb(W, H, seed=0):
    rng = np.random.default_rng(seed)
    base = rng.integers(40, 200, (H // 8 + 1, W // 8 + 1, 3))
    img = np.kron(base, np.ones((8, 8, 1)))[:H, :W]
    return np.clip(img, 1, 254).astype(np.uint8)

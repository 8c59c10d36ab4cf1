"""Analytic RGB-D scenes with a SMOOTH, band-limited, sub-pixel-accurate texture for the registration's metamorphic tests
(tests/test_registration_metamorphic.py).  numpy only, float64, nothing from the oracle or the library.

The texture is a function of the 3-D surface point (a sum of a few sinusoids, longest wavelengths only), sampled exactly at the
point every pixel's ray hits — no texel grid anywhere, so two views of the scene are consistent to the uint8 rounding and the
image gradient is the analytic one.  Geometry: the inside of a box room (nearest positive hit of six planes), or one tilted plane.
"""
import numpy as np


def rot(rx, ry, rz):
    """R = Rz Ry Rx (radians)"""
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def pose(rx=0.0, ry=0.0, rz=0.0, t=(0.0, 0.0, 0.0)):
    T = np.eye(4)
    T[:3, :3] = rot(rx, ry, rz)
    T[:3, 3] = t
    return T


# wavelengths in metres: >= 16 px at 320x240 / fx 264 and 1.5 m (1 px = 5.7 mm), >= 8 px one level up
_WAVES = [((1.0, 0.35, 0.2), 0.31, 0.3), ((-0.3, 1.0, 0.45), 0.37, 1.1), ((0.25, -0.4, 1.0), 0.43, 2.3), ((0.8, 0.7, -0.5), 0.53, 0.7),
          ((-0.6, 0.5, 0.8), 0.29, 4.0)]


def texture(p, contrast=1.0, wavelength=1.0):
    """grey value in (20, 236) at the 3-D points p (..., 3); `wavelength` scales every wave (1.0: 0.29-0.53 m)"""
    g = np.zeros(p.shape[:-1])
    for k, lam, ph in _WAVES:
        k = np.asarray(k, np.float64); k /= np.linalg.norm(k)
        g += np.sin(2.0 * np.pi * (p @ k) / (lam * wavelength) + ph)
    return 128.0 + contrast * (108.0 / len(_WAVES)) * g


ROOM = [((1.0, 0, 0), 1.3), ((-1.0, 0, 0), 2.9), ((0, 1.0, 0), 1.0), ((0, -1.0, 0), 1.9), ((0, 0, 1.0), 1.7), ((0, 0, -1.0), 3.0)]   # n.x = d
PLANE = [((0.28, -0.17, 1.0), 1.45)]


def render(T_wc, W, H, K, scene=ROOM, units=5000.0, contrast=1.0, half=0.0, wavelength=1.0):
    """(rgb uint8 HxWx3, depth uint16 HxW, z float64, points_world) of the scene seen from camera-to-world pose T_wc"""
    fx, fy, cx, cy = K
    u, v = np.meshgrid(np.arange(W, dtype=np.float64) + half, np.arange(H, dtype=np.float64) + half)
    ray_c = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)
    R, t = T_wc[:3, :3], T_wc[:3, 3]
    ray_w = ray_c @ R.T
    best = np.full((H, W), np.inf)
    for n, d in scene:
        n = np.asarray(n, np.float64); nn = np.linalg.norm(n); n = n / nn; d = d / nn
        den = ray_w @ n
        with np.errstate(divide="ignore", invalid="ignore"):
            s = np.where(den > 1e-9, (d - t @ n) / den, np.inf)     # the side the ray LEAVES through: inside of a convex room
        s = np.where(s > 0, s, np.inf)
        best = np.minimum(best, s)
    z = np.where(np.isfinite(best), best, 0.0)                       # ray_c has z = 1: the ray parameter is the camera-frame depth
    pw = t + ray_w * z[..., None]
    g = np.clip(np.rint(texture(pw, contrast, wavelength)), 1, 254).astype(np.uint8)
    rgb = np.repeat(g[..., None], 3, -1)
    rgb[z <= 0] = 0
    depth = np.clip(np.rint(z * units), 0, 65535).astype(np.uint16)
    return np.ascontiguousarray(rgb), depth, z, pw


def reprojection_px(E, G, z, K, stride=4):
    """mean pixel distance between where the estimated and the true relative pose (frame B -> frame A) put frame B's points in A"""
    fx, fy, cx, cy = K
    H, W = z.shape
    u, v = np.meshgrid(np.arange(0, W, stride, dtype=np.float64), np.arange(0, H, stride, dtype=np.float64))
    zz = z[::stride, ::stride]
    ok = zz > 0
    p = np.stack([(u - cx) / fx * zz, (v - cy) / fy * zz, zz, np.ones_like(zz)], -1)[ok]
    def proj(T):
        q = p @ np.asarray(T, np.float64).T
        return np.stack([fx * q[:, 0] / q[:, 2] + cx, fy * q[:, 1] / q[:, 2] + cy], -1)
    d = proj(E) - proj(G)
    return float(np.sqrt((d ** 2).sum(-1)).mean())


def rot_angle_deg(R):
    return float(np.degrees(np.arccos(np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0))))

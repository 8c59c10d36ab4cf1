"""Data formats either side of the hot path — host side, numpy only.

  save_ply              binary PLY, 13 properties, normals negated   Core/src/HRBFFusion.cpp:1737-1853
  save_trajectory       TUM / ICL-NUIM variant / zhou .log / lefloch   Core/src/Utils/TrajectoryManager.cpp:284-373
  load_trajectory_tum   inverse of the TUM writer (for ATE)           Core/src/Utils/TrajectoryManager.cpp:27-120
  load_associations     `ts depth_path ts rgb_path` per line           Core/src/HRBFFusion.cpp:212-238
  ate_rmse              Horn-aligned absolute trajectory error (the reference ships no evaluation script)
"""
import numpy as np


def rotation_to_quaternion(R):
    """x, y, z, w with Eigen::Quaternionf(Matrix3f) branch order (w >= 0 on the trace branch)."""
    R = np.asarray(R, np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0); w = 0.5 * s; s = 0.5 / s
        return np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4); q[i] = 0.5 * s; s = 0.5 / s
    q[3] = (R[k, j] - R[j, k]) * s; q[j] = (R[j, i] + R[i, j]) * s; q[k] = (R[k, i] + R[i, k]) * s
    return q


def quaternion_to_rotation(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def save_trajectory(path, poses, stamps_us=None, fmt="TUM", icl_nuim=False):
    poses = [np.asarray(p, np.float64) for p in poses]
    if stamps_us is None:
        stamps_us = list(range(len(poses)))
    with open(path, "w") as f:
        for i, T in enumerate(poses):
            if fmt == "TUM":
                t = T[:3, 3].copy()
                if icl_nuim:
                    head = "%d " % int(stamps_us[i]); t[1] = -t[1]
                else:
                    head = "%.6f " % (stamps_us[i] / 1000000.0)
                q = rotation_to_quaternion(T[:3, :3])
                f.write(head + "%g %g %g %g %g %g %g\n" % (t[0], t[1], t[2], q[0], q[1], q[2], q[3]))
            elif fmt == "zhou":
                f.write("%d %d %d\n" % (i, i, i + 1))
                for r in range(4):
                    f.write("%f %f %f %f\n" % tuple(T[r]))
            elif fmt == "lefloch":
                f.write("%d " % i + " ".join("%g" % v for v in T.T.ravel()) + " \n")
            else:
                raise ValueError(fmt)


def load_trajectory_tum(path):
    stamps, poses = [], []
    for line in open(path):
        v = line.split()
        if len(v) != 8 or line.startswith("#"):
            continue
        T = np.eye(4); T[:3, 3] = [float(x) for x in v[1:4]]
        T[:3, :3] = quaternion_to_rotation([float(x) for x in v[4:8]])
        stamps.append(float(v[0])); poses.append(T)
    return np.array(stamps), poses


def reference_trajectory_stamp(token):
    """the integer TrajectoryManager::LoadFromFile reads from a TUM / CoRBS time stamp (TrajectoryManager.cpp:166-168): std::remove
    closes the gap the '.' leaves but does not shorten the string, so "1305031102.175304" parses (%llu) as 13050311021753044 — the
    digits without the dot and the last digit once more"""
    kept = token.replace(".", "")
    kept += token[len(kept):]
    digits = ""
    for ch in kept:
        if not ch.isdigit():
            break
        digits += ch
    return int(digits) if digits else 0


def load_trajectory_file(path, fmt="TUM"):
    """the pose files TrajectoryManager::LoadFromFile replays (Core/src/Utils/TrajectoryManager.cpp:61-282): TUM / CoRBS,
    zhou (re-based on the first pose), ICL_NUIM_RT (x mirrored on the left, y on the right); returns a list of 4x4 T_wc"""
    if fmt in ("TUM", "CoRBS"):
        # Eigen::Quaternionf(qw, qx, qy, qz) -> Isometry3f::rotate (TrajectoryManager.cpp:225-231): fp32, the quaternion as read
        # (not normalised), Eigen's toRotationMatrix operation order — the same floats as include/hrbf_io.h loadTrajectoryFile
        out = []
        f = np.float32
        for line in open(path):
            v = line.split()
            if len(v) != 8 or line.startswith("#"):
                continue
            if not line.endswith("\n"):      # `if(file.eof()) break;` before the push_back (TrajectoryManager.cpp:170,207): a last
                break                        # line without a trailing newline is read and dropped
            x, y, z, qx, qy, qz, qw = (f(float(a)) for a in v[1:8])
            tx, ty, tz = f(2) * qx, f(2) * qy, f(2) * qz
            twx, twy, twz, txx, txy, txz = tx * qw, ty * qw, tz * qw, tx * qx, ty * qx, tz * qx
            tyy, tyz, tzz = ty * qy, tz * qy, tz * qz
            out.append(np.array([[f(1) - (tyy + tzz), txy - twz, txz + twy, x], [txy + twz, f(1) - (txx + tzz), tyz - twx, y],
                                 [txz - twy, tyz + twx, f(1) - (txx + tyy), z], [0, 0, 0, 1]], np.float32))
        if not out:
            raise ValueError(path + ": no poses read")
        return out
    vals = open(path).read().split()
    out = []
    if fmt == "zhou":
        for k in range(0, len(vals) - 18, 19):
            out.append(np.array([float(x) for x in vals[k + 3:k + 19]], np.float64).reshape(4, 4))
        if out:
            inv0 = np.linalg.inv(out[0])
            out = [np.eye(4)] + [inv0 @ T for T in out[1:]]
    elif fmt == "ICL_NUIM_RT":
        for k in range(0, len(vals) - 11, 12):
            T = np.eye(4); T[:3, :4] = np.array([float(x) for x in vals[k:k + 12]]).reshape(3, 4)
            out.append(np.diag([-1.0, 1, 1, 1]) @ T @ np.diag([1.0, -1, 1, 1]))
    else:
        raise ValueError("trajectory format %r is not supported (TUM, CoRBS, zhou, ICL_NUIM_RT)" % (fmt,))
    if not out:
        raise ValueError(path + ": no poses read")
    return [np.asarray(T, np.float32) for T in out]


def save_ply(path, surfels, conf_threshold=0.0):
    """surfels: (N,20) float32 in the GlobalModel layout (hrbf_download_map)."""
    s = np.asarray(surfels, np.float32)
    s = s[s[:, 3] > conf_threshold]
    rec = np.zeros(len(s), dtype=[("xyz", "<f4", 3), ("rgb", "u1", 3), ("n", "<f4", 3), ("kmax", "<f4"), ("kmin", "<f4"),
                                  ("radius", "<f4"), ("submap", "<f4")])
    rec["xyz"] = s[:, 0:3]
    c = s[:, 4].astype(np.int64)
    rec["rgb"] = np.stack([(c >> 16) & 255, (c >> 8) & 255, c & 255], 1)
    rec["n"] = -s[:, 8:11]
    rec["kmax"] = s[:, 15]; rec["kmin"] = s[:, 19]; rec["radius"] = s[:, 11]; rec["submap"] = s[:, 5]
    head = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z"
            "\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty float nx\nproperty float ny"
            "\nproperty float nz\nproperty float curvature_max\nproperty float curvature_min\nproperty float radius"
            "\nproperty float submapIndex\nend_header\n" % len(s))
    with open(path, "wb") as f:
        f.write(head.encode()); f.write(rec.tobytes())
    return len(s)


def load_associations(path):
    """TUM associations: `t_depth depth_file t_rgb rgb_file` (the order HRBFFusion.cpp:226-236 reads)."""
    out = []
    for line in open(path):
        v = line.split()
        if len(v) >= 4 and not line.startswith("#"):
            out.append((float(v[0]), v[1], float(v[2]), v[3]))
    return out


def ate_rmse(est, gt, align=True):
    """translation RMSE (m); with align=True after the closed-form rigid alignment (Horn/Umeyama, no scale)."""
    E = np.asarray([np.asarray(p)[:3, 3] for p in est], np.float64)
    G = np.asarray([np.asarray(p)[:3, 3] for p in gt], np.float64)
    if align and len(E) >= 3:
        me, mg = E.mean(0), G.mean(0)
        U, _, Vt = np.linalg.svd((G - mg).T @ (E - me))
        S = np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))])
        R = U @ S @ Vt
        E = (E - me) @ R.T + mg
    return float(np.sqrt(((E - G) ** 2).sum(1).mean()))


# ------------------------------------------------------------------------------------------------
# .klg raw logs — the frame source the reference's GUI feeds processFrame from
# (GUI/src/Tools/RawLogReader.cpp:3-140).  Layout, little endian:
#     int32 numFrames
#     per frame: int64 timestamp, int32 depthSize, int32 imageSize, depthSize bytes, imageSize bytes
# depth is W*H uint16 raw (depthSize == W*H*2) or a zlib stream of it; the image is W*H*3 uint8 raw
# (imageSize == W*H*3), a JPEG (any other positive size) or absent (imageSize == 0 -> black).
import struct
import zlib


class KlgReader:
    """Sequential + random access reader.  `flip_colors` swaps the first and third channel like the reference's
    flipColors flag (logs recorded as BGR).  Frames come back as (timestamp, rgb[H, W, 3] uint8, depth[H, W] uint16),
    i.e. exactly the two buffers HRBFFusion::processFrame borrows."""

    def __init__(self, path, width=640, height=480, flip_colors=False):
        self.path, self.W, self.H, self.flip = path, int(width), int(height), bool(flip_colors)
        self.f = open(path, "rb")
        head = self.f.read(4)
        if len(head) != 4:
            raise ValueError("%s: not a .klg log (no frame count)" % path)
        self.num_frames = struct.unpack("<i", head)[0]
        if self.num_frames < 0:
            raise ValueError("%s: negative frame count" % path)
        self._offsets = [4]          # file offset of frame i, discovered as the log is walked (the reader's filePointers)
        self.current = 0

    def __len__(self):
        return self.num_frames

    def close(self):
        self.f.close()

    def has_more(self):
        return self.current < self.num_frames

    def _header(self, off):
        self.f.seek(off)
        h = self.f.read(16)
        if len(h) != 16:
            raise EOFError("%s: truncated at frame header (offset %d)" % (self.path, off))
        return struct.unpack("<qii", h)

    def _seek_frame(self, i):
        if not 0 <= i < self.num_frames:
            raise IndexError(i)
        while len(self._offsets) <= i:                      # fastForward: skip payloads, remember offsets
            off = self._offsets[-1]
            _, dsz, isz = self._header(off)
            if dsz < 0 or isz < 0:
                raise ValueError("%s: negative payload size at offset %d" % (self.path, off))
            self._offsets.append(off + 16 + dsz + isz)
        return self._offsets[i]

    def read_frame(self, i):
        off = self._seek_frame(i)
        ts, dsz, isz = self._header(off)
        if dsz < 0 or isz < 0:
            raise ValueError("%s: negative payload size at offset %d" % (self.path, off))
        dbuf = self.f.read(dsz)
        ibuf = self.f.read(isz) if isz > 0 else b""
        if len(dbuf) != dsz or len(ibuf) != isz:
            raise EOFError("%s: truncated payload of frame %d" % (self.path, i))
        if len(self._offsets) == i + 1:
            self._offsets.append(off + 16 + dsz + isz)
        n = self.W * self.H
        if dsz != n * 2:
            dbuf = zlib.decompress(dbuf)
            if len(dbuf) != n * 2:
                raise ValueError("%s: frame %d depth inflates to %d bytes, expected %d" % (self.path, i, len(dbuf), n * 2))
        depth = np.frombuffer(dbuf, dtype="<u2").reshape(self.H, self.W).copy()
        if isz == n * 3:
            rgb = np.frombuffer(ibuf, np.uint8).reshape(self.H, self.W, 3).copy()
        elif isz > 0:
            try:
                from PIL import Image
            except ImportError as e:       # pragma: no cover - PIL is present in the supported images
                raise RuntimeError("JPEG-compressed .klg frames need Pillow") from e
            import io as _io
            rgb = np.asarray(Image.open(_io.BytesIO(ibuf)).convert("RGB"), np.uint8)
            if rgb.shape != (self.H, self.W, 3):
                raise ValueError("%s: frame %d JPEG is %s, expected %dx%d" % (self.path, i, rgb.shape, self.W, self.H))
            rgb = rgb.copy()
        else:
            rgb = np.zeros((self.H, self.W, 3), np.uint8)
        if self.flip:
            rgb = rgb[..., ::-1].copy()
        return ts, rgb, depth

    def get_next(self):
        fr = self.read_frame(self.current)
        self.current += 1
        return fr

    def fast_forward(self, frame):
        self.current = min(int(frame), self.num_frames)
        if self.current < self.num_frames:
            self._seek_frame(self.current)

    def __iter__(self):
        for i in range(self.num_frames):
            yield self.read_frame(i)


def write_klg(path, frames, compress_depth=True, jpeg_quality=None):
    """Writer for tests and for converting other sources: frames = iterable of (timestamp, rgb uint8 [H,W,3] or None,
    depth uint16 [H,W]).  jpeg_quality = None stores the image raw."""
    frames = list(frames)
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(frames)))
        for ts, rgb, depth in frames:
            d = np.ascontiguousarray(depth, "<u2").tobytes()
            if compress_depth:
                d = zlib.compress(d)
            if rgb is None:
                im = b""
            elif jpeg_quality is None:
                im = np.ascontiguousarray(rgb, np.uint8).tobytes()
            else:
                from PIL import Image
                import io as _io
                b = _io.BytesIO()
                Image.fromarray(np.ascontiguousarray(rgb, np.uint8)).save(b, format="JPEG", quality=int(jpeg_quality))
                im = b.getvalue()
            f.write(struct.pack("<qii", int(ts), len(d), len(im)))
            f.write(d); f.write(im)

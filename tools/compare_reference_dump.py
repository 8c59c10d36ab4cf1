#!/usr/bin/env python3
"""Pin the oracle against the REFERENCE: load the dumps HRBFFusion3D itself writes and diff them against this
repository's implementation run on the same input.  (the reference cannot be built in the build
container — GL 3.3 context, Pangolin, CUDA, Eigen, OpenCV — so no dump is committed; with one, `parity unpinned`
closes with this one command.)

How to produce the dumps with the reference (any CUDA + GL machine that builds it)
    1. a data directory with the two GPUTest frames and an association file (sensorType 3 reads
       "timestamp depthfile timestamp rgbfile", GUI/src/Tools/RawImageReader.cpp):
           cp GPUTest/{1c,1d,2c,2d}.png data/ ;  printf '0.000000 1d.png 0.000000 1c.png\n0.033333 2d.png 0.033333 2c.png\n' > data/associations.txt
       and a camera file data/GPUTest.yaml:  Camera.fx: 528.0  Camera.fy: 528.0  Camera.cx: 320.0  Camera.cy: 240.0
           Camera.width: 640  Camera.height: 480  Camera.RGB: 1  DepthMapFactor: 5000.0   (+ the ORBextractor.* keys of TUM1.yaml)
    2. GUI/GlobalStateParam.txt: currentWorkingDirectory = ".../data"; sensorType = 3; AssociationFile = "associations.txt";
       parameterFileCvFormat = "GPUTest.yaml"; optimizationUseLocalBA = false; optimizationUseGlobalBA = false;
    3. cd GUI/build && ./HRBFFusion ; let both frames run, then press the GUI buttons
           "saveTexture"  -> IndexMap::downloadTexture  (prediction_hrbf_<tick-1>.txt, prediction_surfel_<tick-1>.txt),
                             HRBFFusion::downloadTextures (rawMap_attributes.txt),
                             RGBDOdometry::DownloadGPUMaps (prev_map_{0,1,2}.ply)          [GUI/src/HRBF_fusion.cpp:483-489]
           "save"         -> HRBFFusion::savePly("hrbf_globalModel.ply")                    [GUI/src/HRBF_fusion.cpp:476-481]
       and keep hrbf_trajectory.freiburg (written on exit, Utils/TrajectoryManager.cpp:284-373).
       All files land in the data directory (the process chdir()s there).
    4. python tools/compare_reference_dump.py --dumps data --frames data/associations.txt --camera data/GPUTest.yaml

The text dumps carry 6 significant digits (default ostream precision); the comparison is therefore by tolerance
(--rtol, default 2e-5 on top of half a unit in the 6th digit), per file: element count, number of rows that differ,
worst deviation per column.  Exit status 0 = every file present matched.

Engines: `--engine oracle` (default, CPU: oracle/ — what the parity tests trust) or `--engine gpu` (the HIP library).
"""
import argparse
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# ------------------------------------------------------------------------------------------------ loaders
def load_text_table(path, ncol):
    """whitespace-separated rows of `ncol` floats (prediction_*.txt have a double space in them)"""
    rows = []
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) == ncol:
                rows.append([float(x) for x in t])
    return np.asarray(rows, np.float64).reshape(-1, ncol)


def load_raw_map_attributes(path):
    """HRBFFusion::downloadTextures (HRBFFusion.cpp:1701-1730): vx vy vz nx ny nz sqrt(gradient_mag) confidence for
    every pixel, row-major, with vertex.z != 0 and normal.x != 0"""
    return load_text_table(path, 8)


def load_prediction_hrbf(path):
    """IndexMap::downloadTexture (IndexMap.cpp:661-692): global position, global normal, icp weight, k1, k2 for every
    pixel whose predicted curvature k1 is neither 0 nor NaN"""
    return load_text_table(path, 9)


def load_prediction_surfel(path):
    """IndexMap.cpp:696-727: the splatted (index-map) vertex / normal in the global frame, k1, k2 (same pixel rule)"""
    return load_text_table(path, 8)


def load_odometry_ply(path):
    """RGBDOdometry::savefilePLY (RGBDOdometry.cpp:1434-1500), ASCII: x y z nx ny nz curv_max curv_min image_dx image_dy
    rgb_gradient_mag icp_weight per pixel of one pyramid level (invalid pixels = zeros; normals NEGATED)"""
    with open(path) as f:
        n = None
        for line in f:
            if line.startswith("element vertex"):
                n = int(line.split()[2])
            if line.strip() == "end_header":
                break
        data = np.loadtxt(f, dtype=np.float64, ndmin=2)
    assert n is None or data.shape[0] == n, "vertex count mismatch in %s" % path
    return data


def load_model_ply(path):
    """HRBFFusion::savePly (HRBFFusion.cpp:1737-1853), binary little endian: x y z (float) r g b (uchar) nx ny nz
    curvature_max curvature_min radius submapIndex (float) — normals negated on write.  Returns an (n, 13) float array."""
    with open(path, "rb") as f:
        n = 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError("no end_header in %s" % path)
            if line.startswith(b"element vertex"):
                n = int(line.split()[2])
            if line.strip() == b"end_header":
                break
        rec = struct.Struct("<3f3B7f")
        raw = f.read(rec.size * n)
    out = np.zeros((n, 13), np.float64)
    for i in range(n):
        out[i] = rec.unpack_from(raw, i * rec.size)
    return out


def model_ply_rows(surfels, conf_threshold=0.0):
    """the rows savePly would write for an AoS surfel map (20 floats per surfel)"""
    m = np.asarray(surfels, np.float32).reshape(-1, 20)
    keep = m[:, 3] > conf_threshold
    m = m[keep]
    c = m[:, 4].astype(np.int64)
    out = np.zeros((m.shape[0], 13), np.float64)
    out[:, 0:3] = m[:, 0:3]
    out[:, 3] = (c >> 16) & 255; out[:, 4] = (c >> 8) & 255; out[:, 5] = c & 255
    out[:, 6:9] = -m[:, 8:11]
    out[:, 9] = m[:, 15]; out[:, 10] = m[:, 19]; out[:, 11] = m[:, 11]; out[:, 12] = m[:, 5]
    return out


# ------------------------------------------------------------------------------------------------ our side of each dump
def ours_raw_map_attributes(e):
    v = e.get_image("VERTEX_RAW"); n = e.get_image("NORMAL")
    g = e.get_image("GRADIENT_MAG"); c = e.get_image("CONFIDENCE")
    ok = (v[..., 2] != 0) & (n[..., 0] != 0)
    with np.errstate(invalid="ignore"):
        return np.concatenate([v[..., :3][ok], n[..., :3][ok], np.sqrt(g[ok])[:, None], c[ok][:, None]], 1).astype(np.float64)


def _to_global(T, p, nrm):
    T = T.astype(np.float64)
    return p.astype(np.float64) @ T[:3, :3].T + T[:3, 3], nrm.astype(np.float64) @ T[:3, :3].T


def ours_prediction(e, pose):
    k1 = e.get_image("PRED_CURV1")[..., 3]; k2 = e.get_image("PRED_CURV2")[..., 3]
    ok = (k1 != 0) & ~np.isnan(k1)
    pg, ng = _to_global(pose, e.get_image("PRED_VERTEX")[..., :3][ok], e.get_image("PRED_NORMAL")[..., :3][ok])
    w = e.get_image("PRED_ICPWEIGHT")[ok]
    hrbf = np.concatenate([pg, ng, w[:, None], k1[ok][:, None], k2[ok][:, None]], 1)
    sg, sn = _to_global(pose, e.get_image("INDEX_VERTCONF")[..., :3][ok], e.get_image("INDEX_NORMRAD")[..., :3][ok])
    surfel = np.concatenate([sg, sn, k1[ok][:, None], k2[ok][:, None]], 1)
    return hrbf.astype(np.float64), surfel.astype(np.float64)


# ------------------------------------------------------------------------------------------------ diff
def diff_table(name, ref, ours, rtol, order_free=False):
    """six-significant-digit text vs float32: |a - b| <= rtol * max(|a|, |b|) + half a unit of the 6th digit of a"""
    rep = {"file": name, "rows_reference": int(ref.shape[0]), "rows_ours": int(ours.shape[0])}
    if ref.shape[0] != ours.shape[0] or ref.shape[1] != ours.shape[1]:
        rep["match"] = False
        rep["reason"] = "row count differs"
        n = min(ref.shape[0], ours.shape[0])
        ref, ours = ref[:n], ours[:n]
        if n == 0:
            return rep
    a, b = ref, ours
    fin = np.isfinite(a) & np.isfinite(b)
    same_nonfinite = (np.isnan(a) == np.isnan(b)) & (np.isinf(a) == np.isinf(b))
    mag = np.maximum(np.abs(a), np.abs(b))
    with np.errstate(divide="ignore", invalid="ignore"):
        digit = np.where(mag > 0, 0.5 * 10.0 ** (np.floor(np.log10(np.where(mag > 0, mag, 1.0))) - 5), 1e-12)
    tol = rtol * mag + digit
    bad = np.where(fin, np.abs(a - b) > tol, ~same_nonfinite)
    rep["rows_differing"] = int(bad.any(axis=1).sum())
    rep["worst_abs_per_column"] = [float(np.nanmax(np.where(fin[:, j], np.abs(a[:, j] - b[:, j]), 0.0))) for j in range(a.shape[1])]
    rep.setdefault("match", rep["rows_differing"] == 0)
    if rep["rows_differing"]:
        i = int(np.argmax(bad.any(axis=1)))
        rep["first_difference"] = {"row": i, "reference": a[i].tolist(), "ours": b[i].tolist()}
    return rep


# ------------------------------------------------------------------------------------------------ driver
def read_frames(assoc, max_frames):
    from PIL import Image
    from hrbffusion3d_amd import io as hio
    base = os.path.dirname(os.path.abspath(assoc))
    out = []
    for td, fd, tr, fr in hio.load_associations(assoc):
        depth = np.asarray(Image.open(os.path.join(base, fd)), np.uint16)
        rgb = np.asarray(Image.open(os.path.join(base, fr)).convert("RGB"), np.uint8)
        out.append((td, np.ascontiguousarray(rgb), np.ascontiguousarray(depth)))
        if max_frames and len(out) >= max_frames:
            break
    return out


def make_engine(kind, p):
    if kind == "gpu":
        from hrbffusion3d_amd.api import HRBFFusion
        return HRBFFusion(p)
    import oracle_lib
    oracle_lib.build()
    return oracle_lib.Oracle(p, omp=True)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--dumps", required=True, help="directory holding the reference's dump files")
    ap.add_argument("--frames", required=True, help="associations.txt of the frames the reference processed")
    ap.add_argument("--camera", help="OpenCV camera YAML the reference ran with")
    ap.add_argument("--config", help="the GlobalStateParam.txt the reference ran with (tunables)")
    ap.add_argument("--max-frames", type=int, default=0, help="frames processed before the dump (default: all listed)")
    ap.add_argument("--engine", choices=["oracle", "gpu"], default="oracle")
    ap.add_argument("--rtol", type=float, default=2e-5)
    ap.add_argument("--ply-confidence", type=float, default=0.0, help="globalOutputSavePointCloudConfThreshold")
    ap.add_argument("--json", help="write the report here")
    args = ap.parse_args(argv)

    from hrbffusion3d_amd import config as hcfg
    from hrbffusion3d_amd import io as hio
    from hrbffusion3d_amd.params import default_params
    cam = hcfg.camera_from_yaml(args.camera) if args.camera else dict(width=640, height=480, fx=528.0, fy=528.0, cx=320.0,
                                                                      cy=240.0, depth_scale=1.0 / 5000.0, rgb=1)
    kw = hcfg.hrbf_kwargs(hcfg.load_global_state(args.config)) if args.config else {}
    frames = read_frames(args.frames, args.max_frames)
    if not frames:
        raise SystemExit("no frames listed in %s" % args.frames)
    p = default_params(cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], depth_scale=cam["depth_scale"],
                       max_surfels=max(1 << 20, 2 * cam["width"] * cam["height"] + len(frames) * cam["width"] * cam["height"] // 4), **kw)
    e = make_engine(args.engine, p)
    poses, stamps = [], []
    for ts, rgb, depth in frames:
        if not cam["rgb"]:
            rgb = np.ascontiguousarray(rgb[..., ::-1])
        e.process_frame(rgb, depth, int(round(ts * 1e6)))
        poses.append(e.get_pose()); stamps.append(int(round(ts * 1e6)))
    tick = len(frames) + 1                      # HRBFFusion::tick after the last frame; the dumps are named tick - 1
    report = {"engine": args.engine, "frames": len(frames), "files": []}
    d = args.dumps

    def have(name):
        return os.path.exists(os.path.join(d, name))

    if have("rawMap_attributes.txt"):
        report["files"].append(diff_table("rawMap_attributes.txt", load_raw_map_attributes(os.path.join(d, "rawMap_attributes.txt")),
                                          ours_raw_map_attributes(e), args.rtol))
    hrbf, surfel = ours_prediction(e, poses[-1])
    for name, loader, ours in (("prediction_hrbf_%d.txt" % (tick - 1), load_prediction_hrbf, hrbf),
                               ("prediction_surfel_%d.txt" % (tick - 1), load_prediction_surfel, surfel)):
        if have(name):
            report["files"].append(diff_table(name, loader(os.path.join(d, name)), ours, args.rtol))
    if have("hrbf_globalModel.ply"):
        ref = load_model_ply(os.path.join(d, "hrbf_globalModel.ply"))
        report["files"].append(diff_table("hrbf_globalModel.ply", ref, model_ply_rows(e.download_map(), args.ply_confidence), 1e-6))
    for name in ("hrbf_trajectory.freiburg", "hrbf_trajectory_whole.freiburg"):
        if have(name):
            gs, gp = hio.load_trajectory_tum(os.path.join(d, name))
            n = min(len(gp), len(poses))
            ref = np.asarray([np.concatenate([g[:3, 3], hio.rotation_to_quaternion(g[:3, :3])]) for g in gp[:n]])
            our = np.asarray([np.concatenate([q[:3, 3], hio.rotation_to_quaternion(q[:3, :3])]) for q in poses[:n]])
            sgn = np.sign((ref[:, 3:] * our[:, 3:]).sum(1, keepdims=True)); sgn[sgn == 0] = 1
            our[:, 3:] *= sgn                     # q and -q are the same rotation
            rep = diff_table(name, ref, our.astype(np.float64), args.rtol)
            rep["ate_rmse_m"] = hio.ate_rmse(poses[:n], list(gp[:n]), align=False)
            report["files"].append(rep)
    for lvl in range(3):
        name = "prev_map_%d.ply" % lvl
        if have(name):
            ref = load_odometry_ply(os.path.join(d, name))
            report["files"].append({"file": name, "rows_reference": int(ref.shape[0]), "match": None,
                                    "note": "loaded; the per-level odometry maps are internal to both the oracle and the HIP "
                                            "library (not exported through the C-ABI), so only the row count is reported: "
                                            "expected %d" % ((cam["width"] >> lvl) * (cam["height"] >> lvl))})
    e.close()
    checked = [f for f in report["files"] if f.get("match") is not None]
    report["all_match"] = bool(checked) and all(f["match"] for f in checked)
    if not report["files"]:
        report["note"] = "no known dump file found in %s" % d
    txt = json.dumps(report, indent=1)
    if args.json:
        with open(args.json, "w") as f:
            f.write(txt)
    print(txt)
    return 0 if report["all_match"] else 1


if __name__ == "__main__":
    sys.exit(main())

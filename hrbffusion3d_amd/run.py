"""Run a recorded sequence through the MI355X path: the caller's side of HRBFFusion::processFrame
(GUI/src/HRBF_fusion.cpp:190-497 without the GUI): frame source -> processFrame per frame -> trajectory + PLY.

    python -m hrbffusion3d_amd.run --klg log.klg --out traj.freiburg --ply map.ply
    python -m hrbffusion3d_amd.run --tum /data/rgbd_dataset_freiburg1_desk --fx 517.3 --fy 516.5 --cx 318.6 --cy 255.3 \
           --out traj.freiburg --groundtruth /data/.../groundtruth.txt
    python -m hrbffusion3d_amd.run --synthetic 200 --noise --out traj.freiburg        # the bench stream, with ATE
    python -m hrbffusion3d_amd.run --config GlobalStateParam.txt --out traj.freiburg  # the reference's own settings:
           # the parameter file (Core/src/Utils/parameterFile.h) names the data directory, the frame source (sensorType
           # 2 = .klg, 3 = associations.txt) and the OpenCV camera YAML (intrinsics, resolution, DepthMapFactor)

Frame sources: a .klg raw log (io.KlgReader), a TUM directory with associations.txt (PNG decoding needs Pillow), or the
synthetic stream.  There is no CPU fallback: the HIP library and a gfx950 device are required.
"""
import argparse
import os
import sys
import time

import numpy as np

from . import io as hio
from . import synth
from .params import default_params


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--config", help="the reference's GlobalStateParam.txt: frame source, camera YAML and every tunable "
                                      "come from it (explicit flags below still override)")
    src.add_argument("--klg", help=".klg raw log (RawLogReader format)")
    src.add_argument("--tum", help="TUM RGB-D directory containing associations.txt (ts depth ts rgb)")
    src.add_argument("--synthetic", type=int, metavar="N", help="N frames of the synthetic stream")
    ap.add_argument("--camera", help="OpenCV camera YAML (Camera.fx/fy/cx/cy/width/height, DepthMapFactor), e.g. TUM1.yaml; "
                                      "with --config the file it names is used unless this is given")
    ap.add_argument("--data-dir", help="with --config: where the data lives on THIS machine when currentWorkingDirectory "
                                        "of the parameter file does not exist here")
    ap.add_argument("--width", type=int, default=None); ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--fx", type=float, default=None); ap.add_argument("--fy", type=float, default=None)
    ap.add_argument("--cx", type=float, default=None); ap.add_argument("--cy", type=float, default=None)
    ap.add_argument("--depth-factor", type=float, default=None, help="raw units per metre (DepthMapFactor)")
    ap.add_argument("--flip-colors", action="store_true", help=".klg stored BGR")
    ap.add_argument("--noise", action="store_true", help="synthetic: Kinect-style depth noise + drop-outs")
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--max-surfels", type=int, default=4 * 1024 * 1024)
    ap.add_argument("--icp-weight", type=float, default=None); ap.add_argument("--rgb-only", action="store_true")
    ap.add_argument("--no-so3", action="store_true"); ap.add_argument("--fast-odom", action="store_true")
    ap.add_argument("--out", help="trajectory file (TUM format; --icl-nuim for that variant)")
    ap.add_argument("--icl-nuim", action="store_true")
    ap.add_argument("--ply", help="write the final surfel map as binary PLY")
    ap.add_argument("--ply-confidence", type=float, default=0.0)
    ap.add_argument("--groundtruth", help="TUM-format ground truth: report the Horn-aligned ATE RMSE")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    return apply_reference_config(args)


def apply_reference_config(args):
    """fill in what the reference's parameter file / camera YAML say; explicit flags win; then the plain defaults"""
    from . import config as hcfg
    args.param_overrides = {}
    cam_file = args.camera
    if args.config:
        g = hcfg.load_global_state(args.config)
        base = args.data_dir or os.path.dirname(os.path.abspath(args.config))
        args.param_overrides = hcfg.hrbf_kwargs(g)
        if cam_file is None and g.get("parameterFileCvFormat"):
            cam_file = hcfg.resolve(g, "parameterFileCvFormat", base)
        st = g.get("sensorType", 3)
        if st == 2:
            args.klg = hcfg.resolve(g, "klgFileName", base)
        elif st == 3:
            args.tum = os.path.dirname(hcfg.resolve(g, "AssociationFile", base)) or base
            args.assoc_name = os.path.basename(g.get("AssociationFile", "associations.txt"))
        else:
            raise SystemExit("sensorType %r (live camera) has no counterpart here" % (st,))
        if g.get("globalInputICLNUIMDataset"):
            args.icl_nuim = True
        # MainController::run (GUI/src/HRBF_fusion.cpp:190-239): start fast-forwards source and tick, the skip is applied once
        # (tick jump + fusion weight), the end frame bounds the tick; globalInputLoadTrajectory replays a pose file
        args.start_frame = int(g.get("globalStartFrame", 0) or 0)
        args.frames_to_skip = int(g.get("globalFrameToSkip", 0) or 0)
        if g.get("globalEndFrame", -1) > 0:
            args.end_tick = int(g["globalEndFrame"])
        if g.get("globalInputLoadTrajectory"):
            args.replay = (hcfg.resolve(g, "globalInputTrajectoryFile", base), g.get("globalInputTrajectoryFormat", "TUM"))
        if g.get("optimizationUseLocalBA") or g.get("optimizationUseGlobalBA"):
            sys.stderr.write("note: optimizationUseLocalBA / GlobalBA are set in %s; the sparse ORB back-end is out of "
                             "scope here (front-end only, as BASELINE configs 2 / 3 specify)\n" % args.config)
    if cam_file:
        cam = hcfg.camera_from_yaml(cam_file)
        for k in ("width", "height", "fx", "fy", "cx", "cy"):
            if getattr(args, k) is None:
                setattr(args, k, cam[k])
        if args.depth_factor is None:
            args.depth_factor = 1.0 / cam["depth_scale"]
        if not cam["rgb"]:
            args.flip_colors = True
    for k, v in (("width", 640), ("height", 480), ("fx", 528.0), ("fy", 528.0), ("cx", 320.0), ("cy", 240.0), ("depth_factor", 5000.0)):
        if getattr(args, k) is None:
            setattr(args, k, v)
    return args


def frame_source(args):
    """yields (timestamp_us, rgb uint8 [H,W,3], depth uint16 [H,W], ground-truth pose or None)"""
    if args.klg:
        r = hio.KlgReader(args.klg, args.width, args.height, flip_colors=args.flip_colors)
        for ts, rgb, depth in r:
            yield ts, rgb, depth, None
        r.close()
    elif args.tum:
        from PIL import Image
        for td, fd, tr, fr in hio.load_associations(os.path.join(args.tum, getattr(args, "assoc_name", "associations.txt"))):
            depth = np.asarray(Image.open(os.path.join(args.tum, fd)), np.uint16)
            rgb = np.asarray(Image.open(os.path.join(args.tum, fr)).convert("RGB"), np.uint8)
            # int64_t(t * 1000000.0): truncated, not rounded (GUI/src/Tools/RawImageReader.cpp:93)
            yield int(np.float64(td) * np.float64(1000000.0)), np.ascontiguousarray(rgb), np.ascontiguousarray(depth), None
    else:
        for k in range(args.synthetic):
            rgb, depth, T = synth.frame(k, args.width, args.height, noise=args.noise, depth_units=args.depth_factor)
            yield k * 33333, rgb, depth, T


def match_groundtruth(stamps_s, gt_stamps, gt_poses, max_dt=0.02):
    """nearest ground-truth pose per estimate (TUM's associate.py rule, 20 ms)"""
    idx = np.searchsorted(gt_stamps, stamps_s)
    pairs = []
    for i, (t, j) in enumerate(zip(stamps_s, idx)):
        best = None
        for c in (j - 1, j):
            if 0 <= c < len(gt_stamps) and (best is None or abs(gt_stamps[c] - t) < abs(gt_stamps[best] - t)):
                best = c
        if best is not None and abs(gt_stamps[best] - t) <= max_dt:
            pairs.append((i, best))
    return pairs


def main(argv=None):
    args = parse(argv)
    from .api import HRBFFusion
    if args.synthetic is not None:
        args.fx, args.fy, args.cx, args.cy = synth.intrinsics(args.width, args.height)
    kw = dict(args.param_overrides)
    kw.update(max_surfels=args.max_surfels, depth_scale=1.0 / args.depth_factor)
    if args.icp_weight is not None:
        kw["icp_weight"] = args.icp_weight
    if args.rgb_only:
        kw["rgb_only"] = 1
    if args.no_so3:
        kw["so3"] = 0
    if args.fast_odom:
        kw["fast_odom"] = 1
    p = default_params(args.width, args.height, args.fx, args.fy, args.cx, args.cy, **kw)
    fus = HRBFFusion(p, device=args.device)
    poses, stamps, gts, file_stamps = [], [], [], []
    replay = None
    if getattr(args, "replay", None):
        replay = hio.load_trajectory_file(*args.replay)          # HRBFFusion.cpp:55-59: LoadFromFile, currPose = poses[0]
        fus.set_load_trajectory(1)
        fus.set_pose(replay[0])
    start, skip, end_tick = getattr(args, "start_frame", 0), getattr(args, "frames_to_skip", 0), getattr(args, "end_tick", 65535)
    t0 = time.perf_counter()
    n = 0
    src = enumerate(frame_source(args))
    skip_to = 0        # source index below which frames are passed over (fastForward)
    for i, (ts, rgb, depth, T) in src:
        if i < skip_to:
            continue
        if fus.tick >= end_tick:
            break
        if fus.tick < start:
            fus.set_tick(start)
            skip_to = start
            if i < start:
                continue
        wm = float(skip + 1)
        if skip > 0:
            fus.set_tick(fus.tick + skip)
            skip_to = i + 1 + skip
            skip = 0
        if n == 0 and T is not None and replay is None:
            fus.set_pose(T)          # the synthetic stream starts at its analytic pose
        if replay is not None and fus.tick > 1:
            if fus.tick - 1 >= len(replay):
                raise SystemExit("globalInputLoadTrajectory: the trajectory file has fewer poses than frames")
            fus.set_pose(replay[fus.tick - 1])                   # HRBFFusion.cpp:1105-1108
        if fus.tick > 1 and replay is None:
            file_stamps.append(ts)       # TrajectoryManager::timstamp: pushed from the second tick on (HRBFFusion.cpp:1131-1132), not at tick 1 (:1060)
        fus.process_frame(rgb, depth, ts, wm)
        poses.append(fus.get_pose()); stamps.append(ts); gts.append(T)
        n += 1
        if args.max_frames and n >= args.max_frames:
            break
    fus.synchronize()
    dt = time.perf_counter() - t0
    report = {"frames": n, "seconds": dt, "fps_including_io_and_pose_readback": n / dt if dt > 0 else 0.0,
              "surfels": fus.surfel_count()}
    if args.out:
        # the reference's file: `poses` has an entry for every frame, `timstamp` none for a frame processed at tick 1 (HRBFFusion.cpp:1060
        # against :1131-1132), so line i then carries the stamp of frame i + 1 and the last line what follows the vector (here: its
        # index, like include/HRBFFusion.h).  The ATE below pairs pose i with frame i's own stamp.
        hio.save_trajectory(args.out, poses, stamps_us=[file_stamps[i] if i < len(file_stamps) else i for i in range(len(poses))],
                            fmt="TUM", icl_nuim=args.icl_nuim)
    if args.ply:
        report["ply_vertices"] = hio.save_ply(args.ply, fus.download_map(), conf_threshold=args.ply_confidence)
    if args.groundtruth:
        gs, gp = hio.load_trajectory_tum(args.groundtruth)
        pairs = match_groundtruth(np.asarray(stamps, np.float64) / 1e6, gs, gp)
        if len(pairs) >= 3:
            report["ate_rmse_m"] = hio.ate_rmse([poses[i] for i, _ in pairs], [gp[j] for _, j in pairs], align=True)
            report["ate_pairs"] = len(pairs)
    elif all(g is not None for g in gts) and n >= 3:
        report["ate_rmse_m"] = hio.ate_rmse(poses, gts, align=False)
    fus.close()
    print(report)
    return report


if __name__ == "__main__":
    sys.exit(0 if main() else 1)

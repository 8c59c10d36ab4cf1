"""BASELINE configs 2 / 3 on their own files: ICL-NUIM living-room kt2 and TUM fr1/desk in the layouts their
publishers ship, prepared for the reference's caller loop (GlobalStateParam.txt + camera YAML + association file,
GUI/src/HRBF_fusion.cpp:35-54, Core/src/HRBFFusion.cpp:212-270), and evaluated the way the benchmarks' own tools do
(time-stamp association within 20 ms, Horn alignment, translation RMSE).  Host logic only (numpy; Pillow for the writers).

The datasets are NOT in the build image (no network).  Everything here therefore runs today on the synthetic stream
WRITTEN TO DISK IN BOTH NATIVE LAYOUTS (`write_tum_layout`, `write_icl_layout`), so that real data adds pixels, not code paths:

  TUM RGB-D `rgbd_dataset_freiburg1_desk/`            ICL-NUIM `living_room_traj2_frei_png/` (TUM-compatible PNG package)
    rgb/<stamp>.png       8-bit RGB 640x480             rgb/<k>.png            8-bit RGB 640x480, k = 0 ..
    depth/<stamp>.png     16-bit grey, 5000 / m         depth/<k>.png          16-bit grey, 5000 / m
    rgb.txt, depth.txt    `stamp file` lists, '#' head  associations.txt       `k depth/k.png k rgb/k.png`
    groundtruth.txt       `stamp tx ty tz qx qy qz qw`  livingRoom2.gt.freiburg `k tx ty tz qx qy qz qw`, k = 1 ..
                          (mocap rate, '#' header)      camera: fx 481.20, fy -480.00, cx 319.5, cy 239.5 (image y up)
    camera (ROS default the benchmark recommends for fr1): 517.3, 516.5, 318.6, 255.3

These layout facts come from the datasets' documentation (external to /root/reference; SURVEY.md §8d flags them).  What the
REFERENCE does with such a directory is cited where it is mirrored:
  * association lines are `t_depth depth_file t_rgb rgb_file`, the first stamp is the frame's (HRBFFusion.cpp:226-236,
    GUI/src/Tools/RawImageReader.cpp:16-30); the stamp handed to processFrame is int64(t * 1e6), TRUNCATED (RawImageReader.cpp:93);
  * `globalInputICLNUIMDataset = true` negates ty and prints an integer stamp on save (TrajectoryManager.cpp:325-330): the
    sequence is run with a POSITIVE fy (the images' y axis points up, so the reconstruction is the mirror image in y).
"""
import os

import numpy as np

from . import io as hio

# GUI/GlobalStateParam.txt as shipped (every `key = value` of the file, in file order) — restated as data so that a run "with
# the reference's settings" needs no file from /root/reference at run time; tests/test_datasets.py holds this table to the
# reference's own file, key by key, wherever the reference checkout is present.
REFERENCE_GUI_SETTINGS = [
    ("currentWorkingDirectory", '"."'), ("sensorType", "3"), ("klgFileName", '"fr1_desk.klg"'),
    ("AssociationFile", '"associations.txt"'), ("parameterFileCvFormat", '"TUM1.yaml"'),
    ("optimizationUseLocalBA", "true"), ("optimizationUseGlobalBA", "true"), ("optimizationVocabularyFile", '""'),
    ("preprocessingUsebilateralFilter", "true"), ("preprocessingInitRadiusMultiplier", "4.0"),
    ("preprocessingCurvEstimationWindow", "3.0"), ("preprocessingCurvValidThreshold", "300"),
    ("preprocessingNormalEstimationPCA", "1.0"), ("preprocessingUseConfEval", "0.0"), ("preprocessingConfEvalEpsilon", "1000.0"),
    ("registrationPreAlignSO3", "true"), ("registrationJointICPWeight", "10.0"), ("registrationICPUseSparseICP", "false"),
    ("registrationUsePlaneConstraint", "false"), ("registrationICPUseCoorespondenceSearch", "false"),
    ("registrationICPNeighborSearchRadius", "2.0"), ("registrationICPUseWeightedICP", "true"),
    ("registrationICPCurvWeightImpactControl", "10"), ("registrationICPErrorThreshold", "5e-05"),
    ("registrationICPCovarianceThreshold", "1e-05"), ("registrationColorUseRGBGrad", "false"),
    ("registrationColorPhotoThreshold", "115"), ("preictionWindowMultiplier", "3.0"), ("preictionMinNeighbors", "6.0"),
    ("preictionMaxNeighbors", "10.0"), ("preictionConfThreshold", "3.0"), ("fusionMergeWindowMultiplier", "2.0"),
    ("fusionCleanWindowMultiplier", "2.0"), ("globalConfidenceThreshold", "5.0"), ("globalDenseEnoughThresh", "0.75"),
    ("globalDepthCutoff", "3.5"), ("globalInputICLNUIMDataset", "false"), ("globalInputLoadTrajectory", "false"),
    ("globalInputTrajectoryFormat", '"TUM"'), ("globalInputTrajectoryFile", '"hrbf_trajectory_whole.freiburg"'),
    ("globalOutputSaveTrjectoryFile", "true"), ("globalOutputSaveTrjectoryFileType", "TUM"),
    ("globalOutputCalculateMeanDistWithGroundTruth", "false"), ("globalOutputSavePointCloudConfThreshold", "0.0"),
    ("globalOutputsaveTimings", "false"), ("globalStartFrame", "0"), ("globalEndFrame", "-1"), ("globalFrameToSkip", "0"),
    ("globalExportFramePeriod", "false"), ("globalExportFrameStart", "0"), ("globalExportFrameEnd", "1"),
]
# the keys a run on another machine / another sequence has to change (everything else stays the reference's)
SEQUENCE_KEYS = ("currentWorkingDirectory", "klgFileName", "AssociationFile", "parameterFileCvFormat",
                 "optimizationVocabularyFile", "globalInputICLNUIMDataset")
# BASELINE configs 2 / 3: "loop-closure off (pure front-end)" — the sparse ORB back-end is out of scope
FRONT_END_ONLY = {"optimizationUseLocalBA": "false", "optimizationUseGlobalBA": "false"}

TUM_FR1 = dict(fx=517.3, fy=516.5, cx=318.6, cy=255.3, width=640, height=480, factor=5000.0)
ICL_NUIM = dict(fx=481.2, fy=480.0, cx=319.5, cy=239.5, width=640, height=480, factor=5000.0)   # the reference runs it with +fy

SEQUENCES = {
    # name: (kind, directory names tried under the dataset root)
    "icl_nuim_lr_kt2": ("icl", ("living_room_traj2_frei_png", "living_room_traj2n_frei_png", "lr_kt2", "icl_nuim_lr_kt2", "lr-kt2")),
    "tum_fr1_desk": ("tum", ("rgbd_dataset_freiburg1_desk", "fr1_desk", "tum_fr1_desk", "freiburg1_desk")),
}
GT_NAMES = {"icl": ("livingRoom2.gt.freiburg", "livingRoom2n.gt.freiburg", "traj2.gt.freiburg", "groundtruth.txt"),
            "tum": ("groundtruth.txt",)}


def find_sequence(root, name):
    """directory of sequence `name` under `root` (or `root` itself when it already is that sequence), else None"""
    kind, names = SEQUENCES[name]
    if not root:
        return None
    for cand in names:
        for d in (os.path.join(root, cand), os.path.join(root, cand, cand)):
            if os.path.isdir(os.path.join(d, "rgb")) and os.path.isdir(os.path.join(d, "depth")):
                return d
    if os.path.basename(os.path.normpath(root)) in names and os.path.isdir(os.path.join(root, "rgb")):
        return root
    return None


def read_file_list(path):
    """TUM `rgb.txt` / `depth.txt`: `stamp file` per line, '#' comments -> [(stamp, file)]"""
    out = []
    for line in open(path):
        line = line.split("#", 1)[0].replace(",", " ").split()
        if len(line) >= 2:
            out.append((float(line[0]), line[1]))
    return out


def associate(first, second, max_dt=0.02, offset=0.0):
    """the benchmark's associate.py: every (a, b) with |a - (b + offset)| < max_dt, best differences first, each stamp used once;
    returns index pairs sorted by the first list's stamp"""
    a = np.asarray([s for s, _ in first], np.float64); b = np.asarray([s for s, _ in second], np.float64) + offset
    cand = []
    order = np.argsort(b, kind="stable")            # the files are sorted by time; a list that is not is searched through its sorted view
    bs = b[order]
    j0 = np.searchsorted(bs, a - max_dt, "left"); j1 = np.searchsorted(bs, a + max_dt, "right")
    for i in range(len(a)):
        for jj in range(j0[i], j1[i]):
            j = int(order[jj])
            d = abs(a[i] - b[j])
            if d < max_dt:
                cand.append((d, a[i], b[j], i, j))       # associate.py sorts (difference, first stamp, second stamp)
    cand.sort()
    cand = [(d, i, j) for d, _, _, i, j in cand]
    used_a, used_b, pairs = set(), set(), []
    for _, i, j in cand:
        if i not in used_a and j not in used_b:
            used_a.add(i); used_b.add(j); pairs.append((i, j))
    pairs.sort()
    return pairs


def reference_frame_stamp(t):
    """the int64 the reference's association reader hands to processFrame: int64_t(t * 1000000.0), i.e. truncated
    (GUI/src/Tools/RawImageReader.cpp:93)"""
    return int(np.float64(t) * np.float64(1000000.0))


def write_global_state(path, overrides=None):
    """a GlobalStateParam.txt with the reference's GUI settings and `overrides` ({key: text as it should stand in the file})"""
    ov = dict(overrides or {})
    unknown = set(ov) - {k for k, _ in REFERENCE_GUI_SETTINGS}
    if unknown:
        raise KeyError("not a GlobalStateParam key: %s" % ", ".join(sorted(unknown)))
    with open(path, "w") as f:
        f.write("## written by hrbffusion3d_amd.datasets: GUI/GlobalStateParam.txt's settings; changed keys: %s\n"
                % (", ".join(sorted(ov)) or "none"))
        for k, v in REFERENCE_GUI_SETTINGS:
            f.write("%s = %s;\n" % (k, ov.get(k, v)))


def write_camera_yaml(path, cam, rgb_order=1):
    """the OpenCV-FileStorage keys the reference reads (GUI/src/HRBF_fusion.cpp:44-54, HRBFFusion.cpp:682-781)"""
    with open(path, "w") as f:
        f.write("%YAML:1.0\n\n# Camera calibration (pixels of this resolution)\n")
        for k in ("fx", "fy", "cx", "cy"):
            f.write("Camera.%s: %r\n" % (k, float(cam[k])))
        f.write("\nCamera.width: %d\nCamera.height: %d\n\n# colour order of the images (0: BGR, 1: RGB)\nCamera.RGB: %d\n\n"
                "# raw depth units per metre\nDepthMapFactor: %r\n" % (cam["width"], cam["height"], rgb_order, float(cam["factor"])))


def load_groundtruth(path):
    """`stamp tx ty tz qx qy qz qw` lines ('#' comments) -> (stamps float64 [n], poses list of 4x4)"""
    return hio.load_trajectory_tum(path)


def evaluate_ate(stamps_s, poses, gt_stamps, gt_poses, max_dt=0.02):
    """the benchmark's evaluate_ate.py: associate estimate and ground truth by stamp (20 ms), rigid alignment (Horn), RMSE of the
    translational differences.  Returns dict(rmse_m, pairs, mean_m, median_m, max_m)."""
    pairs = associate([(s, None) for s in stamps_s], [(s, None) for s in gt_stamps], max_dt)
    if len(pairs) < 3:
        return dict(rmse_m=None, pairs=len(pairs))
    E = np.asarray([np.asarray(poses[i])[:3, 3] for i, _ in pairs], np.float64)
    G = np.asarray([np.asarray(gt_poses[j])[:3, 3] for _, j in pairs], np.float64)
    me, mg = E.mean(0), G.mean(0)
    U, _, Vt = np.linalg.svd((G - mg).T @ (E - me))
    R = U @ np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))]) @ Vt
    d = np.sqrt((((E - me) @ R.T + mg - G) ** 2).sum(1))
    return dict(rmse_m=float(np.sqrt((d ** 2).mean())), pairs=len(pairs), mean_m=float(d.mean()), median_m=float(np.median(d)),
                max_m=float(d.max()))


def load_saved_trajectory(path, icl_nuim=False, frame_stamps_s=None):
    """a trajectory file as the reference saves it (TrajectoryManager.cpp:313-344) -> (stamps in seconds, poses).  The ICL-NUIM
    variant undoes what the writer did: ty negated back is NOT wanted (the negated value is the one comparable with the ground
    truth), and its integer stamp is the frame's int64 microsecond stamp (`int(timstamp[i])`), i.e. seconds * 1e6."""
    s, p = hio.load_trajectory_tum(path)
    if icl_nuim:
        s = s / 1e6
    # the reference pushes a pose for EVERY frame but a stamp only from the second frame on (HRBFFusion.cpp:1060 against
    # :1131-1132), so its file pairs pose i with the stamp of frame i + 1 and the last pose with whatever follows the vector;
    # include/HRBFFusion.h keeps that.  Whoever evaluates such a file has to re-pair it: pose i belongs to frame i
    if frame_stamps_s is not None and len(frame_stamps_s) == len(s):
        s = np.asarray(frame_stamps_s, np.float64)
    return s, p


def prepare(seq_dir, work_dir, kind, max_frames=0):
    """make `work_dir` the reference's currentWorkingDirectory for sequence `seq_dir` without writing into the (possibly read-only)
    dataset: symlinks rgb/ and depth/, writes GlobalStateParam.txt, the camera YAML and — for a TUM sequence that ships none —
    associations.txt from depth.txt x rgb.txt.  Returns a dict describing the run."""
    os.makedirs(work_dir, exist_ok=True)
    for sub in ("rgb", "depth"):
        link = os.path.join(work_dir, sub)
        if not os.path.lexists(link):
            os.symlink(os.path.abspath(os.path.join(seq_dir, sub)), link)
    cam = dict(TUM_FR1 if kind == "tum" else ICL_NUIM)
    yaml_name = "TUM1.yaml" if kind == "tum" else "ICL.yaml"
    write_camera_yaml(os.path.join(work_dir, yaml_name), cam)
    assoc_src = os.path.join(seq_dir, "associations.txt")
    assoc = os.path.join(work_dir, "associations.txt")
    if os.path.isfile(assoc_src):
        lines = [l for l in open(assoc_src).read().splitlines() if l.strip() and not l.startswith("#")]
        # both publishers' tools also emit `t rgb_file t depth_file`; the reference wants the depth file first (HRBFFusion.cpp:226-236)
        out = []
        for l in lines:
            v = l.split()
            if len(v) >= 4 and "rgb" in v[1] and "depth" in v[3]:
                v = [v[2], v[3], v[0], v[1]]
            out.append(" ".join(v[:4]))
    else:
        dl = read_file_list(os.path.join(seq_dir, "depth.txt")); rl = read_file_list(os.path.join(seq_dir, "rgb.txt"))
        out = ["%.6f %s %.6f %s" % (dl[i][0], dl[i][1], rl[j][0], rl[j][1]) for i, j in associate(dl, rl)]
    if max_frames:
        out = out[:max_frames]
    with open(assoc, "w") as f:      # no comment line: the reference's reader would take one for a frame (RawImageReader.cpp:12-30)
        f.write("\n".join(out) + "\n")
    write_global_state(os.path.join(work_dir, "GlobalStateParam.txt"), dict(
        FRONT_END_ONLY, currentWorkingDirectory='"%s"' % os.path.abspath(work_dir), parameterFileCvFormat='"%s"' % yaml_name,
        globalInputICLNUIMDataset="true" if kind == "icl" else "false"))
    gt = next((os.path.join(seq_dir, n) for n in GT_NAMES[kind] if os.path.isfile(os.path.join(seq_dir, n))), None)
    if gt is None and kind == "icl":     # the ground truth is a separate download: accept it beside the directory too
        gt = next((os.path.join(os.path.dirname(os.path.normpath(seq_dir)), n) for n in GT_NAMES[kind]
                   if os.path.isfile(os.path.join(os.path.dirname(os.path.normpath(seq_dir)), n))), None)
    return dict(kind=kind, work_dir=work_dir, config=os.path.join(work_dir, "GlobalStateParam.txt"), camera=cam, frames=len(out),
                groundtruth=gt, icl_nuim=kind == "icl", stamps_s=[float(l.split()[0]) for l in out])


def read_frames(info, count=None):
    """(stamp_us, rgb, depth) of the prepared sequence's first `count` frames, decoded with Pillow (the Python twin of
    include/hrbf_io.h's AssociationReader)"""
    from PIL import Image
    wd = info["work_dir"]
    for k, (td, fd, tr, fr) in enumerate(hio.load_associations(os.path.join(wd, "associations.txt"))):
        if count is not None and k >= count:
            break
        depth = np.ascontiguousarray(np.asarray(Image.open(os.path.join(wd, fd)), np.uint16))
        rgb = np.ascontiguousarray(np.asarray(Image.open(os.path.join(wd, fr)).convert("RGB"), np.uint8))
        yield reference_frame_stamp(td), rgb, depth


# ------------------------------------------------------------------------------------------------ writers (synthetic twins)
def _save_pair(d, rgb_name, depth_name, rgb, depth):
    from PIL import Image
    Image.fromarray(rgb, "RGB").save(os.path.join(d, rgb_name))
    Image.fromarray(depth).save(os.path.join(d, depth_name))          # 16-bit grey ("I;16")


def _gt_line(stamp_text, T):
    q = hio.rotation_to_quaternion(np.asarray(T, np.float64)[:3, :3])
    t = np.asarray(T, np.float64)[:3, 3]
    return "%s %.4f %.4f %.4f %.4f %.4f %.4f %.4f\n" % (stamp_text, t[0], t[1], t[2], q[0], q[1], q[2], q[3])


def write_tum_layout(d, frames, t0=1305031452.791720, with_associations=False, gt_rate=100.0, pose_of=None):
    """frames: [(rgb, depth, T_wc)] -> a directory in the TUM RGB-D layout.  Like the real recordings the colour and depth
    streams carry DIFFERENT stamps (depth ~ 12 ms after colour, 30 Hz with jitter below a millisecond), file names are the
    stamps, and groundtruth.txt is a separate, denser stream (`gt_rate` Hz, 4 decimals, three '#' header lines) sampled
    from `pose_of(time in frames)` (default: interpolation is not needed — the synthetic path is analytic)."""
    from . import synth
    os.makedirs(os.path.join(d, "rgb"), exist_ok=True); os.makedirs(os.path.join(d, "depth"), exist_ok=True)
    rng = np.random.default_rng(20240229)
    rl, dl = [], []
    for k, (rgb, depth, _) in enumerate(frames):
        tr = t0 + k / 30.0 + float(rng.uniform(-4e-4, 4e-4))
        td = tr + 0.0123 + float(rng.uniform(-4e-4, 4e-4))
        rn, dn = "rgb/%.6f.png" % tr, "depth/%.6f.png" % td
        _save_pair(d, rn, dn, rgb, depth)
        rl.append((tr, rn)); dl.append((td, dn))
    for name, lst, what in (("rgb.txt", rl, "color images"), ("depth.txt", dl, "depth maps")):
        with open(os.path.join(d, name), "w") as f:
            f.write("# %s\n# file: 'synthetic twin of rgbd_dataset_freiburg1_desk.bag'\n# timestamp filename\n" % what)
            for s, n in lst:
                f.write("%.6f %s\n" % (s, n))
    # the depth image is the frame (its stamp is the frame's): ground truth of frame k is exact at depth stamp k
    pose_of = pose_of or (lambda x: synth.camera_pose(x))
    with open(os.path.join(d, "groundtruth.txt"), "w") as f:
        f.write("# ground truth trajectory\n# file: 'synthetic twin'\n# timestamp tx ty tz qx qy qz qw\n")
        t_end = dl[-1][0] + 0.05
        g = dl[0][0] - 0.05
        k_of = lambda t: np.interp(t, [s for s, _ in dl], np.arange(len(dl), dtype=np.float64))
        while g < t_end:
            f.write(_gt_line("%.4f" % g, pose_of(float(k_of(g)))))
            g += 1.0 / gt_rate
    if with_associations:
        with open(os.path.join(d, "associations.txt"), "w") as f:
            for i, j in associate(dl, rl):
                f.write("%.6f %s %.6f %s\n" % (dl[i][0], dl[i][1], rl[j][0], rl[j][1]))
    return dict(depth=dl, rgb=rl)


def write_icl_layout(d, frames, gt_name="livingRoom2.gt.freiburg"):
    """frames: [(rgb, depth, T_wc)] RENDERED WITH A NEGATIVE fy (synth.ICL_NUIM_NEG: the publisher's camera, image y up) -> the
    ICL-NUIM TUM-compatible PNG package: rgb/k.png, depth/k.png, associations.txt `k depth/k.png k rgb/k.png` from k = 0,
    ground truth `k tx ty tz qx qy qz qw` from k = 1 (the publisher's file has no line for frame 0)."""
    os.makedirs(os.path.join(d, "rgb"), exist_ok=True); os.makedirs(os.path.join(d, "depth"), exist_ok=True)
    with open(os.path.join(d, "associations.txt"), "w") as f:
        for k, (rgb, depth, _) in enumerate(frames):
            _save_pair(d, "rgb/%d.png" % k, "depth/%d.png" % k, rgb, depth)
            f.write("%d depth/%d.png %d rgb/%d.png\n" % (k, k, k, k))
    with open(os.path.join(d, gt_name), "w") as f:
        for k, (_, _, T) in enumerate(frames):
            if k >= 1:
                f.write(_gt_line("%d" % k, T))


def params_for(info, max_surfels=4 * 1024 * 1024):
    """hrbf_params of a prepared sequence, derived exactly like the runners derive them (run.py `--config`, tools/hrbf_run.cpp
    paramsFrom): GlobalStateParam.txt -> config.hrbf_kwargs, camera YAML -> intrinsics, size and depth scale"""
    from . import config as hcfg
    from .params import default_params
    g = hcfg.load_global_state(info["config"])
    cam = hcfg.camera_from_yaml(hcfg.resolve(g, "parameterFileCvFormat", info["work_dir"]))
    kw = dict(hcfg.hrbf_kwargs(g))
    kw.update(max_surfels=max_surfels, depth_scale=cam["depth_scale"])
    return default_params(cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], **kw)

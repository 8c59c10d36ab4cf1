"""The reference's own configuration files, read the way the reference reads them, so that BASELINE configs 2 / 3 run
"with the reference's settings" from the files a maintainer already has:

* `GUI/GlobalStateParam.txt` — `key = value;` lines parsed by `ParameterFile` (Core/src/Utils/parameterFile.h:29-75,
  186-230) into the typed fields of `GlobalStateParam` (Core/src/Utils/GlobalStateParams.h:12-63);
* the OpenCV-FileStorage camera file named by `parameterFileCvFormat` (e.g. TUM1.yaml), read by
  `MainController` (GUI/src/HRBF_fusion.cpp:44-54: Camera.fx/fy/cx/cy/width/height) and
  `HRBFFusion::LoadCameraParaAndInitORBExtractor` (Core/src/HRBFFusion.cpp:682-781: DepthMapFactor, Camera.RGB).

Pure host logic (no GPU, no OpenCV).  `hrbf_kwargs()` maps the fields onto `hrbf_params` (include/hrbf_mi355.h) the way
the GUI passes them to the HRBFFusion constructor (GUI/src/HRBF_fusion.cpp:87-96,174-181).
"""
import os
import re

# (type, name) in the order of X_GLOBAL_PARAM_FIELDS (GlobalStateParams.h:12-63)
GLOBAL_PARAM_FIELDS = [
    ("str", "currentWorkingDirectory"), ("int", "sensorType"), ("str", "klgFileName"), ("str", "AssociationFile"),
    ("str", "parameterFileCvFormat"), ("bool", "optimizationUseLocalBA"), ("bool", "optimizationUseGlobalBA"),
    ("str", "optimizationVocabularyFile"), ("bool", "preprocessingUsebilateralFilter"),
    ("float", "preprocessingInitRadiusMultiplier"), ("float", "preprocessingCurvEstimationWindow"),
    ("float", "preprocessingCurvValidThreshold"), ("int", "preprocessingUseConfEval"),
    ("float", "preprocessingConfEvalEpsilon"), ("bool", "registrationPreAlignSO3"),
    ("float", "registrationJointICPWeight"), ("bool", "registrationICPUseSparseICP"),
    ("bool", "registrationUsePlaneConstraint"), ("bool", "registrationICPUseCoorespondenceSearch"),
    ("int", "registrationICPNeighborSearchRadius"), ("bool", "registrationICPUseWeightedICP"),
    ("float", "registrationICPCurvWeightImpactControl"), ("float", "registrationICPErrorThreshold"),
    ("float", "registrationICPCovarianceThreshold"), ("bool", "registrationColorUseRGBGrad"),
    ("float", "registrationColorPhotoThreshold"), ("float", "preictionWindowMultiplier"),
    ("int", "preictionMinNeighbors"), ("int", "preictionMaxNeighbors"), ("float", "preictionConfThreshold"),
    ("float", "fusionMergeWindowMultiplier"), ("float", "fusionCleanWindowMultiplier"),
    ("float", "globalConfidenceThreshold"), ("float", "globalDenseEnoughThresh"), ("float", "globalDepthCutoff"),
    ("bool", "globalInputICLNUIMDataset"), ("bool", "globalInputLoadTrajectory"),
    ("str", "globalInputTrajectoryFormat"), ("str", "globalInputTrajectoryFile"),
    ("bool", "globalOutputSaveTrjectoryFile"), ("str", "globalOutputSaveTrjectoryFileType"),
    ("bool", "globalOutputCalculateMeanDistWithGroundTruth"), ("float", "globalOutputSavePointCloudConfThreshold"),
    ("bool", "globalOutputsaveTimings"), ("int", "globalStartFrame"), ("int", "globalEndFrame"),
    ("int", "globalFrameToSkip"), ("bool", "globalExportFramePeriod"), ("int", "globalExportFrameStart"),
    ("int", "globalExportFrameEnd"), ("float", "preprocessingNormalEstimationPCA"),
]

_STRIP = " \t\";\r\n"     # ParameterFile::removeSpecialCharacters


def _remove_comments(line):
    """ParameterFile::removeComments: cut at the first "//", "#" or ";" that is not enclosed in a pair of double quotes
    (each marker is searched once, from the start of what is left)"""
    q = [i for i, c in enumerate(line) if c == '"']
    for marker in ("//", "#", ";"):
        at = line.find(marker)
        if at < 0:
            continue
        inside = any(q[j] < at < q[j + 1] for j in range(0, len(q) - 1, 2))
        if not inside:
            line = line[:at]
    return line


def parse_parameter_file(path):
    """name -> raw string value, later lines overriding earlier ones (a std::map assignment)"""
    values = {}
    with open(path, "r", errors="replace") as f:
        for raw in f:
            line = _remove_comments(raw.rstrip("\n")).strip(_STRIP)
            if len(line) <= 1:
                continue
            at = line.find("=")
            if at < 0:
                continue      # "No seperator found in line"
            name = line[:at].strip(_STRIP); value = line[at + 1:].strip(_STRIP)
            if not name:
                continue
            values[name] = value
    return values


_NUM = re.compile(r"^\s*[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?)")


def _convert(kind, text):
    """util::convertTo (stringUtilConvert.h:15-90): bool = everything but "false" / "False" / "0"; int = std::stoi and
    float = std::stof, i.e. the longest numeric prefix ("6.0" -> int 6, "0.0" -> int 0)"""
    if kind == "str":
        return text
    if kind == "bool":
        return text not in ("false", "False", "0")
    m = _NUM.match(text)
    if not m:
        return 0 if kind == "int" else 0.0
    if kind == "int":
        mi = re.match(r"^\s*[-+]?\d+", text)
        return int(mi.group(0)) if mi else 0
    return float(m.group(0))


def load_global_state(path):
    """typed GlobalStateParam fields found in the file (fields the file lacks are absent: the reference prints
    "skip param name" and keeps an uninitialised member)"""
    raw = parse_parameter_file(path)
    out = {}
    for kind, name in GLOBAL_PARAM_FIELDS:
        if name in raw:
            out[name] = _convert(kind, raw[name])
    return out


def parse_camera_yaml(path):
    """flat `key: value` scalars of an OpenCV FileStorage YAML (the `%YAML:1.0` header and `---` are skipped).
    Returns a dict with floats / ints / strings; keys as written (Camera.fx, DepthMapFactor, ...)."""
    out = {}
    with open(path, "r", errors="replace") as f:
        for raw in f:
            line = raw.split("#", 1)[0].strip()
            if not line or line.startswith("%") or line.startswith("---"):
                continue
            if ":" not in line:
                continue
            k, v = line.split(":", 1)
            k = k.strip(); v = v.strip().strip('"')
            if not k or not v:
                continue
            try:
                out[k] = int(v)
            except ValueError:
                try:
                    out[k] = float(v)
                except ValueError:
                    out[k] = v
    return out


def camera_from_yaml(path):
    """(width, height, fx, fy, cx, cy, depth_scale, rgb_order) as the reference derives them: a missing / zero
    DepthMapFactor means 1, otherwise metres = raw / DepthMapFactor (HRBFFusion.cpp:772-780)"""
    y = parse_camera_yaml(path)
    need = ("Camera.fx", "Camera.fy", "Camera.cx", "Camera.cy", "Camera.width", "Camera.height")
    missing = [k for k in need if k not in y]
    if missing:
        raise ValueError("%s lacks %s" % (path, ", ".join(missing)))
    factor = float(y.get("DepthMapFactor", 0.0))
    depth_scale = 1.0 if abs(factor) < 1e-5 else 1.0 / factor
    return dict(width=int(y["Camera.width"]), height=int(y["Camera.height"]), fx=float(y["Camera.fx"]),
                fy=float(y["Camera.fy"]), cx=float(y["Camera.cx"]), cy=float(y["Camera.cy"]), depth_scale=depth_scale,
                rgb=int(y.get("Camera.RGB", 1)))


def hrbf_kwargs(g):
    """GlobalStateParam fields -> keyword overrides for params.default_params, following the reads on the path:
    HRBF_fusion.cpp:87-96 (ctor arguments), HRBFFusion.cpp:1263-1345, RGBDOdometry.cpp, IndexMap.cpp:413-518,
    GlobalModel.cpp:551-688.  Unknown / absent fields keep the defaults of GUI/GlobalStateParam.txt."""
    m = {
        "globalConfidenceThreshold": ("confidence_threshold", float),
        "globalDepthCutoff": ("depth_cutoff", float),
        "registrationJointICPWeight": ("icp_weight", float),
        "registrationPreAlignSO3": ("so3", int),
        "preprocessingUsebilateralFilter": ("use_bilateral", int),
        "preprocessingInitRadiusMultiplier": ("init_radius_multiplier", float),
        "preprocessingCurvEstimationWindow": ("curv_estimation_window", float),
        "preprocessingCurvValidThreshold": ("curv_valid_threshold", float),
        "preprocessingNormalEstimationPCA": ("normal_estimation_pca", float),
        "preprocessingUseConfEval": ("use_conf_eval", int),
        "preprocessingConfEvalEpsilon": ("conf_eval_epsilon", float),
        "registrationICPUseCoorespondenceSearch": ("icp_use_corr_search", int),
        "registrationICPNeighborSearchRadius": ("icp_search_radius", int),
        "registrationICPUseWeightedICP": ("icp_use_weighted", int),
        "registrationICPCurvWeightImpactControl": ("icp_curv_weight_lambda", float),
        "registrationColorUseRGBGrad": ("rgb_use_grad_weight", int),
        "registrationICPUseSparseICP": ("use_sparse_icp", int),
        "preictionWindowMultiplier": ("predict_window_multiplier", float),
        "preictionMinNeighbors": ("predict_min_neighbors", int),
        "preictionMaxNeighbors": ("predict_max_neighbors", int),
        "preictionConfThreshold": ("predict_conf_threshold", float),
        "fusionCleanWindowMultiplier": ("clean_window_multiplier", float),
        "globalDenseEnoughThresh": ("dense_enough_thresh", float),
        "globalInputLoadTrajectory": ("load_trajectory", int),
    }
    return {dst: cast(g[src]) for src, (dst, cast) in m.items() if src in g}


def resolve(g, name, base=None):
    """a file the parameter file names, relative to currentWorkingDirectory like the reference's chdir
    (GUI/src/HRBF_fusion.cpp:41) — or to `base` when that directory does not exist on this machine"""
    p = g.get(name, "")
    if not p or os.path.isabs(p):
        return p
    cwd = g.get("currentWorkingDirectory", "")
    if cwd and os.path.isdir(cwd):
        return os.path.join(cwd, p)
    return os.path.join(base, p) if base else p

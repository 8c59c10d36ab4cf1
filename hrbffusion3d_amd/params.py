"""ctypes mirror of `hrbf_params` (include/hrbf_mi355.h) and the image / stage enums.

Defaults are the reference's: HRBFFusion ctor arguments as the GUI passes them
(GUI/src/HRBF_fusion.cpp:87-96,174-181) and GUI/GlobalStateParam.txt:20-81.
"""
import ctypes as C


class HrbfParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("depth_scale", C.c_float),
        ("confidence_threshold", C.c_float), ("depth_cutoff", C.c_float), ("icp_weight", C.c_float),
        ("fast_odom", C.c_int32), ("so3", C.c_int32), ("frame_to_frame_rgb", C.c_int32),
        ("rgb_only", C.c_int32), ("pyramid", C.c_int32),
        ("max_depth_processed", C.c_float),
        ("use_bilateral", C.c_int32), ("init_radius_multiplier", C.c_float),
        ("curv_estimation_window", C.c_float), ("curv_valid_threshold", C.c_float),
        ("normal_estimation_pca", C.c_float), ("use_conf_eval", C.c_int32), ("conf_eval_epsilon", C.c_float),
        ("icp_use_corr_search", C.c_int32), ("icp_search_radius", C.c_int32), ("icp_use_weighted", C.c_int32),
        ("icp_curv_weight_lambda", C.c_float), ("rgb_use_grad_weight", C.c_int32), ("use_sparse_icp", C.c_int32),
        ("predict_window_multiplier", C.c_float), ("predict_min_neighbors", C.c_int32),
        ("predict_max_neighbors", C.c_int32), ("predict_conf_threshold", C.c_float),
        ("clean_window_multiplier", C.c_float), ("dense_enough_thresh", C.c_float),
        ("max_surfels", C.c_int32), ("load_trajectory", C.c_int32),
    ]


def default_params(width=640, height=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, depth_scale=1.0 / 5000.0,
                   **overrides):
    p = HrbfParams()
    p.width, p.height = width, height
    p.fx, p.fy, p.cx, p.cy = fx, fy, cx, cy
    p.depth_scale = depth_scale
    p.confidence_threshold = 5.0
    p.depth_cutoff = 3.5
    p.icp_weight = 10.0
    p.fast_odom, p.so3, p.frame_to_frame_rgb, p.rgb_only, p.pyramid = 0, 1, 0, 0, 1
    p.max_depth_processed = 20.0
    p.use_bilateral = 1
    p.init_radius_multiplier = 4.0
    p.curv_estimation_window = 3.0
    p.curv_valid_threshold = 300.0
    p.normal_estimation_pca = 1.0
    p.use_conf_eval = 0
    p.conf_eval_epsilon = 1000.0
    p.icp_use_corr_search = 0
    p.icp_search_radius = 2
    p.icp_use_weighted = 1
    p.icp_curv_weight_lambda = 10.0
    p.rgb_use_grad_weight = 0
    p.use_sparse_icp = 0
    p.predict_window_multiplier = 3.0
    p.predict_min_neighbors = 6
    p.predict_max_neighbors = 10
    p.predict_conf_threshold = 3.0
    p.clean_window_multiplier = 2.0
    p.dense_enough_thresh = 0.75
    p.max_surfels = 4 * 1024 * 1024
    p.load_trajectory = 0
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


# hrbf_image enum: name -> (id, numpy dtype, channels)
IMAGES = {}
_img = [
    ("DEPTH_FILTERED", "f4", 1), ("DEPTH_METRIC", "f4", 1), ("DEPTH_METRIC_FILTERED", "f4", 1),
    ("VERTEX_RAW", "f4", 4), ("VERTEX_FILTERED", "f4", 4), ("NORMAL", "f4", 4), ("NORMAL_PCA", "f4", 4),
    ("RADIUS", "f4", 1), ("CURV1", "f4", 4), ("CURV2", "f4", 4), ("GRADIENT_MAG", "f4", 1),
    ("CONFIDENCE", "f4", 1), ("INDEX", "u4", 1), ("INDEX_VERTCONF", "f4", 4), ("INDEX_COLORTIME", "f4", 4),
    ("INDEX_NORMRAD", "f4", 4), ("INDEX_CURVMAX", "f4", 4), ("INDEX_CURVMIN", "f4", 4),
    ("PRED_IMAGE", "u1", 4), ("PRED_VERTEX", "f4", 4), ("PRED_NORMAL", "f4", 4), ("PRED_CURV1", "f4", 4),
    ("PRED_CURV2", "f4", 4), ("PRED_TIME", "u4", 1), ("PRED_ICPWEIGHT", "f4", 1),
    ("FILL_IMAGE", "u1", 4), ("FILL_VERTEX", "f4", 4), ("FILL_NORMAL", "f4", 4), ("FILL_CURV1", "f4", 4),
    ("FILL_CURV2", "f4", 4), ("FILL_ICPWEIGHT", "f4", 1),
]
for _i, (_n, _d, _c) in enumerate(_img):
    IMAGES[_n] = (_i, _d, _c)
# images of the optional extension (hrbf_fit_curvature; allocated on first use, not part of the reference's set): kept out of
# IMAGES so that loops over the reference's images do not meet them
EXT_IMAGES = {"FIT_CURV1": (len(_img), "f4", 4), "FIT_CURV2": (len(_img) + 1, "f4", 4), "FIT_NORMAL": (len(_img) + 2, "f4", 4)}

STAGES = {n: i for i, n in enumerate([
    "FILTER_DEPTH", "METRICISE", "VERTEX_NORMAL_RADIUS", "CURVATURE", "CONFIDENCE", "INITIALISE",
    "PREDICT_INDICES", "FUSE", "CLEAN", "PREDICT_HRBF", "FILLIN", "ODOMETRY"])}

"""Wrong VALUES at the boundary — sizes that do not match, enums out of range, counts beyond the capacity, parameters outside what the
kernels are built for — on a live context, each case in its own process: an error status (or a defined, harmless result), never a signal,
never a hang, and the context still processes the next frame.  tests/gpu_probe_abi_zero_args.py is the NULL / zero half of the survey.

    python tests/gpu_probe_abi_bad_values.py        # prints one line per case that died, hung or was accepted although it must not be
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (name, python statements run with: g (HRBFFusion after two frames), lib, h, C, np, P = pixels, cap; must end by setting `rc`),
# then `expect`: "error" = a non-zero status is required, "any" = any status as long as nothing dies
CASES = [
    ("get_image: image id 99", "buf = np.zeros(4 * P, np.float32); rc = lib.hrbf_get_image(h, 99, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))", "error"),
    ("get_image: image id -1", "buf = np.zeros(4 * P, np.float32); rc = lib.hrbf_get_image(h, -1, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))", "error"),
    ("get_image: buffer one byte short", "buf = np.zeros(4 * P, np.float32); rc = lib.hrbf_get_image(h, 3, buf.ctypes.data_as(C.c_void_p), C.c_size_t(16 * P - 1))", "error"),
    ("set_image: a larger buffer (the image's bytes are taken)", "buf = np.zeros(4 * P + 1, np.float32); rc = lib.hrbf_set_image(h, 3, buf.ctypes.data_as(C.c_void_p), C.c_size_t(16 * P + 1))", "any"),
    ("set_image: size of another image", "buf = np.zeros(P, np.float32); rc = lib.hrbf_set_image(h, 3, buf.ctypes.data_as(C.c_void_p), C.c_size_t(4 * P))", "error"),
    ("run_stage: stage 99", "rc = lib.hrbf_run_stage(h, 99)", "error"),
    ("run_stage: stage -1", "rc = lib.hrbf_run_stage(h, -1)", "error"),
    ("upload_map: one surfel beyond the capacity", "m = np.zeros((cap + 1, 20), np.float32); rc = lib.hrbf_upload_map(h, m.ctypes.data_as(C.c_void_p), C.c_size_t(cap + 1))", "error"),
    ("upload_map: count (size_t)-1", "m = np.zeros((8, 20), np.float32); rc = lib.hrbf_upload_map(h, m.ctypes.data_as(C.c_void_p), C.c_size_t(2 ** 64 - 1))", "error"),
    ("download_map: room for fewer surfels than the map holds", "m = np.zeros((8, 20), np.float32); rc = lib.hrbf_download_map(h, m.ctypes.data_as(C.c_void_p), C.c_size_t(8))", "error"),
    ("update_model: n = -1", "d = np.zeros(16, np.float32); rc = lib.hrbf_update_model(h, d.ctypes.data_as(C.c_void_p), -1)", "error"),
    ("update_model: n = 1201 (the reference's texture holds 1200)", "d = np.zeros(16 * 1201, np.float32); rc = lib.hrbf_update_model(h, d.ctypes.data_as(C.c_void_p), 1201)", "error"),
    ("update_model: n = 2^30 with 16 floats behind the pointer", "d = np.zeros(16, np.float32); rc = lib.hrbf_update_model(h, d.ctypes.data_as(C.c_void_p), 1 << 30)", "error"),
    ("set_active_submaps: n = -5", "a = np.ones(4, np.uint8); rc = lib.hrbf_set_active_submaps(h, a.ctypes.data_as(C.c_void_p), -5)", "error"),
    ("set_index_submap: -1", "rc = lib.hrbf_set_index_submap(h, -1)", "error"),
    ("set_index_submap: 2^24 + 1 (not a float32 integer any more)", "rc = lib.hrbf_set_index_submap(h, (1 << 24) + 1)", "error"),
    ("comm_init: rank >= world", "id = (C.c_uint8 * 128)(); rc = lib.hrbf_comm_init(h, 5, 2, id)", "error"),
    ("comm_init: world 0", "rc = lib.hrbf_comm_init(h, -1, 0, None)", "error"),
    ("map_shard_init: 100 virtual shards", "g.upload_map(np.zeros((0, 20), np.float32)); lib.hrbf_comm_init(h, -1, 100, None); rc = lib.hrbf_map_shard_init(h, 1); lib.hrbf_comm_init(h, -1, 1, None); lib.hrbf_map_shard_init(h, 0)", "error"),
    ("map_shard_init: mode 7", "g.upload_map(np.zeros((0, 20), np.float32)); lib.hrbf_comm_init(h, -1, 2, None); rc = lib.hrbf_map_shard_init(h, 7); lib.hrbf_comm_init(h, -1, 1, None); lib.hrbf_map_shard_init(h, 0)", "any"),
    ("map_shard_init on a map that is not empty", "lib.hrbf_comm_init(h, -1, 2, None); rc = lib.hrbf_map_shard_init(h, 1); lib.hrbf_comm_init(h, -1, 1, None)", "error"),
    ("map_rebalance without shards", "rc = lib.hrbf_map_rebalance(h)", "any"),
    ("set_fuse_ring_stride: 0", "rc = lib.hrbf_set_fuse_ring_stride(h, 0)", "error"),
    ("get_fuse_ring: max_frames -1", "a = np.zeros(8, np.float32); s = np.zeros(32, np.uint32); rc = lib.hrbf_get_fuse_ring(h, -1, a.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p)); rc = 1 if rc < 0 else 0", "error"),
    ("get_pose_log: frames that were never run", "o = np.zeros(16 * 4, np.float32); rc = lib.hrbf_get_pose_log(h, C.c_uint32(1000), C.c_uint32(4), o.ctypes.data_as(C.c_void_p), 0)", "any"),
    ("get_pose_log: count 2^31", "o = np.zeros(16 * 4, np.float32); rc = lib.hrbf_get_pose_log(h, C.c_uint32(0), C.c_uint32(1 << 31), o.ctypes.data_as(C.c_void_p), 0)", "error"),
    ("set_tick: negative", "rc = lib.hrbf_set_tick(h, -7)", "any"),
    ("set_pose: NaN", "T = np.full(16, np.nan, np.float32); rc = lib.hrbf_set_pose(h, T.ctypes.data_as(C.c_void_p))", "any"),
    ("set_hrbf_fit_params: window 0", "rc = lib.hrbf_set_hrbf_fit_params(h, 0, C.c_float(1.25), C.c_float(0.1), C.c_float(3.0))", "error"),
    ("set_hrbf_fit_params: window 9", "rc = lib.hrbf_set_hrbf_fit_params(h, 9, C.c_float(1.25), C.c_float(0.1), C.c_float(3.0))", "error"),
    ("set_hrbf_fit_params: support NaN", "rc = lib.hrbf_set_hrbf_fit_params(h, 2, C.c_float(float('nan')), C.c_float(0.1), C.c_float(3.0))", "error"),
    ("fit_curvature: window 9", "ms = C.c_float(); rc = lib.hrbf_fit_curvature(h, 9, C.c_float(1.25), C.c_float(1e-6), C.c_float(3.0), C.byref(ms))", "error"),
    ("weight multiplier NaN", "rgb, d, _ = synth.frame(5, 160, 120); rc = lib.hrbf_process_frame(h, rgb.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), C.c_int64(0), C.c_float(float('nan')))", "any"),
    ("set_weighting: negative", "rc = lib.hrbf_set_weighting(h, C.c_float(-1.0))", "any"),
    ("set_icp_weight: NaN", "rc = lib.hrbf_set_icp_weight(h, C.c_float(float('nan')))", "any"),
    ("set_depth_cutoff: -1", "rc = lib.hrbf_set_depth_cutoff(h, C.c_float(-1.0))", "any"),
]

CREATE = [   # hrbf_create with parameters the library is not built for: every one must be refused
    ("width 0", dict(width=0)), ("width 164 (not a multiple of 8)", dict(width=164)), ("height -120", dict(height=-120)),
    ("max_surfels 0", dict(max_surfels=0)), ("max_surfels -5", dict(max_surfels=-5)), ("fx 0", dict(fx=0.0)), ("fx NaN", dict(fx="nan")),
    ("depth_scale 0", dict(depth_scale=0.0)), ("depth_scale NaN", dict(depth_scale="nan")),
    ("curv_estimation_window 4", dict(curv_estimation_window=4.0)), ("curv_estimation_window 0", dict(curv_estimation_window=0.0)),
    ("predict_window_multiplier 5", dict(predict_window_multiplier=5.0)), ("predict_window_multiplier NaN", dict(predict_window_multiplier="nan")),
    ("clean_window_multiplier 9", dict(clean_window_multiplier=9.0)), ("clean_window_multiplier 0", dict(clean_window_multiplier=0.0)),
    ("clean_window_multiplier -2", dict(clean_window_multiplier=-2.0)), ("icp_search_radius 50", dict(icp_search_radius=50)),
    ("icp_search_radius -1", dict(icp_search_radius=-1)), ("predict_max_neighbors 1000", dict(predict_max_neighbors=1000)),
    ("predict_min_neighbors -3", dict(predict_min_neighbors=-3)), ("device 99", dict(_device=99)),
]

# accepted on purpose: degenerate but defined (an empty window finds nothing, a search radius of 50 is a 101 x 101 search, the neighbour counts
# are thresholds on at most 49 candidates) — and the oracle does the same with them; they only must not bring the context down
DEFINED = {"curv_estimation_window 0", "clean_window_multiplier 0", "icp_search_radius 50", "predict_max_neighbors 1000", "predict_min_neighbors -3"}

CHILD = """
import ctypes as C, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from hrbffusion3d_amd import synth
from hrbffusion3d_amd.api import HRBFFusion, HrbfError
from hrbffusion3d_amd.params import default_params
if %r == "create":
    kw = {k: (float("nan") if v == "nan" else v) for k, v in %r.items()}
    dev = kw.pop("_device", 0)
    try:
        g = HRBFFusion(default_params(160, 120, *synth.intrinsics(160, 120), max_surfels=1 << 16) if not kw else
                       default_params(**{**dict(width=160, height=120, fx=132.0, fy=132.0, cx=80.0, cy=60.0, max_surfels=1 << 16), **kw}), device=dev)
    except HrbfError as e:
        print("RET refused:", str(e)[:100], flush=True); sys.exit(0)
    rgb, d, _ = synth.frame(0, 160, 120)
    try:
        for k in range(3):
            g.process_frame(rgb, d)
        g.synchronize(); print("RET accepted, 3 frames ran, status", g.status(), flush=True)
    except HrbfError as e:
        print("RET accepted, then:", str(e)[:100], flush=True)
    sys.exit(0)
g = HRBFFusion(default_params(160, 120, *synth.intrinsics(160, 120), max_surfels=1 << 16))
for k in range(2):
    rgb, d, _ = synth.frame(k, 160, 120, noise=True); g.process_frame(rgb, d)
lib, h, P, cap = g.lib, g.h, 160 * 120, 1 << 16
for f in ("hrbf_get_image", "hrbf_set_image", "hrbf_upload_map", "hrbf_download_map", "hrbf_update_model", "hrbf_set_active_submaps", "hrbf_comm_init",
          "hrbf_get_fuse_ring", "hrbf_get_pose_log", "hrbf_set_pose", "hrbf_set_hrbf_fit_params", "hrbf_fit_curvature", "hrbf_process_frame",
          "hrbf_set_weighting", "hrbf_set_icp_weight", "hrbf_set_depth_cutoff", "hrbf_run_stage", "hrbf_set_index_submap", "hrbf_map_shard_init",
          "hrbf_set_fuse_ring_stride", "hrbf_set_tick", "hrbf_map_rebalance"):
    getattr(lib, f).argtypes = None; getattr(lib, f).restype = C.c_int
%s
print("RET", rc, flush=True)
rgb, d, _ = synth.frame(2, 160, 120, noise=True)
try:
    g.process_frame(rgb, d); g.synchronize(); print("NEXT frame ok", flush=True)
except Exception as e:
    print("NEXT frame:", repr(e)[:160], flush=True)
"""


def survey():
    findings = []
    tests = os.path.join(ROOT, "tests")
    for name, code, expect in CASES:
        try:
            p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, tests, "case", {}, code)], capture_output=True, text=True, timeout=120)
        except subprocess.TimeoutExpired:
            findings.append("%s: hung" % name); continue
        lines = [l for l in p.stdout.splitlines() if l.startswith(("RET", "NEXT"))]
        if p.returncode != 0 or len(lines) < 2:
            err = [l for l in p.stderr.strip().splitlines() if l.strip()]
            findings.append("%s: exit %d after %s | %s" % (name, p.returncode, lines, err[-1][:160] if err else ""))
        elif expect == "error" and lines[0].split()[1] == "0":
            findings.append("%s: accepted (status 0)" % name)
    for name, kw in CREATE:
        try:
            p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, tests, "create", kw, "")], capture_output=True, text=True, timeout=120)
        except subprocess.TimeoutExpired:
            findings.append("create, %s: hung" % name); continue
        lines = [l for l in p.stdout.splitlines() if l.startswith("RET")]
        if p.returncode != 0 or not lines:
            err = [l for l in p.stderr.strip().splitlines() if l.strip()]
            findings.append("create, %s: exit %d | %s" % (name, p.returncode, err[-1][:160] if err else ""))
        elif "refused" not in lines[0] and name not in DEFINED:
            findings.append("create, %s: %s" % (name, lines[0][4:]))
    return findings


def main():
    f = survey()
    print("%d calls with wrong values, %d creations with parameters out of range: %d findings" % (len(CASES), len(CREATE), len(f)))
    for x in f:
        print("  " + x)
    return 1 if f else 0


if __name__ == "__main__":
    sys.exit(main())

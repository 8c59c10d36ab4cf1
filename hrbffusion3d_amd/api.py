"""Host-side Python mirror of the reference's `HRBFFusion` interface over the C-ABI of
libhrbf_mi355.so (include/hrbf_mi355.h).

Method names follow Core/src/HRBFFusion.h:82-534 (processFrame, getCurrPose, getTick, ...) in
snake_case.  There is NO CPU fallback: importing is free, but constructing HRBFFusion without the
HIP library or without a gfx950 device raises.
"""
import ctypes as C
import os

import numpy as np

from .params import HrbfParams, IMAGES, EXT_IMAGES, STAGES, default_params  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("HRBF_LIB", "libhrbf_mi355.so"))   # HRBF_LIB: experiment builds
_lib = None

EXPORTS = [
    "hrbf_default_params", "hrbf_create", "hrbf_destroy", "hrbf_last_error", "hrbf_version",
    "hrbf_process_frame", "hrbf_process_frame_device", "hrbf_synchronize", "hrbf_get_pose", "hrbf_set_pose",
    "hrbf_get_tick", "hrbf_surfel_count", "hrbf_download_map", "hrbf_upload_map", "hrbf_last_icp",
    "hrbf_last_weighting", "hrbf_set_rgb_only", "hrbf_set_icp_weight", "hrbf_set_pyramid", "hrbf_set_fast_odom",
    "hrbf_set_so3", "hrbf_set_frame_to_frame_rgb", "hrbf_set_confidence_threshold", "hrbf_set_depth_cutoff",
    "hrbf_image_bytes", "hrbf_get_image", "hrbf_set_image", "hrbf_enable_timing", "hrbf_get_timings",
    "hrbf_get_fuse_stats", "hrbf_upload_frame", "hrbf_run_stage", "hrbf_set_tick", "hrbf_set_weighting",
    "hrbf_set_index_submap", "hrbf_set_active_submaps", "hrbf_update_model",
    "hrbf_so3_step", "hrbf_rgb_residual", "hrbf_rgb_step",
    "hrbf_icp_step", "hrbf_icp_step_sparse", "hrbf_update_lambda_map", "hrbf_comm_unique_id", "hrbf_comm_init", "hrbf_peer_unique_id", "hrbf_comm_init_peer", "hrbf_peer_release_id", "hrbf_map_shard_init", "hrbf_map_rebalance", "hrbf_download_gids", "hrbf_shard_counts", "hrbf_hash_owner", "hrbf_hash_renumber_count", "hrbf_gn_graph_captures", "hrbf_fit_curvature", "hrbf_set_hrbf_fit", "hrbf_set_hrbf_fit_params", "hrbf_get_hrbf_fit", "hrbf_shard_exchange_mode",
    "hrbf_rebalance_plan", "hrbf_local_surfel_count", "hrbf_set_row_sharding", "hrbf_comm_stats",
    "hrbf_initialise", "hrbf_predict_indices", "hrbf_fuse", "hrbf_clean", "hrbf_predict_hrbf", "hrbf_dense_enough", "hrbf_bootstrap", "hrbf_get_fuse_ring", "hrbf_reset_fuse_ring", "hrbf_set_load_trajectory",
    "hrbf_get_fuse_ring_parts", "hrbf_get_status", "hrbf_frames_enqueued", "hrbf_frames_completed", "hrbf_get_pose_log",
    "hrbf_probe_single_workgroup_iteration", "hrbf_probe_sqrt_rounding", "hrbf_probe_exp_scaling", "hrbf_probe_division", "hrbf_set_fuse_ring_stride",
]


class HrbfError(RuntimeError):
    pass


class CommCounters(C.Structure):
    """mirror of hrbf_comm_counters (include/hrbf_mi355.h)"""
    _fields_ = [("transport", C.c_int32), ("world", C.c_int32), ("rank", C.c_int32), ("frames", C.c_int32)] + \
               [(n, C.c_uint64) for n in ("limb_allreduce", "limb_allreduce_bytes", "key_min_reduce", "key_min_reduce_bytes", "allgather",
                                          "allgather_bytes", "word_allreduce", "word_allreduce_bytes", "send", "send_bytes", "recv",
                                          "recv_bytes", "host_barriers")]


TRANSPORTS = {0: "none", 1: "rccl", 2: "shm", 3: "virtual"}


def load_library():
    """dlopen libhrbf_mi355.so (built by hrbffusion3d_amd.build); raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HrbfError("%s not found: run `python -m hrbffusion3d_amd.build` (hipcc, gfx950). "
                        "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    lib.hrbf_last_error.restype = C.c_char_p
    lib.hrbf_version.restype = C.c_char_p
    lib.hrbf_default_params.argtypes = [C.POINTER(HrbfParams), i32, i32, f32, f32, f32, f32, f32]
    lib.hrbf_default_params.restype = None
    lib.hrbf_create.argtypes = [C.POINTER(HrbfParams), i32, C.POINTER(vp)]
    lib.hrbf_destroy.argtypes = [vp]; lib.hrbf_destroy.restype = None
    lib.hrbf_process_frame.argtypes = [vp, vp, vp, C.c_int64, f32]
    lib.hrbf_process_frame_device.argtypes = [vp, vp, vp, C.c_int64, f32]
    lib.hrbf_synchronize.argtypes = [vp]
    lib.hrbf_get_pose.argtypes = [vp, vp]; lib.hrbf_set_pose.argtypes = [vp, vp]
    lib.hrbf_get_tick.argtypes = [vp]; lib.hrbf_set_tick.argtypes = [vp, i32]
    lib.hrbf_set_index_submap.argtypes = [vp, i32]; lib.hrbf_set_active_submaps.argtypes = [vp, vp, i32]
    lib.hrbf_update_model.argtypes = [vp, vp, i32]
    lib.hrbf_so3_step.argtypes = [vp, vp, vp, i32, i32] + [vp] * 6
    lib.hrbf_rgb_residual.argtypes = [vp, f32] + [vp] * 6 + [i32, i32] + [vp] * 6
    lib.hrbf_rgb_step.argtypes = [vp, vp, vp, f32, vp, f32, f32, vp, vp, i32, i32, i32, vp, vp, vp]
    lib.hrbf_surfel_count.argtypes = [vp]; lib.hrbf_surfel_count.restype = C.c_uint32
    lib.hrbf_download_map.argtypes = [vp, vp, C.c_size_t]; lib.hrbf_upload_map.argtypes = [vp, vp, C.c_size_t]
    lib.hrbf_last_icp.argtypes = [vp, vp, vp]; lib.hrbf_last_weighting.argtypes = [vp, vp]
    for n in ("rgb_only", "pyramid", "fast_odom", "so3", "frame_to_frame_rgb"):
        getattr(lib, "hrbf_set_" + n).argtypes = [vp, i32]
    for n in ("icp_weight", "confidence_threshold", "depth_cutoff", "weighting"):
        getattr(lib, "hrbf_set_" + n).argtypes = [vp, f32]
    lib.hrbf_image_bytes.argtypes = [vp, i32]; lib.hrbf_image_bytes.restype = C.c_size_t
    lib.hrbf_get_image.argtypes = [vp, i32, vp, C.c_size_t]; lib.hrbf_set_image.argtypes = [vp, i32, vp, C.c_size_t]
    lib.hrbf_enable_timing.argtypes = [vp, i32]; lib.hrbf_get_timings.argtypes = [vp, vp]
    lib.hrbf_get_fuse_stats.argtypes = [vp, vp]
    lib.hrbf_upload_frame.argtypes = [vp, vp, vp]; lib.hrbf_run_stage.argtypes = [vp, i32]
    lib.hrbf_bootstrap.argtypes = [vp, vp, vp]
    lib.hrbf_get_fuse_ring.argtypes = [vp, i32, vp, vp]; lib.hrbf_reset_fuse_ring.argtypes = [vp]
    lib.hrbf_get_fuse_ring_parts.argtypes = [vp, i32, vp, vp, vp]; lib.hrbf_get_status.argtypes = [vp, vp, i32]
    lib.hrbf_frames_enqueued.argtypes = [vp]; lib.hrbf_frames_enqueued.restype = C.c_uint32
    lib.hrbf_frames_completed.argtypes = [vp]; lib.hrbf_frames_completed.restype = C.c_uint32
    lib.hrbf_get_pose_log.argtypes = [vp, C.c_uint32, C.c_uint32, vp, i32]
    lib.hrbf_probe_single_workgroup_iteration.argtypes = [vp, i32, i32, vp]
    lib.hrbf_probe_sqrt_rounding.argtypes = [vp, vp]
    lib.hrbf_probe_exp_scaling.argtypes = [vp, vp]
    lib.hrbf_probe_division.argtypes = [vp, vp]
    lib.hrbf_set_fuse_ring_stride.argtypes = [vp, i32]
    lib.hrbf_set_load_trajectory.argtypes = [vp, i32]
    lib.hrbf_icp_step.argtypes = [vp] + [vp] * 6 + [vp] * 2 + [f32] * 4 + [vp] * 5 + [i32, i32, f32, f32, i32, vp, vp, vp]
    lib.hrbf_icp_step_sparse.argtypes = [vp] + [vp] * 6 + [vp] * 2 + [f32] * 4 + [vp] * 5 + [i32, i32, f32, f32, i32, vp, vp, vp, vp, vp, vp]
    lib.hrbf_update_lambda_map.argtypes = [vp] + [vp] * 9 + [i32, i32]
    lib.hrbf_comm_unique_id.argtypes = [vp]; lib.hrbf_comm_init.argtypes = [vp, i32, i32, vp]
    lib.hrbf_peer_unique_id.argtypes = [vp]; lib.hrbf_comm_init_peer.argtypes = [vp, i32, i32, vp]; lib.hrbf_peer_release_id.argtypes = [vp]
    lib.hrbf_map_shard_init.argtypes = [vp, i32]; lib.hrbf_map_rebalance.argtypes = [vp]
    lib.hrbf_download_gids.argtypes = [vp, vp, C.c_size_t]; lib.hrbf_shard_counts.argtypes = [vp, vp]
    lib.hrbf_hash_owner.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, i32]; lib.hrbf_hash_renumber_count.argtypes = [vp]; lib.hrbf_gn_graph_captures.argtypes = [vp]; lib.hrbf_fit_curvature.argtypes = [vp, i32, f32, f32, f32, vp]; lib.hrbf_set_hrbf_fit.argtypes = [vp, i32]; lib.hrbf_set_hrbf_fit_params.argtypes = [vp, i32, f32, f32, f32]; lib.hrbf_get_hrbf_fit.argtypes = [vp] * 6; lib.hrbf_shard_exchange_mode.argtypes = [vp]
    lib.hrbf_set_row_sharding.argtypes = [vp, i32]
    lib.hrbf_comm_stats.argtypes = [vp, C.c_void_p, i32]
    lib.hrbf_initialise.argtypes = [vp, vp]; lib.hrbf_predict_hrbf.argtypes = [vp]
    lib.hrbf_dense_enough.argtypes = [vp, vp]
    lib.hrbf_predict_indices.argtypes = [vp, vp, i32, f32, i32]; lib.hrbf_fuse.argtypes = [vp, vp, i32, f32, i32]
    lib.hrbf_clean.argtypes = [vp, vp, i32, f32, f32]
    lib.hrbf_rebalance_plan.argtypes = [vp, i32, vp, vp, vp]
    lib.hrbf_local_surfel_count.argtypes = [vp]; lib.hrbf_local_surfel_count.restype = C.c_uint32
    _lib = lib
    return lib


def rebalance_plan(counts):
    """hrbf_rebalance_plan: the even re-cut of contiguous shards -> (new_counts, [(src, dst, src_off, dst_off, len)])"""
    lib = load_library()
    c = np.ascontiguousarray(counts, np.uint32)
    G = len(c)
    new = np.zeros(G, np.uint32); moves = np.zeros((2 * G, 5), np.uint32); n = C.c_int(0)
    rc = lib.hrbf_rebalance_plan(_p(c), G, _p(new), _p(moves), C.byref(n))
    if rc != 0:
        raise HrbfError("hrbf_rebalance_plan failed with status %d" % rc)
    return new, [tuple(int(v) for v in m) for m in moves[:n.value]]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class HRBFFusion:
    """`HRBFFusion` (Core/src/HRBFFusion.h:82-534) on one MI355X."""

    def __init__(self, params=None, device=0):
        self.lib = load_library()
        self.params = params if params is not None else default_params()
        self.W, self.H = self.params.width, self.params.height
        h = C.c_void_p()
        self._check(self.lib.hrbf_create(C.byref(self.params), device, C.byref(h)))
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise HrbfError("hrbf status %d: %s" % (rc, self.lib.hrbf_last_error().decode()))
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.lib.hrbf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- primary entry (HRBFFusion.h:110-113)
    def process_frame(self, rgb, depth, ts=0, weight_multiplier=1.0):
        rgb = np.ascontiguousarray(rgb, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
        if rgb.size != self.W * self.H * 3 or depth.size != self.W * self.H:
            raise ValueError("frame size does not match the context resolution")
        return self._check(self.lib.hrbf_process_frame(self.h, _p(rgb), _p(depth), ts, weight_multiplier))

    def process_frame_device(self, d_rgb_ptr, d_depth_ptr, ts=0, weight_multiplier=1.0):
        return self._check(self.lib.hrbf_process_frame_device(self.h, C.c_void_p(d_rgb_ptr), C.c_void_p(d_depth_ptr),
                                                              ts, weight_multiplier))

    def synchronize(self):
        return self._check(self.lib.hrbf_synchronize(self.h))

    def upload_frame(self, rgb, depth):
        rgb = np.ascontiguousarray(rgb, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
        return self._check(self.lib.hrbf_upload_frame(self.h, _p(rgb), _p(depth)))

    def bootstrap(self, rgb, depth):
        rgb = np.ascontiguousarray(rgb, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
        return self._check(self.lib.hrbf_bootstrap(self.h, _p(rgb), _p(depth)))

    def run_stage(self, name):
        return self._check(self.lib.hrbf_run_stage(self.h, STAGES[name]))

    # -- getters (GUI/src/HRBF_fusion.cpp:235-497)
    def get_pose(self):
        o = np.zeros(16, np.float32)
        self._check(self.lib.hrbf_get_pose(self.h, _p(o)))
        return o.reshape(4, 4).T.copy()

    def set_pose(self, T):
        a = np.ascontiguousarray(np.asarray(T, np.float32).T)
        self._check(self.lib.hrbf_set_pose(self.h, _p(a)))

    @property
    def tick(self):
        return self.lib.hrbf_get_tick(self.h)

    def set_tick(self, t):
        self._check(self.lib.hrbf_set_tick(self.h, t))

    def set_weighting(self, w):
        self._check(self.lib.hrbf_set_weighting(self.h, w))

    # row-sharded registration over RCCL; rank < 0 = virtual ranks in one process (test hook)
    @staticmethod
    def comm_unique_id():
        lib = load_library()
        buf = (C.c_uint8 * 128)()
        if lib.hrbf_comm_unique_id(buf) != 0:
            raise HrbfError(lib.hrbf_last_error().decode())
        return bytes(buf)

    def comm_init(self, rank, world, unique_id=None):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        self._check(self.lib.hrbf_comm_init(self.h, int(rank), int(world), buf))

    # the sharded map without RCCL: rendezvous through a POSIX shared-memory segment, images peer-mapped with hipIpcMemHandle;
    # also works with several ranks on ONE GPU (which RCCL refuses), which is how the path is tested on a single device
    @staticmethod
    def peer_unique_id():
        lib = load_library()
        buf = (C.c_uint8 * 128)()
        if lib.hrbf_peer_unique_id(buf) != 0:
            raise HrbfError(lib.hrbf_last_error().decode())
        return bytes(buf)

    @staticmethod
    def peer_release_id(unique_id):
        """remove a rendezvous segment that will not be used (rank 0's context does it otherwise); False if it is already gone"""
        lib = load_library()
        return lib.hrbf_peer_release_id((C.c_uint8 * 128).from_buffer_copy(unique_id)) == 0

    def comm_init_peer(self, rank, world, unique_id):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.lib.hrbf_comm_init_peer(self.h, int(rank), int(world), buf))

    # submap bookkeeping + rigid map correction (GlobalModel::updateModel), SURVEY §8f-3
    # GlobalModel / IndexMap operators under the reference's names (explicit pose / time / cut-offs)
    @staticmethod
    def _pose_arg(T):
        return None if T is None else _p(np.ascontiguousarray(np.asarray(T, np.float32).T.reshape(-1)))

    def initialise(self, init_pose=None):
        self._check(self.lib.hrbf_initialise(self.h, self._pose_arg(init_pose)))

    def predict_indices(self, pose=None, time=0, depth_cutoff=0.0, index_submap=-1):
        self._check(self.lib.hrbf_predict_indices(self.h, self._pose_arg(pose), int(time), float(depth_cutoff), int(index_submap)))

    def fuse(self, pose=None, time=0, depth_cutoff=0.0, index_submap=-1):
        self._check(self.lib.hrbf_fuse(self.h, self._pose_arg(pose), int(time), float(depth_cutoff), int(index_submap)))

    def clean(self, pose=None, time=0, conf_threshold=-1.0, max_depth=0.0):
        self._check(self.lib.hrbf_clean(self.h, self._pose_arg(pose), int(time), float(conf_threshold), float(max_depth)))

    def predict_hrbf(self):
        self._check(self.lib.hrbf_predict_hrbf(self.h))

    def dense_enough(self):
        """Resize::vertex + HRBFFusion::denseEnough on the current PRED_VERTEX image (HRBFFusion.cpp:974-988, 1069-1070)"""
        d = C.c_int(0)
        self._check(self.lib.hrbf_dense_enough(self.h, C.byref(d)))
        return bool(d.value)

    def map_shard_init(self, enable=True, partition="ranges"):
        """cut the surfel map over the ranks of comm_init (SURVEY §8e sharding 2); the map must be empty.
        partition: "ranges" = contiguous ranges of the global order, "hash" = ownership by spatial hash of the surfel's cell"""
        mode = 0 if not enable else (2 if partition == "hash" else 1)
        self._check(self.lib.hrbf_map_shard_init(self.h, mode))

    def comm_stats(self, reset=False):
        """what the library's own communicator is and what the sharded paths have issued since the last reset (hrbf_comm_stats):
        a dict of the hrbf_comm_counters fields, `transport` as a name"""
        o = CommCounters()
        self._check(self.lib.hrbf_comm_stats(self.h, C.byref(o), 1 if reset else 0))
        d = {n: int(getattr(o, n)) for n, _ in CommCounters._fields_}
        d["transport"] = TRANSPORTS.get(d["transport"], str(d["transport"]))
        return d

    def shard_counts(self):
        """(partition, counts): partition 0 one map | 1 contiguous ranges | 2 spatial hash; live surfel counts of the G shards"""
        o = np.zeros(8, np.uint32)
        mode = self.lib.hrbf_shard_counts(self.h, _p(o))
        if mode < 0:
            self._check(mode)
        return mode, o

    def fit_curvature(self, window=2, support=1.25, ridge=1e-6, jump=3.0, timed=False):
        """EXTENSION (no reference counterpart): true Hermite-RBF fit per pixel on the matrix core -> images FIT_CURV1 / FIT_CURV2 /
        FIT_NORMAL; returns the kernel time in ms when timed"""
        ms = C.c_float(0.0)
        self._check(self.lib.hrbf_fit_curvature(self.h, int(window), float(support), float(ridge), float(jump), C.byref(ms) if timed else None))
        return float(ms.value) if timed else None

    def set_hrbf_fit(self, enable):
        """EXTENSION, off by default: process_frame takes the live frame's curvatures from the true Hermite-RBF fit (results are then not the reference's)"""
        self._check(self.lib.hrbf_set_hrbf_fit(self.h, int(bool(enable))))

    def set_hrbf_fit_params(self, window=2, support=1.25, ridge=0.1, jump=3.0):
        self._check(self.lib.hrbf_set_hrbf_fit_params(self.h, int(window), float(support), float(ridge), float(jump)))

    def get_hrbf_fit(self):
        """(enabled, window, support, ridge, jump) of the in-frame Hermite-RBF fit option (an extension: off by default)"""
        e, w = C.c_int(0), C.c_int(0)
        s_, r, j = C.c_float(0), C.c_float(0), C.c_float(0)
        self._check(self.lib.hrbf_get_hrbf_fit(self.h, C.byref(e), C.byref(w), C.byref(s_), C.byref(r), C.byref(j)))
        return bool(e.value), w.value, s_.value, r.value, j.value

    def gn_graph_captures(self):
        """how often the Gauss-Newton loop was captured into a hipGraph (2 in a steady run: one per image parity)"""
        return int(self.lib.hrbf_gn_graph_captures(self.h))

    def hash_renumber_count(self):
        return int(self.lib.hrbf_hash_renumber_count(self.h))

    def shard_exchange_mode(self):
        """0 not sharded over ranks, 1 peer-mapped images, 2 packed records on request, 3 packed records agreed by the ranks"""
        return int(self.lib.hrbf_shard_exchange_mode(self.h))

    def download_gids(self):
        """hash ownership, one shard per rank: the global-order ids of the rank's surfels (order of download_map)"""
        n = self.local_surfel_count()
        o = np.zeros(n, np.uint32)
        if n:
            self._check(self.lib.hrbf_download_gids(self.h, _p(o), n))
        return o

    def set_row_sharding(self, enable):
        self._check(self.lib.hrbf_set_row_sharding(self.h, int(bool(enable))))

    def map_rebalance(self):
        self._check(self.lib.hrbf_map_rebalance(self.h))

    def local_surfel_count(self):
        return int(self.lib.hrbf_local_surfel_count(self.h))

    def set_index_submap(self, idx):
        self._check(self.lib.hrbf_set_index_submap(self.h, int(idx)))

    def set_active_submaps(self, active):
        """byte mask indexed by submap id (IndexMap::lActiveKFID); None / empty = all active"""
        a = np.ascontiguousarray(np.asarray([] if active is None else active, np.uint8))
        self._check(self.lib.hrbf_set_active_submaps(self.h, _p(a) if a.size else None, int(a.size)))

    def update_model(self, deltas):
        """deltas: (n, 4, 4) row-major numpy matrices, one rigid correction per submap id"""
        d = np.ascontiguousarray(np.asarray(deltas, np.float32).reshape(-1, 4, 4).transpose(0, 2, 1))   # -> column-major
        self._check(self.lib.hrbf_update_model(self.h, _p(d), int(d.shape[0])))

    def get_weighting(self):
        w = C.c_float()
        self._check(self.lib.hrbf_last_weighting(self.h, C.byref(w)))
        return w.value

    def surfel_count(self):
        return int(self.lib.hrbf_surfel_count(self.h))

    def download_map(self):
        """the local range(s) in global order: the whole map unless this context is one rank of a sharded map"""
        n = self.local_surfel_count()
        o = np.zeros((n, 20), np.float32)
        if n:
            self._check(self.lib.hrbf_download_map(self.h, _p(o), n))
        return o

    def upload_map(self, m):
        m = np.ascontiguousarray(m, np.float32)
        self._check(self.lib.hrbf_upload_map(self.h, _p(m), m.shape[0]))

    def get_image(self, name):
        i, dt, ch = IMAGES[name] if name in IMAGES else EXT_IMAGES[name]
        shape = (self.H, self.W, ch) if ch > 1 else (self.H, self.W)
        o = np.zeros(shape, np.dtype(dt))
        self._check(self.lib.hrbf_get_image(self.h, i, _p(o), o.nbytes))
        return o

    def set_image(self, name, a):
        i, dt, ch = IMAGES[name]
        a = np.ascontiguousarray(a, np.dtype(dt))
        self._check(self.lib.hrbf_set_image(self.h, i, _p(a), a.nbytes))

    def last_icp(self):
        e = C.c_float(); n = C.c_float()
        self._check(self.lib.hrbf_last_icp(self.h, C.byref(e), C.byref(n)))
        return e.value, n.value

    def fuse_stats(self):
        o = np.zeros(4, np.uint32)
        self._check(self.lib.hrbf_get_fuse_stats(self.h, _p(o)))
        return o

    def fuse_ring(self, max_frames=1024):
        ms = np.zeros(max_frames, np.float32); st = np.zeros((max_frames, 4), np.uint32)
        n = self.lib.hrbf_get_fuse_ring(self.h, max_frames, _p(ms), _p(st))
        if n < 0:
            raise HrbfError("hrbf_get_fuse_ring failed")
        return ms[:n].copy(), st[:n].copy()

    def fuse_ring_parts(self, max_frames=1024):
        """(merge_ms, stream_ms, stats8) per frame: F2 = k_apply_merges, F3 = k_clean_flags + k_fuse_stream;
        stats8 = {in, merged, appended, out, -, -, moved, status}"""
        mm = np.zeros(max_frames, np.float32); ms = np.zeros(max_frames, np.float32); st = np.zeros((max_frames, 8), np.uint32)
        n = self.lib.hrbf_get_fuse_ring_parts(self.h, max_frames, _p(mm), _p(ms), _p(st))
        if n < 0:
            raise HrbfError("hrbf_get_fuse_ring_parts failed")
        return mm[:n].copy(), ms[:n].copy(), st[:n].copy()

    def probe_single_workgroup_iteration(self, level=2, iters=4):
        """ms one 256-thread workgroup needs for the pixel work of `iters` Gauss-Newton iterations of `level`"""
        ms = C.c_float()
        self._check(self.lib.hrbf_probe_single_workgroup_iteration(self.h, level, iters, C.byref(ms)))
        return ms.value

    def probe_sqrt_rounding(self):
        """exhaustive device-side check of k_predict_hrbf's square-root shortcut, see hrbf_probe_sqrt_rounding"""
        o = np.zeros(6, np.uint64)
        self._check(self.lib.hrbf_probe_sqrt_rounding(self.h, _p(o)))
        return o

    def probe_exp_scaling(self):
        """(mismatches, cases) of v_ldexp_f32 against hd_expf's two-step scaling, see hrbf_probe_exp_scaling"""
        o = np.zeros(2, np.uint64)
        self._check(self.lib.hrbf_probe_exp_scaling(self.h, _p(o)))
        return o

    def set_fuse_ring_stride(self, every_nth_frame):
        """record the fuse ring (four event records + a statistics copy, ~22 us) only every n-th frame"""
        self._check(self.lib.hrbf_set_fuse_ring_stride(self.h, int(every_nth_frame)))

    def probe_division(self):
        """(mismatches, cases) of k_curvature's unscaled division against the compiler's, see hrbf_probe_division"""
        o = np.zeros(2, np.uint64)
        self._check(self.lib.hrbf_probe_division(self.h, _p(o)))
        return o

    def frames_completed(self):
        """frames whose pose has landed in the pinned trajectory ring (never blocks)"""
        return int(self.lib.hrbf_frames_completed(self.h))

    def pose_log(self, first=0, count=None, wait=True):
        """poses (n, 4, 4) of frames [first, first + count) from the device-written trajectory ring"""
        if count is None:
            count = int(self.lib.hrbf_frames_enqueued(self.h)) - first
        out = np.zeros((max(count, 1), 16), np.float32)
        n = self.lib.hrbf_get_pose_log(self.h, first, count, _p(out), int(bool(wait)))
        if n < 0:
            raise HrbfError(self.lib.hrbf_last_error().decode())
        return out[:n].reshape(n, 4, 4).transpose(0, 2, 1).copy()

    STATUS_CAPACITY, STATUS_INTERNAL_BOUND, STATUS_SO3_TIMEOUT, STATUS_FUSE_TIMEOUT = 1, 2, 4, 8

    def status(self, clear=False):
        """sticky condition bits (HRBF_STATUS_*); synchronises"""
        v = np.zeros(1, np.uint32)
        self._check(self.lib.hrbf_get_status(self.h, _p(v), int(bool(clear))))
        return int(v[0])

    def reset_fuse_ring(self):
        self.lib.hrbf_reset_fuse_ring(self.h)

    def set_load_trajectory(self, v):
        self._check(self.lib.hrbf_set_load_trajectory(self.h, int(v)))

    def enable_timing(self, on=True):
        """False/0: off, True/1: Stopwatch regions + fuse ring, 2: fuse ring only"""
        self._check(self.lib.hrbf_enable_timing(self.h, int(on)))

    def timings(self):
        o = np.zeros(8, np.float32)
        self._check(self.lib.hrbf_get_timings(self.h, _p(o)))
        return o

    # live-tunable setters (GUI/src/HRBF_fusion.cpp:448-456)
    def set_rgb_only(self, v): self._check(self.lib.hrbf_set_rgb_only(self.h, int(v)))
    def set_icp_weight(self, v): self._check(self.lib.hrbf_set_icp_weight(self.h, float(v)))
    def set_pyramid(self, v): self._check(self.lib.hrbf_set_pyramid(self.h, int(v)))
    def set_fast_odom(self, v): self._check(self.lib.hrbf_set_fast_odom(self.h, int(v)))
    def set_so3(self, v): self._check(self.lib.hrbf_set_so3(self.h, int(v)))
    def set_frame_to_frame_rgb(self, v): self._check(self.lib.hrbf_set_frame_to_frame_rgb(self.h, int(v)))
    def set_confidence_threshold(self, v): self._check(self.lib.hrbf_set_confidence_threshold(self.h, float(v)))
    def set_depth_cutoff(self, v): self._check(self.lib.hrbf_set_depth_cutoff(self.h, float(v)))

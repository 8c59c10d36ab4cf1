"""ctypes binding of the CPU oracle (oracle/_build/liboracle*.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess
import numpy as np

from hrbffusion3d_amd.params import HrbfParams, IMAGES, STAGES

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ODIR = os.path.join(_ROOT, "oracle")


def build():
    subprocess.check_call(["make", "-s", "-C", _ODIR], stdout=subprocess.DEVNULL)


def load(omp=False):
    # the OpenMP build spawns one thread per visible CPU by default; on a 256-CPU GPU box that made a QVGA frame take 7.5 s
    # instead of 0.05 s (measured).  Bound it before libgomp initialises, unless the caller has chosen a number.
    if omp and "OMP_NUM_THREADS" not in os.environ:
        os.environ["OMP_NUM_THREADS"] = str(max(1, min(16, len(os.sched_getaffinity(0)))))
    name = "liboracle_omp.so" if omp else "liboracle.so"
    path = os.path.join(_ODIR, "_build", name)
    # tools/mutation_report.py only: a deliberately MISREAD oracle (oracle/Makefile `mutants`), to show that the metamorphic tests fail on it
    if os.environ.get("HRBF_ORACLE_MUTANT"):
        path = os.path.join(_ODIR, "_build", "liboracle_mutant_%d.so" % int(os.environ["HRBF_ORACLE_MUTANT"]))
    # tests/test_oracle_sanitized.py only: the ASan + UBSan build (make -C oracle san; the process runs under LD_PRELOAD=libasan)
    if os.environ.get("HRBF_ORACLE_SAN"):
        path = os.path.join(_ODIR, "_build", "liboracle_san.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.orc_create.restype = C.c_void_p
    lib.orc_create.argtypes = [C.POINTER(HrbfParams)]
    for fn in ("orc_destroy",):
        getattr(lib, fn).argtypes = [C.c_void_p]
    lib.orc_process_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float]
    lib.orc_upload_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_run_stage.argtypes = [C.c_void_p, C.c_int]
    lib.orc_bootstrap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_get_pose.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_set_pose.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_get_tick.argtypes = [C.c_void_p]
    lib.orc_set_tick.argtypes = [C.c_void_p, C.c_int]
    lib.orc_set_weighting.argtypes = [C.c_void_p, C.c_float]
    lib.orc_set_fragment_texcoords.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_get_odo_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_get_pyramid.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]; lib.orc_get_pyramid.restype = C.c_size_t
    lib.orc_set_index_submap.argtypes = [C.c_void_p, C.c_int]
    lib.orc_set_switch.argtypes = [C.c_void_p, C.c_int, C.c_float]
    lib.orc_set_active_submaps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_update_model.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_so3_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6
    lib.orc_rgb_residual.argtypes = [C.c_float] + [C.c_void_p] * 6 + [C.c_int, C.c_int] + [C.c_void_p] * 6
    lib.orc_rgb_step.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_get_weighting.argtypes = [C.c_void_p]
    lib.orc_dense_enough.argtypes = [C.c_void_p]
    lib.orc_get_weighting.restype = C.c_float
    lib.orc_surfel_count.argtypes = [C.c_void_p]
    lib.orc_surfel_count.restype = C.c_uint32
    lib.orc_download_map.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.orc_upload_map.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.orc_image_bytes.argtypes = [C.c_void_p, C.c_int]
    lib.orc_image_bytes.restype = C.c_size_t
    lib.orc_get_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    lib.orc_set_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    lib.orc_last_icp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_get_fuse_stats.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_get_timings.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_hrbf_value.restype = C.c_float
    lib.orc_hrbf_value.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.orc_hrbf_gradient.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.orc_hrbf_hessian.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.orc_expf.restype = C.c_float; lib.orc_expf.argtypes = [C.c_float]
    lib.orc_f2i.restype = C.c_int; lib.orc_f2i.argtypes = [C.c_float]
    lib.orc_f2u.restype = C.c_uint; lib.orc_f2u.argtypes = [C.c_float]
    lib.orc_d2l.restype = C.c_longlong; lib.orc_d2l.argtypes = [C.c_double]
    lib.orc_encode_color.restype = C.c_float; lib.orc_encode_color.argtypes = [C.c_float] * 3
    lib.orc_acosf.restype = C.c_float; lib.orc_acosf.argtypes = [C.c_float]
    lib.orc_atan2f.restype = C.c_float; lib.orc_atan2f.argtypes = [C.c_float, C.c_float]
    lib.orc_sincosf.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
    lib.orc_sincos.argtypes = [C.c_double, C.c_void_p, C.c_void_p]
    lib.orc_acos.restype = C.c_double; lib.orc_acos.argtypes = [C.c_double]
    lib.orc_acc_test.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.orc_solve6.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_icp_step_sparse.argtypes = [C.c_void_p] * 6 + [C.c_void_p] * 2 + [C.c_float] * 4 + [C.c_void_p] * 5 + \
        [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int] + [C.c_void_p] * 6
    lib.orc_update_lambda_map.argtypes = [C.c_void_p] * 9 + [C.c_int, C.c_int]
    lib.orc_icp_step_search.argtypes = [C.c_void_p] * 6 + [C.c_void_p] * 2 + [C.c_float] * 4 + [C.c_void_p] * 5 + \
        [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4
    lib.orc_sparse_shrink_factor.restype = C.c_float; lib.orc_sparse_shrink_factor.argtypes = [C.c_float]
    lib.orc_sparse_shrunk_count.restype = C.c_int64; lib.orc_sparse_shrunk_count.argtypes = [C.c_void_p]
    lib.orc_icp_step.argtypes = [C.c_void_p] * 6 + [C.c_void_p] * 2 + [C.c_float] * 4 + [C.c_void_p] * 5 + \
        [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Same method names as hrbffusion3d_amd.api.HRBFFusion so parity tests read symmetrically."""

    def __init__(self, params, omp=False):
        self.lib = load(omp)
        self.params = params
        self.W, self.H = params.width, params.height
        self.h = self.lib.orc_create(C.byref(params))

    def close(self):
        if self.h:
            self.lib.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_frame(self, rgb, depth, ts=0, weight_multiplier=1.0):
        rgb = np.ascontiguousarray(rgb, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
        assert rgb.size == self.W * self.H * 3 and depth.size == self.W * self.H
        return self.lib.orc_process_frame(self.h, _p(rgb), _p(depth), ts, weight_multiplier)

    def upload_frame(self, rgb, depth):
        rgb = np.ascontiguousarray(rgb, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
        return self.lib.orc_upload_frame(self.h, _p(rgb), _p(depth))

    def bootstrap(self, rgb, depth):
        rgb = np.ascontiguousarray(rgb, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
        return self.lib.orc_bootstrap(self.h, _p(rgb), _p(depth))

    def run_stage(self, name):
        r = self.lib.orc_run_stage(self.h, STAGES[name])
        assert r == 0
        return r

    def get_pose(self):
        o = np.zeros(16, np.float32); self.lib.orc_get_pose(self.h, _p(o)); return o.reshape(4, 4).T.copy()

    def set_pose(self, T):
        a = np.ascontiguousarray(np.asarray(T, np.float32).T); self.lib.orc_set_pose(self.h, _p(a))

    @property
    def tick(self):
        return self.lib.orc_get_tick(self.h)

    def set_tick(self, t):
        self.lib.orc_set_tick(self.h, t)

    def set_weighting(self, w):
        self.lib.orc_set_weighting(self.h, w)

    PYRAMIDS = {"vmap_g": 0, "nmap_g": 1, "ck1_g": 2, "ck2_g": 3, "vmap_c": 4, "nmap_c": 5, "ck1_c": 6, "ck2_c": 7, "icpw": 8, "last_depth": 9,
                "next_depth": 10, "last_image": 11, "next_image": 12, "prev_image": 13, "dIdx": 14, "dIdy": 15}

    def odo_trace(self):
        """test hook: one row of 128 doubles per SO3 / Gauss-Newton iteration of the last registration (oracle.h)"""
        out = np.zeros((40, 128), np.float64)
        n = self.lib.orc_get_odo_trace(self.h, _p(out), 40)
        return out[:n]

    def pyramid(self, name, level):
        """test hook: a level of the registration pyramids; maps come back as (rows, cols, 4)"""
        which = self.PYRAMIDS[name]
        r, c = self.H >> level, self.W >> level
        if which < 8:
            a = np.zeros((4, r, c), np.float32)
        else:
            a = np.zeros((r, c), np.float32 if which <= 10 else (np.uint8 if which <= 13 else np.int16))
        assert self.lib.orc_get_pyramid(self.h, which, level, _p(a), a.nbytes) == a.nbytes
        return np.ascontiguousarray(np.moveaxis(a, 0, -1)) if which < 8 else a

    def set_fragment_texcoords(self, tc):
        """test hook: the texcoord a rasteriser interpolated for every pixel ((H, W, 2) float32), None = correctly rounded"""
        if tc is None:
            self.lib.orc_set_fragment_texcoords(self.h, None)
            return
        tc = np.ascontiguousarray(tc, np.float32)
        assert tc.shape == (self.H, self.W, 2)
        assert self.lib.orc_set_fragment_texcoords(self.h, _p(tc)) == 0

    def set_index_submap(self, idx):
        self.lib.orc_set_index_submap(self.h, int(idx))

    SWITCHES = ("rgb_only", "icp_weight", "pyramid", "fast_odom", "so3", "frame_to_frame_rgb", "confidence_threshold", "depth_cutoff")

    def set_switch(self, name, v):
        """the boundary's run-time setters (HRBFFusion.set_<name> on the library's side)"""
        self.lib.orc_set_switch(self.h, self.SWITCHES.index(name), float(v))

    def set_active_submaps(self, active):
        a = np.ascontiguousarray(np.asarray([] if active is None else active, np.uint8))
        self.lib.orc_set_active_submaps(self.h, _p(a) if a.size else None, int(a.size))

    def update_model(self, deltas):
        d = np.ascontiguousarray(np.asarray(deltas, np.float32).reshape(-1, 4, 4).transpose(0, 2, 1))
        self.lib.orc_update_model(self.h, _p(d), int(d.shape[0]))

    def get_weighting(self):
        return self.lib.orc_get_weighting(self.h)

    def surfel_count(self):
        return int(self.lib.orc_surfel_count(self.h))

    def sparse_shrunk_count(self):
        return int(self.lib.orc_sparse_shrunk_count(self.h))

    def download_map(self):
        n = self.surfel_count()
        o = np.zeros((n, 20), np.float32)
        if n:
            assert self.lib.orc_download_map(self.h, _p(o), n) == 0
        return o

    def upload_map(self, m):
        m = np.ascontiguousarray(m, np.float32)
        assert self.lib.orc_upload_map(self.h, _p(m), m.shape[0]) == 0

    def get_image(self, name):
        i, dt, ch = IMAGES[name]
        shape = (self.H, self.W, ch) if ch > 1 else (self.H, self.W)
        o = np.zeros(shape, np.dtype(dt))
        assert self.lib.orc_get_image(self.h, i, _p(o), o.nbytes) == 0
        return o

    def set_image(self, name, a):
        i, dt, ch = IMAGES[name]
        a = np.ascontiguousarray(a, np.dtype(dt))
        assert self.lib.orc_set_image(self.h, i, _p(a), a.nbytes) == 0

    def dense_enough(self):
        return bool(self.lib.orc_dense_enough(self.h))

    def last_icp(self):
        e = C.c_float(); n = C.c_float()
        self.lib.orc_last_icp(self.h, C.byref(e), C.byref(n)); return e.value, n.value

    def fuse_stats(self):
        o = np.zeros(4, np.uint32); self.lib.orc_get_fuse_stats(self.h, _p(o)); return o

    def timings(self):
        o = np.zeros(8, np.float64); self.lib.orc_get_timings(self.h, _p(o)); return o

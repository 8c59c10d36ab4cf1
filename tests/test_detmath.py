"""hrbf_detmath.h: accuracy of the deterministic transcendentals vs libm and exactness of the
order-independent accumulator (CPU side; the GPU side is covered by test_parity_gpu.py)."""
import ctypes as C
import math
from fractions import Fraction

import numpy as np
import pytest


def _ulp(ref):
    r = np.float32(ref)
    return abs(float(np.nextafter(r, np.float32(np.inf))) - float(r)) or 1e-45


def test_expf_accuracy(oracle_lib_built):
    lib = oracle_lib_built.load()
    xs = np.linspace(-87.0, 88.0, 20001).astype(np.float32)
    worst = max(abs(lib.orc_expf(float(x)) - math.exp(float(x))) / _ulp(math.exp(float(x))) for x in xs)
    assert worst <= 4.0
    assert lib.orc_expf(0.0) == 1.0
    assert lib.orc_expf(-200.0) == 0.0
    assert math.isinf(lib.orc_expf(100.0))


def test_float_to_int_is_defined_for_nan_and_out_of_range(oracle_lib_built):
    """hd_cvt_i32: what C, GLSL and CUDA leave undefined is stated (NaN -> 0, saturating: the GPUs' conversion, not x86's INT_MIN), and the
    colour word of a surfel merged at total confidence 0 (every channel 0 / 0) is therefore 0 on both sides (found by
    tests/gpu_fuzz_params.py, draw 13 of seed 31: the oracle said -2^31, the kernel 0)"""
    lib = oracle_lib_built.load()
    nan, inf = float("nan"), float("inf")
    assert [lib.orc_f2i(x) for x in (nan, -nan, inf, -inf, 3e9, -3e9, 2147483520.0, -2147483648.0)] == \
        [0, 0, 2147483647, -2147483648, 2147483647, -2147483648, 2147483520, -2147483648]
    assert [lib.orc_f2i(x) for x in (0.0, -0.0, 2.9, -2.9, 255.0, 16777215.0)] == [0, 0, 2, -2, 255, 16777215]      # truncation, as (int)
    assert lib.orc_encode_color(nan, nan, nan) == 0.0
    # the quadrant count of hd_sincos (found by the oracle under UBSan: an SE3 step of 1.9e20 rad solved from garbage images)
    assert [lib.orc_d2l(x) for x in (nan, inf, -inf, 1.9e20, -1.9e20, 9.2233720368547e18, -7.9, 2.0 ** 62)] == \
        [0, 2 ** 63 - 1, -2 ** 63, 2 ** 63 - 1, -2 ** 63, 9223372036854700032, -7, 2 ** 62]
    s, c = C.c_double(), C.c_double()
    for x in (1.9e20, -3e300, 2.0 ** 80):            # no angle any more, but a defined pair of finite numbers, the same on every call
        lib.orc_sincos(x, C.byref(s), C.byref(c)); first = (s.value, c.value)
        lib.orc_sincos(x, C.byref(s), C.byref(c))
        assert (s.value, c.value) == first
    # uint(): a submap id, an init time (found by tests/gpu_fuzz_stages.py: -1.0 was 0xFFFFFFFF on the host, 1e30 was 0)
    assert [lib.orc_f2u(x) for x in (nan, -1.0, -0.0, -inf, 0.9, 7.0, 4294967040.0, 4294967296.0, 1e30, inf)] == \
        [0, 0, 0, 0, 0, 7, 4294967040, 4294967295, 4294967295, 4294967295]
    assert lib.orc_encode_color(1.0, 0.5, 0.0) == float((255 << 16) + (128 << 8))       # rint: 127.5 -> 128 (ties to even)
    assert lib.orc_encode_color(nan, 1.0, nan) == float(255 << 8)


def test_acosf_atan2f_accuracy(oracle_lib_built):
    lib = oracle_lib_built.load()
    xs = np.linspace(-1.0, 1.0, 20001).astype(np.float32)
    worst = max(abs(lib.orc_acosf(float(x)) - math.acos(float(x))) / _ulp(math.acos(float(x)) or 1.0) for x in xs[1:-1])
    assert worst <= 4.0
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1, 1, (20000, 2)).astype(np.float32)
    worst = max(abs(lib.orc_atan2f(float(y), float(x)) - math.atan2(float(y), float(x))) /
                _ulp(math.atan2(float(y), float(x))) for y, x in pts)
    assert worst <= 4.0


def test_sincos(oracle_lib_built):
    lib = oracle_lib_built.load()
    s = C.c_float(); c = C.c_float()
    for x in np.linspace(-20, 20, 4001).astype(np.float32):
        lib.orc_sincosf(float(x), C.byref(s), C.byref(c))
        assert abs(s.value - math.sin(float(x))) < 2e-7 and abs(c.value - math.cos(float(x))) < 2e-7
    sd = C.c_double(); cd = C.c_double()
    for x in np.linspace(-20, 20, 4001):
        lib.orc_sincos(float(x), C.byref(sd), C.byref(cd))
        assert abs(sd.value - math.sin(x)) < 5e-16 and abs(cd.value - math.cos(x)) < 5e-16
    for x in np.linspace(-1, 1, 4001):
        assert abs(lib.orc_acos(float(x)) - math.acos(x)) < 1e-15


def test_accumulator_is_exact_and_order_independent(oracle_lib_built):
    lib = oracle_lib_built.load()
    rng = np.random.default_rng(1)
    v = (rng.standard_normal(50000) * np.exp(rng.uniform(-20, 20, 50000))).astype(np.float32)
    v[::17] = -v[::17] * 1e6
    out = C.c_double()

    def acc(a):
        a = np.ascontiguousarray(a, np.float32)
        lib.orc_acc_test(a.ctypes.data_as(C.c_void_p), a.size, C.byref(out))
        return out.value

    a = acc(v)
    assert a == acc(v[::-1]) == acc(rng.permutation(v))          # bitwise equal under any order
    # equals the exact rational sum of round_half_even(p * 2^40) / 2^40, rounded once to double
    q = sum(int(round(Fraction(float(x)) * (1 << 40))) for x in v[:4000])   # python round = half-even
    assert acc(v[:4000]) == float(Fraction(q, 1 << 40))


def test_accumulator_cancellation(oracle_lib_built):
    lib = oracle_lib_built.load()
    out = C.c_double()
    a = np.array([1e20, 3.5, -1e20, -5.0, 2.0 ** -30], np.float32)
    lib.orc_acc_test(a.ctypes.data_as(C.c_void_p), a.size, C.byref(out))
    assert out.value == -1.5 + 2.0 ** -30


# ---- the literal-rule helpers (hd_tap_texel, hd_window_axis, hd_halfpixel_walk, hd_gl_point_window_coord) -------------------
# Oracle and kernels SHARE these (include/hrbf_detmath.h), so bit equality between the two cannot see an error in them; the executed
# shader fixtures can (tests/test_ref_glsl.py, since round 4 at 640 x 480).  Here each rule is additionally held to an independent
# numpy emulation of the shader's fp32 arithmetic at EVERY column / row of the sizes in use.
SIZES = [640, 480, 1280, 960, 320, 240, 160, 120, 512, 424, 848, 256, 128]


def _lib(oracle_lib_built):
    lib = oracle_lib_built.load()
    lib.orc_tap_texel.argtypes = [C.c_int, C.c_int]
    lib.orc_window_samples.argtypes = [C.c_float, C.c_int, C.c_float, C.c_void_p]
    lib.orc_halfpixel_walk_samples.argtypes = [C.c_float, C.c_int, C.c_float, C.c_void_p]
    lib.orc_gl_point_window_coord.argtypes = [C.c_float, C.c_float, C.c_void_p]; lib.orc_gl_point_window_coord.restype = C.c_float
    return lib


def test_tap_texel_is_floor_of_the_fp32_product_at_every_row(oracle_lib_built):
    """depth_bilateral.frag:51-54: texture(s, float(c) / n) with NEAREST reads texel floor(fl(fl(c / n) * n)) — c - 1 at rows
    {63, 125, 126, 127, 250, 252, 254} of a 480-high image (what llvmpipe AND softpipe execute), c at 640 / 512 / 256"""
    lib = _lib(oracle_lib_built)
    f = np.float32
    low = {}
    for n in SIZES:
        c = np.arange(n, dtype=f)
        want = np.clip(np.floor((c / f(n)) * f(n)), 0, n - 1).astype(int)
        got = np.array([lib.orc_tap_texel(int(k), n) for k in range(n)])
        assert np.array_equal(got, want), n
        low[n] = np.nonzero(got != np.arange(n))[0].tolist()
    assert low[480] == [63, 125, 126, 127, 250, 252, 254] and low[120] == [63] and low[640] == [] and low[512] == [] and low[256] == []


def _np_window(t, n, win):
    f = np.float32
    s_ = f(1) / f(n)
    lo, hi = max(f(0), f(t - f(s_ * f(win)))), min(f(1), f(t + f(s_ * f(win))))
    i, out = lo, []
    while i <= hi:
        out.append(int(min(max(np.floor(f(i * f(n))), 0), n - 1))); i = f(i + s_)
    return out


@pytest.mark.parametrize("win", [3.0, 2.0])
def test_window_axis_is_the_shaders_float_stepped_loop_at_every_pixel(oracle_lib_built, win):
    """geometry.glsl:198-207 / depth_curvature_gradient.frag:54-63: for (i = max(0, t - s win); i <= min(1, t + s win); i += s),
    texel floor(i * n), t the correctly rounded fragment texcoord AND the host's uv attribute (data.vert's coordinate)"""
    lib = _lib(oracle_lib_built)
    f = np.float32
    buf = (C.c_int * 16)()
    short = {}
    for n in SIZES:
        cnt = []
        for p in range(n):
            for which, t in enumerate((f((f(p) + f(0.5)) / f(n)), f(np.float64(f(p) / f(n)) + 1.0 / float(2 * f(n))))):
                k = lib.orc_window_samples(float(t), n, win, buf)
                want = _np_window(t, n, win)
                assert k == len(want) and list(buf[:k]) == want, (n, p, float(t), k, want)
                if which == 0:
                    cnt.append(k)                 # under the fragment shaders' coordinate
        short[n] = int((np.array(cnt[8:-8]) < 2 * win + 1).sum())
    assert short[256] == 0 and short[128] == 0 and short[512] == 0           # exact at power-of-two sizes
    if win == 3.0:       # the LAST sample is dropped at about 40 % of the columns of a 640-wide and half of the rows of a 480-high image
        assert 230 < short[640] < 260 and 240 < short[480] < 265 and 70 < short[160] < 90, short


def test_halfpixel_walk_and_point_snap_follow_their_fp32_definitions(oracle_lib_built):
    """copy_unstable.vert:106-108: for (i = x / n - step wm; i < x / n + step wm; i += step), step = (1 / n) / 2 — 2 wm samples in
    exact arithmetic, one more for about a third of the positions in fp32; and the rasteriser's 1/256-pixel snap of a point's window
    coordinate (hd_gl_point_window_coord)"""
    lib = _lib(oracle_lib_built)
    f = np.float32
    buf = (C.c_int * 16)()
    rng = np.random.default_rng(3)
    for n in (640, 480, 160, 128):
        extra = 0
        xs = np.concatenate([rng.uniform(1, n - 1, 4000), np.arange(1, n - 1) + 0.5]).astype(f)
        for x in xs:
            for wm in (2.0, 1.0, 2.25):
                step = f(f(1) / f(n)) * f(0.5)
                reach = f(step * f(wm)); c = f(x / f(n))
                i, hi, want = f(c - reach), f(c + reach), []
                while i < hi:
                    want.append(int(min(max(np.floor(f(i * f(n))), 0), n - 1))); i = f(i + step)
                k = lib.orc_halfpixel_walk_samples(float(x), n, wm, buf)
                assert k == len(want) and list(buf[:k]) == want, (n, float(x), wm)
                if wm == 2.0:
                    extra += k == 5
        assert (extra == 0) == (n == 128), (n, extra)       # exact at the power-of-two size, a fifth sample elsewhere
    clip = C.c_int(0)
    for ext in (640.0, 480.0):
        u = rng.uniform(-2, ext + 2, 20000).astype(f)
        for v in u[:4000]:
            got = lib.orc_gl_point_window_coord(float(v), ext, C.byref(clip))
            half = f(ext) * f(0.5)
            ndc = f(f(v - half) / half)
            w = f(f(ndc * half) + half)
            want = f(np.floor(f(f(w * f(256)) + f(0.5))) * f(1.0 / 256.0))
            assert got == want and bool(clip.value) == (not (-1.0 <= ndc <= 1.0)), (float(v), got, want)

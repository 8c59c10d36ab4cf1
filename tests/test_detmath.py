"""hrbf_detmath.h: accuracy of the deterministic transcendentals vs libm and exactness of the
order-independent accumulator (CPU side; the GPU side is covered by test_parity_gpu.py)."""
import ctypes as C
import math
from fractions import Fraction

import numpy as np


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

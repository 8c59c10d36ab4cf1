/*
 * hrbf_detmath.h — the arithmetic contract of the hrbf-mi355 hot path.
 *
 * The reference (YabinXuTUD/HRBFFusion3D) evaluates exp/acos/atan/sin/cos in GLSL and
 * CUDA fast-math (Core/src/CMakeLists.txt:75: --ftz --prec-div=false --prec-sqrt=false), i.e.
 * with driver-defined precision, and reduces its ICP/RGB/SO3 normal equations in fp32 with a
 * launch-shape dependent summation order (Core/src/Cuda/reduce.cu:58-184).  Neither is
 * reproducible across devices.  This header pins both so that the CPU oracle (oracle/), one
 * MI355X and N MI355X produce bit-identical results:
 *
 *   1. hd_* transcendentals: fixed polynomial kernels built only from +,-,*,/ , sqrt and
 *      explicit fma — all IEEE correctly rounded on gfx950 and x86-64 when the translation
 *      unit is compiled with -ffp-contract=off (both build recipes do).  Accuracy (float
 *      versions <= 4 ulp over the ranges the path uses, double versions <= 5e-16 abs.) is
 *      checked against libm in tests/test_detmath.py.
 *   2. hd_acc128: an order-independent exact accumulator.  A float addend p contributes
 *      round_half_even(p * 2^40) to a 128-bit two's-complement integer, so sums are
 *      associative and commutative: wavefront shuffles, LDS trees, global atomics and RCCL
 *      all-reduce (as int64 limbs) give the same bits as a sequential CPU loop.
 *
 * Plain C99 / HIP dual-compilable.  No reference code is used here.
 */
#ifndef HRBF_DETMATH_H_
#define HRBF_DETMATH_H_

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define HD_FN __host__ __device__ __forceinline__
#else
#define HD_FN static inline
#endif

/* ---------------------------------------------------------------- bit casts */
HD_FN uint32_t hd_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
HD_FN float hd_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
HD_FN uint64_t hd_d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
HD_FN double hd_u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

HD_FN float hd_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
HD_FN double hd_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
HD_FN float hd_rintf(float a) { return __builtin_rintf(a); }
HD_FN double hd_rint(double a) { return __builtin_rint(a); }
HD_FN float hd_sqrtf(float a) { return __builtin_sqrtf(a); }
HD_FN double hd_sqrt(double a) { return __builtin_sqrt(a); }
HD_FN float hd_fabsf(float a) { return __builtin_fabsf(a); }
HD_FN float hd_floorf(float a) { return __builtin_floorf(a); }
/* float -> int where the operand can be NaN or out of range (the colour word of a surfel merged at total confidence 0: 0 / 0).  C and
 * GLSL leave that conversion undefined; x86 returns INT_MIN, the GPUs the reference runs on and v_cvt_i32_f32 return 0 for NaN and
 * saturate.  The contract takes the GPUs' result and states it, so that the host restatement does not inherit the host's. */
HD_FN int hd_cvt_i32(float a)
{
    return a != a ? 0 : (a >= 2147483648.0f ? 2147483647 : (a <= -2147483648.0f ? (-2147483647 - 1) : (int)a));
}
/* ... and float -> unsigned (GLSL uint(): a surfel's submap id, its init time — index_map.vert:41, predict_hrbf.frag:296): NaN and
 * negative values -> 0, saturating above (v_cvt_u32_f32; x86 goes through a 64-bit conversion and wraps: -1.0f -> 0xFFFFFFFF) */
HD_FN unsigned hd_cvt_u32(float a)
{
    return !(a > 0.0f) ? 0u : (a >= 4294967296.0f ? 0xFFFFFFFFu : (unsigned)a);
}
/* double -> int64 for the quadrant count of hd_sincos: an angle of 1e20 (a Gauss-Newton step on garbage input) is not an angle any
 * more, but the conversion must still be the same number on the host and on the device (found by the oracle under UBSan) */
HD_FN long long hd_cvt_i64(double a)
{
    return a != a ? 0 : (a >= 9223372036854775808.0 ? 9223372036854775807LL : (a <= -9223372036854775808.0 ? (-9223372036854775807LL - 1) : (long long)a));
}

/* One axis of the shaders' float-stepped window loops, literally (geometry.glsl:198-207 getNormalPCA,
 * depth_curvature_gradient.frag:54-63):
 *     float step = 1.0f / n;                    // indexXStep
 *     float lo = max(0.0f, t - step * win), hi = min(1.0f, t + step * win);
 *     for (float i = lo; i <= hi; i += step) sample(texel floor(i * n), position i * n)
 * with t = (p + 0.5) / n, the texture coordinate the rasteriser interpolates for pixel p (taken correctly rounded).  In exact
 * arithmetic that is 2 win + 1 samples; in fp32 the accumulated i overshoots hi by an ulp for about 40 % of the columns of a
 * 640-wide and 53 % of the rows of a 480-high image, and the LAST sample is then not taken — on any IEEE implementation.
 * The reference's shaders executed on Mesa llvmpipe (oracle/ref_glsl, profiles/r03_ref_glsl_vga_report.txt) show exactly this
 * pattern (this rule reproduces its window at all but 4 of 640 columns and 4 of 480 rows, where llvmpipe's interpolated t is an
 * ulp off the correctly rounded one), and the curvature it yields differs from the nominal 7 x 7 window's by 5 % in the median.
 * At power-of-two sizes every quantity above is exact and the loop takes the nominal samples.
 * Callers iterate `for (float i = w.lo; i <= w.hi; i += w.step)` and take texel hd_window_texel(i, n). */
typedef struct hd_window { float lo, hi, step; } hd_window;
HD_FN hd_window hd_window_axis_t(float t, int n, float win)
{
    hd_window w;
    const float fn = (float)n;
    w.step = 1.0f / fn;
    const float reach = w.step * win;
    const float lo = t - reach, hi = t + reach;
    w.lo = lo > 0.0f ? lo : 0.0f;
    w.hi = hi < 1.0f ? hi : 1.0f;
    return w;
}
HD_FN float hd_uv_fragment(int p, int n) { return ((float)p + 0.5f) / (float)n; }   /* a fragment shader's texcoord, correctly rounded */
HD_FN hd_window hd_window_axis(int p, int n, float win) { return hd_window_axis_t(hd_uv_fragment(p, n), n, win); }
/* The texture coordinate a VERTEX shader of the map passes receives for pixel p (data.vert, init_unstableTex.vert): an attribute
 * the reference's host computes once, `((float)i / (float)width) + 1.0 / (2 * (float)width)` stored as float
 * (GlobalModel.cpp:88-97) — a float quotient, a double sum, one more rounding.  It equals hd_uv_fragment at power-of-two sizes
 * and differs from it by an ulp at 171 of 640 columns and 139 of 480 rows; `x = texcoord.x * cols` (data.vert:66) is then not
 * exactly p + 0.5 at 103 / 117 of them, and the float-stepped loops that start from it (getNormalPCA in data.vert:88) drop
 * their last sample at other columns than the fragment shader's do.  Found by executing the shaders at 640 x 480
 * (tests/golden/make_ref_glsl.py --vga-map-report). */
/* x = texcoord.x * cols as the shaders form it (p + 0.5 exactly at power-of-two sizes) */
HD_FN float hd_px_fragment(int p, int n) { return hd_uv_fragment(p, n) * (float)n; }
HD_FN float hd_uv_attribute(int p, int n)
{
    const float fn = (float)n;
    return (float)((double)((float)p / fn) + 1.0 / (double)(2.0f * fn));
}
HD_FN float hd_px_attribute(int p, int n) { return hd_uv_attribute(p, n) * (float)n; }
HD_FN int hd_window_texel(float i, int n)   /* NEAREST filtering, CLAMP_TO_EDGE */
{
    int t = hd_cvt_i32(hd_floorf(i * (float)n));
    return t < 0 ? 0 : (t > n - 1 ? n - 1 : t);
}

/* The texel a NEAREST lookup at texture coordinate float(c) / n reads (depth_bilateral.frag:51-54 and depth_guass.frag: the taps of
 * the depth filters sit exactly ON a texel's left / upper edge): floor(fl(fl(c / n) * n)).  That is c wherever the fp32 product
 * rounds back to c — everywhere at 640, 512, 256 ... — and c - 1 where it stays below: rows {63, 125, 126, 127, 250, 252, 254} of a
 * 480-high image, row 63 of a 120-high one.  Both rasterisers the image has (Mesa llvmpipe and softpipe, independent code for
 * texture addressing) execute exactly this; rounds 1-3 read texel c ("what a fixed-point texture unit yields") and differed from
 * the executed shaders within +-6 rows of those. */
HD_FN int hd_tap_texel(int c, int n) { return hd_window_texel((float)c / (float)n, n); }

/* The half-pixel walk of the clean pass around a surfel's projection x (copy_unstable.vert:85-108, scale = 1):
 *     step = (1 / cols) * 0.5;  for (i = x / cols - step * wm; i < x / cols + step * wm; i += step)  sample texel floor(i * cols)
 * is 2 wm samples in exact arithmetic.  In fp32 the accumulated i can end an ulp BELOW the bound, and the loop then takes one more
 * sample — at x + wm/2, the texel to the right of the window — for roughly a third of the surfels at 640 x 480.  Rounds 1-3 called
 * that "driver arithmetic" and kept the exact count; but every fp32 implementation takes the extra sample for SOME surfels, so the
 * exact rule removes systematically fewer surfels than the reference does (`count > 8`, `zCount > 4`): on the 640 x 480 GPUTest
 * frame the executed shader removes 607 outliers and the exact rule 425.  Taken literally, with IEEE division (GLSL allows a
 * division 2.5 ulp: which individual surfels get the extra sample is implementation-defined, their share is not).
 * Callers iterate `for (float i = w.lo; i < w.hi; i += w.step)` and take texel hd_window_texel(i, n). */
typedef struct hd_walk { float lo, hi, step; } hd_walk;
HD_FN hd_walk hd_halfpixel_walk(float x, int n, float wm)
{
    hd_walk w;
    const float fn = (float)n;
    w.step = (1.0f / fn) * 0.5f;
    const float reach = w.step * wm;
    const float c = x / fn;
    w.lo = c - reach;
    w.hi = c + reach;
    return w;
}

/* Where the GL rasteriser places a 1-pixel point (IndexMap::predictIndices, index_map.vert:57-60): the shader emits the
 * normalised device coordinate (u - extent/2) / (extent/2); the fixed-function viewport transform maps it back to a
 * window coordinate, which is snapped to the sub-pixel grid (GL_SUBPIXEL_BITS = 8 on NVIDIA hardware and on Mesa,
 * round half up) before the point is rasterised; the fragment is the pixel that contains the snapped centre.  Executing
 * the reference's shaders on Mesa llvmpipe (oracle/ref_glsl) reproduces exactly this rule: 0 index-map mismatches on
 * 124 116 drawn surfels, against 991 for floor(u).  It matters most when the camera has not moved: the reference's
 * vertices sit on integer pixel coordinates (depth_vertex_normal_radius.frag:25-29), i.e. exactly on pixel corners,
 * and only the snap puts each back on its own pixel.
 * Returns the snapped window coordinate; *clipped != 0 when the point's centre is outside the view volume in this
 * axis (-1 <= ndc <= 1, points are clipped by their centre). */
HD_FN float hd_gl_point_window_coord(float u, float extent, int *clipped)
{
    const float half = extent * 0.5f;
    const float ndc = (u - half) / half;
    *clipped = !(ndc >= -1.0f && ndc <= 1.0f);
    const float w = ndc * half + half;
    return hd_floorf(w * 256.0f + 0.5f) * (1.0f / 256.0f);
}
HD_FN int hd_isnanf(float a) { return a != a; }
HD_FN float hd_nanf(void) { return hd_u2f(0x7fffffffu); }

/* ---------------------------------------------------------------- expf */
HD_FN float hd_expf(float x0)
{
    /* branch-free: the core runs on a clamped copy (identical to x0 whenever the core's result is the one returned),
       the three special cases are selected at the end — 169 calls per pixel in the bilateral filter */
    const int is_nan = x0 != x0, over = x0 > 88.7228f, under = x0 < -103.9f;
    const float x = is_nan ? 0.0f : (over ? 88.7228f : (under ? -103.9f : x0));
    float kf = hd_rintf(x * 1.44269504088896341f);
    float r = hd_fmaf(kf, -0.693359375f, x);
    r = hd_fmaf(kf, 2.12194440e-4f, r);
    /* exp(r) on [-ln2/2, ln2/2], degree-6 Horner */
    float p = 1.3981999507e-3f;
    p = hd_fmaf(p, r, 8.3334519073e-3f);
    p = hd_fmaf(p, r, 4.1665795894e-2f);
    p = hd_fmaf(p, r, 1.6666665459e-1f);
    p = hd_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    p = hd_fmaf(p, r2, r);
    p = p + 1.0f;
    int k = (int)kf;
    /* scale by 2^k in two exact steps so that results may go subnormal correctly */
    int k1 = k / 2, k2 = k - k1;
    float s1 = hd_u2f((uint32_t)(k1 + 127) << 23);
    float s2 = hd_u2f((uint32_t)(k2 + 127) << 23);
    const float core = (p * s1) * s2;
    return is_nan ? x0 : (over ? hd_u2f(0x7f800000u) : (under ? 0.0f : core));
}

/* hd_expf for arguments known to be <= 0 and not NaN (the bilateral weights, 169 per pixel): the NaN / overflow selects
   of hd_expf can never fire there, so they are dropped; every other step — and therefore every result bit — is hd_expf's */
HD_FN float hd_expf_nonpos(float x0)
{
    const int under = x0 < -103.9f;
    const float x = under ? -103.9f : x0;
    float kf = hd_rintf(x * 1.44269504088896341f);
    float r = hd_fmaf(kf, -0.693359375f, x);
    r = hd_fmaf(kf, 2.12194440e-4f, r);
    float p = 1.3981999507e-3f;
    p = hd_fmaf(p, r, 8.3334519073e-3f);
    p = hd_fmaf(p, r, 4.1665795894e-2f);
    p = hd_fmaf(p, r, 1.6666665459e-1f);
    p = hd_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    p = hd_fmaf(p, r2, r);
    p = p + 1.0f;
    int k = (int)kf;
    int k1 = k / 2, k2 = k - k1;
    float s1 = hd_u2f((uint32_t)(k1 + 127) << 23);
    float s2 = hd_u2f((uint32_t)(k2 + 127) << 23);
    const float core = (p * s1) * s2;
    return under ? 0.0f : core;
}

/* ---------------------------------------------------------------- asin/acos (float) */
HD_FN float hd_asin_kernel(float x, float z) /* asin(x) for |x|<=0.5, z=x*x */
{
    float p = 4.2163199048e-2f;
    p = hd_fmaf(p, z, 2.4181311049e-2f);
    p = hd_fmaf(p, z, 4.5470025998e-2f);
    p = hd_fmaf(p, z, 7.4953002686e-2f);
    p = hd_fmaf(p, z, 1.6666752422e-1f);
    return hd_fmaf(p * z, x, x);
}

HD_FN float hd_acosf(float x)
{
    if (x != x) return x;
    if (x < -1.0f || x > 1.0f) return hd_nanf();
    if (x > 0.5f) {
        float z = 0.5f * (1.0f - x);
        float s = hd_sqrtf(z);
        return 2.0f * hd_asin_kernel(s, z);
    }
    if (x < -0.5f) {
        float z = 0.5f * (1.0f + x);
        float s = hd_sqrtf(z);
        return 3.14159265358979f - 2.0f * hd_asin_kernel(s, z);
    }
    return 1.57079632679490f - hd_asin_kernel(x, x * x);
}

/* ---------------------------------------------------------------- atan/atan2 (float) */
HD_FN float hd_atanf_pos(float x) /* x >= 0 */
{
    float y0, t;
    if (x > 2.414213562373095f) { y0 = 1.57079632679490f; t = -1.0f / x; }
    else if (x > 0.4142135623730950f) { y0 = 0.785398163397448f; t = (x - 1.0f) / (x + 1.0f); }
    else { y0 = 0.0f; t = x; }
    float z = t * t;
    float p = 8.05374449538e-2f;
    p = hd_fmaf(p, z, -1.38776856032e-1f);
    p = hd_fmaf(p, z, 1.99777106478e-1f);
    p = hd_fmaf(p, z, -3.33329491539e-1f);
    float r = hd_fmaf(p * z, t, t);
    return y0 + r;
}

HD_FN float hd_atan2f(float y, float x)
{
    if (x != x || y != y) return hd_nanf();
    if (x == 0.0f) {
        if (y > 0.0f) return 1.57079632679490f;
        if (y < 0.0f) return -1.57079632679490f;
        return 0.0f;
    }
    float a = hd_atanf_pos(hd_fabsf(y / x));
    if (x < 0.0f) a = 3.14159265358979f - a;
    return (y < 0.0f) ? -a : a;
}

/* ---------------------------------------------------------------- sin/cos (float), |x| < ~8192 */
HD_FN void hd_sincosf(float x, float *s, float *c)
{
    float ax = hd_fabsf(x);
    float jf = hd_rintf(ax * 0.636619772367581f); /* 2/pi */
    int j = hd_cvt_i32(jf);   /* saturating beyond the stated range: only j & 3 is used, and it must be the same bits on both sides */
    float r = hd_fmaf(jf, -1.5703125f, ax);
    r = hd_fmaf(jf, -4.837512969970703125e-4f, r);
    r = hd_fmaf(jf, -7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = hd_fmaf(ps, z, 8.3321608736e-3f);
    ps = hd_fmaf(ps, z, -1.6666654611e-1f);
    float sr = hd_fmaf(ps * z, r, r);
    float pc = 2.443315711809948e-5f;
    pc = hd_fmaf(pc, z, -1.388731625493765e-3f);
    pc = hd_fmaf(pc, z, 4.166664568298827e-2f);
    float cr = hd_fmaf(pc * z, z, hd_fmaf(-0.5f, z, 1.0f));
    float ss, cc;
    switch (j & 3) {
        case 0: ss = sr; cc = cr; break;
        case 1: ss = cr; cc = -sr; break;
        case 2: ss = -sr; cc = -cr; break;
        default: ss = -cr; cc = sr; break;
    }
    *s = (x < 0.0f) ? -ss : ss;
    *c = cc;
}
HD_FN float hd_sinf(float x) { float s, c; hd_sincosf(x, &s, &c); return s; }
HD_FN float hd_cosf(float x) { float s, c; hd_sincosf(x, &s, &c); return c; }

/* ---------------------------------------------------------------- sin/cos/acos (double) */
HD_FN void hd_sincos(double x, double *s, double *c)
{
    double ax = x < 0.0 ? -x : x;
    double jf = hd_rint(ax * 0.63661977236758134308);
    long long j = hd_cvt_i64(jf);
    /* pi/2 in three parts (Cody-Waite) */
    double r = hd_fma(jf, -1.57079632673412561417e+00, ax);
    r = hd_fma(jf, -6.07710050650619224932e-11, r);
    r = hd_fma(jf, -2.02226624879595063154e-21, r);
    double z = r * r;
    double ps = 1.58962301576546568060e-10;
    ps = hd_fma(ps, z, -2.50507477628578072866e-8);
    ps = hd_fma(ps, z, 2.75573136213857245213e-6);
    ps = hd_fma(ps, z, -1.98412698295895385996e-4);
    ps = hd_fma(ps, z, 8.33333333332211858878e-3);
    ps = hd_fma(ps, z, -1.66666666666666307295e-1);
    double sr = hd_fma(ps * z, r, r);
    double pc = -1.13585365213876817300e-11;
    pc = hd_fma(pc, z, 2.08757008419747316778e-9);
    pc = hd_fma(pc, z, -2.75573141792967388112e-7);
    pc = hd_fma(pc, z, 2.48015872888517045348e-5);
    pc = hd_fma(pc, z, -1.38888888888730564116e-3);
    pc = hd_fma(pc, z, 4.16666666666665929218e-2);
    double cr = hd_fma(pc * z, z, hd_fma(-0.5, z, 1.0));
    double ss, cc;
    switch ((int)(j & 3)) {
        case 0: ss = sr; cc = cr; break;
        case 1: ss = cr; cc = -sr; break;
        case 2: ss = -sr; cc = -cr; break;
        default: ss = -cr; cc = sr; break;
    }
    *s = (x < 0.0) ? -ss : ss;
    *c = cc;
}

HD_FN double hd_asin_kernel_d(double x, double z) /* asin(x), |x| <= 0.5, z = x*x */
{
    /* odd Taylor series asin(x) = x + x^3 P(z), 27 terms: truncation < 1e-18 for z <= 0.25 */
    double p = 1.96503361627728368941e-03;
    p = hd_fma(p, z, 2.07766103251816759007e-03);
    p = hd_fma(p, z, 2.20147397371013835848e-03);
    p = hd_fma(p, z, 2.33809189211197504532e-03);
    p = hd_fma(p, z, 2.48944867824688357769e-03);
    p = hd_fma(p, z, 2.65787063820729007810e-03);
    p = hd_fma(p, z, 2.84617840110894205694e-03);
    p = hd_fma(p, z, 3.05782164925803064820e-03);
    p = hd_fma(p, z, 3.29705950347348487883e-03);
    p = hd_fma(p, z, 3.56920539382593474467e-03);
    p = hd_fma(p, z, 3.88096455883766905046e-03);
    p = hd_fma(p, z, 4.24090709367936323504e-03);
    p = hd_fma(p, z, 4.66014348691509618788e-03);
    p = hd_fma(p, z, 5.15330968231990458467e-03);
    p = hd_fma(p, z, 5.74003767084192359493e-03);
    p = hd_fma(p, z, 6.44721031188964874975e-03);
    p = hd_fma(p, z, 7.31252587359884544810e-03);
    p = hd_fma(p, z, 8.39033580961681506316e-03);
    p = hd_fma(p, z, 9.76160952919407839956e-03);
    p = hd_fma(p, z, 1.15518008961397050660e-02);
    p = hd_fma(p, z, 1.39648437500000006939e-02);
    p = hd_fma(p, z, 1.73527644230769238776e-02);
    p = hd_fma(p, z, 2.23721590909090918553e-02);
    p = hd_fma(p, z, 3.03819444444444440590e-02);
    p = hd_fma(p, z, 4.46428571428571438484e-02);
    p = hd_fma(p, z, 7.49999999999999972244e-02);
    p = hd_fma(p, z, 1.66666666666666657415e-01);
    return hd_fma(p * z, x, x);
}

HD_FN double hd_acos(double x)
{
    if (x != x) return x;
    if (x >= 1.0) return 0.0;
    if (x <= -1.0) return 3.14159265358979323846;
    if (x > 0.5) {
        double z = 0.5 * (1.0 - x);
        double s = hd_sqrt(z);
        /* reduce once more if s > 0.5 is impossible here: z <= 0.25 -> s <= 0.5 */
        return 2.0 * hd_asin_kernel_d(s, z);
    }
    if (x < -0.5) {
        double z = 0.5 * (1.0 + x);
        double s = hd_sqrt(z);
        return 3.14159265358979323846 - 2.0 * hd_asin_kernel_d(s, z);
    }
    return 1.57079632679489661923 - hd_asin_kernel_d(x, x * x);
}

/* ---------------------------------------------------------------- exact accumulator */
#define HD_ACC_FRAC_BITS 40

typedef struct { uint64_t lo; int64_t hi; } hd_acc128;

HD_FN void hd_acc_zero(hd_acc128 *a) { a->lo = 0; a->hi = 0; }

HD_FN void hd_acc_add(hd_acc128 *a, hd_acc128 b)
{
    uint64_t lo = a->lo + b.lo;
    int64_t carry = lo < b.lo ? 1 : 0;
    a->lo = lo;
    a->hi = (int64_t)((uint64_t)a->hi + (uint64_t)b.hi + (uint64_t)carry);
}

/* q = round_half_even(p * 2^40) as 128-bit two's complement; non-finite -> 0 */
HD_FN hd_acc128 hd_acc_from_f32(float p)
{
    hd_acc128 q; q.lo = 0; q.hi = 0;
    uint32_t b = hd_f2u(p);
    uint32_t e = (b >> 23) & 0xffu;
    uint64_t m = b & 0x7fffffu;
    if (e == 255u) return q;
    if (e) m |= 0x800000u; else e = 1u;
    int sh = (int)e - 150 + HD_ACC_FRAC_BITS;
    uint64_t lo, hi;
    if (sh >= 0) {
        if (sh > 100) return q;                 /* |p| >= 2^84 contributes nothing, like inf/NaN */
        if (sh == 0) { lo = m; hi = 0; }
        else if (sh < 64) { lo = m << sh; hi = m >> (64 - sh); }
        else { lo = 0; hi = m << (sh - 64); }
    } else {
        int r = -sh;
        hi = 0;
        if (r > 25) lo = 0;
        else {
            uint64_t qv = m >> r;
            uint64_t rem = m & ((1ull << r) - 1ull);
            uint64_t half = 1ull << (r - 1);
            if (rem > half || (rem == half && (qv & 1ull))) qv++;
            lo = qv;
        }
    }
    if (b >> 31) {
        lo = ~lo + 1ull;
        hi = ~hi + (lo == 0 ? 1ull : 0ull);
    }
    q.lo = lo; q.hi = (int64_t)hi;
    return q;
}

HD_FN void hd_acc_add_f32(hd_acc128 *a, float p) { hd_acc_add(a, hd_acc_from_f32(p)); }

HD_FN double hd_acc_to_double(hd_acc128 a)
{
    uint64_t lo = a.lo, hi = (uint64_t)a.hi;
    int neg = (a.hi < 0);
    if (neg) { lo = ~lo + 1ull; hi = ~hi + (lo == 0 ? 1ull : 0ull); }
    double v = (double)hi * 18446744073709551616.0 + (double)lo;
    v = v * 9.094947017729282379150390625e-13; /* 2^-40 */
    return neg ? -v : v;
}

/* ---------------------------------------------------------------- limb form (parallel-friendly)
 * The same exact integer split into three 40-bit limbs, each carried in an int64:
 *     Q = l0 + l1 * 2^40 + l2 * 2^80.
 * Limbs of up to 2^23 addends can be summed independently with plain 64-bit adds (wave shuffles,
 * LDS, atomics, RCCL int64 all-reduce) and recombined once at the end with hd_limbs_combine. */
typedef struct { int64_t l0, l1, l2; } hd_limbs;

HD_FN hd_limbs hd_limbs_from_f32(float p)
{
    hd_acc128 q = hd_acc_from_f32(p);
    const uint64_t M40 = (1ull << 40) - 1ull;
    hd_limbs r;
    r.l0 = (int64_t)(q.lo & M40);
    r.l1 = (int64_t)(((q.lo >> 40) | ((uint64_t)q.hi << 24)) & M40);
    r.l2 = q.hi >> 16;
    return r;
}

HD_FN hd_acc128 hd_limbs_combine(int64_t s0, int64_t s1, int64_t s2)
{
    hd_acc128 a, t;
    a.lo = (uint64_t)s0; a.hi = s0 < 0 ? -1 : 0;
    t.lo = (uint64_t)s1 << 40; t.hi = s1 >> 24;
    hd_acc_add(&a, t);
    t.lo = 0; t.hi = (int64_t)((uint64_t)s2 << 16);
    hd_acc_add(&a, t);
    return a;
}

/* ---------------------------------------------------------------- 25-bit signed limb form (wave-level)
 * Q split into five SIGNED limbs at bit offsets 0, 24, 49, 74, 99:
 *     Q = d0 + d1 2^24 + d2 2^49 + d3 2^74 + d4 2^99,   |d0| <= 2^24, |d1..d4| < 2^25, all with the sign of p.
 * 64 lanes x 2^25 stays inside an int32, so a wave64 reduces each limb with plain 32-bit DPP adds, no carries.
 * The split is done in fp32 (exact: scaling by powers of two, truncation, and removal of leading bits are
 * all error-free), about 20 VALU instructions, no branches.  hd_limbs25_to_limbs turns limb SUMS (int64)
 * back into the 40-bit limb form. */
typedef struct { int32_t d0, d1, d2, d3, d4; } hd_limbs25;

HD_FN hd_limbs25 hd_limbs25_from_f32(float p)
{
    hd_limbs25 l;
    const uint32_t mag = hd_f2u(p) & 0x7fffffffu;
    const float a = mag < ((127u + 84u) << 23) ? p : 0.0f;    /* non-finite or |p| >= 2^84 -> 0 (hd_acc_from_f32) */
    const float h4 = __builtin_truncf(a * 0x1p-59f);
    float r = hd_fmaf(-h4, 0x1p59f, a);
    const float h3 = __builtin_truncf(r * 0x1p-34f);
    r = hd_fmaf(-h3, 0x1p34f, r);
    const float h2 = __builtin_truncf(r * 0x1p-9f);
    r = hd_fmaf(-h2, 0x1p9f, r);
    const float h1 = __builtin_truncf(r * 0x1p16f);
    r = hd_fmaf(-h1, 0x1p-16f, r);
    const float h0 = hd_rintf(r * 0x1p40f);                       /* half-even (default rounding mode) */
    l.d0 = (int32_t)h0; l.d1 = (int32_t)h1; l.d2 = (int32_t)h2; l.d3 = (int32_t)h3; l.d4 = (int32_t)h4;
    return l;
}

/* the same split when the caller knows |p| < 2^34 (d3 = d4 = 0) or |p| < 2^9 (d2 = d3 = d4 = 0): the skipped
 * steps of hd_limbs25_from_f32 would have produced exact zeros, so the low limbs are bit-identical */
HD_FN void hd_limbs25_low3(float a, int32_t *d0, int32_t *d1, int32_t *d2)
{
    const float h2 = __builtin_truncf(a * 0x1p-9f);
    float r = hd_fmaf(-h2, 0x1p9f, a);
    const float h1 = __builtin_truncf(r * 0x1p16f);
    r = hd_fmaf(-h1, 0x1p-16f, r);
    const float h0 = hd_rintf(r * 0x1p40f);
    *d0 = (int32_t)h0; *d1 = (int32_t)h1; *d2 = (int32_t)h2;
}
HD_FN void hd_limbs25_low2(float a, int32_t *d0, int32_t *d1)
{
    const float h1 = __builtin_truncf(a * 0x1p16f);
    const float r = hd_fmaf(-h1, 0x1p-16f, a);
    const float h0 = hd_rintf(r * 0x1p40f);
    *d0 = (int32_t)h0; *d1 = (int32_t)h1;
}

/* five limb SUMS (|s_j| < 2^39) -> Q as 40-bit limbs */
HD_FN hd_limbs hd_limbs25_to_limbs(int64_t s0, int64_t s1, int64_t s2, int64_t s3, int64_t s4)
{
    hd_acc128 a, t;
    a.lo = (uint64_t)s0; a.hi = s0 < 0 ? -1 : 0;
    t.lo = (uint64_t)s1 << 24; t.hi = s1 >> 40; hd_acc_add(&a, t);
    t.lo = (uint64_t)s2 << 49; t.hi = s2 >> 15; hd_acc_add(&a, t);
    t.lo = 0; t.hi = (int64_t)((uint64_t)s3 << 10); hd_acc_add(&a, t);
    t.lo = 0; t.hi = (int64_t)((uint64_t)s4 << 35); hd_acc_add(&a, t);
    const uint64_t M40 = (1ull << 40) - 1ull;
    hd_limbs r;
    r.l0 = (int64_t)(a.lo & M40);
    r.l1 = (int64_t)(((a.lo >> 40) | ((uint64_t)a.hi << 24)) & M40);
    r.l2 = a.hi >> 16;
    return r;
}

/* ---------------------------------------------------------------- sparse-ICP shrink factor
   ICPReduction::thrink (Core/src/Cuda/reduce.cu:302-315) for the only parameters the reference ever sets
   (p = 0.5, mu = 10, three fixed-point sweeps; reduce.cu:652-654): z = factor * h.  With p = 1/2 the two pow() calls
   are pow(x, -1.5) and pow(beta, -0.5); they are restated with the correctly rounded sqrt and division so device and
   oracle agree bit for bit.  alpha_a = 0.1^(2/3) and hTilde = alpha_a + 0.05 * alpha_a^(-1/2) are the fp32 roundings
   of those expressions. */
#define HD_SPARSE_MU 10.0f
HD_FN float hd_sparse_shrink_factor(float hnorm)
{
    const float alpha_a = 0.21544346f, h_tilde = 0.3231652f, p_over_mu = 0.05f;
    if (hnorm <= h_tilde) return 0.0f;
    float beta = (alpha_a / hnorm + 1.0f) / 2.0f;
    const float hp = 1.0f / (hnorm * hd_sqrtf(hnorm));
    for (int i = 0; i < 3; ++i) beta = 1.0f - (p_over_mu * hp) * (1.0f / hd_sqrtf(beta));
    return beta;
}

#endif /* HRBF_DETMATH_H_ */

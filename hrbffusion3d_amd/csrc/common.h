// common.h — shared device/host declarations of libhrbf_mi355 (gfx950 only).
//
// Arithmetic contract: this translation unit is compiled with -ffp-contract=off; every fused
// multiply-add is an explicit hd_fmaf.  +,-,*,/ and sqrt are IEEE correctly rounded on gfx950
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), so the kernels reproduce the CPU
// oracle bit for bit as long as the order of operations below is kept (tests/test_parity_*.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hrbf_mi355.h"
#include "../../include/hrbf_detmath.h"

#define HRBF_NUM_PYRS 3

struct f3 { float x, y, z; };

__host__ __device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__host__ __device__ __forceinline__ f3 xyz(float4 a) { return mk3(a.x, a.y, a.z); }
__host__ __device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__host__ __device__ __forceinline__ float len3(f3 a) { return hd_sqrtf(dot3(a, a)); }
__host__ __device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__host__ __device__ __forceinline__ f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ __forceinline__ f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ __forceinline__ f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__host__ __device__ __forceinline__ f3 normalize3(f3 a)
{
    float l = len3(a);
    return mk3(a.x / l, a.y / l, a.z / l);
}

// rigid transform passed by value to kernels: row-major 3x3 + translation
struct Rigid {
    float r[9];
    float t[3];
};
__host__ __device__ __forceinline__ f3 rot_mul(const Rigid &m, f3 v)
{
    return mk3((m.r[0] * v.x + m.r[1] * v.y) + m.r[2] * v.z, (m.r[3] * v.x + m.r[4] * v.y) + m.r[5] * v.z,
               (m.r[6] * v.x + m.r[7] * v.y) + m.r[8] * v.z);
}
__host__ __device__ __forceinline__ f3 xform(const Rigid &m, f3 v)
{
    f3 r = rot_mul(m, v);
    return mk3(r.x + m.t[0], r.y + m.t[1], r.z + m.t[2]);
}
__host__ __device__ __forceinline__ f3 m33_mul(const float *R, f3 v)
{
    return mk3((R[0] * v.x + R[1] * v.y) + R[2] * v.z, (R[3] * v.x + R[4] * v.y) + R[5] * v.z,
               (R[6] * v.x + R[7] * v.y) + R[8] * v.z);
}
// cofactor inverse of a rigid transform (stands in for Eigen's Matrix4f::inverse())
__host__ __device__ __forceinline__ Rigid rigid_inverse(const Rigid &m)
{
    float a = m.r[0], b = m.r[1], c = m.r[2], d = m.r[3], e = m.r[4], f = m.r[5], g = m.r[6], h = m.r[7],
          i = m.r[8];
    float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    float det = (a * c00 + b * c01) + c * c02;
    float id = 1.0f / det;
    Rigid o;
    o.r[0] = c00 * id; o.r[1] = (c * h - b * i) * id; o.r[2] = (b * f - c * e) * id;
    o.r[3] = c01 * id; o.r[4] = (a * i - c * g) * id; o.r[5] = (c * d - a * f) * id;
    o.r[6] = c02 * id; o.r[7] = (b * g - a * h) * id; o.r[8] = (a * e - b * d) * id;
    o.t[0] = -((o.r[0] * m.t[0] + o.r[1] * m.t[1]) + o.r[2] * m.t[2]);
    o.t[1] = -((o.r[3] * m.t[0] + o.r[4] * m.t[1]) + o.r[5] * m.t[2]);
    o.t[2] = -((o.r[6] * m.t[0] + o.r[7] * m.t[1]) + o.r[8] * m.t[2]);
    return o;
}

__host__ __device__ __forceinline__ float encode_color_bytes(int r, int g, int b)
{
    return (float)((((r << 8) + g) << 8) + b);
}
__host__ __device__ __forceinline__ float encode_color(f3 c)
{
    /* hd_cvt_i32: NaN (a merge at total confidence 0) -> 0, the conversion C leaves undefined (hrbf_detmath.h) */
    int rgb = hd_cvt_i32(hd_rintf(c.x * 255.0f));
    rgb = (int)((uint32_t)rgb << 8) + hd_cvt_i32(hd_rintf(c.y * 255.0f));
    rgb = (int)((uint32_t)rgb << 8) + hd_cvt_i32(hd_rintf(c.z * 255.0f));
    return (float)rgb;
}
__host__ __device__ __forceinline__ f3 decode_color(float c)
{
    int ci = hd_cvt_i32(c);
    return mk3((float)((ci >> 16) & 0xFF) / 255.0f, (float)((ci >> 8) & 0xFF) / 255.0f, (float)(ci & 0xFF) / 255.0f);
}
__host__ __device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// surfels.glsl:19-34 getRadius, :37-46 confidence
__host__ __device__ __forceinline__ float get_radius(float depth, float norm_z, float camz, float camw)
{
    float meanFocal = ((1.0f / hd_fabsf(camz)) + (1.0f / hd_fabsf(camw))) / 2.0f;
    float radius = (depth / meanFocal) * 1.41421356237f;
    float radius_n = radius / hd_fabsf(norm_z);
    float two_r = 2.0f * radius;
    return two_r < radius_n ? two_r : radius_n;
}
__host__ __device__ __forceinline__ float radial_confidence(float x, float y, float cx, float cy, float max_dist,
                                                            float weighting)
{
    float px = x - cx, py = y - cy;
    float radialDist = hd_sqrtf(px * px + py * py) / max_dist;
    return hd_expf(-(radialDist * radialDist) / 0.72f) * weighting;
}

// camera block passed by value to kernels
struct Cam {
    int W, H;
    float fx, fy, cx, cy;
    float camz, camw;   // (float)(1.0/fx), (float)(1.0/fy) — the reference's cam.zw uniforms
    float max_dist;     // sqrt((H/2)^2 + (W/2)^2)
};

const char *hrbf_set_error(const char *fmt, ...);
#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            hrbf_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));      \
            return HRBF_ERR_DEVICE;                                                                  \
        }                                                                                            \
    } while (0)

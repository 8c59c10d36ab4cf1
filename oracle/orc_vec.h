/* orc_vec.h — GLSL-style vector helpers with a pinned operation order (oracle only). */
#ifndef ORC_VEC_H_
#define ORC_VEC_H_
#include "oracle.h"

static inline f3 v3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f4 v4(float x, float y, float z, float w) { f4 r = {x, y, z, w}; return r; }
static inline f3 xyz(f4 a) { return v3(a.x, a.y, a.z); }
static inline float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float len3(f3 a) { return sqrtf(dot3(a, a)); }
static inline f3 cross3(f3 a, f3 b)
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline f3 sub3(f3 a, f3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 add3(f3 a, f3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 scale3(f3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline f3 normalize3(f3 a) { float l = len3(a); return v3(a.x / l, a.y / l, a.z / l); }

/* rigid transform helpers on column-major 4x4 float: M(r,c) = m[c*4+r] */
static inline float M4(const float *m, int r, int c) { return m[c * 4 + r]; }
static inline f3 rot_mul(const float *m, f3 v)   /* mat3(m) * v */
{
    return v3((M4(m, 0, 0) * v.x + M4(m, 0, 1) * v.y) + M4(m, 0, 2) * v.z,
              (M4(m, 1, 0) * v.x + M4(m, 1, 1) * v.y) + M4(m, 1, 2) * v.z,
              (M4(m, 2, 0) * v.x + M4(m, 2, 1) * v.y) + M4(m, 2, 2) * v.z);
}
static inline f3 xform(const float *m, f3 v)     /* (m * vec4(v,1)).xyz */
{
    f3 r = rot_mul(m, v);
    return v3(r.x + M4(m, 0, 3), r.y + M4(m, 1, 3), r.z + M4(m, 2, 3));
}
/* inverse of a rigid 4x4: 3x3 cofactor inverse of the linear part, t' = -(R^-1 t).
   Stands in for Eigen's Matrix4f::inverse() (GlobalModel.cpp:486,575, IndexMap.cpp:207). */
static inline void rigid_inverse(const float *m, float *o)
{
    float a = M4(m, 0, 0), b = M4(m, 0, 1), c = M4(m, 0, 2);
    float d = M4(m, 1, 0), e = M4(m, 1, 1), f = M4(m, 1, 2);
    float g = M4(m, 2, 0), h = M4(m, 2, 1), i = M4(m, 2, 2);
    float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    float det = (a * c00 + b * c01) + c * c02;
    float id = 1.0f / det;
    float r00 = c00 * id, r01 = (c * h - b * i) * id, r02 = (b * f - c * e) * id;
    float r10 = c01 * id, r11 = (a * i - c * g) * id, r12 = (c * d - a * f) * id;
    float r20 = c02 * id, r21 = (b * g - a * h) * id, r22 = (a * e - b * d) * id;
    float tx = M4(m, 0, 3), ty = M4(m, 1, 3), tz = M4(m, 2, 3);
    o[0] = r00; o[1] = r10; o[2] = r20; o[3] = 0.0f;
    o[4] = r01; o[5] = r11; o[6] = r21; o[7] = 0.0f;
    o[8] = r02; o[9] = r12; o[10] = r22; o[11] = 0.0f;
    o[12] = -((r00 * tx + r01 * ty) + r02 * tz);
    o[13] = -((r10 * tx + r11 * ty) + r12 * tz);
    o[14] = -((r20 * tx + r21 * ty) + r22 * tz);
    o[15] = 1.0f;
}
static inline void mat4_mul(const float *a, const float *b, float *o) /* o = a*b (column-major) */
{
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            o[c * 4 + r] = ((M4(a, r, 0) * M4(b, 0, c) + M4(a, r, 1) * M4(b, 1, c)) + M4(a, r, 2) * M4(b, 2, c)) +
                           M4(a, r, 3) * M4(b, 3, c);
}

/* color.glsl:19-34 */
static inline float encode_color_bytes(int r, int g, int b) { return (float)((((r << 8) + g) << 8) + b); }
static inline float encode_color(f3 c)
{
    /* hd_cvt_i32: NaN (a merge at total confidence 0) -> 0, the conversion C leaves undefined (hrbf_detmath.h) */
    int rgb = hd_cvt_i32(rintf(c.x * 255.0f));
    rgb = (int)((uint32_t)rgb << 8) + hd_cvt_i32(rintf(c.y * 255.0f));
    rgb = (int)((uint32_t)rgb << 8) + hd_cvt_i32(rintf(c.z * 255.0f));
    return (float)rgb;
}
static inline f3 decode_color(float c)
{
    int ci = hd_cvt_i32(c);
    return v3((float)((ci >> 16) & 0xFF) / 255.0f, (float)((ci >> 8) & 0xFF) / 255.0f, (float)(ci & 0xFF) / 255.0f);
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

#endif

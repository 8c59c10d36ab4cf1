// pca_normal.h — getNormalPCA (geometry.glsl:63-244) as a device function over any row-major depth array: the LDS tile of
// k_vertex_normal_radius (P3, the fragment shader's texcoord) and the filtered depth image itself in k_associate (data.vert's
// recomputation with the vertex attribute's texcoord, where that differs).
#pragma once
#include "common.h"

__device__ __forceinline__ f3 roots2(float b, float cc)
{
    float d = b * b - 4.0f * cc;
    if (d < 0.0f) d = 0.0f;
    float sd = hd_sqrtf(d);
    return mk3(0.0f, 0.5f * (b + sd), 0.5f * (b - sd));
}

__device__ __forceinline__ f3 compute_roots(float m00, float m10, float m20, float m11, float m21, float m22)
{
    float c0 = (((m00 * m11 * m22 + 2.0f * m10 * m20 * m21) - m00 * m21 * m21) - m11 * m20 * m20) - m22 * m10 * m10;
    float c1 = ((((m00 * m11 - m10 * m10) + m00 * m22) - m20 * m20) + m11 * m22) - m21 * m21;
    float c2 = (m00 + m11) + m22;
    if (hd_fabsf(c0) < 0.000001f) return roots2(c2, c1);
    const float s_inv3 = 1.0f / 3.0f;
    const float s_sqrt3 = 1.7320508075688772f;
    float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    float rho = hd_sqrtf(-a_over_3);
    float theta = hd_atan2f(hd_sqrtf(-q), half_b) * s_inv3;
    float st, ct;
    hd_sincosf(theta, &st, &ct);
    f3 r;
    r.x = c2_over_3 + 2.0f * rho * ct;
    r.y = c2_over_3 - rho * (ct + s_sqrt3 * st);
    r.z = c2_over_3 - rho * (ct - s_sqrt3 * st);
    if (r.x >= r.y) { float t = r.x; r.x = r.y; r.y = t; }
    if (r.y >= r.z) {
        float t = r.y; r.y = r.z; r.z = t;
        if (r.x >= r.y) { float t1 = r.x; r.x = r.y; r.y = t1; }
    }
    if (r.x <= 0.0f) return roots2(c2, c1);
    return r;
}

// getNormalPCA (geometry.glsl:190-244) over the staged depth tile (tile texel (gx, gy) at tile[(gy - oy) * TW + (gx - ox)]), for the
// texture coordinate (tx, ty) the calling shader has: the float-stepped walk, literally (hd_window_axis_t) — the last sample of an
// axis is not taken where the accumulated coordinate overshoots the bound by an ulp; a sample's vertex sits at the float position
// i * cols
__device__ __forceinline__ f3 pca_normal_tile(const float *tile, int TW, int ox, int oy, int W, int H, float tx, float ty, float zc,
                                              float cx, float cy, float camz, float camw)
{
    const hd_window wx = hd_window_axis_t(tx, W, 3.0f), wy = hd_window_axis_t(ty, H, 3.0f);
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
    int cnt = 0;
    for (float fi = wx.lo; fi <= wx.hi; fi += wx.step) {
        const int lx = hd_window_texel(fi, W) - ox;
        const float xf = fi * (float)W;
        for (float fj = wy.lo; fj <= wy.hi; fj += wy.step) {
            const int ly = hd_window_texel(fj, H) - oy;
            const float yf = fj * (float)H;
            float z = tile[ly * TW + lx];
            if (z > 0.3f && hd_fabsf(z - zc) < 0.05f) {
                float X = (xf - cx) * z * camz;
                float Y = (yf - cy) * z * camw;
                a0 += X * X; a1 += X * Y; a2 += X * z; a3 += Y * Y; a4 += Y * z; a5 += z * z;
                a6 += X; a7 += Y; a8 += z;
                cnt++;
            }
        }
    }
    f3 n = mk3(0.0f, 0.0f, 0.0f);
    if (cnt >= 8) {
        float fn = (float)cnt;
        a0 /= fn; a1 /= fn; a2 /= fn; a3 /= fn; a4 /= fn; a5 /= fn; a6 /= fn; a7 /= fn; a8 /= fn;
        float m00 = a0 - a6 * a6, m10 = a1 - a6 * a7, m20 = a2 - a6 * a8;
        float m11 = a3 - a7 * a7, m21 = a4 - a7 * a8, m22 = a5 - a8 * a8;
        float s01 = m00 > m10 ? m00 : m10, s23 = m20 > m11 ? m20 : m11;
        float s0123 = s01 > s23 ? s01 : s23;
        float s45 = m21 > m22 ? m21 : m22;
        float scale = s0123 > s45 ? s0123 : s45;
        float n00 = m00 / scale, n10 = m10 / scale, n20 = m20 / scale, n11 = m11 / scale, n21 = m21 / scale,
              n22 = m22 / scale;
        f3 ev = compute_roots(m00, m10, m20, m11, m21, m22);
        float eigenvalue = ev.x * scale;
        n00 -= eigenvalue; n11 -= eigenvalue; n22 -= eigenvalue;
        f3 row0 = mk3(n00, n10, n20), row1 = mk3(n10, n11, n21), row2 = mk3(n20, n21, n22);
        f3 v1 = cross3(row0, row1), v2 = cross3(row0, row2), v3 = cross3(row1, row2);
        float l1 = len3(v1), l2 = len3(v2), l3 = len3(v3);
        f3 nrm;
        if (l1 >= l2 && l1 >= l3) nrm = v1;
        else if (l2 >= l1 && l2 >= l3) nrm = v2;
        else nrm = v3;
        if (nrm.z < 0.0f) nrm = mk3(-nrm.x, -nrm.y, -nrm.z);
        n = normalize3(nrm);
    }
    return n;
}


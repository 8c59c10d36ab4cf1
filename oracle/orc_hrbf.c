/*
 * orc_hrbf.c — the closed-form HRBF implicit (oracle; test infrastructure only).
 * Follows Core/src/Shaders/hrbfbase.glsl:7-195 of the reference: Wendland phi(r) = (1-r)^4 (4r+1),
 * coefficients sol_i = 10 * n_i, support rho_i = surfel radius.
 */
#include "oracle.h"
#include "orc_vec.h"

/* hrbfbase.glsl:20-34 getWeightD */
static inline f3 weight_d(float vx, float vy, float vz, float d2, float support)
{
    float T2 = support * support;
    if (d2 > T2 || d2 == 0.0f) return v3(0.0f, 0.0f, 0.0f);
    float invT2 = 1.0f / T2;
    float r = sqrtf(d2 * invT2);
    float s = 1.0f - r;
    float s3 = s * s * s;
    float t = -20.0f * s3 * invT2;
    return v3(vx * t, vy * t, vz * t);
}

/* hrbfbase.glsl:37-69 getWeightH */
static inline void weight_h(float vx, float vy, float vz, float d2, float support, float h[9])
{
    float T2 = support * support;
    if (d2 > T2) { for (int i = 0; i < 9; ++i) h[i] = 0.0f; return; }
    if (d2 == 0.0f) {
        h[0] = h[4] = h[8] = -20.0f / T2;
        h[1] = h[2] = h[3] = h[5] = h[6] = h[7] = 0.0f;
        return;
    }
    float r = sqrtf(d2 / T2);
    float s = 1.0f - r;
    float s2 = s * s;
    float t1 = 20.0f * s2 / (T2 * T2 * r);
    float t2 = -r * s * T2;
    float vx2 = vx * vx, vy2 = vy * vy, vz2 = vz * vz;
    h[0] = t1 * (3.0f * vx2 + t2);
    h[1] = t1 * 3.0f * vx * vy;
    h[2] = t1 * 3.0f * vx * vz;
    h[3] = h[1];
    h[4] = t1 * (3.0f * vy2 + t2);
    h[5] = t1 * 3.0f * vy * vz;
    h[6] = h[2];
    h[7] = h[5];
    h[8] = t1 * (3.0f * vz2 + t2);
}

/* hrbfbase.glsl:72-124 getWeightT */
static inline void weight_t(float vx, float vy, float vz, float d2, float support, float t[27])
{
    float T2 = support * support;
    if (d2 > T2 || d2 == 0.0f) { for (int i = 0; i < 27; ++i) t[i] = 0.0f; return; }
    float r = sqrtf(d2 / T2);
    float s = 1.0f - r;
    float s2 = r - 2.0f + 1.0f / r;
    float s3 = 60.0f / (T2 * T2);
    float s4 = 1.0f / (r * r);
    float prx = vx / (T2 * r);
    float pry = vy / (T2 * r);
    float prz = vz / (T2 * r);
    float qx = prx - s4 * prx, qy = pry - s4 * pry, qz = prz - s4 * prz;
    float tss = T2 * s * s;
    t[0] = s3 * (tss * prx + 2.0f * vx * s2 + vx * vx * qx);
    t[1] = s3 * vy * (qx * vx + s2);
    t[2] = s3 * vz * (qx * vx + s2);
    t[3] = s3 * (tss * pry + vx * vx * qy);
    t[4] = s3 * vx * (qy * vy + s2);
    t[5] = s3 * vx * vz * qy;
    t[6] = s3 * (tss * prz + vx * vx * qz);
    t[7] = s3 * vx * vy * qz;
    t[8] = s3 * vx * (qz * vz + s2);
    t[9] = t[1];
    t[10] = s3 * (tss * prx + vy * vy * qx);
    t[11] = s3 * vy * vz * qx;
    t[12] = t[4];
    t[13] = s3 * (tss * pry + 2.0f * vy * s2 + vy * vy * qy);
    t[14] = s3 * vz * (qy * vy + s2);
    t[15] = t[7];
    t[16] = s3 * (tss * prz + vy * vy * qz);
    t[17] = s3 * vy * (qz * vz + s2);
    t[18] = t[2];
    t[19] = t[11];
    t[20] = s3 * (tss * prx + vz * vz * qx);
    t[21] = t[5];
    t[22] = t[14];
    t[23] = s3 * (tss * pry + vz * vz * qy);
    t[24] = t[8];
    t[25] = t[17];
    t[26] = s3 * (tss * prz + 2.0f * vz * s2 + vz * vz * qz);
}

/* hrbfbase.glsl:126-145 hrbfvalue */
float orc_hrbf_value(const float p[3], const f4 *vc, const f4 *nr, int n, int *nsupport)
{
    float value = 0.0f;
    int ns = 0;
    for (int i = 0; i < n; ++i) {
        float sx = 10.0f * nr[i].x, sy = 10.0f * nr[i].y, sz = 10.0f * nr[i].z;
        float vx = p[0] - vc[i].x, vy = p[1] - vc[i].y, vz = p[2] - vc[i].z;
        float d2 = (vx * vx + vy * vy) + vz * vz;
        float support = nr[i].w;
        if (support * support < d2) continue;
        f3 g = weight_d(vx, vy, vz, d2, support);
        value -= (g.x * sx + g.y * sy) + g.z * sz;
        ns++;
    }
    if (nsupport) *nsupport = ns;
    return value;
}

/* hrbfbase.glsl:147-166 hrbfgradient */
void orc_hrbf_gradient(const float p[3], const f4 *vc, const f4 *nr, int n, float out[3])
{
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;
    for (int i = 0; i < n; ++i) {
        float sx = 10.0f * nr[i].x, sy = 10.0f * nr[i].y, sz = 10.0f * nr[i].z;
        float vx = p[0] - vc[i].x, vy = p[1] - vc[i].y, vz = p[2] - vc[i].z;
        float d2 = (vx * vx + vy * vy) + vz * vz;
        float h[9];
        weight_h(vx, vy, vz, d2, nr[i].w, h);
        gx -= (sx * h[0] + sy * h[1]) + sz * h[2];
        gy -= (sx * h[3] + sy * h[4]) + sz * h[5];
        gz -= (sx * h[6] + sy * h[7]) + sz * h[8];
    }
    out[0] = gx; out[1] = gy; out[2] = gz;
}

/* hrbfbase.glsl:168-195 hrbfHessianMatrix (symmetric fill as in the reference) */
void orc_hrbf_hessian(const float p[3], const f4 *vc, const f4 *nr, int n, float g[9])
{
    for (int i = 0; i < 9; ++i) g[i] = 0.0f;
    for (int i = 0; i < n; ++i) {
        float sx = 10.0f * nr[i].x, sy = 10.0f * nr[i].y, sz = 10.0f * nr[i].z;
        float vx = p[0] - vc[i].x, vy = p[1] - vc[i].y, vz = p[2] - vc[i].z;
        float d2 = (vx * vx + vy * vy) + vz * vz;
        float hw[27];
        weight_t(vx, vy, vz, d2, nr[i].w, hw);
        g[0] -= (sx * hw[0] + sy * hw[1]) + sz * hw[2];
        g[1] -= (sx * hw[3] + sy * hw[4]) + sz * hw[5];
        g[2] -= (sx * hw[6] + sy * hw[7]) + sz * hw[8];
        g[4] -= (sx * hw[12] + sy * hw[13]) + sz * hw[14];
        g[5] -= (sx * hw[15] + sy * hw[16]) + sz * hw[17];
        g[8] -= (sx * hw[24] + sy * hw[25]) + sz * hw[26];
    }
    g[3] = g[1]; g[6] = g[2]; g[7] = g[5];
}

/*
 * orc_odo.c — frame-to-model RGB-D odometry (oracle; test infrastructure only).
 * Follows RGBDOdometry (Core/src/Utils/RGBDOdometry.cpp:183-247,660-1249), the CUDA kernels of
 * Core/src/Cuda/reduce.cu:253-1359 and Core/src/Cuda/cudafuncs.cu:57-1028, and
 * OdometryProvider (Core/src/Utils/OdometryProvider.h:35-93).
 *
 * Build-defined arithmetic where the reference leans on Eigen / nvcc fast-math (parity unpinned):
 *  - all normal-equation sums use the exact accumulator (hrbf_detmath.h), then are rounded to
 *    float like the reference's host_data[] (reduce.cu:674-692);
 *  - LDLT = diagonal-pivoted LDL^T written here; 3x3/4x4 inverses by cofactors;
 *  - sin/cos/acos = hd_* ; normalized(n) = n * (1/sqrt(n.n)).
 */
/* ORC_MUTANT (default 0 = the oracle): deliberate MISREADINGS of the reference, one per value, compiled only into
   oracle/_build/liboracle_mutant_<k>.so by `make mutants` — never into the oracle.  tools/mutation_report.py runs the metamorphic
   tests of tests/test_registration_metamorphic.py against each of them: a test suite that is meant to catch a misread Jacobian
   column, sign, weight or frame has to FAIL on these (profiles/r05_metamorphic_mutation_report.txt says which do). */
#ifndef ORC_MUTANT
#define ORC_MUTANT 0
#endif
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "orc_vec.h"

#define PL(m, k, y, x) ((m).p[((size_t)(k) * (m).rows + (y)) * (m).cols + (x)])

static inline f3 m33_mul(const float *R, f3 v)   /* row-major mat33 * v (operators.cuh) */
{
    return v3((R[0] * v.x + R[1] * v.y) + R[2] * v.z, (R[3] * v.x + R[4] * v.y) + R[5] * v.z,
              (R[6] * v.x + R[7] * v.y) + R[8] * v.z);
}

/* ------------------------------------------------------------------ map building (O1) */
/* copyMapsKernel cudafuncs.cu:344-383 */
static void copy_maps(const f4 *vsrc, const f4 *nsrc, orc_planar *vd, orc_planar *nd)
{
    int rows = vd->rows, cols = vd->cols;
    const float qn = hd_nanf();
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            f4 v = vsrc[y * cols + x], n = nsrc[y * cols + x];
            f4 vo = v4(qn, qn, qn, qn), no = vo;
#if ORC_MUTANT == 32     /* copyMaps validity on the vertex alone: `nsrc.w > 0` dropped (cudafuncs.cu:366) */
            if (!(v.z == 0.0f)) { vo = v; no = n; }
#else
            if (!(v.z == 0.0f) && n.w > 0.0f) { vo = v; no = n; }
#endif
            PL(*vd, 0, y, x) = vo.x; PL(*vd, 1, y, x) = vo.y; PL(*vd, 2, y, x) = vo.z; PL(*vd, 3, y, x) = vo.w;
            PL(*nd, 0, y, x) = no.x; PL(*nd, 1, y, x) = no.y; PL(*nd, 2, y, x) = no.z; PL(*nd, 3, y, x) = no.w;
        }
}
/* copyCurvatureMapKernel cudafuncs.cu:405-431 */
static void copy_curv(const f4 *src, orc_planar *d, float thr)
{
    int rows = d->rows, cols = d->cols;
    const float qn = hd_nanf();
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            f4 s = src[y * cols + x], o = v4(qn, qn, qn, qn);
#if ORC_MUTANT == 33     /* curvature kept when kappa < threshold, without `> -threshold` (cudafuncs.cu:420) */
            if (s.w < thr && !hd_isnanf(s.w)) o = s;
#else
            if (s.w < thr && s.w > -thr && !hd_isnanf(s.w)) o = s;
#endif
            PL(*d, 0, y, x) = o.x; PL(*d, 1, y, x) = o.y; PL(*d, 2, y, x) = o.z; PL(*d, 3, y, x) = o.w;
        }
}
/* resizeMapKernel<normalize> cudafuncs.cu:526-587; unwritten planes of a NaN texel keep their old
   contents in the reference — they are never read because plane 0 gates every consumer; here they
   are set to NaN. */
static void resize_map(const orc_planar *in, orc_planar *out, int normalize)
{
    const float qn = hd_nanf();
    for (int y = 0; y < out->rows; ++y)
        for (int x = 0; x < out->cols; ++x) {
            int xs = x * 2, ys = y * 2;
            float x00 = PL(*in, 0, ys, xs), x01 = PL(*in, 0, ys, xs + 1), x10 = PL(*in, 0, ys + 1, xs),
                  x11 = PL(*in, 0, ys + 1, xs + 1);
#if ORC_MUTANT == 30     /* 2 x 2 resize averaging the VALID taps (like the depth pyramid) instead of "NaN if any tap is NaN" (cudafuncs.cu:549-556) */
            {
                const float t4[4] = {x00, x01, x10, x11};
                int nv = 0; float r4[4] = {0, 0, 0, 0};
                for (int q = 0; q < 4; ++q) if (!hd_isnanf(t4[q])) {
                    ++nv;
                    for (int k = 0; k < 4; ++k) r4[k] += PL(*in, k, ys + (q >> 1), xs + (q & 1));
                }
                if (nv > 0 && nv < 4) {
                    for (int k = 0; k < 4; ++k) r4[k] /= (float)nv;
                    if (normalize) { float inv = 1.0f / sqrtf((r4[0] * r4[0] + r4[1] * r4[1]) + r4[2] * r4[2]); r4[0] *= inv; r4[1] *= inv; r4[2] *= inv; }
                    for (int k = 0; k < 4; ++k) PL(*out, k, y, x) = r4[k];
                    continue;
                }
            }
#endif
            if (hd_isnanf(x00) || hd_isnanf(x01) || hd_isnanf(x10) || hd_isnanf(x11)) {
                PL(*out, 0, y, x) = qn; PL(*out, 1, y, x) = qn; PL(*out, 2, y, x) = qn; PL(*out, 3, y, x) = qn;
                continue;
            }
            float r[4];
            for (int k = 0; k < 4; ++k)
                r[k] = (((PL(*in, k, ys, xs) + PL(*in, k, ys, xs + 1)) + PL(*in, k, ys + 1, xs)) +
                        PL(*in, k, ys + 1, xs + 1)) / 4.0f;
#if ORC_MUTANT == 31     /* the resized normal left as the plain average: no renormalisation (cudafuncs.cu:573-581) */
            normalize = 0;
#endif
            if (normalize) {
                float inv = 1.0f / sqrtf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
                r[0] *= inv; r[1] *= inv; r[2] *= inv;
            }
            for (int k = 0; k < 4; ++k) PL(*out, k, y, x) = r[k];
        }
}
/* resizeCMapKernel cudafuncs.cu:618-674 (validity on the w plane) */
static void resize_cmap(const orc_planar *in, orc_planar *out)
{
    const float qn = hd_nanf();
    for (int y = 0; y < out->rows; ++y)
        for (int x = 0; x < out->cols; ++x) {
            int xs = x * 2, ys = y * 2;
            float w00 = PL(*in, 3, ys, xs), w01 = PL(*in, 3, ys, xs + 1), w10 = PL(*in, 3, ys + 1, xs),
                  w11 = PL(*in, 3, ys + 1, xs + 1);
            if (hd_isnanf(w00) || hd_isnanf(w01) || hd_isnanf(w10) || hd_isnanf(w11)) {
                PL(*out, 0, y, x) = qn; PL(*out, 1, y, x) = qn; PL(*out, 2, y, x) = qn; PL(*out, 3, y, x) = qn;
                continue;
            }
            for (int k = 0; k < 4; ++k)
                PL(*out, k, y, x) = (((PL(*in, k, ys, xs) + PL(*in, k, ys, xs + 1)) + PL(*in, k, ys + 1, xs)) +
                                     PL(*in, k, ys + 1, xs + 1)) / 4.0f;
#if ORC_MUTANT == 58     /* the resized principal direction renormalised like a normal (resizeCMapKernel takes the plain mean, cudafuncs.cu:618-674) */
            {
                float inv = 1.0f / sqrtf((PL(*out, 0, y, x) * PL(*out, 0, y, x) + PL(*out, 1, y, x) * PL(*out, 1, y, x)) + PL(*out, 2, y, x) * PL(*out, 2, y, x));
                if (inv < 1.0e30f) { PL(*out, 0, y, x) *= inv; PL(*out, 1, y, x) *= inv; PL(*out, 2, y, x) *= inv; }
            }
#endif
        }
}
/* tranformMapsKernel cudafuncs.cu:213-257 (in place), tranformCurvMapsKernel :279-322 */
static void transform_map(orc_planar *m, const float *R, f3 t, int add_t)
{
    for (int y = 0; y < m->rows; ++y)
        for (int x = 0; x < m->cols; ++x) {
            float vx = PL(*m, 0, y, x);
            if (hd_isnanf(vx)) continue;
            f3 v = v3(vx, PL(*m, 1, y, x), PL(*m, 2, y, x));
            f3 o = m33_mul(R, v);
            if (add_t) o = add3(o, t);
            PL(*m, 0, y, x) = o.x; PL(*m, 1, y, x) = o.y; PL(*m, 2, y, x) = o.z;
        }
}
/* copyicpWeightMapKernel :452-470, resizeicpWeightMapKernel :694-726 */
static void copy_icpw(const float *src, float *dst, int n)
{
    const float qn = hd_nanf();
#if ORC_MUTANT == 34     /* icp weight kept when >= 0 instead of > 0 (cudafuncs.cu:462) */
    for (int i = 0; i < n; ++i) dst[i] = src[i] >= 0.0f ? src[i] : qn;
#else
    for (int i = 0; i < n; ++i) dst[i] = src[i] > 0.0f ? src[i] : qn;
#endif
}
static void resize_icpw(const float *in, int irows, int icols, float *out)
{
    const float qn = hd_nanf();
    int orows = irows / 2, ocols = icols / 2;
    for (int y = 0; y < orows; ++y)
        for (int x = 0; x < ocols; ++x) {
            float a = in[(2 * y) * icols + 2 * x], b = in[(2 * y) * icols + 2 * x + 1],
                  cc = in[(2 * y + 1) * icols + 2 * x], d = in[(2 * y + 1) * icols + 2 * x + 1];
            out[y * ocols + x] = (hd_isnanf(a) || hd_isnanf(b) || hd_isnanf(cc) || hd_isnanf(d))
                                     ? qn : (((a + b) + cc) + d) / 4.0f;
        }
}
/* verticesToDepthKernel :874-885 */
static void vertices_to_depth(const f4 *v, float *dst, int n, float cutoff)
{
    const float qn = hd_nanf();
#if ORC_MUTANT == 49     /* no far cut-off in verticesToDepth: only z <= 0 is invalid (cudafuncs.cu:881) */
    for (int i = 0; i < n; ++i) { float z = v[i].z; (void)cutoff; dst[i] = (z <= 0.0f) ? qn : z; }
#else
    for (int i = 0; i < n; ++i) { float z = v[i].z; dst[i] = (z > cutoff || z <= 0.0f) ? qn : z; }
#endif
}
static const float GK[25] = {1, 4, 6, 4, 1, 4, 16, 24, 16, 4, 6, 24, 36, 24, 6, 4, 16, 24, 16, 4, 1, 4, 6, 4, 1};
/* pyrDownKernelGaussF :493-524 (including its border/index quirk) */
static void pyrdown_gauss_f(const float *src, int srows, int scols, float *dst)
{
    int drows = srows / 2, dcols = scols / 2;
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            int tx = 2 * x + 3 < scols - 1 ? 2 * x + 3 : scols - 1;
            int ty = 2 * y + 3 < srows - 1 ? 2 * y + 3 : srows - 1;
            float sum = 0.0f; int count = 0;
            for (int cy = (2 * y - 2 > 0 ? 2 * y - 2 : 0); cy < ty; ++cy)
                for (int cx = (2 * x - 2 > 0 ? 2 * x - 2 : 0); cx < tx; ++cx) {
                    float s = src[cy * scols + cx];
                    if (!hd_isnanf(s)) {
                        float g = GK[(ty - cy - 1) * 5 + (tx - cx - 1)];
                        sum += s * g;
                        count += (int)g;
                    }
                }
#if ORC_MUTANT == 20     /* the depth pyramid by plain subsampling instead of the NaN-aware 5 x 5 binomial */
            dst[y * dcols + x] = src[(2 * y) * scols + 2 * x];
#else
            dst[y * dcols + x] = sum / (float)count;
#endif
        }
}
/* pyrDownKernelIntensityGauss :818-848 */
static void pyrdown_gauss_u8(const uint8_t *src, int srows, int scols, uint8_t *dst)
{
    int drows = srows / 2, dcols = scols / 2;
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            int tx = 2 * x + 3 < scols - 1 ? 2 * x + 3 : scols - 1;
            int ty = 2 * y + 3 < srows - 1 ? 2 * y + 3 : srows - 1;
            float sum = 0.0f; int count = 0;
            for (int cy = (2 * y - 2 > 0 ? 2 * y - 2 : 0); cy < ty; ++cy)
                for (int cx = (2 * x - 2 > 0 ? 2 * x - 2 : 0); cx < tx; ++cx) {
                    uint8_t s = src[cy * scols + cx];
#if ORC_MUTANT == 48     /* the intensity pyramid averaging EVERY tap: black (= no data) pixels darken their neighbourhood (cudafuncs.cu:836-841) */
                    if (1) {
#else
                    if (s > 0) {
#endif
                        float g = GK[(ty - cy - 1) * 5 + (tx - cx - 1)];
                        sum += (float)s * g;
                        count += (int)g;
                    }
                }
            dst[y * dcols + x] = count > 0 ? (uint8_t)(int)(sum / (float)count) : 0;
        }
}
/* bgr2IntensityKernel :896-911 — channel x = R as uploaded (HRBFFusion.cpp:1010) */
static inline uint8_t intensity(int r, int g, int b)
{
#if ORC_MUTANT == 25     /* a different but CONSISTENT image: the textbook luma 0.299 R + 0.587 G + 0.114 B */
    float v = (float)r * 0.299f;
    v = v + (float)g * 0.587f;
    v = v + (float)b * 0.114f;
#else
    float v = (float)r * 0.114f;
    v = v + (float)g * 0.299f;
    v = v + (float)b * 0.587f;
#endif
    return (uint8_t)(int)v;
}
/* applyKernel (Sobel) :927-954 incl. the running kernelIndex quirk at borders */
static void sobel(const uint8_t *src, int rows, int cols, int16_t *dx, int16_t *dy)
{
#if ORC_MUTANT == 26     /* the two Sobel kernels swapped: dIdx holds the vertical derivative */
    static const float gy[9] = {1, 0, -1, 2, 0, -2, 1, 0, -1};
    static const float gx[9] = {1, 2, 1, 0, 0, 0, -1, -2, -1};
#else
    static const float gx[9] = {1, 0, -1, 2, 0, -2, 1, 0, -1};
    static const float gy[9] = {1, 2, 1, 0, 0, 0, -1, -2, -1};
#endif
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float dxv = 0.0f, dyv = 0.0f;
            int ki = 8;
            for (int j = (y - 1 > 0 ? y - 1 : 0); j <= (y + 1 < rows - 1 ? y + 1 : rows - 1); ++j)
                for (int i = (x - 1 > 0 ? x - 1 : 0); i <= (x + 1 < cols - 1 ? x + 1 : cols - 1); ++i) {
                    float s = (float)src[j * cols + i];
#if ORC_MUTANT == 57     /* the kernel entry that BELONGS to the tap (index from the tap's offset) instead of the reference's running index, which
                            slips at the image border where the window is cut (cudafuncs.cu:937-946) */
                    ki = 8 - ((j - (y - 1)) * 3 + (i - (x - 1)));
                    dxv += s * gx[ki];
                    dyv += s * gy[ki];
#else
                    dxv += s * gx[ki];
                    dyv += s * gy[ki];
                    --ki;
#endif
                }
            dx[y * cols + x] = (int16_t)dxv;
            dy[y * cols + x] = (int16_t)dyv;
        }
}

void orc_odo_init_model(orc_ctx *c, const f4 *vtex, const f4 *ntex, const uint8_t *img4,
                        const f4 *k1tex, const f4 *k2tex, const float *icpw_tex)
{
    /* initICPModel RGBDOdometry.cpp:204-247 */
    float R[9]; f3 t;
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) R[r * 3 + k] = M4(c->pose, r, k);
    t = v3(M4(c->pose, 0, 3), M4(c->pose, 1, 3), M4(c->pose, 2, 3));
    copy_maps(vtex, ntex, &c->vmap_g[0], &c->nmap_g[0]);
    for (int i = 1; i < ORC_NUM_PYRS; ++i) {
        resize_map(&c->vmap_g[i - 1], &c->vmap_g[i], 0);
        resize_map(&c->nmap_g[i - 1], &c->nmap_g[i], 1);
    }
    /* ORC_MUTANT 27: the model normals moved like points (R n + t); 28: the model vertices rotated only (R v) — cudafuncs.cu:230,246 */
    for (int i = 0; i < ORC_NUM_PYRS; ++i) { transform_map(&c->vmap_g[i], R, t, ORC_MUTANT != 28); transform_map(&c->nmap_g[i], R, t, ORC_MUTANT == 27); }
    /* initRGBModel :689-693 -> populateRGBDData :660-687 */
    vertices_to_depth(vtex, c->last_depth[0], c->P, 6.0f);
    for (int i = 0; i + 1 < ORC_NUM_PYRS; ++i) pyrdown_gauss_f(c->last_depth[i], c->H >> i, c->W >> i, c->last_depth[i + 1]);
    for (int i = 0; i < c->P; ++i) c->last_image[0][i] = intensity(img4[i * 4], img4[i * 4 + 1], img4[i * 4 + 2]);
    for (int i = 0; i + 1 < ORC_NUM_PYRS; ++i) pyrdown_gauss_u8(c->last_image[i], c->H >> i, c->W >> i, c->last_image[i + 1]);
    /* initCurvatureModel :723-757 */
    copy_curv(k1tex, &c->ck1_g[0], c->prm.curv_valid_threshold);
    copy_curv(k2tex, &c->ck2_g[0], c->prm.curv_valid_threshold);
    for (int i = 1; i < ORC_NUM_PYRS; ++i) { resize_cmap(&c->ck1_g[i - 1], &c->ck1_g[i]); resize_cmap(&c->ck2_g[i - 1], &c->ck2_g[i]); }
#if ORC_MUTANT != 29     /* 29: the model's principal directions left in the camera frame (transformCurvMaps skipped, cudafuncs.cu:279-322) */
    for (int i = 0; i < ORC_NUM_PYRS; ++i) { transform_map(&c->ck1_g[i], R, t, 0); transform_map(&c->ck2_g[i], R, t, 0); }
#endif
    /* initICPweight :759-775 */
    copy_icpw(icpw_tex, c->icpw[0], c->P);
    for (int i = 1; i < ORC_NUM_PYRS; ++i) resize_icpw(c->icpw[i - 1], c->H >> (i - 1), c->W >> (i - 1), c->icpw[i]);
}

void orc_odo_init_live(orc_ctx *c)
{
    /* initICP(vertex,normal) :183-202 */
    copy_maps(c->vertex_filtered, c->normal, &c->vmap_c[0], &c->nmap_c[0]);
    for (int i = 1; i < ORC_NUM_PYRS; ++i) {
        resize_map(&c->vmap_c[i - 1], &c->vmap_c[i], 0);
        resize_map(&c->nmap_c[i - 1], &c->nmap_c[i], 1);
    }
    /* initRGB :695-699 */
    vertices_to_depth(c->vertex_filtered, c->next_depth[0], c->P, 6.0f);
    for (int i = 0; i + 1 < ORC_NUM_PYRS; ++i) pyrdown_gauss_f(c->next_depth[i], c->H >> i, c->W >> i, c->next_depth[i + 1]);
    for (int i = 0; i < c->P; ++i) c->next_image[0][i] = intensity(c->rgb[i * 3], c->rgb[i * 3 + 1], c->rgb[i * 3 + 2]);
    for (int i = 0; i + 1 < ORC_NUM_PYRS; ++i) pyrdown_gauss_u8(c->next_image[i], c->H >> i, c->W >> i, c->next_image[i + 1]);
    /* initCurvature :701-721 */
    copy_curv(c->curv1, &c->ck1_c[0], c->prm.curv_valid_threshold);
    copy_curv(c->curv2, &c->ck2_c[0], c->prm.curv_valid_threshold);
    for (int i = 1; i < ORC_NUM_PYRS; ++i) { resize_cmap(&c->ck1_c[i - 1], &c->ck1_c[i]); resize_cmap(&c->ck2_c[i - 1], &c->ck2_c[i]); }
}

/* initFirstRGB :777-794 */
void orc_odo_init_first_rgb(orc_ctx *c)
{
    for (int i = 0; i < c->P; ++i) c->last_next_image[0][i] = intensity(c->rgb[i * 3], c->rgb[i * 3 + 1], c->rgb[i * 3 + 2]);
    for (int i = 0; i + 1 < ORC_NUM_PYRS; ++i)
        pyrdown_gauss_u8(c->last_next_image[i], c->H >> i, c->W >> i, c->last_next_image[i + 1]);
}

/* ------------------------------------------------------------------ small dense linear algebra
 * Diagonal-pivoted LDL^T (the reference calls Eigen's ldlt().solve, RGBDOdometry.cpp:1173-1185: own factorisation, DESIGN §8
 * deviations).  One reciprocal per pivot: the device runs this on ONE lane, where each of the 21 divisions of the textbook
 * form (15 column scalings + 6 diagonal solves) is a ~30-instruction dependent chain in fp64. */
#define DEF_LDLT(NAME, T)                                                                        \
    static void NAME(int n, const T *Ain, const T *b, T *x)                                      \
    {                                                                                            \
        T A[36]; int perm[6]; T y[6]; T rdiag[6];                                                \
        for (int i = 0; i < n * n; ++i) A[i] = Ain[i];                                           \
        for (int i = 0; i < n; ++i) perm[i] = i;                                                 \
        for (int k = 0; k < n; ++k) {                                                            \
            int piv = k; T best = A[k * n + k] < 0 ? -A[k * n + k] : A[k * n + k];               \
            for (int i = k + 1; i < n; ++i) {                                                    \
                T v = A[i * n + i] < 0 ? -A[i * n + i] : A[i * n + i];                           \
                if (v > best) { best = v; piv = i; }                                             \
            }                                                                                    \
            if (piv != k) {                                                                      \
                for (int j = 0; j < n; ++j) { T t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; } \
                for (int j = 0; j < n; ++j) { T t = A[j * n + k]; A[j * n + k] = A[j * n + piv]; A[j * n + piv] = t; } \
                int tp = perm[k]; perm[k] = perm[piv]; perm[piv] = tp;                           \
            }                                                                                    \
            T d = A[k * n + k];                                                                  \
            rdiag[k] = 0;                                                                        \
            if (d == 0) continue;                                                                \
            const T rd = 1 / d;   /* ONE division per pivot; columns and the diagonal solve multiply by it */ \
            rdiag[k] = rd;                                                                       \
            for (int i = k + 1; i < n; ++i) A[i * n + k] = A[i * n + k] * rd;                    \
            for (int i = k + 1; i < n; ++i)                                                      \
                for (int j = k + 1; j <= i; ++j) {                                               \
                    A[i * n + j] = A[i * n + j] - A[i * n + k] * d * A[j * n + k];               \
                    A[j * n + i] = A[i * n + j];                                                 \
                }                                                                                \
        }                                                                                        \
        for (int i = 0; i < n; ++i) y[i] = b[perm[i]];                                           \
        for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) y[i] = y[i] - A[i * n + j] * y[j]; \
        for (int i = 0; i < n; ++i) y[i] = (A[i * n + i] == 0) ? 0 : y[i] * rdiag[i];           \
        for (int i = n - 1; i >= 0; --i) for (int j = i + 1; j < n; ++j) y[i] = y[i] - A[j * n + i] * y[j]; \
        for (int i = 0; i < n; ++i) x[perm[i]] = y[i];                                           \
    }
DEF_LDLT(ldlt_d, double)
DEF_LDLT(ldlt_f, float)

void orc_solve6(const double A[36], const double b[6], double x[6]) { ldlt_d(6, A, b, x); }

static void inv3d(const double *m, double *o)   /* row-major 3x3 cofactor inverse */
{
    double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    double c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    double det = (a * c00 + b * c01) + c * c02;
    double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (c * h - b * i) * id; o[2] = (b * f - c * e) * id;
    o[3] = c01 * id; o[4] = (a * i - c * g) * id; o[5] = (c * d - a * f) * id;
    o[6] = c02 * id; o[7] = (b * g - a * h) * id; o[8] = (a * e - b * d) * id;
}
static void mul3d(const double *a, const double *b, double *o)
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
        o[r * 3 + c] = (a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c]) + a[r * 3 + 2] * b[6 + c];
}
static void mul3f(const float *a, const float *b, float *o)
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
        o[r * 3 + c] = (a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c]) + a[r * 3 + 2] * b[6 + c];
}
/* OdometryProvider::rodrigues OdometryProvider.h:35-71 */
static void rodrigues(const double *src, double *R)
{
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    double rx = src[0], ry = src[1], rz = src[2];
    double theta = sqrt((rx * rx + ry * ry) + rz * rz);
    if (theta >= 2.2204460492503131e-16) {
        double s, cth; hd_sincos(theta, &s, &cth);
        double c1 = 1.0 - cth, itheta = 1.0 / theta;
        rx *= itheta; ry *= itheta; rz *= itheta;
        double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        double rxm[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int k = 0; k < 9; ++k) R[k] = (cth * I[k] + c1 * rrt[k]) + s * rxm[k];
    }
}

/* ------------------------------------------------------------------ O4: icpStep reduce.cu:253-693 */
typedef struct { hd_acc128 a[29]; } acc29;
/* side images of the sparse (ADMM) variant, one level: multiplier lambdaMap, shrunk residual z_thrinkMap, corresICP */
typedef struct { f3 *lambda, *z; int32_t *corres; int64_t *shrunk; } orc_sparse;

/* the rejection rule of reduce.cu:383: `sine > angleThres || dist > distThres`, and three misreadings of it */
#if ORC_MUTANT == 35     /* the distance threshold met by the SQUARED distance */
#define ICP_REJECT(sine, dist, nc, np, vp, vg) ((sine) > angleThres || (dist) * (dist) > distThres)
#elif ORC_MUTANT == 36   /* the angle threshold met by 1 - cosine instead of the sine */
#define ICP_REJECT(sine, dist, nc, np, vp, vg) (1.0f - dot3(nc, np) > angleThres || (dist) > distThres)
#elif ORC_MUTANT == 37   /* the distance threshold on the depth difference instead of the Euclidean distance */
#define ICP_REJECT(sine, dist, nc, np, vp, vg) ((sine) > angleThres || fabsf((vp).z - (vg).z) > distThres)
#else
#define ICP_REJECT(sine, dist, nc, np, vp, vg) ((sine) > angleThres || (dist) > distThres)
#endif
#if ORC_MUTANT == 47     /* the shrink operator of the l1 norm (soft threshold 1 / mu) instead of l_p, p = 0.5 (reduce.cu:302-315, :652) */
static inline float orc_shrink(float hnorm) { return hnorm <= 1.0f / HD_SPARSE_MU ? 0.0f : 1.0f - (1.0f / HD_SPARSE_MU) / hnorm; }
#else
static inline float orc_shrink(float hnorm) { return hd_sparse_shrink_factor(hnorm); }
#endif
static void icp_pixel(const float *Rcurr, f3 tcurr, const orc_planar *vc, const orc_planar *nc,
                      const orc_planar *k1c, const orc_planar *k2c, const float *Rpi, f3 tprev,
                      float fx, float fy, float cx, float cy, const orc_planar *vg, const orc_planar *ng,
                      const orc_planar *k1g, const orc_planar *k2g, const float *icpw, float distThres,
                      float angleThres, int use_search, int radius, int use_weight, const orc_sparse *sp, int32_t *probe,
                      int x, int y, float out[29])
{
    int rows = vc->rows, cols = vc->cols;
    for (int i = 0; i < 29; ++i) out[i] = 0.0f;
    if (probe) { probe[2 * (y * cols + x)] = -1; probe[2 * (y * cols + x) + 1] = -1; }   /* corresICP of the plain variant (reduce.cu:444-456) */
    if (sp) {   /* getProducts writes both side outputs for every pixel before the found test (reduce.cu:455-471) */
        sp->z[y * cols + x] = v3(0, 0, 0);
        sp->corres[2 * (y * cols + x)] = -1; sp->corres[2 * (y * cols + x) + 1] = -1;
    }
    f3 vcur = v3(PL(*vc, 0, y, x), PL(*vc, 1, y, x), PL(*vc, 2, y, x));
    f3 ncur = v3(PL(*nc, 0, y, x), PL(*nc, 1, y, x), PL(*nc, 2, y, x));
    float ck1 = PL(*k1c, 3, y, x), ck2 = PL(*k2c, 3, y, x);
    if (hd_isnanf(vcur.x) || hd_isnanf(ncur.x) || hd_isnanf(ck1) || hd_isnanf(ck2)) return;
    f3 vg_ = add3(m33_mul(Rcurr, vcur), tcurr);
#if ORC_MUTANT == 15     /* the live vertex projected as if the previous camera stood at the origin of the tracker's frame */
    f3 vcp = vg_;
#else
    f3 vcp = m33_mul(Rpi, sub3(vg_, tprev));
#endif
    float fu = vcp.x * fx / vcp.z + cx, fv = vcp.y * fy / vcp.z + cy;
    if (hd_isnanf(fu) || hd_isnanf(fv)) return;
    if (!(fu > -1.0e9f && fu < 1.0e9f && fv > -1.0e9f && fv < 1.0e9f)) return;
#if ORC_MUTANT == 14     /* the associated texel by truncation instead of __float2int_rn */
    int ux = (int)fu, uy = (int)fv;
#else
    int ux = (int)rintf(fu), uy = (int)rintf(fv);
#endif
    if (ux < 0 || uy < 0 || ux >= cols || uy >= rows || vcp.z < 0.0f) return;
    f3 ncur_g = m33_mul(Rcurr, ncur);
    const int R = use_search ? radius : 0;
    int found = 0, bx = -1, by = -1;
    f3 bv = v3(0, 0, 0), bn = v3(0, 0, 0);
    float p_smallest = 1e8f;
    /* two passes like the reference: D_p_R = max dist over accepted candidates, then the cost */
    float DpR = -1e8f;
    for (int cy_ = uy - R; cy_ < uy + R + 1; ++cy_)
        for (int cx_ = ux - R; cx_ < ux + R + 1; ++cx_) {
            if (cx_ < 0 || cy_ < 0 || cx_ >= cols || cy_ >= rows) continue;
            f3 vp = v3(PL(*vg, 0, cy_, cx_), PL(*vg, 1, cy_, cx_), PL(*vg, 2, cy_, cx_));
            f3 np = v3(PL(*ng, 0, cy_, cx_), PL(*ng, 1, cy_, cx_), PL(*ng, 2, cy_, cx_));
            float c1 = PL(*k1g, 3, cy_, cx_), c2 = PL(*k2g, 3, cy_, cx_);
            if (hd_isnanf(vp.x) || hd_isnanf(np.x) || hd_isnanf(c1) || hd_isnanf(c2)) continue;
            float dist = len3(sub3(vp, vg_)), sine = len3(cross3(ncur_g, np));
            if (ICP_REJECT(sine, dist, ncur_g, np, vp, vg_)) continue;
            if (dist > DpR) DpR = dist;
        }
    for (int cy_ = uy - R; cy_ < uy + R + 1; ++cy_)
        for (int cx_ = ux - R; cx_ < ux + R + 1; ++cx_) {
            if (cx_ < 0 || cy_ < 0 || cx_ >= cols || cy_ >= rows) continue;
            f3 vp = v3(PL(*vg, 0, cy_, cx_), PL(*vg, 1, cy_, cx_), PL(*vg, 2, cy_, cx_));
            f3 np = v3(PL(*ng, 0, cy_, cx_), PL(*ng, 1, cy_, cx_), PL(*ng, 2, cy_, cx_));
            float c1 = PL(*k1g, 3, cy_, cx_), c2 = PL(*k2g, 3, cy_, cx_);
            if (hd_isnanf(vp.x) || hd_isnanf(np.x) || hd_isnanf(c1) || hd_isnanf(c2)) continue;
            float dist = len3(sub3(vp, vg_)), sine = len3(cross3(ncur_g, np));
            if (ICP_REJECT(sine, dist, ncur_g, np, vp, vg_)) continue;
            float p = 1.0f;
            if (use_search) {
                float a1 = fabsf(c1), a2 = fabsf(c2);
                float ckmax = a1 > a2 ? a1 : a2;
#if ORC_MUTANT == 39     /* D_p normalised by the distance threshold instead of the window's largest accepted distance (reduce.cu:421) */
                float D_p = dist / distThres;
#else
                float D_p = dist / DpR;
#endif
                float D_n = 1.0f - dot3(np, ncur_g);
                float D_c = 1.0f - hd_expf(-fabsf(c1 - ck1) / ckmax) * hd_expf(-fabsf(c2 - ck2) / ckmax);
                p = (0.333f * D_p + 0.333f * D_n) + 0.333f * D_c;
            }
#if ORC_MUTANT == 38     /* ties go to the LAST candidate in raster order (`<=` for the reference's `<`, reduce.cu:429) */
            if (p <= p_smallest) { bx = cx_; by = cy_; bv = vp; bn = np; p_smallest = p; }
#else
            if (p < p_smallest) { bx = cx_; by = cy_; bv = vp; bn = np; p_smallest = p; }
#endif
            found = 1;
        }
    if (!found) return;
    if (probe) { probe[2 * (y * cols + x)] = bx; probe[2 * (y * cols + x) + 1] = by; }
    f3 s_cp = m33_mul(Rpi, sub3(vg_, tprev));
    f3 d_cp = m33_mul(Rpi, sub3(bv, tprev));
#if ORC_MUTANT == 1      /* the matched normal left in the tracker's world frame */
    f3 n_cp = bn;
#else
    f3 n_cp = m33_mul(Rpi, bn);
#endif
    if (sp) {   /* sparse ICP (reduce.cu:479-492): the target moves by the shrunk residual minus the scaled multiplier */
        const int k = y * cols + x;
        sp->corres[2 * k] = bx; sp->corres[2 * k + 1] = by;
        f3 lm = v3(sp->lambda[k].x / HD_SPARSE_MU, sp->lambda[k].y / HD_SPARSE_MU, sp->lambda[k].z / HD_SPARSE_MU);
#if ORC_MUTANT == 40     /* h = s - d - lambda / mu (the multiplier with the opposite sign, reduce.cu:482) */
        f3 h = sub3(sub3(s_cp, d_cp), lm);
#else
        f3 h = add3(sub3(s_cp, d_cp), lm);
#endif
        float beta = orc_shrink(len3(h));
        if (beta != 0.0f && sp->shrunk) {   /* test evidence only: how often the non-trivial branch ran */
#pragma omp atomic
            ++*sp->shrunk;
        }
        f3 z = v3(beta * h.x, beta * h.y, beta * h.z);
#if ORC_MUTANT == 41     /* the target moved by z + lambda / mu instead of z - lambda / mu (reduce.cu:485) */
        d_cp = add3(add3(d_cp, z), lm);
#else
        d_cp = sub3(add3(d_cp, z), lm);
#endif
        sp->z[k] = z;
    }
    float weight = 1.0f;
    /* every candidate can score NaN (flat patch: ckmax = 0): the reference then keeps corres = (-1,-1) and reads
       uninitialised locals and icpWeight[-1][-1]; the products are all zero here (bv = bn = 0), the weight is defined 0 */
    if (use_weight) { float w = bx < 0 ? 0.0f : icpw[by * cols + bx]; weight = hd_isnanf(w) ? 0.0f : w; }
    float row[7];
#if ORC_MUTANT == 13     /* the rotational columns of the ICP row as n x s instead of s x n */
    f3 cr = cross3(n_cp, s_cp);
#else
    f3 cr = cross3(s_cp, n_cp);
#endif
    row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z; row[3] = cr.x; row[4] = cr.y; row[5] = cr.z;
#if ORC_MUTANT == 2      /* residual n . (d - s) instead of n . (s - d) */
    row[6] = dot3(n_cp, sub3(d_cp, s_cp));
#else
    row[6] = dot3(n_cp, sub3(s_cp, d_cp));
#endif
    int k = 0;
    for (int i = 0; i < 6; ++i) for (int j = i; j < 7; ++j) out[k++] = weight * row[i] * row[j];
    out[27] = weight * row[6] * row[6];
    out[28] = 1.0f;
}

static void unpack27(const double *s, float *A, float *b)
{
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            float value = (float)s[shift++];
            if (j == 6) b[i] = value; else A[j * 6 + i] = A[i * 6 + j] = value;
        }
}

static void icp_step(const float *Rcurr, f3 tcurr, const orc_planar *vc, const orc_planar *nc,
                     const orc_planar *k1c, const orc_planar *k2c, const float *Rpi, f3 tprev,
                     float fx, float fy, float cx, float cy, const orc_planar *vg, const orc_planar *ng,
                     const orc_planar *k1g, const orc_planar *k2g, const float *icpw, float distThres,
                     float angleThres, int use_search, int radius, int use_weight, const orc_sparse *sp, int32_t *probe,
                     double sums[29])
{
    acc29 tot; for (int i = 0; i < 29; ++i) hd_acc_zero(&tot.a[i]);
#pragma omp parallel
    {
        acc29 loc; for (int i = 0; i < 29; ++i) hd_acc_zero(&loc.a[i]);
#pragma omp for schedule(static) nowait
        for (int y = 0; y < vc->rows; ++y)
            for (int x = 0; x < vc->cols; ++x) {
                float o[29];
                icp_pixel(Rcurr, tcurr, vc, nc, k1c, k2c, Rpi, tprev, fx, fy, cx, cy, vg, ng, k1g, k2g, icpw,
                          distThres, angleThres, use_search, radius, use_weight, sp, probe, x, y, o);
                if (o[28] != 0.0f) for (int i = 0; i < 29; ++i) hd_acc_add_f32(&loc.a[i], o[i]);
            }
#pragma omp critical
        for (int i = 0; i < 29; ++i) hd_acc_add(&tot.a[i], loc.a[i]);
    }
    for (int i = 0; i < 29; ++i) sums[i] = hd_acc_to_double(tot.a[i]);
}

/* updateLambdaMapKernel (cudafuncs.cu:1030-1080): lambda += mu * (s' - d - z) at the pose just solved for, where the
   last icpStep found a correspondence — tested as corresp.x > 0, so matches in column 0 never update (kept) */
static void sparse_update_lambda(const float *Rcurr, f3 tcurr, const orc_planar *vc, const float *Rpi, f3 tprev,
                                 const orc_planar *vg, const orc_sparse *sp)
{
    const int rows = vc->rows, cols = vc->cols;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const int k = y * cols + x, ux = sp->corres[2 * k], uy = sp->corres[2 * k + 1];
            if (ux <= 0) continue;
            f3 vcur = v3(PL(*vc, 0, y, x), PL(*vc, 1, y, x), PL(*vc, 2, y, x));
            f3 vlp = m33_mul(Rpi, sub3(add3(m33_mul(Rcurr, vcur), tcurr), tprev));
            f3 vp = m33_mul(Rpi, sub3(v3(PL(*vg, 0, uy, ux), PL(*vg, 1, uy, ux), PL(*vg, 2, uy, ux)), tprev));
            f3 d = sub3(sub3(vlp, vp), sp->z[k]);
#if ORC_MUTANT == 42     /* lambda - mu * Delta (cudafuncs.cu:1066-1067 with the opposite sign) */
            sp->lambda[k] = sub3(sp->lambda[k], v3(HD_SPARSE_MU * d.x, HD_SPARSE_MU * d.y, HD_SPARSE_MU * d.z));
#else
            sp->lambda[k] = add3(sp->lambda[k], v3(HD_SPARSE_MU * d.x, HD_SPARSE_MU * d.y, HD_SPARSE_MU * d.z));
#endif
        }
}

float orc_sparse_shrink_factor(float hnorm) { return orc_shrink(hnorm); }
int64_t orc_sparse_shrunk_count(const orc_ctx *c) { return c->sp_shrunk; }

int orc_icp_step(const float Rcurr[9], const float tcurr[3], const float *vmap_curr, const float *nmap_curr,
                 const float *ck1_curr, const float *ck2_curr, const float Rprev_inv[9], const float tprev[3],
                 float fx, float fy, float cx, float cy, const float *vmap_g_prev, const float *nmap_g_prev,
                 const float *ck1_g_prev, const float *ck2_g_prev, const float *icp_weight_prev, int rows,
                 int cols, float dist_thresh, float angle_thresh, int use_weight, double A_out[36],
                 double b_out[6], double residual_out[2])
{
    orc_planar vc = {rows, cols, (float *)vmap_curr}, nc = {rows, cols, (float *)nmap_curr};
    orc_planar k1c = {rows, cols, (float *)ck1_curr}, k2c = {rows, cols, (float *)ck2_curr};
    orc_planar vg = {rows, cols, (float *)vmap_g_prev}, ng = {rows, cols, (float *)nmap_g_prev};
    orc_planar k1g = {rows, cols, (float *)ck1_g_prev}, k2g = {rows, cols, (float *)ck2_g_prev};
    double s[29];
    icp_step(Rcurr, v3(tcurr[0], tcurr[1], tcurr[2]), &vc, &nc, &k1c, &k2c, Rprev_inv,
             v3(tprev[0], tprev[1], tprev[2]), fx, fy, cx, cy, &vg, &ng, &k1g, &k2g, icp_weight_prev,
             dist_thresh, angle_thresh, 0, 0, use_weight, NULL, NULL, s);
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            double v = s[shift++];
            if (j == 6) b_out[i] = v; else A_out[j * 6 + i] = A_out[i * 6 + j] = v;
        }
    residual_out[0] = s[27]; residual_out[1] = s[28];
    return 0;
}

/* icpStep with icp_if_use_coorespondence_search = true (reduce.cu:357-430) as a stand-alone operator that also returns corresICP
   (2 int32 per pixel, (-1, -1) = none): the windowed search's choice is an observable of its own */
int orc_icp_step_search(const float Rcurr[9], const float tcurr[3], const float *vmap_curr, const float *nmap_curr,
                        const float *ck1_curr, const float *ck2_curr, const float Rprev_inv[9], const float tprev[3],
                        float fx, float fy, float cx, float cy, const float *vmap_g_prev, const float *nmap_g_prev,
                        const float *ck1_g_prev, const float *ck2_g_prev, const float *icp_weight_prev, int rows,
                        int cols, float dist_thresh, float angle_thresh, int use_weight, int use_search, int radius,
                        int32_t *corres_out, double A_out[36], double b_out[6], double residual_out[2])
{
    orc_planar vc = {rows, cols, (float *)vmap_curr}, nc = {rows, cols, (float *)nmap_curr};
    orc_planar k1c = {rows, cols, (float *)ck1_curr}, k2c = {rows, cols, (float *)ck2_curr};
    orc_planar vg = {rows, cols, (float *)vmap_g_prev}, ng = {rows, cols, (float *)nmap_g_prev};
    orc_planar k1g = {rows, cols, (float *)ck1_g_prev}, k2g = {rows, cols, (float *)ck2_g_prev};
    double s[29];
    icp_step(Rcurr, v3(tcurr[0], tcurr[1], tcurr[2]), &vc, &nc, &k1c, &k2c, Rprev_inv,
             v3(tprev[0], tprev[1], tprev[2]), fx, fy, cx, cy, &vg, &ng, &k1g, &k2g, icp_weight_prev,
             dist_thresh, angle_thresh, use_search, radius, use_weight, NULL, corres_out, s);
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            double v = s[shift++];
            if (j == 6) b_out[i] = v; else A_out[j * 6 + i] = A_out[i * 6 + j] = v;
        }
    residual_out[0] = s[27]; residual_out[1] = s[28];
    return 0;
}

/* icpStep with useSparse = true and updateLambdaMap as stand-alone operators on host images (lambda / z: 3 interleaved
   floats per pixel, corres: 2 int32 per pixel) — the seam hrbf_icp_step_sparse / hrbf_update_lambda_map is checked against */
int orc_icp_step_sparse(const float Rcurr[9], const float tcurr[3], const float *vmap_curr, const float *nmap_curr,
                        const float *ck1_curr, const float *ck2_curr, const float Rprev_inv[9], const float tprev[3],
                        float fx, float fy, float cx, float cy, const float *vmap_g_prev, const float *nmap_g_prev,
                        const float *ck1_g_prev, const float *ck2_g_prev, const float *icp_weight_prev, int rows,
                        int cols, float dist_thresh, float angle_thresh, int use_weight, const float *lambda_map,
                        float *z_map_out, int32_t *corres_out, double A_out[36], double b_out[6], double residual_out[2])
{
    orc_planar vc = {rows, cols, (float *)vmap_curr}, nc = {rows, cols, (float *)nmap_curr};
    orc_planar k1c = {rows, cols, (float *)ck1_curr}, k2c = {rows, cols, (float *)ck2_curr};
    orc_planar vg = {rows, cols, (float *)vmap_g_prev}, ng = {rows, cols, (float *)nmap_g_prev};
    orc_planar k1g = {rows, cols, (float *)ck1_g_prev}, k2g = {rows, cols, (float *)ck2_g_prev};
    orc_sparse sp = {(f3 *)lambda_map, (f3 *)z_map_out, corres_out, NULL};
    double s[29];
    icp_step(Rcurr, v3(tcurr[0], tcurr[1], tcurr[2]), &vc, &nc, &k1c, &k2c, Rprev_inv,
             v3(tprev[0], tprev[1], tprev[2]), fx, fy, cx, cy, &vg, &ng, &k1g, &k2g, icp_weight_prev,
             dist_thresh, angle_thresh, 0, 0, use_weight, &sp, NULL, s);
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            double v = s[shift++];
            if (j == 6) b_out[i] = v; else A_out[j * 6 + i] = A_out[i * 6 + j] = v;
        }
    residual_out[0] = s[27]; residual_out[1] = s[28];
    return 0;
}
int orc_update_lambda_map(const float Rcurr[9], const float tcurr[3], const float *vmap_curr, const float Rprev_inv[9],
                          const float tprev[3], const float *vmap_g_prev, const int32_t *corres, const float *z_map,
                          float *lambda_map, int rows, int cols)
{
    orc_planar vc = {rows, cols, (float *)vmap_curr}, vg = {rows, cols, (float *)vmap_g_prev};
    orc_sparse sp = {(f3 *)lambda_map, (f3 *)z_map, (int32_t *)corres, NULL};
    sparse_update_lambda(Rcurr, v3(tcurr[0], tcurr[1], tcurr[2]), &vc, Rprev_inv, v3(tprev[0], tprev[1], tprev[2]), &vg, &sp);
    return 0;
}

/* ------------------------------------------------------------------ O3: RGBResidual reduce.cu:957-1154 */
static void rgb_residual_core(int rows, int cols, const int16_t *dIdx, const int16_t *dIdy, const float *lastDepth,
                              const float *nextDepth, const uint8_t *lastImage, const uint8_t *nextImage, float minScale,
                              const float *krk, f3 kt, int16_t *corres, float *corres_diff, int64_t *count_out,
                              int64_t *sigma_out)
{
    const float maxDepthDelta = 0.07f;
    int64_t cnt = 0, sig = 0;
#pragma omp parallel for schedule(static) reduction(+ : cnt, sig)
    for (int i = 0; i < rows; ++i)
        for (int j0 = 0; j0 < cols; ++j0) {
            int k = i * cols + j0;
            int16_t *co = &corres[(size_t)k * 6];
            co[0] = co[1] = co[2] = co[3] = co[4] = co[5] = 0;
            corres_diff[k] = 0.0f;
#if ORC_MUTANT != 52     /* 52: no border margin — every pixel of the image is a candidate (reduce.cu:999 drops the last 5 columns and the last row) */
            if (!(j0 < cols - 5 && i < rows - 1)) continue;
#endif
            int valid = 1;
            for (int u = (i - 2 > 0 ? i - 2 : 0); u < (i + 2 < rows ? i + 2 : rows); ++u)
                for (int v = (j0 - 2 > 0 ? j0 - 2 : 0); v < (j0 + 2 < cols ? j0 + 2 : cols); ++v)
                    valid = valid && (nextImage[u * cols + v] > 0);
#if ORC_MUTANT == 50     /* the "not an isolated pixel" window dropped: black live pixels nearby do not matter (reduce.cu:1003-1010) */
            valid = 1;
#endif
            if (!valid) continue;
            int valx = dIdx[k], valy = dIdy[k];
            float mTwo = (float)((valx * valx) + (valy * valy));
            if (!(mTwo >= minScale)) continue;
            int y = i, x = j0;
            float d1 = nextDepth[y * cols + x];
            if (hd_isnanf(d1)) continue;
            float td1 = d1 * ((krk[6] * (float)x + krk[7] * (float)y) + krk[8]) + kt.z;
            float fu = (d1 * ((krk[0] * (float)x + krk[1] * (float)y) + krk[2]) + kt.x) / td1;
            float fv = (d1 * ((krk[3] * (float)x + krk[4] * (float)y) + krk[5]) + kt.y) / td1;
            if (!(fu > -1.0e9f && fu < 1.0e9f && fv > -1.0e9f && fv < 1.0e9f)) continue;
#if ORC_MUTANT == 4      /* the model image looked up at the truncated instead of the nearest texel */
            int u0 = (int)floorf(fu), v0 = (int)floorf(fv);
#else
            int u0 = (int)rintf(fu), v0 = (int)rintf(fv);
#endif
            if (!(u0 >= 0 && v0 >= 0 && u0 < cols && v0 < rows)) continue;
            float d0 = lastDepth[v0 * cols + u0];
#if ORC_MUTANT == 16     /* the depth gate on the live pixel's own depth instead of its depth in the model camera */
            if (d0 > 0.0f && fabsf(d1 - d0) <= maxDepthDelta && lastImage[v0 * cols + u0] != 0) {
#else
#if ORC_MUTANT == 51     /* a black model pixel accepted as a correspondence (`lastImage != 0` dropped, reduce.cu:1039) */
            if (d0 > 0.0f && fabsf(td1 - d0) <= maxDepthDelta) {
#else
            if (d0 > 0.0f && fabsf(td1 - d0) <= maxDepthDelta && lastImage[v0 * cols + u0] != 0) {
#endif
#endif
                float diff = (float)nextImage[y * cols + x] - (float)lastImage[v0 * cols + u0];
                co[0] = (int16_t)u0; co[1] = (int16_t)v0; co[2] = (int16_t)x; co[3] = (int16_t)y; co[4] = 1;
                corres_diff[k] = diff;
                cnt += 1;
                sig += (int64_t)(diff * diff);
            }
        }
    *count_out = cnt; *sigma_out = sig;
}
static void rgb_residual(orc_ctx *c, int lvl, float minScale, const float *krk, f3 kt, int64_t *count_out,
                         int64_t *sigma_out)
{
    rgb_residual_core(c->H >> lvl, c->W >> lvl, c->dIdx[lvl], c->dIdy[lvl], c->last_depth[lvl], c->next_depth[lvl],
                      c->last_image[lvl], c->next_image[lvl], minScale, krk, kt, c->corres, c->corres_diff, count_out,
                      sigma_out);
}

/* ------------------------------------------------------------------ O5: RGBReduction reduce.cu:697-896 */
static void rgb_step_core(int rows, int cols, const int16_t *corres, const float *corres_diff, const f3 *cloud,
                          const int16_t *dIdxl, const int16_t *dIdyl, float sigma, float fx, float fy, int use_grad,
                          double sums[29])
{
#if ORC_MUTANT == 5      /* Sobel scale 1 / 2^2 instead of 1 / 2^3 */
    const float sobelScale = 0.25f;
#else
    const float sobelScale = 0.125f;
#endif
    acc29 tot; for (int i = 0; i < 29; ++i) hd_acc_zero(&tot.a[i]);
#pragma omp parallel
    {
        acc29 loc; for (int i = 0; i < 29; ++i) hd_acc_zero(&loc.a[i]);
#pragma omp for schedule(static) nowait
        for (int k = 0; k < rows * cols; ++k) {
            const int16_t *co = &corres[(size_t)k * 6];
            if (!co[4]) continue;
            float diff = corres_diff[k];
#if ORC_MUTANT == 19     /* the photometric weight 1 / sigma: no down-weighting of large residuals */
            float w = sigma;
#else
            float w = sigma + fabsf(diff);
#endif
            w = w > 1.19209290e-07f ? 1.0f / w : 1.0f;
#if ORC_MUTANT != 55     /* 55: the rgbOnly signal sigma == -1 not honoured: w stays 1 / (sigma + |diff|) (reduce.cu:737-740) */
            if (sigma == -1.0f) w = 1.0f;
#endif
            float row[7];
            row[6] = -w * diff;
#if ORC_MUTANT == 54     /* the row's 3-D point read at the LIVE pixel (`one`) instead of the model pixel (`zero`) (reduce.cu:744-746) */
            f3 cp = cloud[co[3] * cols + co[2]];
#else
            f3 cp = cloud[co[1] * cols + co[0]];
#endif
            float invz = 1.0f / cp.z;
#if ORC_MUTANT == 12     /* the gradient read at the model pixel (`zero`) instead of the live pixel (`one`) */
            float dIx = w * sobelScale * (float)dIdxl[co[1] * cols + co[0]];
            float dIy = w * sobelScale * (float)dIdyl[co[1] * cols + co[0]];
#else
            float dIx = w * sobelScale * (float)dIdxl[co[3] * cols + co[2]];
            float dIy = w * sobelScale * (float)dIdyl[co[3] * cols + co[2]];
#endif
            float v0 = dIx * fx * invz, v1 = dIy * fy * invz;
#if ORC_MUTANT == 18     /* the depth column of the photometric row without its second division by z */
            float v2 = -(v0 * cp.x + v1 * cp.y);
#else
            float v2 = -(v0 * cp.x + v1 * cp.y) * invz;
#endif
            row[0] = v0; row[1] = v1; row[2] = v2;
#if ORC_MUTANT == 3      /* the rotational columns of the photometric row with the cross product the other way round */
            row[3] = cp.z * v1 - cp.y * v2;
#else
            row[3] = -cp.z * v1 + cp.y * v2;
#endif
            row[4] = cp.z * v0 - cp.x * v2;
            row[5] = -cp.y * v0 + cp.x * v1;
#if ORC_MUTANT == 3
            row[4] = -row[4]; row[5] = -row[5];
#endif
            float rw = 1.0f;
            if (use_grad) {
                float gm = sqrtf(dIx * dIx + dIy * dIy);
#if ORC_MUTANT == 56     /* the gradient weight with the ratio upside down: exp(-0.5 (grad / 10)^2) (reduce.cu:757-758) */
                rw = hd_expf(-0.5f * (gm / 10.0f) * (gm / 10.0f));
#else
                rw = hd_expf(-0.5f * (10.0f / gm) * (10.0f / gm));
#endif
            }
            int q = 0;
            for (int i = 0; i < 6; ++i) for (int j = i; j < 7; ++j) hd_acc_add_f32(&loc.a[q++], rw * row[i] * row[j]);
            hd_acc_add_f32(&loc.a[27], rw * row[6] * row[6]);
            hd_acc_add_f32(&loc.a[28], 1.0f);
        }
#pragma omp critical
        for (int i = 0; i < 29; ++i) hd_acc_add(&tot.a[i], loc.a[i]);
    }
    for (int i = 0; i < 29; ++i) sums[i] = hd_acc_to_double(tot.a[i]);
}
static void rgb_step(orc_ctx *c, int lvl, float sigma, float fx, float fy, double sums[29])
{
    rgb_step_core(c->H >> lvl, c->W >> lvl, c->corres, c->corres_diff, c->cloud[lvl], c->dIdx[lvl], c->dIdy[lvl], sigma,
                  fx, fy, c->prm.rgb_use_grad_weight, sums);
}

/* ------------------------------------------------------------------ O2: SO3Reduction reduce.cu:1156-1359 */
static inline void so3_grad(const uint8_t *img, int cols, int x, int y, float *gx, float *gy)
{
    float actu = (float)img[y * cols + x];
    float back = (float)img[y * cols + x - 1], fore = (float)img[y * cols + x + 1];
    *gx = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
    back = (float)img[(y - 1) * cols + x]; fore = (float)img[(y + 1) * cols + x];
    *gy = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
}

static void so3_step(const uint8_t *lastImage, const uint8_t *nextImage, int rows, int cols,
                     const float *basis, const float *kinv, const float *krlr, double sums[11])
{
    hd_acc128 tot[11]; for (int i = 0; i < 11; ++i) hd_acc_zero(&tot[i]);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            f3 un = v3((float)x, (float)y, 1.0f);
            f3 wp = m33_mul(basis, un);
            float fu = wp.x / wp.z, fv = wp.y / wp.z;
            if (!(fu > -1.0e9f && fu < 1.0e9f && fv > -1.0e9f && fv < 1.0e9f)) continue;
            int wx = (int)rintf(fu), wy = (int)rintf(fv);
            if (!(wx >= 1 && wx < cols - 1 && wy >= 1 && wy < rows - 1 && x >= 1 && x < cols - 1 && y >= 1 &&
                  y < rows - 1))
                continue;
            float gnx, gny, glx, gly;
            so3_grad(nextImage, cols, wx, wy, &gnx, &gny);
            so3_grad(lastImage, cols, x, y, &glx, &gly);
#if ORC_MUTANT == 53     /* the SO3 row's image gradient from the warped live image alone, not the mean of both images' (reduce.cu:1224-1225) */
            float gx = gnx, gy = gny; (void)glx; (void)gly;
#else
            float gx = (gnx + glx) / 2.0f, gy = (gny + gly) / 2.0f;
#endif
            f3 point = m33_mul(kinv, un);
            float z2 = point.z * point.z;
            float a = krlr[0], b = krlr[1], cc = krlr[2], d = krlr[3], e = krlr[4], f = krlr[5], g = krlr[6],
                  h = krlr[7], i_ = krlr[8];
            float fxp = (float)x, fyp = (float)y;
            f3 lp = v3((((point.z * (d * gy + a * gx)) - (gy * g * fyp)) - (gx * g * fxp)) / z2,
                       (((point.z * (e * gy + b * gx)) - (gy * h * fyp)) - (gx * h * fxp)) / z2,
                       (((point.z * (f * gy + cc * gx)) - (gy * i_ * fyp)) - (gx * i_ * fxp)) / z2);
            f3 jr = cross3(lp, point);
            float row[4] = {jr.x, jr.y, jr.z,
#if ORC_MUTANT == 11     /* SO3 residual last - next ... with the sign of next - last */
                            ((float)nextImage[wy * cols + wx] - (float)lastImage[y * cols + x])};
#else
                            -((float)nextImage[wy * cols + wx] - (float)lastImage[y * cols + x])};
#endif
            int q = 0;
            for (int i = 0; i < 3; ++i) for (int j = i; j < 4; ++j) hd_acc_add_f32(&tot[q++], row[i] * row[j]);
            hd_acc_add_f32(&tot[9], row[3] * row[3]);
            hd_acc_add_f32(&tot[10], 1.0f);
        }
    for (int i = 0; i < 11; ++i) sums[i] = hd_acc_to_double(tot[i]);
}

/* operator-level seams on caller-provided host arrays (cudafuncs.cuh:118-162): so3Step, computeRgbResidual, rgbStep */
int orc_so3_step(const uint8_t *lastImage, const uint8_t *nextImage, int rows, int cols, const float basis[9],
                 const float kinv[9], const float krlr[9], double A_out[9], double b_out[3], double residual_out[2])
{
    double s[11];
    so3_step(lastImage, nextImage, rows, cols, basis, kinv, krlr, s);
    int shift = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 4; ++j) {
            double v = s[shift++];
            if (j == 3) b_out[i] = v; else A_out[j * 3 + i] = A_out[i * 3 + j] = v;
        }
    residual_out[0] = s[9]; residual_out[1] = s[10];
    return 0;
}
int orc_rgb_residual(float minScale, const int16_t *dIdx, const int16_t *dIdy, const float *lastDepth,
                     const float *nextDepth, const uint8_t *lastImage, const uint8_t *nextImage, int rows, int cols,
                     const float kt[3], const float krkinv[9], int16_t *corres_out, float *diff_out, long long *count,
                     long long *sigma)
{
    int64_t c = 0, sg = 0;
    rgb_residual_core(rows, cols, dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage, minScale, krkinv,
                      v3(kt[0], kt[1], kt[2]), corres_out, diff_out, &c, &sg);
    *count = (long long)c; *sigma = (long long)sg;
    return 0;
}
int orc_rgb_step(const int16_t *corres, const float *corres_diff, float sigma, const float *cloud, float fx, float fy,
                 const int16_t *dIdx, const int16_t *dIdy, int use_grad_weight, int rows, int cols, double A_out[36],
                 double b_out[6], double residual_out[2])
{
    double s[29];
    rgb_step_core(rows, cols, corres, corres_diff, (const f3 *)cloud, dIdx, dIdy, sigma, fx, fy, use_grad_weight, s);
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            double v = s[shift++];
            if (j == 6) b_out[i] = v; else A_out[j * 6 + i] = A_out[i * 6 + j] = v;
        }
    residual_out[0] = s[27]; residual_out[1] = s[28];
    return 0;
}

/* projectPointsKernel cudafuncs.cu:995-1013 */
static void project_cloud(const float *depth, int rows, int cols, float fx, float fy, float cx, float cy, f3 *cloud)
{
    float invFx = 1.0f / fx, invFy = 1.0f / fy;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float z = depth[y * cols + x];
            cloud[y * cols + x] = v3(((float)x - cx) * z * invFx, ((float)y - cy) * z * invFy, z);
        }
}

/* ------------------------------------------------------------------ O6: getIncrementalTransformation */
int orc_get_odo_trace(const orc_ctx *c, double *out, int max_rows)
{
    const int n = c->odo_trace_n < max_rows ? c->odo_trace_n : max_rows;
    memcpy(out, c->odo_trace, sizeof(double) * 128 * (size_t)n);
    return n;
}
size_t orc_get_pyramid(const orc_ctx *c, int which, int level, void *out, size_t bytes)
{
    if (level < 0 || level >= ORC_NUM_PYRS) return 0;
    const size_t n = (size_t)(c->H >> level) * (size_t)(c->W >> level);
    const orc_planar *pl[8] = {c->vmap_g, c->nmap_g, c->ck1_g, c->ck2_g, c->vmap_c, c->nmap_c, c->ck1_c, c->ck2_c};
    const void *src = NULL; size_t b = 0;
    if (which >= 0 && which < 8) { src = pl[which][level].p; b = 16 * n; }
    else if (which == 8) { src = c->icpw[level]; b = 4 * n; }
    else if (which == 9) { src = c->last_depth[level]; b = 4 * n; }
    else if (which == 10) { src = c->next_depth[level]; b = 4 * n; }
    else if (which == 11) { src = c->last_image[level]; b = n; }
    else if (which == 12) { src = c->next_image[level]; b = n; }
    else if (which == 13) { src = c->last_next_image[level]; b = n; }
    else if (which == 14) { src = c->dIdx[level]; b = 2 * n; }
    else if (which == 15) { src = c->dIdy[level]; b = 2 * n; }
    if (!src || bytes < b) return 0;
    memcpy(out, src, b);
    return b;
}

void orc_odo_track(orc_ctx *c)
{
    c->odo_trace_n = 0;
    const int rgbOnly = c->prm.rgb_only;
    const float icpWeight = c->prm.icp_weight;
    const int icp = !rgbOnly && icpWeight > 0.0f;
    const int rgb = rgbOnly || icpWeight < 100.0f;
    float Rprev[9], Rcurr[9]; f3 tprev, tcurr;
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) Rprev[r * 3 + k] = M4(c->pose, r, k);
    tprev = v3(M4(c->pose, 0, 3), M4(c->pose, 1, 3), M4(c->pose, 2, 3));
    memcpy(Rcurr, Rprev, sizeof(Rprev)); tcurr = tprev;

    if (rgb) for (int i = 0; i < ORC_NUM_PYRS; ++i) sobel(c->next_image[i], c->H >> i, c->W >> i, c->dIdx[i], c->dIdy[i]);

    double resultR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (c->prm.so3) {
        const int lvl = 2, div = 1 << lvl;
        float R_lr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        double K[9] = {0}, Kinv[9];
        K[0] = c->prm.fx / div; K[4] = c->prm.fy / div; K[2] = c->prm.cx / div; K[5] = c->prm.cy / div; K[8] = 1;
        inv3d(K, Kinv);
        float lastError = 3.402823466e+38f / 2.0f, lastCount = 3.402823466e+38f / 2.0f;
        double lastResultR[9]; memcpy(lastResultR, resultR, sizeof(resultR));
        for (int it = 0; it < 10; ++it) {
            double KR[9], Hm[9];
#if ORC_MUTANT == 24     /* the SO3 homography with the rotation transposed: K R^T K^-1 */
            { double Rtr[9]; for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) Rtr[r * 3 + k] = resultR[k * 3 + r]; mul3d(K, Rtr, KR); }
            mul3d(KR, Kinv, Hm);
#else
            mul3d(K, resultR, KR); mul3d(KR, Kinv, Hm);
#endif
            float basis[9], kinvf[9], krlr[9];
            for (int k = 0; k < 9; ++k) { basis[k] = (float)Hm[k]; kinvf[k] = (float)Kinv[k]; krlr[k] = (float)KR[k]; }
            double s[11];
            so3_step(c->last_next_image[lvl], c->next_image[lvl], c->H >> lvl, c->W >> lvl, basis, kinvf, krlr, s);
            float jtj[9], jtr[3];
            int shift = 0;
            for (int i = 0; i < 3; ++i)
                for (int j = i; j < 4; ++j) {
                    float value = (float)s[shift++];
                    if (j == 3) jtr[i] = value; else jtj[j * 3 + i] = jtj[i * 3 + j] = value;
                }
            if (c->odo_trace_n < 40) {
                double *tr = c->odo_trace[c->odo_trace_n++];
                memset(tr, 0, sizeof(double) * 128);
                tr[0] = -1; tr[1] = it;
                for (int k = 0; k < 9; ++k) tr[96 + k] = resultR[k];
                for (int k = 0; k < 9; ++k) tr[2 + k] = jtj[k];
                for (int k = 0; k < 3; ++k) tr[11 + k] = jtr[k];
                tr[14] = s[9]; tr[15] = s[10];
            }
            float res0 = (float)s[9], res1 = (float)s[10];
            float so3err = sqrtf(res0) / res1, so3cnt = res1;
            if (so3err < lastError && lastCount == so3cnt) break;
            else if (so3err > lastError + 0.001f) { memcpy(resultR, lastResultR, sizeof(resultR)); break; }
            lastError = so3err; lastCount = so3cnt; memcpy(lastResultR, resultR, sizeof(resultR));
            float delta[3];
            ldlt_f(3, jtj, jtr, delta);
            double dd[3] = {delta[0], delta[1], delta[2]}, rotU[9];
            rodrigues(dd, rotU);
            float rotUf[9], tmp[9];
            for (int k = 0; k < 9; ++k) rotUf[k] = (float)rotU[k];
            mul3f(rotUf, R_lr, tmp); memcpy(R_lr, tmp, sizeof(tmp));
            for (int k = 0; k < 9; ++k) resultR[k] = R_lr[k];
        }
    }
    int iterations[3];
    iterations[0] = c->prm.fast_odom ? 3 : 10;
    iterations[1] = c->prm.pyramid ? 5 : 0;
    iterations[2] = c->prm.pyramid ? 4 : 0;
#if ORC_MUTANT == 21     /* the schedule 10 / 5 / 4 given to the levels the other way round (most iterations on the coarsest) */
    if (c->prm.pyramid) { iterations[0] = 4; iterations[2] = c->prm.fast_odom ? 3 : 10; }
#endif

    float Rprev_inv[9];
    {   /* Eigen 3x3 float inverse -> cofactor inverse */
        float a = Rprev[0], b = Rprev[1], cc = Rprev[2], d = Rprev[3], e = Rprev[4], f = Rprev[5], g = Rprev[6],
              h = Rprev[7], i = Rprev[8];
        float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
        float det = (a * c00 + b * c01) + cc * c02, id = 1.0f / det;
        Rprev_inv[0] = c00 * id; Rprev_inv[1] = (cc * h - b * i) * id; Rprev_inv[2] = (b * f - cc * e) * id;
        Rprev_inv[3] = c01 * id; Rprev_inv[4] = (a * i - cc * g) * id; Rprev_inv[5] = (cc * d - a * f) * id;
        Rprev_inv[6] = c02 * id; Rprev_inv[7] = (b * g - a * h) * id; Rprev_inv[8] = (a * e - b * d) * id;
    }
    double Rt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   /* resultRt, row-major */
    if (c->prm.so3) for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) Rt[r * 4 + k] = resultR[r * 3 + k];

    const float distThres = 0.1f, angleThres = 0.3420201433f; /* sin(20 deg), RGBDOdometry.h:65-66 */
    const float minGrad[3] = {5, 3, 1};
    const double sobelScale = 0.125;
    float res_icp[2] = {0, 0};

    for (int i = ORC_NUM_PYRS - 1; i >= 0; --i) {
        const int div = 1 << i;
#if ORC_MUTANT == 10     /* the principal point not divided by 2^level */
        const float fxl = c->prm.fx / div, fyl = c->prm.fy / div, cxl = c->prm.cx, cyl = c->prm.cy;
#else
        const float fxl = c->prm.fx / div, fyl = c->prm.fy / div, cxl = c->prm.cx / div, cyl = c->prm.cy / div;
#endif
        if (rgb) project_cloud(c->last_depth[i], c->H >> i, c->W >> i, fxl, fyl, cxl, cyl, c->cloud[i]);
        double K[9] = {0}, Kinv[9];
        K[0] = fxl; K[4] = fyl; K[2] = cxl; K[5] = cyl; K[8] = 1;
        inv3d(K, Kinv);
        float lastRGBError = 3.402823466e+38f;
        const int sparse = icp && c->prm.use_sparse_icp;
        orc_sparse sp = {c->sp_lambda[i], c->sp_z[i], c->sp_corres[i], &c->sp_shrunk};
        if (sparse) memset(sp.lambda, 0, sizeof(f3) * (size_t)(c->H >> i) * (c->W >> i));   /* RGBDOdometry.cpp:964-977 */
        for (int j = 0; j < iterations[i]; ++j) {
            double state0[28];
            memcpy(state0, Rt, sizeof(Rt));
            for (int k = 0; k < 9; ++k) state0[16 + k] = Rcurr[k];
            state0[25] = tcurr.x; state0[26] = tcurr.y; state0[27] = tcurr.z;
            /* Rt = resultRt.inverse() (rigid: cofactor inverse of the linear part) */
            double L[9], Li[9], ti[3];
            for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) L[r * 3 + k] = Rt[r * 4 + k];
            inv3d(L, Li);
            for (int r = 0; r < 3; ++r)
                ti[r] = -((Li[r * 3] * Rt[3] + Li[r * 3 + 1] * Rt[7]) + Li[r * 3 + 2] * Rt[11]);
#if ORC_MUTANT == 17     /* the photometric warp built from resultRt itself instead of its inverse (RGBDOdometry.cpp:981) */
            memcpy(Li, L, sizeof(L)); ti[0] = Rt[3]; ti[1] = Rt[7]; ti[2] = Rt[11];
#endif
            double KR[9], KRK[9];
            mul3d(K, Li, KR); mul3d(KR, Kinv, KRK);
            float krk[9]; for (int k = 0; k < 9; ++k) krk[k] = (float)KRK[k];
            double Ktd[3];
            for (int r = 0; r < 3; ++r) Ktd[r] = (K[r * 3] * ti[0] + K[r * 3 + 1] * ti[1]) + K[r * 3 + 2] * ti[2];
            f3 kt = v3((float)Ktd[0], (float)Ktd[1], (float)Ktd[2]);

            int64_t sigma = 0, rgbSize = 0, sigma_unwrapped = 0;
            if (rgb) {
#if ORC_MUTANT == 23     /* the gradient threshold compared unsquared with the squared magnitude */
                float minScale = (float)((double)minGrad[i] / sobelScale);
#else
                float minScale = (float)(((double)minGrad[i] * (double)minGrad[i]) / (sobelScale * sobelScale));
#endif
                rgb_residual(c, i, minScale, krk, kt, &rgbSize, &sigma);
                /* `int sigma` (RGBDOdometry.cpp:994) summed in int2 on the device (reduce.cu:985-1046): wraps beyond 2^31 */
                sigma_unwrapped = sigma;
                sigma = (int64_t)(int32_t)(uint32_t)(uint64_t)sigma;
            }
            /* RGBDOdometry.cpp:1017-1018 incl. the precedence quirk */
#if ORC_MUTANT == 8      /* sigma as the rms residual: what the expression looks like it means, not what its precedence says */
            float sigmaVal = sqrtf((float)sigma / (float)(rgbSize == 0 ? 1 : rgbSize));
#else
            float sigmaVal = sqrtf(((float)sigma / (float)rgbSize == 0.0f) ? 1.0f : (float)rgbSize);
#endif
            float rgbError = (float)(sqrt((double)sigma) / (double)(rgbSize == 0 ? 1 : rgbSize));
            if (rgbOnly && rgbError > lastRGBError) break;
            lastRGBError = rgbError;
            if (rgbOnly) sigmaVal = -1.0f;

            float A_icp[36], b_icp[6], A_rgb[36], b_rgb[6];
            memset(A_icp, 0, sizeof(A_icp)); memset(b_icp, 0, sizeof(b_icp));
            memset(A_rgb, 0, sizeof(A_rgb)); memset(b_rgb, 0, sizeof(b_rgb));
            if (icp) {
                double s[29];
                icp_step(Rcurr, tcurr, &c->vmap_c[i], &c->nmap_c[i], &c->ck1_c[i], &c->ck2_c[i], Rprev_inv, tprev,
                         fxl, fyl, cxl, cyl, &c->vmap_g[i], &c->nmap_g[i], &c->ck1_g[i], &c->ck2_g[i], c->icpw[i],
                         distThres, angleThres, c->prm.icp_use_corr_search, c->prm.icp_search_radius,
                         c->prm.icp_use_weighted, sparse ? &sp : NULL, NULL, s);
                unpack27(s, A_icp, b_icp);
                res_icp[0] = (float)s[27]; res_icp[1] = (float)s[28];
            }
            c->last_icp_error = sqrtf(res_icp[0]) / res_icp[1];
            c->last_icp_count = res_icp[1];
            if (rgb) {
                double s[29];
                rgb_step(c, i, sigmaVal, fxl, fyl, s);
                unpack27(s, A_rgb, b_rgb);
            }
            double lastA[36], lastb[6], result[6];
            if (icp && rgb) {
                double w = icpWeight, ww = w * w;
#if ORC_MUTANT == 6      /* A_rgb + w A_icp (the weight not squared on the matrix) */
                for (int k = 0; k < 36; ++k) lastA[k] = (double)A_rgb[k] + w * (double)A_icp[k];
#else
                for (int k = 0; k < 36; ++k) lastA[k] = (double)A_rgb[k] + ww * (double)A_icp[k];
#endif
#if ORC_MUTANT == 7      /* b_rgb + w^2 b_icp (a consistent weighting: NOT what the reference does) */
                for (int k = 0; k < 6; ++k) lastb[k] = (double)b_rgb[k] + ww * (double)b_icp[k];
#else
                for (int k = 0; k < 6; ++k) lastb[k] = (double)b_rgb[k] + w * (double)b_icp[k];
#endif
            } else if (icp) {
                for (int k = 0; k < 36; ++k) lastA[k] = A_icp[k];
                for (int k = 0; k < 6; ++k) lastb[k] = b_icp[k];
            } else {
                for (int k = 0; k < 36; ++k) lastA[k] = A_rgb[k];
                for (int k = 0; k < 6; ++k) lastb[k] = b_rgb[k];
            }
            ldlt_d(6, lastA, lastb, result);
            if (c->odo_trace_n < 40) {
                double *tr = c->odo_trace[c->odo_trace_n++];
                tr[0] = i; tr[1] = j;
                for (int k = 0; k < 36; ++k) { tr[2 + k] = A_icp[k]; tr[44 + k] = A_rgb[k]; }
                for (int k = 0; k < 6; ++k) { tr[38 + k] = b_icp[k]; tr[80 + k] = b_rgb[k]; tr[86 + k] = result[k]; }
                tr[92] = res_icp[1]; tr[93] = (double)rgbSize; tr[94] = (double)sigma; tr[95] = res_icp[0];
                memcpy(tr + 96, state0, sizeof(state0));
                tr[124] = (double)sigma_unwrapped;
            }
            /* computeUpdateSE3 OdometryProvider.h:73-93 */
            double rv[3] = {result[3], result[4], result[5]}, Ru[9], U[16], N[16];
            rodrigues(rv, Ru);
            for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) U[r * 4 + k] = Ru[r * 3 + k]; U[r * 4 + 3] = result[r]; }
            U[12] = U[13] = U[14] = 0; U[15] = 1;
            for (int r = 0; r < 4; ++r) for (int k = 0; k < 4; ++k)
#if ORC_MUTANT == 22     /* the increment composed on the right: resultRt * update instead of update * resultRt (OdometryProvider.h:91) */
                N[r * 4 + k] = ((Rt[r * 4] * U[k] + Rt[r * 4 + 1] * U[4 + k]) + Rt[r * 4 + 2] * U[8 + k]) + Rt[r * 4 + 3] * U[12 + k];
#else
                N[r * 4 + k] = ((U[r * 4] * Rt[k] + U[r * 4 + 1] * Rt[4 + k]) + U[r * 4 + 2] * Rt[8 + k]) + U[r * 4 + 3] * Rt[12 + k];
#endif
            memcpy(Rt, N, sizeof(N));
            float oR[9], ot[3];
            for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) oR[r * 3 + k] = (float)Rt[r * 4 + k]; ot[r] = (float)Rt[r * 4 + 3]; }
            /* currentT = [Rprev|tprev] * rgbOdom.inverse(); inverse = (R^T, -R^T t) */
            float iR[9], it_[3];
#if ORC_MUTANT == 9      /* T_prev * dT instead of T_prev * dT^-1 */
            for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) iR[r * 3 + k] = oR[r * 3 + k];
            for (int r = 0; r < 3; ++r) it_[r] = ot[r];
#else
            for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) iR[r * 3 + k] = oR[k * 3 + r];
            for (int r = 0; r < 3; ++r) it_[r] = -((iR[r * 3] * ot[0] + iR[r * 3 + 1] * ot[1]) + iR[r * 3 + 2] * ot[2]);
#endif
            mul3f(Rprev, iR, Rcurr);
            f3 rt = m33_mul(Rprev, v3(it_[0], it_[1], it_[2]));
            tcurr = add3(rt, tprev);
            if (sparse)   /* RGBDOdometry.cpp:1206-1227 */
                sparse_update_lambda(Rcurr, tcurr, &c->vmap_c[i], Rprev_inv, tprev, &c->vmap_g[i], &sp);
        }
    }
    if (rgb) {
        f3 d = sub3(tcurr, tprev);
#if ORC_MUTANT != 43     /* 43: no 0.3 m guard (RGBDOdometry.cpp:1232-1236) */
        if (len3(d) > 0.3f) { memcpy(Rcurr, Rprev, sizeof(Rprev)); tcurr = tprev; }
#endif
    }
    if (c->prm.so3)
        for (int i = 0; i < ORC_NUM_PYRS; ++i) { uint8_t *t = c->last_next_image[i]; c->last_next_image[i] = c->next_image[i]; c->next_image[i] = t; }
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) c->pose[k * 4 + r] = Rcurr[r * 3 + k];
    c->pose[12] = tcurr.x; c->pose[13] = tcurr.y; c->pose[14] = tcurr.z;
}

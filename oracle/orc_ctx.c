/*
 * orc_ctx.c — context + per-frame orchestration (oracle; test infrastructure only).
 * Follows HRBFFusion::processFrame / predict (Core/src/HRBFFusion.cpp:991-1260) with the sparse
 * back-end off (optimizationUseLocalBA/GlobalBA = false).
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>
/* ORC_MUTANT: see orc_odo.c (deliberate misreadings for tools/mutation_report.py; 0 = the oracle) */
#ifndef ORC_MUTANT
#define ORC_MUTANT 0
#endif
#include "oracle.h"
#include "orc_vec.h"

void orc_filter_depth(orc_ctx *c);
void orc_metricise(orc_ctx *c);
void orc_vertex_normal_radius(orc_ctx *c);
void orc_curvature(orc_ctx *c);
void orc_confidence(orc_ctx *c);
void orc_initialise(orc_ctx *c);
void orc_predict_indices(orc_ctx *c);
void orc_fuse(orc_ctx *c);
void orc_clean(orc_ctx *c);
void orc_predict_hrbf(orc_ctx *c);
void orc_fillin(orc_ctx *c);
int orc_dense_enough(const orc_ctx *c);
void orc_odo_init_model(orc_ctx *c, const f4 *vtex, const f4 *ntex, const uint8_t *img4, const f4 *k1tex,
                        const f4 *k2tex, const float *icpw_tex);
void orc_odo_init_live(orc_ctx *c);
void orc_odo_init_first_rgb(orc_ctx *c);
void orc_odo_track(orc_ctx *c);

static double now_ms(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static void *zalloc(size_t n) { void *p = calloc(1, n ? n : 1); return p; }
static void planar_alloc(orc_planar *m, int rows, int cols)
{
    m->rows = rows; m->cols = cols; m->p = (float *)zalloc(sizeof(float) * 4 * rows * cols);
}

orc_ctx *orc_create(const hrbf_params *p)
{
    orc_ctx *c = (orc_ctx *)zalloc(sizeof(orc_ctx));
    c->prm = *p;
    c->W = p->width; c->H = p->height; c->P = c->W * c->H;
    const int P = c->P;
    c->tick = 1;
    for (int i = 0; i < 16; ++i) c->pose[i] = c->prev_pose[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    c->weighting = 1.0f;
    c->rgb = zalloc(P * 3); c->depth_raw = zalloc(P * 2);
    c->depth_filtered = zalloc(P * 4); c->depth_metric = zalloc(P * 4); c->depth_metric_filtered = zalloc(P * 4);
#define A4(x) c->x = (f4 *)zalloc(sizeof(f4) * P)
    A4(vertex_raw); A4(vertex_filtered); A4(normal); A4(normal_pca); A4(normal_opt); A4(curv1); A4(curv2);
    A4(im_vertconf); A4(im_colortime); A4(im_normrad); A4(im_curvmax); A4(im_curvmin);
    A4(pr_vertex); A4(pr_normal); A4(pr_curv1); A4(pr_curv2);
    A4(fi_vertex); A4(fi_normal); A4(fi_curv1); A4(fi_curv2);
#undef A4
    c->radius = zalloc(P * 4); c->gradmag = zalloc(P * 4); c->confidence = zalloc(P * 4);
    c->idx = zalloc(P * 4); c->pr_time = zalloc(P * 4); c->pr_icpw = zalloc(P * 4); c->fi_icpw = zalloc(P * 4);
    c->pr_image = zalloc(P * 4); c->fi_image = zalloc(P * 4);
    c->cap = (uint32_t)p->max_surfels;
    c->map[0] = (f4 *)zalloc(sizeof(f4) * 5 * (size_t)c->cap);
    c->map[1] = (f4 *)zalloc(sizeof(f4) * 5 * (size_t)c->cap);
    c->Q = (c->W / 2) * (c->H / 2);
    c->rec = (f4 *)zalloc(sizeof(f4) * 5 * c->Q);
    c->rec_flag = zalloc(sizeof(int32_t) * c->Q); c->rec_best = zalloc(sizeof(uint32_t) * c->Q);
    for (int i = 0; i < ORC_NUM_PYRS; ++i) {
        int r = c->H >> i, w = c->W >> i;
        planar_alloc(&c->vmap_g[i], r, w); planar_alloc(&c->nmap_g[i], r, w);
        planar_alloc(&c->ck1_g[i], r, w); planar_alloc(&c->ck2_g[i], r, w);
        planar_alloc(&c->vmap_c[i], r, w); planar_alloc(&c->nmap_c[i], r, w);
        planar_alloc(&c->ck1_c[i], r, w); planar_alloc(&c->ck2_c[i], r, w);
        c->icpw[i] = zalloc(sizeof(float) * r * w);
        c->sp_lambda[i] = zalloc(sizeof(f3) * r * w); c->sp_z[i] = zalloc(sizeof(f3) * r * w);
        c->sp_corres[i] = zalloc(sizeof(int32_t) * 2 * r * w);
        c->last_depth[i] = zalloc(sizeof(float) * r * w); c->next_depth[i] = zalloc(sizeof(float) * r * w);
        c->last_image[i] = zalloc(r * w); c->next_image[i] = zalloc(r * w); c->last_next_image[i] = zalloc(r * w);
        c->dIdx[i] = zalloc(2 * r * w); c->dIdy[i] = zalloc(2 * r * w);
        c->cloud[i] = zalloc(sizeof(f3) * r * w);
    }
    c->corres = zalloc(sizeof(int16_t) * 6 * P); c->corres_diff = zalloc(sizeof(float) * P);
    return c;
}

void orc_destroy(orc_ctx *c)
{
    if (!c) return;
    free(c->submap_active); free(c->frag_tc);
    free(c->rgb); free(c->depth_raw); free(c->depth_filtered); free(c->depth_metric); free(c->depth_metric_filtered);
    free(c->vertex_raw); free(c->vertex_filtered); free(c->normal); free(c->normal_pca); free(c->normal_opt);
    free(c->curv1); free(c->curv2); free(c->im_vertconf); free(c->im_colortime); free(c->im_normrad);
    free(c->im_curvmax); free(c->im_curvmin); free(c->pr_vertex); free(c->pr_normal); free(c->pr_curv1);
    free(c->pr_curv2); free(c->fi_vertex); free(c->fi_normal); free(c->fi_curv1); free(c->fi_curv2);
    free(c->radius); free(c->gradmag); free(c->confidence); free(c->idx); free(c->pr_time); free(c->pr_icpw);
    free(c->fi_icpw); free(c->pr_image); free(c->fi_image); free(c->map[0]); free(c->map[1]); free(c->rec);
    free(c->rec_flag); free(c->rec_best);
    for (int i = 0; i < ORC_NUM_PYRS; ++i) {
        free(c->vmap_g[i].p); free(c->nmap_g[i].p); free(c->ck1_g[i].p); free(c->ck2_g[i].p);
        free(c->vmap_c[i].p); free(c->nmap_c[i].p); free(c->ck1_c[i].p); free(c->ck2_c[i].p);
        free(c->sp_lambda[i]); free(c->sp_z[i]); free(c->sp_corres[i]);
        free(c->icpw[i]); free(c->last_depth[i]); free(c->next_depth[i]); free(c->last_image[i]);
        free(c->next_image[i]); free(c->last_next_image[i]); free(c->dIdx[i]); free(c->dIdy[i]); free(c->cloud[i]);
    }
    free(c->corres); free(c->corres_diff);
    free(c);
}

int orc_upload_frame(orc_ctx *c, const uint8_t *rgb, const uint16_t *depth)
{
    memcpy(c->rgb, rgb, (size_t)c->P * 3);
    memcpy(c->depth_raw, depth, (size_t)c->P * 2);
    return 0;
}

/* HRBFFusion::rodrigues2 (HRBFFusion.cpp:2004-2050) without the JacobiSVD re-orthonormalisation
   (input is already a rotation to ~1e-7; Eigen is not available — parity unpinned). */
static f3 rodrigues2(const float *R /* row-major 3x3 */)
{
    double rx = (double)R[7] - (double)R[5], ry = (double)R[2] - (double)R[6], rz = (double)R[3] - (double)R[1];
    double s = sqrt(((rx * rx + ry * ry) + rz * rz) * 0.25);
    double cth = ((double)((R[0] + R[4]) + R[8]) - 1.0) * 0.5;
    cth = cth > 1.0 ? 1.0 : (cth < -1.0 ? -1.0 : cth);
    double theta = hd_acos(cth);
    if (s < 1e-5) {
        if (cth > 0) rx = ry = rz = 0;
        else {
            double t;
            t = ((double)R[0] + 1.0) * 0.5; rx = sqrt(t > 0.0 ? t : 0.0);
            t = ((double)R[4] + 1.0) * 0.5; ry = sqrt(t > 0.0 ? t : 0.0) * (R[1] < 0 ? -1.0 : 1.0);
            t = ((double)R[8] + 1.0) * 0.5; rz = sqrt(t > 0.0 ? t : 0.0) * (R[2] < 0 ? -1.0 : 1.0);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt((rx * rx + ry * ry) + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = 1.0 / (2.0 * s);
        vth *= theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    return v3((float)rx, (float)ry, (float)rz);
}

static void registration(orc_ctx *c)
{
    /* HRBFFusion.cpp:1069-1103 */
    int shouldFillIn = !orc_dense_enough(c);
    const f4 *vtex = shouldFillIn ? c->fi_vertex : c->pr_vertex;
    const f4 *ntex = shouldFillIn ? c->fi_normal : c->pr_normal;
    const uint8_t *img = (shouldFillIn || c->prm.frame_to_frame_rgb) ? c->fi_image : c->pr_image;
    const f4 *k1 = shouldFillIn ? c->fi_curv1 : c->pr_curv1;
    const f4 *k2 = shouldFillIn ? c->fi_curv2 : c->pr_curv2;
    const float *iw = shouldFillIn ? c->fi_icpw : c->pr_icpw;
    orc_odo_init_model(c, vtex, ntex, img, k1, k2, iw);
    orc_odo_init_live(c);
    orc_odo_track(c);
}

int orc_process_frame(orc_ctx *c, const uint8_t *rgb, const uint16_t *depth, int64_t ts, float wmul)
{
    (void)ts;
    double t0, t1;
    orc_upload_frame(c, rgb, depth);
    t0 = now_ms();
    orc_filter_depth(c); orc_metricise(c); orc_vertex_normal_radius(c); orc_curvature(c);
    t1 = now_ms(); c->timings_ms[0] = t1 - t0;
    c->timings_ms[1] = c->timings_ms[2] = 0;
    if (c->tick == 1) {
        orc_initialise(c);
        orc_odo_init_first_rgb(c);
    } else {
        float lastPose[16]; memcpy(lastPose, c->prev_pose, sizeof(lastPose));
        t0 = now_ms();
        if (!c->prm.load_trajectory) registration(c);
        t1 = now_ms(); c->timings_ms[1] = t1 - t0;
        /* velocity weighting HRBFFusion.cpp:1112-1123 */
        float inv[16], diff[16];
        rigid_inverse(c->pose, inv); mat4_mul(inv, lastPose, diff);
        f3 dt = v3(diff[12], diff[13], diff[14]);
        float Rm[9];
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) Rm[r * 3 + k] = M4(diff, r, k);
        float a = len3(dt), b = len3(rodrigues2(Rm));
#if ORC_MUTANT == 44     /* the translation alone: no max with the rotation angle (HRBFFusion.cpp:1116) */
        float weighting = a; (void)b;
#else
        float weighting = a > b ? a : b;
#endif
        const float largest = 0.01f, minWeight = 0.5f;
        if (weighting > largest) weighting = largest;
        float wv = 1.0f - (weighting / largest);
#if ORC_MUTANT == 45     /* no lower clamp at minWeight (HRBFFusion.cpp:1124) */
        c->weighting = wv * wmul;
#elif ORC_MUTANT == 46   /* weightMultiplier inside the clamp: max((1 - w / largest) * wMul, minWeight) */
        c->weighting = (wv * wmul > minWeight ? wv * wmul : minWeight);
#else
        c->weighting = (wv > minWeight ? wv : minWeight) * wmul;
#endif
        orc_confidence(c);
        if (!c->prm.rgb_only) {
            orc_predict_indices(c);
            t0 = now_ms();
            orc_fuse(c);
            t1 = now_ms(); c->timings_ms[2] = t1 - t0;
            orc_predict_indices(c);
            t0 = now_ms();
            orc_clean(c);
            c->timings_ms[4] = now_ms() - t0;
        }
    }
    /* predict() HRBFFusion.cpp:1244-1260 */
    orc_predict_indices(c);
    t0 = now_ms();
    orc_predict_hrbf(c);
    c->timings_ms[3] = now_ms() - t0;
    orc_fillin(c);
    memcpy(c->prev_pose, c->pose, sizeof(c->pose));
    c->tick++;
    return 0;
}

int orc_bootstrap(orc_ctx *c, const uint8_t *rgb, const uint16_t *depth)
{
    orc_upload_frame(c, rgb, depth);
    orc_filter_depth(c); orc_metricise(c); orc_vertex_normal_radius(c); orc_curvature(c);
    orc_odo_init_first_rgb(c);
    orc_confidence(c);
    orc_predict_indices(c); orc_predict_hrbf(c); orc_fillin(c);
    memcpy(c->prev_pose, c->pose, sizeof(c->pose));
    c->tick = 2;
    return 0;
}

int orc_run_stage(orc_ctx *c, int stage)
{
    switch (stage) {
        case HRBF_STAGE_FILTER_DEPTH: orc_filter_depth(c); break;
        case HRBF_STAGE_METRICISE: orc_metricise(c); break;
        case HRBF_STAGE_VERTEX_NORMAL_RADIUS: orc_vertex_normal_radius(c); break;
        case HRBF_STAGE_CURVATURE: orc_curvature(c); break;
        case HRBF_STAGE_CONFIDENCE: orc_confidence(c); break;
        case HRBF_STAGE_INITIALISE: orc_initialise(c); orc_odo_init_first_rgb(c); break;
        case HRBF_STAGE_PREDICT_INDICES: orc_predict_indices(c); break;
        case HRBF_STAGE_FUSE: orc_fuse(c); break;
        case HRBF_STAGE_CLEAN: orc_clean(c); break;
        case HRBF_STAGE_PREDICT_HRBF: orc_predict_hrbf(c); break;
        case HRBF_STAGE_FILLIN: orc_fillin(c); break;
        case HRBF_STAGE_ODOMETRY: registration(c); break;
        default: return -1;
    }
    return 0;
}

void orc_get_pose(orc_ctx *c, float o[16]) { memcpy(o, c->pose, 64); }
void orc_set_pose(orc_ctx *c, const float in[16]) { memcpy(c->pose, in, 64); }
int orc_get_tick(orc_ctx *c) { return c->tick; }
void orc_set_tick(orc_ctx *c, int t) { c->tick = t; }
void orc_set_index_submap(orc_ctx *c, int idx) { c->index_submap = idx; }
/* the run-time switches of the boundary (hrbf_set_rgb_only ... hrbf_set_depth_cutoff, the reference's setters HRBFFusion.h:150-215):
 * which = 0 rgb_only, 1 icp_weight, 2 pyramid, 3 fast_odom, 4 so3, 5 frame_to_frame_rgb, 6 confidence_threshold, 7 depth_cutoff */
void orc_set_switch(orc_ctx *c, int which, float v)
{
    switch (which) {
    case 0: c->prm.rgb_only = (int)v; break;
    case 1: c->prm.icp_weight = v; break;
    case 2: c->prm.pyramid = (int)v; break;
    case 3: c->prm.fast_odom = (int)v; break;
    case 4: c->prm.so3 = (int)v; break;
    case 5: c->prm.frame_to_frame_rgb = (int)v; break;
    case 6: c->prm.confidence_threshold = v; break;
    case 7: c->prm.depth_cutoff = v; break;
    default: break;
    }
}
void orc_set_active_submaps(orc_ctx *c, const uint8_t *active, int n)
{
    free(c->submap_active); c->submap_active = NULL; c->n_submap_active = 0;
    if (active && n > 0) {
        c->submap_active = (uint8_t *)malloc((size_t)n);
        memcpy(c->submap_active, active, (size_t)n);
        c->n_submap_active = n;
    }
}
void orc_set_weighting(orc_ctx *c, float w) { c->weighting = w; }
float orc_get_weighting(orc_ctx *c) { return c->weighting; }
uint32_t orc_surfel_count(orc_ctx *c) { return c->count; }
void orc_last_icp(orc_ctx *c, float *e, float *n) { *e = c->last_icp_error; *n = c->last_icp_count; }
void orc_get_fuse_stats(orc_ctx *c, uint32_t o[4]) { memcpy(o, c->fuse_stats, 16); }
void orc_get_timings(orc_ctx *c, double o[8]) { memcpy(o, c->timings_ms, sizeof(double) * 8); }

int orc_download_map(orc_ctx *c, float *out, size_t cap)
{
    if (cap < c->count) return -1;
    memcpy(out, c->map[c->target], sizeof(f4) * 5 * (size_t)c->count);
    return 0;
}
int orc_upload_map(orc_ctx *c, const float *in, size_t n)
{
    if (n > c->cap) return -1;
    memcpy(c->map[c->target], in, sizeof(f4) * 5 * n);
    c->count = (uint32_t)n;
    return 0;
}

static void *img_ptr(orc_ctx *c, int which, size_t *bytes)
{
    size_t P = (size_t)c->P;
    switch (which) {
#define I1(ID, F) case ID: *bytes = P * 4; return c->F;
#define I4(ID, F) case ID: *bytes = P * 16; return c->F;
        I1(HRBF_IMG_DEPTH_FILTERED, depth_filtered) I1(HRBF_IMG_DEPTH_METRIC, depth_metric)
        I1(HRBF_IMG_DEPTH_METRIC_FILTERED, depth_metric_filtered)
        I4(HRBF_IMG_VERTEX_RAW, vertex_raw) I4(HRBF_IMG_VERTEX_FILTERED, vertex_filtered)
        I4(HRBF_IMG_NORMAL, normal) I4(HRBF_IMG_NORMAL_PCA, normal_pca) I1(HRBF_IMG_RADIUS, radius)
        I4(HRBF_IMG_CURV1, curv1) I4(HRBF_IMG_CURV2, curv2) I1(HRBF_IMG_GRADIENT_MAG, gradmag)
        I1(HRBF_IMG_CONFIDENCE, confidence) I1(HRBF_IMG_INDEX, idx)
        I4(HRBF_IMG_INDEX_VERTCONF, im_vertconf) I4(HRBF_IMG_INDEX_COLORTIME, im_colortime)
        I4(HRBF_IMG_INDEX_NORMRAD, im_normrad) I4(HRBF_IMG_INDEX_CURVMAX, im_curvmax)
        I4(HRBF_IMG_INDEX_CURVMIN, im_curvmin)
        I1(HRBF_IMG_PRED_IMAGE, pr_image) I4(HRBF_IMG_PRED_VERTEX, pr_vertex) I4(HRBF_IMG_PRED_NORMAL, pr_normal)
        I4(HRBF_IMG_PRED_CURV1, pr_curv1) I4(HRBF_IMG_PRED_CURV2, pr_curv2) I1(HRBF_IMG_PRED_TIME, pr_time)
        I1(HRBF_IMG_PRED_ICPWEIGHT, pr_icpw)
        I1(HRBF_IMG_FILL_IMAGE, fi_image) I4(HRBF_IMG_FILL_VERTEX, fi_vertex) I4(HRBF_IMG_FILL_NORMAL, fi_normal)
        I4(HRBF_IMG_FILL_CURV1, fi_curv1) I4(HRBF_IMG_FILL_CURV2, fi_curv2) I1(HRBF_IMG_FILL_ICPWEIGHT, fi_icpw)
#undef I1
#undef I4
        default: *bytes = 0; return NULL;
    }
}
size_t orc_image_bytes(orc_ctx *c, int which) { size_t b; img_ptr(c, which, &b); return b; }
int orc_get_image(orc_ctx *c, int which, void *out, size_t bytes)
{
    size_t b; void *p = img_ptr(c, which, &b);
    if (!p || bytes < b) return -1;
    memcpy(out, p, b); return 0;
}
int orc_set_image(orc_ctx *c, int which, const void *in, size_t bytes)
{
    size_t b; void *p = img_ptr(c, which, &b);
    if (!p || bytes < b) return -1;
    memcpy(p, in, b); return 0;
}

/* TEST HOOK.  The fragment shaders' `texcoord` is a varying the rasteriser interpolates over the full-screen quad; how it rounds
   is the GL implementation's (llvmpipe: an ulp off the correctly rounded (p + 0.5) / n at 85 % of the pixels of a 640 x 480 target,
   softpipe: at other pixels).  The oracle takes it correctly rounded; with the coordinates a rasteriser actually produced handed in
   here (H x W x 2 floats, NULL to go back) P3 / P4 must reproduce that rasteriser's execution of the shaders at EVERY pixel
   (tests/test_ref_glsl.py, 640 x 480). */
int orc_set_fragment_texcoords(orc_ctx *c, const float *tc)
{
    free(c->frag_tc); c->frag_tc = NULL;
    if (!tc) return 0;
    c->frag_tc = (float *)malloc(sizeof(float) * 2 * (size_t)c->P);
    if (!c->frag_tc) return -1;
    memcpy(c->frag_tc, tc, sizeof(float) * 2 * (size_t)c->P);
    return 0;
}
/* the literal-rule helpers of include/hrbf_detmath.h, exported so that tests can hold each of them to an independent numpy emulation
   at every column / row of the sizes in use (tests/test_detmath.py) — oracle and kernels share them, so oracle == kernel cannot */
int orc_tap_texel(int c, int n) { return hd_tap_texel(c, n); }
int orc_window_samples(float t, int n, float win, int *texels /* >= 16 */)
{
    const hd_window w = hd_window_axis_t(t, n, win);
    int k = 0;
    for (float i = w.lo; i <= w.hi; i += w.step) { if (k < 16) texels[k] = hd_window_texel(i, n); ++k; }
    return k;
}
int orc_halfpixel_walk_samples(float x, int n, float wm, int *texels /* >= 16 */)
{
    const hd_walk w = hd_halfpixel_walk(x, n, wm);
    int k = 0;
    for (float i = w.lo; i < w.hi; i += w.step) { if (k < 16) texels[k] = hd_window_texel(i, n); ++k; }
    return k;
}
float orc_gl_point_window_coord(float u, float extent, int *clipped) { return hd_gl_point_window_coord(u, extent, clipped); }
float orc_uv_attribute(int p, int n) { return hd_uv_attribute(p, n); }
float orc_uv_fragment(int p, int n) { return hd_uv_fragment(p, n); }
float orc_expf(float x) { return hd_expf(x); }
int orc_f2i(float x) { return hd_cvt_i32(x); }
unsigned orc_f2u(float x) { return hd_cvt_u32(x); }
long long orc_d2l(double x) { return hd_cvt_i64(x); }
float orc_encode_color(float r, float g, float b) { return encode_color(v3(r, g, b)); }
float orc_acosf(float x) { return hd_acosf(x); }
float orc_atan2f(float y, float x) { return hd_atan2f(y, x); }
void orc_sincosf(float x, float *s, float *c) { hd_sincosf(x, s, c); }
void orc_sincos(double x, double *s, double *c) { hd_sincos(x, s, c); }
double orc_acos(double x) { return hd_acos(x); }
void orc_acc_test(const float *v, int n, double *out)
{
    hd_acc128 a; hd_acc_zero(&a);
    for (int i = 0; i < n; ++i) hd_acc_add_f32(&a, v[i]);
    *out = hd_acc_to_double(a);
}

/* color.glsl:19-34 round trip over all 2^24 colours; returns the number of failures */
int orc_color_roundtrip_failures(void)
{
    int bad = 0;
    for (int c = 0; c < (1 << 24); ++c) {
        f3 d = decode_color((float)c);
        if (encode_color(d) != (float)c) bad++;
        if (encode_color_bytes((c >> 16) & 255, (c >> 8) & 255, c & 255) != (float)c) bad++;
    }
    return bad;
}

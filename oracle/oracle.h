/*
 * oracle.h — CPU restatement ("oracle") of the HRBF-Fusion per-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under hrbffusion3d_amd/ or include/ may call into this
 * library; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PINNING.  The reference has no tests and no golden vectors, and its C++/CUDA cannot be built here (Pangolin, CUDA, Eigen,
 * OpenCV absent).  Its GLSL shaders can be EXECUTED: oracle/ref_glsl runs the shader files of
 * /root/reference/Core/src/Shaders, verbatim but for three respelled tokens, on Mesa llvmpipe with the GL calls of the
 * reference's host code, and tests/golden/ref_glsl holds what they wrote.  tests/test_ref_glsl.py compares this oracle (and
 * the HIP path) with those outputs pass by pass on identical inputs:
 *   GLSL rows (P1-P5, M1, F1-F4, H1-H3, f-3 of SURVEY §8a): PINNED to the executed reference — discrete results exact
 *     (validity, z-test winners, merge / create / remove decisions, map order), floats within the stated ulp bounds
 *     (the shaders' exp / acos / division are llvmpipe's, not hrbf_detmath.h's).
 *   CUDA rows (O1-O6: pyramids, so3Step, residuals, icpStep, rgbStep, the Gauss-Newton loop): PARITY UNPINNED by execution
 *     (no nvcc; a hipify build would be a stand-in) — pinned only by analytic known-answer tests (tests/test_oracle_*.py),
 *     independent fp64 numpy evaluations, the intrinsics KATs (tests/kat_projection.py, tests/test_intrinsics_kat.py) and 84
 *     metamorphic / analytic tests whose answers come from the mathematics the reference's code states
 *     (tests/test_registration_metamorphic*.py), themselves tested against 58 deliberately misread builds of this oracle
 *     (#if ORC_MUTANT in orc_odo.c / orc_ctx.c; tools/mutation_report.py: 58 of 58 caught).
 *   tools/compare_reference_dump.py diffs against dump files the reference itself writes, should a reference run exist.
 *
 * Plain C99, scalar, single-thread (OpenMP over pixels/surfels when built with -fopenmp; results
 * are identical because every reduction goes through the exact accumulator of hrbf_detmath.h).
 */
#ifndef HRBF_ORACLE_H_
#define HRBF_ORACLE_H_

#include <stdint.h>
#include <stddef.h>
#include "../include/hrbf_mi355.h"   /* hrbf_params, hrbf_image, hrbf_stage enums (types only) */
#include "../include/hrbf_detmath.h"

typedef struct { float x, y, z, w; } f4;
typedef struct { float x, y, z; } f3;

#define ORC_NUM_PYRS 3

typedef struct orc_planar {   /* DeviceArray2D<float>(rows*4, cols), planar SoA (RGBDOdometry.cpp:128-136) */
    int rows, cols;
    float *p;                 /* 4*rows*cols */
} orc_planar;

typedef struct orc_ctx {
    hrbf_params prm;
    int W, H, P;
    int tick;
    float pose[16];           /* column-major T_wc */
    float prev_pose[16];      /* pose at the end of the previous frame (lastPose, HRBFFusion.cpp:1064) */
    float weighting;
    float last_icp_error, last_icp_count;
    int index_submap;
    uint8_t *submap_active;   /* NULL = every submap active; else n_submap_active flags (IndexMap.cpp:222-237) */
    int n_submap_active;
    /* inputs */
    uint8_t *rgb;             /* P*3 */
    uint16_t *depth_raw;      /* P */
    /* preprocessing images */
    float *depth_filtered, *depth_metric, *depth_metric_filtered;
    f4 *vertex_raw, *vertex_filtered, *normal, *normal_pca, *normal_opt, *curv1, *curv2;
    float *frag_tc;   /* test hook (orc_set_fragment_texcoords): the rasteriser's interpolated texcoord per pixel, or NULL = correctly rounded */
    float *radius, *gradmag, *confidence;
    /* index map */
    uint32_t *idx;
    f4 *im_vertconf, *im_colortime, *im_normrad, *im_curvmax, *im_curvmin;
    /* prediction */
    uint8_t *pr_image;        /* P*4 */
    f4 *pr_vertex, *pr_normal, *pr_curv1, *pr_curv2;
    uint32_t *pr_time;
    float *pr_icpw;
    /* fill-in */
    uint8_t *fi_image;
    f4 *fi_vertex, *fi_normal, *fi_curv1, *fi_curv2;
    float *fi_icpw;
    /* surfel map: AoS 5 x f4, two ping-pong buffers */
    f4 *map[2];
    int target;
    uint32_t count, cap;
    /* fuse records: one per quarter-grid pixel, column-major order */
    f4 *rec;                  /* Q*5 */
    int32_t *rec_flag;        /* 0 none, 1 merge, 2 new */
    uint32_t *rec_best;
    int Q;
    uint32_t fuse_stats[4];   /* in, merged, appended, out */
    /* odometry state (RGBDOdometry members) */
    orc_planar vmap_g[ORC_NUM_PYRS], nmap_g[ORC_NUM_PYRS], ck1_g[ORC_NUM_PYRS], ck2_g[ORC_NUM_PYRS];
    orc_planar vmap_c[ORC_NUM_PYRS], nmap_c[ORC_NUM_PYRS], ck1_c[ORC_NUM_PYRS], ck2_c[ORC_NUM_PYRS];
    float *icpw[ORC_NUM_PYRS];
    f3 *sp_lambda[ORC_NUM_PYRS], *sp_z[ORC_NUM_PYRS];   /* sparse ICP: lambdaMap, z_thrinkMap, corresICP */
    int32_t *sp_corres[ORC_NUM_PYRS];
    int64_t sp_shrunk;        /* pixels x iterations whose shrink factor was non-zero, since creation */
    float *last_depth[ORC_NUM_PYRS], *next_depth[ORC_NUM_PYRS];
    uint8_t *last_image[ORC_NUM_PYRS], *next_image[ORC_NUM_PYRS], *last_next_image[ORC_NUM_PYRS];
    int16_t *dIdx[ORC_NUM_PYRS], *dIdy[ORC_NUM_PYRS];
    f3 *cloud[ORC_NUM_PYRS];
    int16_t *corres;          /* P * 6 : zero.x zero.y one.x one.y valid pad ; diff in corres_diff */
    float *corres_diff;
    double timings_ms[8];
    /* test hook (tests/test_registration_fp64.py): per Gauss-Newton iteration of the last orc_odo_track {level, iteration, A_icp[36],
       b_icp[6], A_rgb[36], b_rgb[6], increment[6], icp inliers, rgb count, rgb sigma, icp residual} = 96 doubles, then the state the
       iteration STARTED from: resultRt[16] (row-major), Rcurr[9], tcurr[3]; SO3 iterations {-1, iteration, jtj[9], jtr[3], residual,
       count} and resultR[9] at [96..105) */
    double odo_trace[40][128];
    int odo_trace_n;
} orc_ctx;

#ifdef __cplusplus
extern "C" {
#endif

orc_ctx *orc_create(const hrbf_params *p);
void orc_destroy(orc_ctx *c);
int orc_process_frame(orc_ctx *c, const uint8_t *rgb, const uint16_t *depth, int64_t ts, float wmul);
int orc_upload_frame(orc_ctx *c, const uint8_t *rgb, const uint16_t *depth);
int orc_run_stage(orc_ctx *c, int stage);
int orc_bootstrap(orc_ctx *c, const uint8_t *rgb, const uint16_t *depth);
void orc_get_pose(orc_ctx *c, float out16[16]);
void orc_set_pose(orc_ctx *c, const float in16[16]);
int orc_get_tick(orc_ctx *c);
void orc_set_tick(orc_ctx *c, int t);
void orc_set_index_submap(orc_ctx *c, int idx);
void orc_set_switch(orc_ctx *c, int which, float v);   /* the boundary's run-time setters, see orc_ctx.c */
void orc_set_active_submaps(orc_ctx *c, const uint8_t *active, int n);   /* n = 0: all active */
/* GlobalModel::updateModel (GlobalModel.cpp:690-767 -> update_delta_trans.vert:41-104) */
void orc_update_model(orc_ctx *c, const float *delta16_colmajor, int n);
void orc_set_weighting(orc_ctx *c, float w);
int orc_set_fragment_texcoords(orc_ctx *c, const float *tc);   /* test hook, see orc_ctx.c */
int orc_get_odo_trace(const orc_ctx *c, double *out, int max_rows);   /* test hook: rows of 128 doubles, returns the count */
/* test hook: a level of the registration pyramids as orc_odo_init_* left them.  which: 0-3 model v / n / k1 / k2 (global frame), 4-7 live
   v / n / k1 / k2 (planar 4 x rows x cols floats), 8 icp weight, 9 last depth, 10 next depth (floats), 11 last / 12 next / 13 previous
   intensity (bytes), 14 dIdx, 15 dIdy (int16, valid after a track).  Returns the byte count, 0 if `bytes` is too small. */
size_t orc_get_pyramid(const orc_ctx *c, int which, int level, void *out, size_t bytes);
float orc_get_weighting(orc_ctx *c);
uint32_t orc_surfel_count(orc_ctx *c);
int orc_download_map(orc_ctx *c, float *out, size_t cap);
int orc_upload_map(orc_ctx *c, const float *in, size_t n);
size_t orc_image_bytes(orc_ctx *c, int which);
int orc_get_image(orc_ctx *c, int which, void *out, size_t bytes);
int orc_set_image(orc_ctx *c, int which, const void *in, size_t bytes);
void orc_last_icp(orc_ctx *c, float *err, float *cnt);
void orc_get_fuse_stats(orc_ctx *c, uint32_t out[4]);
void orc_get_timings(orc_ctx *c, double out[8]);

/* standalone operators */
int orc_icp_step(const float Rcurr[9], const float tcurr[3],
                 const float *vmap_curr, const float *nmap_curr, const float *ck1_curr, const float *ck2_curr,
                 const float Rprev_inv[9], const float tprev[3], float fx, float fy, float cx, float cy,
                 const float *vmap_g_prev, const float *nmap_g_prev, const float *ck1_g_prev,
                 const float *ck2_g_prev, const float *icp_weight_prev, int rows, int cols,
                 float dist_thresh, float angle_thresh, int use_weight,
                 double A_out[36], double b_out[6], double residual_out[2]);
int orc_so3_step(const uint8_t *lastImage, const uint8_t *nextImage, int rows, int cols, const float basis[9],
                 const float kinv[9], const float krlr[9], double A_out[9], double b_out[3], double residual_out[2]);
int orc_rgb_residual(float minScale, const int16_t *dIdx, const int16_t *dIdy, const float *lastDepth,
                     const float *nextDepth, const uint8_t *lastImage, const uint8_t *nextImage, int rows, int cols,
                     const float kt[3], const float krkinv[9], int16_t *corres_out, float *diff_out, long long *count,
                     long long *sigma);
int orc_rgb_step(const int16_t *corres, const float *corres_diff, float sigma, const float *cloud, float fx, float fy,
                 const int16_t *dIdx, const int16_t *dIdy, int use_grad_weight, int rows, int cols, double A_out[36],
                 double b_out[6], double residual_out[2]);
/* HRBF primitives for known-answer tests (hrbfbase.glsl:126-195) */
float orc_hrbf_value(const float p[3], const f4 *vc, const f4 *nr, int n, int *nsupport);
void orc_hrbf_gradient(const float p[3], const f4 *vc, const f4 *nr, int n, float out[3]);
void orc_hrbf_hessian(const float p[3], const f4 *vc, const f4 *nr, int n, float out[9]);
/* detmath exports for tests */
int orc_tap_texel(int c, int n); int orc_window_samples(float t, int n, float win, int *texels);
int orc_halfpixel_walk_samples(float x, int n, float wm, int *texels); float orc_gl_point_window_coord(float u, float extent, int *clipped);
float orc_expf(float x); float orc_acosf(float x); float orc_atan2f(float y, float x);
int orc_f2i(float x); unsigned orc_f2u(float x); long long orc_d2l(double x); float orc_encode_color(float r, float g, float b);   /* hd_cvt_i32 and the colour word built on it */
void orc_sincosf(float x, float *s, float *c); void orc_sincos(double x, double *s, double *c);
double orc_acos(double x);
void orc_acc_test(const float *v, int n, double *out);
/* 6x6 solve and SE3 update used by the GN loop (RGBDOdometry.cpp:1162-1204) */
void orc_solve6(const double A[36], const double b[6], double x[6]);
float orc_sparse_shrink_factor(float hnorm);
int orc_icp_step_sparse(const float Rcurr[9], const float tcurr[3], const float *vmap_curr, const float *nmap_curr,
                        const float *ck1_curr, const float *ck2_curr, const float Rprev_inv[9], const float tprev[3],
                        float fx, float fy, float cx, float cy, const float *vmap_g_prev, const float *nmap_g_prev,
                        const float *ck1_g_prev, const float *ck2_g_prev, const float *icp_weight_prev, int rows,
                        int cols, float dist_thresh, float angle_thresh, int use_weight, const float *lambda_map,
                        float *z_map_out, int32_t *corres_out, double A_out[36], double b_out[6], double residual_out[2]);
/* the windowed correspondence search (reduce.cu:357-430) with its choice per pixel returned (tests only) */
int orc_icp_step_search(const float Rcurr[9], const float tcurr[3], const float *vmap_curr, const float *nmap_curr,
                        const float *ck1_curr, const float *ck2_curr, const float Rprev_inv[9], const float tprev[3],
                        float fx, float fy, float cx, float cy, const float *vmap_g_prev, const float *nmap_g_prev,
                        const float *ck1_g_prev, const float *ck2_g_prev, const float *icp_weight_prev, int rows,
                        int cols, float dist_thresh, float angle_thresh, int use_weight, int use_search, int radius,
                        int32_t *corres_out, double A_out[36], double b_out[6], double residual_out[2]);
int orc_update_lambda_map(const float Rcurr[9], const float tcurr[3], const float *vmap_curr, const float Rprev_inv[9],
                          const float tprev[3], const float *vmap_g_prev, const int32_t *corres, const float *z_map,
                          float *lambda_map, int rows, int cols);
int64_t orc_sparse_shrunk_count(const orc_ctx *c);

#ifdef __cplusplus
}
#endif
#endif

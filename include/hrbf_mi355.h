/*
 * hrbf_mi355.h — C-ABI of libhrbf_mi355.so, the MI355X-native per-frame hot path of
 * HRBF-Fusion (reference: YabinXuTUD/HRBFFusion3D, cited as path:line under /root/reference).
 *
 * Boundary replaced: `HRBFFusion::processFrame` (Core/src/HRBFFusion.h:110-113,
 * Core/src/HRBFFusion.cpp:991-1241), the `GlobalModel` surfel map
 * (Core/src/GlobalModel.h:44-175) and the operator seams underneath it
 * (Core/src/Cuda/cudafuncs.cuh:64-241, Core/src/IndexMap.h:43-68).
 *
 * Conventions
 *  - every function returns HRBF_OK (0) or a negative hrbf_status; nothing calls exit()
 *    (the reference exit(0)s on CUDA errors, Core/src/Cuda/convenience.cuh:64-71).
 *  - images are row-major, pixel (x,y) at index y*width+x; "f4" images are 4 floats/pixel
 *    (the reference's RGBA32F textures); poses are 4x4 float, COLUMN-major (Eigen default),
 *    camera-to-world, like `HRBFFusion::getCurrPose()` (Core/src/HRBFFusion.h).
 *  - a surfel is 20 floats, the reference's interleaved layout (Core/src/Shaders/Vertex.cpp:21-44):
 *      [x y z conf] [rgb24-as-float submapIdx initTime lastTime] [nx ny nz radius]
 *      [k1dir.xyz k1] [k2dir.xyz k2]
 *    Internally the map is five float4 planes (SoA) in HBM; the AoS form exists only at
 *    hrbf_download_map / hrbf_upload_map.
 *  - one context = one GPU = one thread at a time (the reference is single-threaded and
 *    non-reentrant on this path).
 */
#ifndef HRBF_MI355_H_
#define HRBF_MI355_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum hrbf_status {
    HRBF_OK = 0,
    HRBF_ERR_INVALID = -1,   /* bad argument / unsupported parameter combination */
    HRBF_ERR_DEVICE = -2,    /* HIP runtime error, see hrbf_last_error() */
    HRBF_ERR_CAPACITY = -3,  /* surfel map capacity exceeded */
    HRBF_ERR_NODEVICE = -4,  /* no gfx950 device visible */
    HRBF_ERR_COMM = -5       /* RCCL error */
} hrbf_status;

/* Parameters = HRBFFusion ctor arguments (Core/src/HRBFFusion.h:87-94) + the GlobalStateParam
 * fields the path reads (Core/src/Utils/GlobalStateParams.h:12-63, defaults GUI/GlobalStateParam.txt). */
typedef struct hrbf_params {
    int32_t width, height;          /* Resolution singleton */
    float fx, fy, cx, cy;           /* Intrinsics singleton */
    float depth_scale;              /* metres per raw depth unit = 1/DepthMapFactor (HRBFFusion.cpp:772-780) */
    /* ctor */
    float confidence_threshold;     /* `confidence`, GUI passes 5.0 (GUI/src/HRBF_fusion.cpp:87-96) */
    float depth_cutoff;             /* `depthCut`, GUI passes globalDepthCutoff = 3.5 */
    float icp_weight;               /* `icpThresh` = 10 */
    int32_t fast_odom;              /* false */
    int32_t so3;                    /* true  */
    int32_t frame_to_frame_rgb;     /* false */
    int32_t rgb_only;               /* false (HRBFFusion.cpp:38) */
    int32_t pyramid;                /* true  (HRBFFusion.cpp:40) */
    float max_depth_processed;      /* 20.0  (HRBFFusion.cpp:37) */
    /* preprocessing */
    int32_t use_bilateral;          /* preprocessingUsebilateralFilter = true */
    float init_radius_multiplier;   /* 4.0 */
    float curv_estimation_window;   /* 3.0 */
    float curv_valid_threshold;     /* 300 */
    float normal_estimation_pca;    /* 1.0 */
    int32_t use_conf_eval;          /* 0 */
    float conf_eval_epsilon;        /* 1000 */
    /* registration */
    int32_t icp_use_corr_search;    /* false */
    int32_t icp_search_radius;      /* 2 */
    int32_t icp_use_weighted;       /* true */
    float icp_curv_weight_lambda;   /* registrationICPCurvWeightImpactControl = 10 */
    int32_t rgb_use_grad_weight;    /* registrationColorUseRGBGrad = false */
    int32_t use_sparse_icp;         /* registrationICPUseSparseICP = false; 1 = ADMM variant (reduce.cu:302-315,479-492) */
    /* prediction */
    float predict_window_multiplier;/* 3.0 */
    int32_t predict_min_neighbors;  /* 6 */
    int32_t predict_max_neighbors;  /* 10 */
    float predict_conf_threshold;   /* 3.0 */
    /* fusion */
    float clean_window_multiplier;  /* fusionCleanWindowMultiplier = 2.0 */
    float dense_enough_thresh;      /* globalDenseEnoughThresh = 0.75 */
    /* build-specific */
    int32_t max_surfels;            /* map capacity; reference: 4596^2 (GlobalModel.cpp:21-22) */
    int32_t load_trajectory;        /* globalInputLoadTrajectory: poses supplied via hrbf_set_pose before each frame */
} hrbf_params;

typedef struct hrbf_context *hrbf_handle;

void hrbf_default_params(hrbf_params *p, int width, int height, float fx, float fy, float cx, float cy,
                         float depth_scale);

/* lifecycle -------------------------------------------------------------------------------- */
int hrbf_create(const hrbf_params *p, int device, hrbf_handle *out);
void hrbf_destroy(hrbf_handle h);
const char *hrbf_last_error(void);
const char *hrbf_version(void);

/* primary entry: HRBFFusion::processFrame (Core/src/HRBFFusion.h:110-113).
 * rgb: W*H*3 uint8 (R,G,B), depth: W*H uint16 raw units; host pointers borrowed for the call (copied into pinned
 * staging, uploaded asynchronously — the call enqueues the frame and returns). */
int hrbf_process_frame(hrbf_handle h, const uint8_t *rgb, const uint16_t *depth, int64_t timestamp,
                       float weight_multiplier);
/* same, inputs already resident in HBM (device pointers) — what bench.py times.  The call only enqueues and the
 * kernels read d_rgb / d_depth in place (no staging copy): keep both buffers valid and unchanged until
 * hrbf_synchronize() or any blocking getter returns.  The context does not keep them: a stage seam that reads the raw frame
 * (hrbf_run_stage FILTER_DEPTH / METRICISE / INITIALISE / FUSE / FILLIN) refuses to run after this entry until hrbf_upload_frame(). */
int hrbf_process_frame_device(hrbf_handle h, const void *d_rgb, const void *d_depth, int64_t timestamp,
                              float weight_multiplier);
/* block until all work queued on the context's stream is complete */
int hrbf_synchronize(hrbf_handle h);

/* getters the reference's caller uses (GUI/src/HRBF_fusion.cpp:235-497) ---------------------- */
int hrbf_get_pose(hrbf_handle h, float out16[16]);          /* getCurrPose(), column-major T_wc */
int hrbf_set_pose(hrbf_handle h, const float in16[16]);     /* trajectory replay (HRBFFusion.cpp:1105-1108) */
/* The trajectory WITHOUT blocking (trajectory_manager->poses, Core/src/HRBFFusion.h:383-384): every processed frame's
 * pose is appended by the device to a pinned host ring of 65536 entries.  hrbf_frames_enqueued = process_frame calls so
 * far; hrbf_frames_completed = frames whose pose has landed (never blocks); hrbf_get_pose_log copies the poses of frames
 * [first, first + count) that have landed (column-major 4x4 each) and returns how many — wait != 0 drains the stream
 * first.  hrbf_get_pose remains the blocking getCurrPose(). */
uint32_t hrbf_frames_enqueued(hrbf_handle h);
uint32_t hrbf_frames_completed(hrbf_handle h);
int hrbf_get_pose_log(hrbf_handle h, uint32_t first_frame, uint32_t count, float *out16_each, int wait);
int hrbf_get_tick(hrbf_handle h);                           /* getTick() */
uint32_t hrbf_surfel_count(hrbf_handle h);                  /* getGlobalModel().lastCount() */
int hrbf_download_map(hrbf_handle h, float *out, size_t cap_surfels); /* GlobalModel::downloadMap (GlobalModel.cpp:775-804) */
int hrbf_upload_map(hrbf_handle h, const float *in, size_t n_surfels); /* build-specific: seed the map (bench inflation) */
int hrbf_last_icp(hrbf_handle h, float *error, float *count); /* getFrameToModel().lastICPError/lastICPCount */
int hrbf_last_weighting(hrbf_handle h, float *w);
/* live-tunable setters applied every GUI frame (GUI/src/HRBF_fusion.cpp:448-456) */
int hrbf_set_rgb_only(hrbf_handle h, int v);
int hrbf_set_icp_weight(hrbf_handle h, float v);
int hrbf_set_pyramid(hrbf_handle h, int v);
int hrbf_set_fast_odom(hrbf_handle h, int v);
int hrbf_set_so3(hrbf_handle h, int v);
int hrbf_set_frame_to_frame_rgb(hrbf_handle h, int v);
int hrbf_set_confidence_threshold(hrbf_handle h, float v);
int hrbf_set_depth_cutoff(hrbf_handle h, float v);

/* named images = the reference's GPUTexture map (Core/src/GPUTexture.cpp:21-38) + IndexMap /
 * FillIn texture getters (Core/src/IndexMap.h, Core/src/Shaders/FillIn.h) */
typedef enum hrbf_image {
    HRBF_IMG_DEPTH_FILTERED = 0,      /* f1, raw units */
    HRBF_IMG_DEPTH_METRIC,            /* f1 */
    HRBF_IMG_DEPTH_METRIC_FILTERED,   /* f1 */
    HRBF_IMG_VERTEX_RAW,              /* f4 */
    HRBF_IMG_VERTEX_FILTERED,         /* f4 */
    HRBF_IMG_NORMAL,                  /* f4 (after updateNormalRad: NORMAL_OPT) */
    HRBF_IMG_NORMAL_PCA,              /* f4 build-specific: the un-invalidated normal + radius of depth_vertex_normal_radius.frag; the fuse takes it
                                         for a new point's normal where data.vert's own recomputation (data.vert:83-96) has the same inputs, and
                                         recomputes elsewhere (sizes that are no power of two; central differences) */
    HRBF_IMG_RADIUS,                  /* f1 */
    HRBF_IMG_CURV1,                   /* f4 */
    HRBF_IMG_CURV2,                   /* f4 */
    HRBF_IMG_GRADIENT_MAG,            /* f1 */
    HRBF_IMG_CONFIDENCE,              /* f1 */
    HRBF_IMG_INDEX,                   /* u32 */
    HRBF_IMG_INDEX_VERTCONF,          /* f4 */
    HRBF_IMG_INDEX_COLORTIME,         /* f4 */
    HRBF_IMG_INDEX_NORMRAD,           /* f4 */
    HRBF_IMG_INDEX_CURVMAX,           /* f4 */
    HRBF_IMG_INDEX_CURVMIN,           /* f4 */
    HRBF_IMG_PRED_IMAGE,              /* u8x4 */
    HRBF_IMG_PRED_VERTEX,             /* f4 */
    HRBF_IMG_PRED_NORMAL,             /* f4 */
    HRBF_IMG_PRED_CURV1,              /* f4 */
    HRBF_IMG_PRED_CURV2,              /* f4 */
    HRBF_IMG_PRED_TIME,               /* u32 (reference: R16UI) */
    HRBF_IMG_PRED_ICPWEIGHT,          /* f1 */
    HRBF_IMG_FILL_IMAGE,              /* u8x4 */
    HRBF_IMG_FILL_VERTEX,             /* f4 */
    HRBF_IMG_FILL_NORMAL,             /* f4 */
    HRBF_IMG_FILL_CURV1,              /* f4 */
    HRBF_IMG_FILL_CURV2,              /* f4 */
    HRBF_IMG_FILL_ICPWEIGHT,          /* f1 */
    HRBF_IMG_FIT_CURV1,               /* f4 extension (hrbf_fit_curvature): direction of kmax, kmax; w = 1000 where no fit */
    HRBF_IMG_FIT_CURV2,               /* f4 extension: direction of kmin, kmin */
    HRBF_IMG_FIT_NORMAL,              /* f4 extension: normalised gradient of the fitted interpolant, its length */
    HRBF_IMG_COUNT
} hrbf_image;
size_t hrbf_image_bytes(hrbf_handle h, int which);
int hrbf_get_image(hrbf_handle h, int which, void *out, size_t bytes);   /* device -> host */
int hrbf_set_image(hrbf_handle h, int which, const void *in, size_t bytes); /* host -> device (tests) */

/* so3Step / computeRgbResidual / rgbStep seams (Core/src/Cuda/cudafuncs.cuh:118-162, reduce.cu:697-896,957-1359) on
 * caller-provided DEVICE images (row-major rows x cols); matrices are host arrays, row-major 3x3.
 * hrbf_so3_step: A_out 9 doubles row-major, b_out 3, residual_out {sum r^2, count}.
 * hrbf_rgb_residual: corres_out = 6 int16 per pixel {u0, v0, x, y, valid, 0} (the reference's DataTerm zero/one/valid),
 *   diff_out = intensity difference; *count / *sigma = the reference's count and sigmaSum (maxDepthDelta = 0.07).
 * hrbf_rgb_step: cloud = 3 floats per pixel (projectToPointCloud), sobel scale 0.125; A_out 36 doubles row-major, b_out 6,
 *   residual_out {sum (w r)^2, count}. */
int hrbf_so3_step(hrbf_handle h, const uint8_t *last_image, const uint8_t *next_image, int rows, int cols,
                  const float image_basis[9], const float kinv[9], const float krlr[9], double A_out[9], double b_out[3],
                  double residual_out[2]);
int hrbf_rgb_residual(hrbf_handle h, float min_scale, const int16_t *dIdx, const int16_t *dIdy, const float *last_depth,
                      const float *next_depth, const uint8_t *last_image, const uint8_t *next_image, int rows, int cols,
                      const float kt[3], const float krkinv[9], int16_t *corres_out, float *diff_out, long long *count,
                      long long *sigma);
int hrbf_rgb_step(hrbf_handle h, const int16_t *corres, const float *corres_diff, float sigma, const float *cloud,
                  float fx, float fy, const int16_t *dIdx, const int16_t *dIdy, int use_grad_weight, int rows, int cols,
                  double A_out[36], double b_out[6], double residual_out[2]);

/* submap bookkeeping and rigid map correction — the callers either side of the path.
 * hrbf_set_index_submap: HRBFFusion::indexSubmap, the id stamped on surfels created from now on (data.vert).
 * hrbf_set_active_submaps: IndexMap::lActiveKFID as a byte mask (IndexMap.cpp:222-237); surfels of inactive submaps
 *   are not drawn into the index map (index_map.vert:41-45) and cannot cause free-space removals
 *   (copy_unstable.vert:98-134).  n = 0 restores "all active".
 * hrbf_update_model: GlobalModel::updateModel (GlobalModel.cpp:690-767, update_delta_trans.vert): n column-major 4x4
 *   corrections, surfel <- delta[submap(surfel)] applied to position and normal; ids >= n are left unchanged. */
int hrbf_set_index_submap(hrbf_handle h, int index);
int hrbf_set_active_submaps(hrbf_handle h, const uint8_t *active, int n);
int hrbf_update_model(hrbf_handle h, const float *delta16_colmajor, int n);

/* per-region timings of the last frame in ms, named as the reference's Stopwatch regions
 * (Core/src/HRBFFusion.cpp:1016,1063,1196,1248): 0 Initialization 1 Registration 2 Integration 3 Prediction;
 * 4 = the fuse (clean+compact+append) streaming kernel alone.  Requires hrbf_enable_timing(h,1);
 * hrbf_enable_timing(h,2) records only the two events per frame that feed hrbf_get_fuse_ring. */
int hrbf_enable_timing(hrbf_handle h, int on);
int hrbf_get_timings(hrbf_handle h, float out_ms[8]);
/* ring of the last <= 1024 frames (timing enabled): duration in ms of the fuse pass — F2 + F3 of SURVEY.md §8d:
 * k_apply_merges (update.vert) + k_clean_flags + k_fuse_stream (copy_unstable.*), HIP events on the context's stream
 * bracketing exactly those launches — and its {in, merged, appended, out} surfel counts; returns the number of
 * frames written, oldest first.  hrbf_get_fuse_ring_parts gives the two parts separately and 8 statistics words:
 * {in, merged, appended, out, -, -, moved (surfels that changed slot in the in-place compaction), status}. */
int hrbf_get_fuse_ring(hrbf_handle h, int max_frames, float *kernel_ms, uint32_t *stats4);
/* record the ring only on frames whose time stamp is a multiple of every_nth_frame (default 1: every frame).  The four
 * event records and the statistics copy cost the stream about 22 us per recorded frame. */
int hrbf_set_fuse_ring_stride(hrbf_handle h, int every_nth_frame);

int hrbf_get_fuse_ring_parts(hrbf_handle h, int max_frames, float *merge_ms, float *stream_ms, uint32_t *stats8);
int hrbf_reset_fuse_ring(hrbf_handle h);
/* measurement probe (not part of the path): the pixel work of `iters` Gauss-Newton iterations of pyramid `level` (ICP
 * products + RGB products, exact reductions) executed by ONE 256-thread workgroup; kernel time in ms.  Call after at
 * least two processed frames.  DESIGN.md §6 compares it with the three launches an iteration takes. */
int hrbf_probe_single_workgroup_iteration(hrbf_handle h, int level, int iters, float *ms_out);
/* test probe (not part of the path): exhaustive check, on the device, of the square-root shortcut k_predict_hrbf uses
 * (v_sqrt_f32 + two residual tests).  out = {raw == root, raw one ulp low, raw one ulp high, raw further off,
 * shortcut != correctly rounded root for x >= 2^-96, (1 - shortcut) != (1 - root) for x < 2^-96} over all
 * non-negative finite floats; the last three must be 0. */
int hrbf_probe_sqrt_rounding(hrbf_handle h, uint64_t out[6]);
/* test probe: the bilateral filter's exp scales its polynomial by 2^k with one v_ldexp_f32 where hd_expf (and the oracle)
 * multiply by 2^(k/2) and 2^(k - k/2).  out = {mismatches, cases} over every p in [0.5, 2) and k in [-160, 0]. */
int hrbf_probe_exp_scaling(hrbf_handle h, uint64_t out[2]);
/* test probe: k_curvature's division on tame tiles (reciprocal refined once, quotient twice, no v_div_scale / v_div_fixup)
 * against the compiler's correctly rounded division on 2^32 pseudo-random operand pairs spread over the tame ranges.
 * out = {mismatches, cases}. */
int hrbf_probe_division(hrbf_handle h, uint64_t out[2]);
/* build-specific: toggle trajectory replay (globalInputLoadTrajectory) between frames */
int hrbf_set_load_trajectory(hrbf_handle h, int v);
/* sticky condition bits, folded from the device by this (synchronising) call; clear != 0 resets them.  The per-frame
 * path never blocks, so conditions the reference would exit() on are reported here instead of from hrbf_process_frame. */
#define HRBF_STATUS_CAPACITY 1u        /* the map reached max_surfels: new surfels were dropped (cf. HRBF_ERR_CAPACITY) */
#define HRBF_STATUS_INTERNAL_BOUND 2u  /* a fuse pass was launched with a stale host bound on the surfel count and refused to run */
#define HRBF_STATUS_SO3_TIMEOUT 4u     /* the SO3 pre-alignment kernel ran into its poll bound: that frame's pose is NaN */
#define HRBF_STATUS_FUSE_TIMEOUT 8u    /* the in-place compaction ran into its (bounded) tile wait: the map of that frame is not trustworthy */
#define HRBF_STATUS_ID_SPACE 16u       /* hash-owned map: the 32-bit order ids were about to run out and renumbering failed: that frame's clean
                                          pass did not run, hrbf_process_frame returned the error and keeps failing until this is cleared */
#define HRBF_STATUS_COLLECTIVE 32u     /* row-sharded registration: an all-reduce of the limb sums failed (either transport): that frame's
                                          pose was solved from unreduced sums and is not to be trusted */
#define HRBF_STATUS_EXTENSION 64u      /* a frame was processed with an option that has no reference counterpart switched on (hrbf_set_hrbf_fit):
                                          its results are not the reference's */
int hrbf_get_status(hrbf_handle h, uint32_t *flags, int clear);
int hrbf_shard_exchange_mode(hrbf_handle h);   /* see "How the ranks exchange the index map" below */
/* number of surfels that entered / merged / appended / survived in the last frame's fuse pass */
int hrbf_get_fuse_stats(hrbf_handle h, uint32_t out[4]);

/* operator-level seams on the context's own images (isolated parity) ------------ */
typedef enum hrbf_stage {
    HRBF_STAGE_FILTER_DEPTH = 0,      /* filterDepth        HRBFFusion.cpp:1272-1280 */
    HRBF_STAGE_METRICISE,             /* metriciseDepth     :1263-1270 */
    HRBF_STAGE_VERTEX_NORMAL_RADIUS,  /* computeVertexNormalRadius :1329-1345 */
    HRBF_STAGE_CURVATURE,             /* computeCurvatureGradient + updateNormalRad :1282-1310 */
    HRBF_STAGE_CONFIDENCE,            /* VertexConfidence   :1311-1327 (uses last weighting) */
    HRBF_STAGE_INITIALISE,            /* GlobalModel::initialise GlobalModel.cpp:214-288 */
    HRBF_STAGE_PREDICT_INDICES,       /* IndexMap::predictIndices IndexMap.cpp:193-267 */
    HRBF_STAGE_FUSE,                  /* GlobalModel::fuse  GlobalModel.cpp:355-548 */
    HRBF_STAGE_CLEAN,                 /* GlobalModel::clean GlobalModel.cpp:551-688 */
    HRBF_STAGE_PREDICT_HRBF,          /* IndexMap::predictHRBF IndexMap.cpp:413-518 */
    HRBF_STAGE_FILLIN,                /* FillIn::vertex/normal/curvature/image FillIn.cpp:93-297 */
    HRBF_STAGE_ODOMETRY,              /* initICP*.. + getIncrementalTransformation RGBDOdometry.cpp:183-247,660-1249 */
    HRBF_STAGE_COUNT
} hrbf_stage;
int hrbf_upload_frame(hrbf_handle h, const uint8_t *rgb, const uint16_t *depth);
/* build-specific: start tracking against an uploaded map (hrbf_upload_map + hrbf_set_pose): runs the
 * pre-processing of the given frame, initFirstRGB and predict() (HRBFFusion.cpp:1244-1260), then sets
 * tick = 2, i.e. the state the reference is in after its first processFrame. */
int hrbf_bootstrap(hrbf_handle h, const uint8_t *rgb, const uint16_t *depth);
int hrbf_run_stage(hrbf_handle h, int stage);
/* the map / prediction operators under the reference's names and with the arguments it passes explicitly
 * (GlobalModel.h:50-107 initialise / fuse / clean, IndexMap.h:43-68 predictIndices / predictHRBF).  They work on the
 * context's map and images — the reference's GPUTexture arguments, filled by the pre-processing stages or
 * hrbf_set_image.  pose16: column-major T_wc (NULL keeps the current pose); time <= 0, cut-offs <= 0 and
 * index_submap < 0 keep the context's values.  What is passed becomes the context's current state. */
int hrbf_initialise(hrbf_handle h, const float init_pose16[16]);
int hrbf_predict_indices(hrbf_handle h, const float pose16[16], int time, float depth_cutoff, int index_submap);
int hrbf_fuse(hrbf_handle h, const float pose16[16], int time, float depth_cutoff, int index_submap);
int hrbf_clean(hrbf_handle h, const float pose16[16], int time, float conf_threshold, float max_depth);
int hrbf_predict_hrbf(hrbf_handle h);
/* Resize::vertex(indexMap.vertexTexHRBF(), verticesBuff) + HRBFFusion::denseEnough (Shaders/Resize.cpp:106-134, resize.frag,
 * HRBFFusion.cpp:974-988): *dense = 1 when more than dense_enough_thresh of the (width / 20) x (height / 20) cell centres of the
 * predicted vertex map hold a depth.  processFrame takes this decision on the device (shouldFillIn = !dense, :1069-1070);
 * this is the reference's host-side predicate on the context's current PRED_VERTEX image. */
int hrbf_dense_enough(hrbf_handle h, int *dense);
int hrbf_set_tick(hrbf_handle h, int tick);
int hrbf_set_weighting(hrbf_handle h, float w);

/* icpStep seam (Core/src/Cuda/cudafuncs.cuh icpStep / reduce.cu:580-693) on caller-provided DEVICE
 * planar maps: each map is 4 planes of rows*cols floats (x,y,z,w), NaN in plane 0 = invalid.
 * A_out: 36 doubles row-major, b_out: 6 doubles, residual_out: {sum r^2 (weighted), inliers}. */
int hrbf_icp_step(hrbf_handle h, const float Rcurr[9], const float tcurr[3],
                  const float *vmap_curr, const float *nmap_curr, const float *ck1_curr, const float *ck2_curr,
                  const float Rprev_inv[9], const float tprev[3], float fx, float fy, float cx, float cy,
                  const float *vmap_g_prev, const float *nmap_g_prev, const float *ck1_g_prev,
                  const float *ck2_g_prev, const float *icp_weight_prev, int rows, int cols,
                  float dist_thresh, float angle_thresh, int use_weight,
                  double A_out[36], double b_out[6], double residual_out[2]);
/* the same with useSparse = true (reduce.cu:302-315,455-492) and updateLambdaMap (cudafuncs.cu:1030-1111): lambda_map
 * (lambdaMap, in) and z_map_out (z_thrinkMap) are DEVICE images of 3 interleaved floats per pixel, corres_out
 * (corresICP) 2 int32 per pixel, (-1, -1) = no match */
int hrbf_icp_step_sparse(hrbf_handle h, const float Rcurr[9], const float tcurr[3],
                         const float *vmap_curr, const float *nmap_curr, const float *ck1_curr, const float *ck2_curr,
                         const float Rprev_inv[9], const float tprev[3], float fx, float fy, float cx, float cy,
                         const float *vmap_g_prev, const float *nmap_g_prev, const float *ck1_g_prev,
                         const float *ck2_g_prev, const float *icp_weight_prev, int rows, int cols,
                         float dist_thresh, float angle_thresh, int use_weight,
                         const float *lambda_map, float *z_map_out, int32_t *corres_out,
                         double A_out[36], double b_out[6], double residual_out[2]);
int hrbf_update_lambda_map(hrbf_handle h, const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                           const float Rprev_inv[9], const float tprev[3], const float *vmap_g_prev,
                           const int32_t *corres, const float *z_map, float *lambda_map, int rows, int cols);

/* multi-GPU (sharding 1): one process per GPU joins an RCCL communicator (librccl is loaded on first use);
 * afterwards every rank still runs the whole frame on its own full map, but the registration reductions (SO3, RGB
 * residual, ICP and RGB normal equations) cover only the image rows [rank, rank+1) * rows / world of each pyramid
 * level and the exact int64 limb sums are all-reduced (ncclInt64, ncclSum) on the context's stream, 29 times per
 * frame (+ <= 10 for SO3) — every rank then takes the identical step, bit-identical to the single-GPU result.
 * At VGA this is a latency cost, not a speed-up (the reductions are launch-bound); it is the exchange step a
 * spatially sharded map would need and is exercised by tests, while bench.py scales by independent replicas.
 * hrbf_comm_init(h, -1, world, NULL) is a test hook without RCCL: the process plays `world` virtual ranks in turn
 * (world <= 1 returns to the single-GPU path). */
int hrbf_comm_unique_id(uint8_t out128[128]);
int hrbf_comm_init(hrbf_handle h, int rank, int world, const uint8_t id128[128]);

/* multi-GPU (sharding 2): the surfel map itself is cut over the ranks of the communicator above.  The
 * global surfel order — the order GlobalModel holds on one GPU (GlobalModel.cpp:551-688: stable compaction, new
 * surfels appended) — is split into `world` contiguous ranges, one per rank; ids in the index map stay global, so the
 * z-test ties, the association, the merges and the order of the map are the single-GPU ones bit for bit.  Per
 * projection (three per frame) the ranks exchange: ncclAllReduce(min) over the W*H packed u64 z-buffer keys, then
 * ncclAllReduce(sum, as uint32) over the resolved attribute images (one owner per pixel, zeros elsewhere: exact);
 * after the clean pass one ncclAllGather of the `world` live counts.  Association is replicated (it reads images
 * only); merges are applied by the owner; the streaming clean + compaction pass — the HBM-bound part — runs on every
 * rank over its own range only.  New surfels are appended at the end of the order = on the last rank;
 * hrbf_map_rebalance() re-cuts the ranges evenly (synchronous, ncclSend/ncclRecv of the pieces that change owner).
 *   hrbf_map_shard_init(h, 1)  requires hrbf_comm_init and an empty map; with the virtual communicator
 *                              (rank < 0) this one process plays all shards and local kernels stand in for the
 *                              collectives — the test mode, bit-identical to the single-GPU map.
 *   hrbf_map_shard_init(h, 2)  the same, but ownership by SPATIAL HASH of the surfel's cell ("sharded by
 *                              spatial hash of surfel position ... fixed at insertion"; HRBF_HASH_CELL = cell edge in
 *                              metres, default 0.25): a rank owns the surfels whose cell hashes to it, wherever they stand
 *                              in the global order, so the surfels in view — the costly ones — spread over the ranks
 *                              instead of sitting in one range.  Every surfel carries its place in the global order (an
 *                              id that is never renumbered: the seed's row, then next-free + record index), the z-test
 *                              runs in two levels ({depth, local index} per rank, then {depth, id} min-reduced), every rank
 *                              appends the new surfels of its own cells; images (up to the names in the index image),
 *                              pose, and the map merged by id are the single-GPU ones bit for bit.  No re-cut is ever
 *                              needed (hrbf_map_rebalance is a no-op).  Ids grow by W*H/4 per frame; before 32 bits run out
 *                              every id is replaced by its rank in the global order (peer-mapped id planes; all-gathered
 *                              planes under the packed-record exchange) — the order, hence every result, is unchanged.  Should
 *                              that fail (allocation, communicator) the frame's clean pass does not run, hrbf_process_frame
 *                              returns the error, HRBF_STATUS_ID_SPACE is set and later frames fail at once until the status is
 *                              cleared: ids never run past the limit.  hrbf_download_map of a rank returns its own
 *                              surfels, hrbf_download_gids their ids; one process playing all shards returns the merged map.
 *   hrbf_upload_map            always takes the WHOLE map; a rank keeps its slice.
 *   hrbf_surfel_count          global count; hrbf_local_surfel_count / hrbf_download_map: the local range(s).
 *   hrbf_rebalance_plan        the host-side arithmetic of the re-cut (pure function, no device needed):
 *                              moves5[i] = {src shard, dst shard, offset in src, offset in dst, length}, <= 2*G - 1.
 *
 * How the ranks exchange the index map: every rank maps the other ranks' index-map images (hipIpcMemHandle; the
 * ranks of one node) and the OWNER of a pixel's z-test winner writes the winner's attributes straight into every rank's
 * images — no packing, no count exchange, no host read-back; per projection the ranks meet twice on the stream (key
 * min-reduce, "everybody has written").  HRBF_SHARD_EXCHANGE=records selects the older packed-record ncclSend / ncclRecv.
 * Peer mapping needs all ranks on ONE node in one IPC namespace.  Whether it worked is decided COLLECTIVELY at
 * hrbf_map_shard_init: every rank reaches every meeting point whatever failed locally, a one-word all-reduce counts the ranks
 * that could not map, and if there is one ALL ranks drop their mappings and use the packed-record exchange (which has no such
 * restriction) — never some ranks on one transport and some on the other.  hrbf_shard_exchange_mode tells which: 0 not sharded
 * over ranks, 1 peer images, 2 records on request, 3 records by that agreement.
 *
 * hrbf_peer_unique_id / hrbf_comm_init_peer: the same sharded map WITHOUT RCCL — rendezvous, surfel counts and the meeting
 * points go through a POSIX shared-memory segment (the id is its name), the key min-reduce reads the peers' z-buffers.  It
 * exists because RCCL refuses two ranks on one GPU ("Duplicate GPU detected"): with it the whole sharded-map path runs as
 * two PROCESSES on ONE device, bit-identical to the single map (tests/test_peer_shards_gpu.py).  Registration is then not
 * row-sharded by default (every rank reduces the whole image); hrbf_set_row_sharding(h, 1) switches the strips on, with the
 * int64 limb sums all-reduced through the segment (a host round trip per reduction: a correctness path for one-device
 * boxes, not a fast one).  hrbf_map_rebalance is unavailable on this transport.
 * A failed id renumbering (HRBF_STATUS_ID_SPACE) is AGREED between the ranks — all commit the new ids or none — and is final
 * for a map shared by ranks (clearing the status re-arms a retry for a single process only). */
int hrbf_peer_unique_id(uint8_t out128[128]);
int hrbf_comm_init_peer(hrbf_handle h, int rank, int world, const uint8_t id128[128]);
/* a rendezvous id that will not be used after all (rank 0's context removes the segment when it goes away; if rank 0 never
 * joins, this does).  An id serves ONE rendezvous: a second set of contexts on it is refused. */
int hrbf_peer_release_id(const uint8_t id128[128]);
int hrbf_map_shard_init(hrbf_handle h, int enable);   /* 0 off | 1 contiguous ranges | 2 spatial hash */
/* EXTENSION with NO counterpart in the reference (BASELINE config 5's "batched-HRBF small-GEMM on MFMA"; the reference uses the closed
 * form 10 * n_i, hrbfbase.glsl:132): a true Hermite-RBF fit over every pixel's (2 * window + 1)^2 window (window 1 or 2) of the
 * context's VERTEX_FILTERED / NORMAL images — a (4k x 4k) symmetric positive definite Wendland-C4 system per pixel, factored by a
 * blocked Cholesky whose trailing updates run on v_mfma_f32_16x16x4_f32 — and the principal curvatures of the fitted level set at the
 * pixel (images HRBF_IMG_FIT_CURV1 / FIT_CURV2 / FIT_NORMAL).  support: common support radius in units of the window's largest
 * centre distance (1.25); ridge: added to the diagonal (1e-6); jump: centres farther than jump * window pixel footprints from the
 * pixel are left out (3).  ms (nullable): kernel time by HIP events.  Not called by hrbf_process_frame; never changes its results. */
int hrbf_fit_curvature(hrbf_handle h, int window, float support, float ridge, float jump, float *ms);
/* the same fit INSIDE the frame path, as an option beside the closed form (default 0 = the reference's closed form; with 1 the results
 * are no longer the reference's): hrbf_process_frame then takes the live frame's principal curvatures (HRBF_IMG_CURV1 / CURV2: ICP
 * weights, curvature validity, the records of the fusion) from the fitted interpolant, 5 x 5 window, ridge 0.1 (an exact interpolant
 * of noisy normals amplifies their noise), at ~7 ms per 640 x 480 frame */
int hrbf_set_hrbf_fit(hrbf_handle h, int enable);
/* the option's state and parameters, visible to the caller (round-5 advice: a context left in this mode fails parity with nothing
 * to show for it): every frame processed with it raises the sticky HRBF_STATUS_EXTENSION.  Defaults: window 2 (5 x 5), support
 * 1.25, ridge 0.1, jump 3.0 — the arguments of hrbf_fit_curvature */
int hrbf_set_hrbf_fit_params(hrbf_handle h, int window, float support, float ridge, float jump);
int hrbf_get_hrbf_fit(hrbf_handle h, int *enabled, int *window, float *support, float *ridge, float *jump);
int hrbf_gn_graph_captures(hrbf_handle h);    /* times the Gauss-Newton loop was captured into a hipGraph: 2 in a steady run (one per image parity) whatever
                                                weightMultiplier the caller passes per frame (GUI/src/HRBF_fusion.cpp:225); re-captured only when a setter changes the configuration */
int hrbf_hash_renumber_count(hrbf_handle h);   /* hash ownership: times the 32-bit ids were renumbered to ranks (every ~55 000 VGA frames; order unchanged) */
int hrbf_hash_owner(float x, float y, float z, float cell_metres, int n_shards);   /* shard of a surfel inserted at (x, y, z); host code */
int hrbf_shard_counts(hrbf_handle h, uint32_t out[8]);   /* live counts of all shards; returns 0 one map | 1 ranges | 2 hash (negative: error) */
int hrbf_download_gids(hrbf_handle h, uint32_t *out, size_t cap_surfels);   /* hash ownership, one shard per rank: ids of the rank's surfels */
int hrbf_set_row_sharding(hrbf_handle h, int enable);   /* 0: keep the communicator (sharded map) but let every rank reduce the whole image: no registration collectives */
int hrbf_map_rebalance(hrbf_handle h);
int hrbf_rebalance_plan(const uint32_t *counts, int n_shards, uint32_t *new_counts, uint32_t *moves5, int *n_moves);
uint32_t hrbf_local_surfel_count(hrbf_handle h);

/* What the library's OWN communicator is and what it has issued since the last reset — so that a multi-GPU bench line can
 * prove from the library's side that N ranks took part, and a test can hold the per-frame collective count to the model of
 * SURVEY.md §8e ("ncclAllReduce(sum) per GN iteration; ncclAllReduce(min, P x u64) per projection; ncclAllGather of ...").
 * Operations are counted where the sharded path ISSUES them, by meaning, whichever transport carries them: RCCL on the
 * context's stream, the shared-memory segment (a host barrier + a kernel over the peers' buffers), or — one process playing
 * all shards / ranks in turn — a local kernel (transport 3: nothing crosses a wire, the count is that of the exchange steps). */
enum { HRBF_TRANSPORT_NONE = 0, HRBF_TRANSPORT_RCCL = 1, HRBF_TRANSPORT_SHM = 2, HRBF_TRANSPORT_VIRTUAL = 3 };
typedef struct hrbf_comm_counters {
    int32_t transport;              /* HRBF_TRANSPORT_* */
    int32_t world, rank;            /* RCCL: what ncclCommCount / ncclCommUserRank report for the library's communicator (-1 if the entry
                                       points are missing); shm: the segment's; virtual: the number of shards / ranks played, rank 0 */
    int32_t frames;                 /* hrbf_process_frame* calls since the last reset */
    uint64_t limb_allreduce, limb_allreduce_bytes;       /* registration: all-reduce(sum, int64) of the exact limb sums */
    uint64_t key_min_reduce, key_min_reduce_bytes;       /* projection: all-reduce(min, u64) over the W*H z-buffer keys */
    uint64_t allgather, allgather_bytes;                 /* all-gather(u32): live counts, first ids, record counts, handles, id planes (bytes contributed per rank) */
    uint64_t word_allreduce, word_allreduce_bytes;       /* one-word all-reduce(sum, u32): agreement votes, "every owner has written" */
    uint64_t send, send_bytes, recv, recv_bytes;         /* ncclSend / ncclRecv: packed-record exchange, hrbf_map_rebalance */
    uint64_t host_barriers;                              /* shm transport: barriers through the segment */
} hrbf_comm_counters;
int hrbf_comm_stats(hrbf_handle h, hrbf_comm_counters *out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* HRBF_MI355_H_ */

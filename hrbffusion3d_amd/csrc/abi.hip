// abi.hip — C-ABI of libhrbf_mi355 (include/hrbf_mi355.h): context, per-frame orchestration
// (HRBFFusion::processFrame / predict, Core/src/HRBFFusion.cpp:991-1260) and the operator seams.
//
// One context owns one HIP stream; a frame is a fixed sequence of kernel launches on that stream
// with no host synchronisation inside (pose, weighting, should-fill-in and surfel count all live in
// device memory).  The reference ends every GL pass with glFinish() and every CUDA step with
// cudaDeviceSynchronize().
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "common.h"
#include "kernels.h"

static thread_local char g_err[512] = "";
const char *hrbf_set_error(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return g_err;
}
extern "C" const char *hrbf_last_error(void) { return g_err; }
extern "C" const char *hrbf_version(void) { return "hrbf-mi355 0.1 (gfx950)"; }

extern "C" void hrbf_default_params(hrbf_params *p, int width, int height, float fx, float fy, float cx, float cy,
                                    float depth_scale)
{
    memset(p, 0, sizeof(*p));
    p->width = width; p->height = height; p->fx = fx; p->fy = fy; p->cx = cx; p->cy = cy; p->depth_scale = depth_scale;
    p->confidence_threshold = 5.0f; p->depth_cutoff = 3.5f; p->icp_weight = 10.0f;
    p->fast_odom = 0; p->so3 = 1; p->frame_to_frame_rgb = 0; p->rgb_only = 0; p->pyramid = 1;
    p->max_depth_processed = 20.0f;
    p->use_bilateral = 1; p->init_radius_multiplier = 4.0f; p->curv_estimation_window = 3.0f;
    p->curv_valid_threshold = 300.0f; p->normal_estimation_pca = 1.0f; p->use_conf_eval = 0;
    p->conf_eval_epsilon = 1000.0f;
    p->icp_use_corr_search = 0; p->icp_search_radius = 2; p->icp_use_weighted = 1; p->icp_curv_weight_lambda = 10.0f;
    p->rgb_use_grad_weight = 0; p->use_sparse_icp = 0;
    p->predict_window_multiplier = 3.0f; p->predict_min_neighbors = 6; p->predict_max_neighbors = 10;
    p->predict_conf_threshold = 3.0f;
    p->clean_window_multiplier = 2.0f; p->dense_enough_thresh = 0.75f;
    p->max_surfels = 4 * 1024 * 1024; p->load_trajectory = 0;
}

#define HRBF_RING 1024
#define HRBF_MAX_SHARDS 8

// one shard of the surfel map (SURVEY §8e sharding 2): a contiguous range of the global surfel order.  A single-GPU
// context has exactly one; a rank of a sharded map has one (its own); the single-process test mode has all G.
struct MapShard {
    MapPlanes map;              // single copy: the fuse pass compacts in place
    uint32_t count_ub;          // host upper bound of this shard's surfel count
    uint32_t *d_slot;           // per-surfel merge slot (lowest draw-order record wins)
    uint32_t *d_stats;
    uint32_t *d_tile_count[2];  // per-tile keep counts, double buffered: a pass zeroes the buffer the next pass uses
    uint32_t tile_dirty[2];     // entries of each buffer that may be non-zero
    int tile_par;               // buffer the next pass accumulates into
    uint32_t *d_tile_done; uint32_t epoch;   // tile_done holds the epoch (pass counter) of the pass that raised it
    uint8_t *d_keep_flags;      // one byte per surfel + record: result of the clean test (pass A)
    uint32_t *d_class_word;     // CLEAN_CLASS_WORD builds: {class, first window texel} per surfel from the projection in front of pass A
    uint32_t *d_merged_part;    // the merged count of the last fuse, one word per workgroup of k_apply_merges
    // ownership by spatial hash (kernels.h, ShardRef): allocated by hrbf_map_shard_init(c, 2)
    uint32_t *d_gid;            // cap words: the surfel's place in the global order, moved along with the planes
    uint32_t *d_own_local;      // P words: local index of the pixel's winner where this shard owns it
    uint32_t *d_rec_lbest;      // Q words: local index of the record's matched surfel where this shard owns it
    unsigned long long *d_zpriv;   // P keys {depth, local index}: the shard's private z-buffer
};
// sharded map: the private z-buffer of the virtual shards k >= 1 and the winner records ("winners' attributes
// gathered only for hit pixels"): `send` = the records this shard packed, `recv` = those of the other ranks (real mode)
struct ShardScratch {
    unsigned long long *zbuf;
    uint32_t *rec_count;        // [HRBF_MAX_SHARDS]: records packed by each rank (slot `rank` = own count; all-gathered)
    uint32_t *send_idx, *recv_idx;
    float4 *send_f, *recv_f;    // 6 planes of P float4 each
    uint32_t *h_counts;         // pinned: the all-gathered counts (sizes of the variable-length exchange)
};

// Peer link of a sharded map: the ranks of one node map each other's index-map images and z-buffers through
// hipIpcMemHandle, the owner of a pixel's winner writes its attributes into every rank's images (k_resolve_scatter).  Two
// transports meet the ranks: (a) RCCL, when hrbf_comm_init joined a communicator — key min-reduce = ncclAllReduce(min), the
// "everybody has written" point = a one-word ncclAllReduce, all on the context's stream, handles exchanged by ncclAllGather;
// (b) a POSIX shared-memory rendezvous (hrbf_peer_unique_id / hrbf_comm_init_peer) — handles, counts and barriers go through
// the segment, the key min-reduce reads the peers' z-buffers.  (b) needs no RCCL and also runs with several ranks on ONE GPU
// (RCCL refuses that: "Duplicate GPU detected"), which is how the path is tested on a single device.
#define PEER_RED_WORDS 176   // 87 + 87 + 2 limb sums at most per all-reduce (launch_odometry)
#define PEER_BUFS 8   // z-buffer + six image planes (+ the id plane of a hash-owned map: only then exchanged)
struct PeerShm {
    volatile uint32_t arrived;                  // monotone barrier counter
    volatile uint32_t failed;                   // a rank ran into an error (stream, IPC mapping): published BEFORE the barrier it still arrives at
    volatile uint32_t map_failed;               // a rank could not map its peers' images (agreed on after the mapping barrier)
    volatile uint32_t attached;                 // contexts that joined: a segment serves ONE rendezvous of `world` ranks (its barrier counter is never reset)
    volatile uint32_t counts[2][HRBF_PEER_MAX]; // live surfel counts, double buffered by barrier generation
    hipIpcMemHandle_t handles[HRBF_PEER_MAX][PEER_BUFS];
    volatile long long red[2][HRBF_PEER_MAX][PEER_RED_WORDS];   // the ranks' int64 limb sums of a registration all-reduce (double buffered)
};
struct PeerLink {
    int enabled;            // images are peer-mapped (either transport)
    int shm_mode;           // transport (b)
    int rank, world;
    PeerShm *shm; char shm_name[64]; int shm_owner;
    uint32_t gen;           // barriers passed
    PeerImages img;
    void *opened[HRBF_PEER_MAX][PEER_BUFS];   // what hipIpcOpenMemHandle returned (to close)
    uint32_t *d_token;      // one word all-reduced as the on-stream meeting point (transport a); lives in the context's d_comm_scratch
};

// device scratch of a communicator, allocated ONCE by hrbf_comm_init (round-4 advice: no allocation may stand between a rank and a
// collective its peers are about to issue): [the ranks' IPC handle bytes][vote word][meeting token]
#define COMM_SCRATCH_HANDLES 0
#define COMM_SCRATCH_VOTE ((size_t)HRBF_PEER_MAX * PEER_BUFS * sizeof(hipIpcMemHandle_t) / 4)
#define COMM_SCRATCH_TOKEN (COMM_SCRATCH_VOTE + 1)
#define COMM_SCRATCH_WORDS (COMM_SCRATCH_TOKEN + 1)
struct hrbf_context {
    hrbf_params prm;
    PeerLink peer;
    uint32_t *d_comm_scratch;
    float4 *d_fit_curv1, *d_fit_curv2, *d_fit_normal;   // extension (hrbf_fit_curvature): allocated on first use
    int fit_in_frame;           // extension (hrbf_set_hrbf_fit): processFrame takes the live frame's curvatures from the fitted interpolant
    int fit_window; float fit_support, fit_ridge, fit_jump;   // ... with these parameters (hrbf_set_hrbf_fit_params; defaults 2, 1.25, 0.1, 3.0)
    int device;
    hipStream_t stream;
    Cam cam;
    int W, H, P, Q;
    int tick;
    // inputs
    uint8_t *d_rgb; uint16_t *d_depth;
    // images
    float *d_depth_filtered, *d_depth_metric, *d_depth_metric_f, *d_radius, *d_gradmag, *d_confidence;
    float4 *d_vertex_raw, *d_vertex_filtered, *d_normal, *d_normal_opt, *d_normal_pca, *d_curv1, *d_curv2;
    uint32_t *d_idx; unsigned long long *d_zbuf;
    float4 *d_im_vertconf, *d_im_colortime, *d_im_normrad, *d_im_curvmax, *d_im_curvmin;
    uint8_t *d_pr_image, *d_fi_image;
    float4 *d_pr_vertex, *d_pr_normal, *d_pr_curv1, *d_pr_curv2, *d_fi_vertex, *d_fi_normal, *d_fi_curv1, *d_fi_curv2;
    uint32_t *d_pr_time; float *d_pr_icpw, *d_fi_icpw;
    // map
    MapShard sh[HRBF_MAX_SHARDS];   // local shards (nsh of them)
    int nsh;                    // local shard count: 1, or G in the single-process test mode
    int G;                      // global shard count (1 = the map is not sharded)
    int shard_first;            // global index of sh[0] (the rank in real mode)
    int shard_real;             // one shard per rank, reductions through RCCL
    int rows_replicated;        // communicator present but registration NOT row-sharded (hrbf_set_row_sharding(h, 0))
    ShardScratch x;
    int target;                 // index of the live row of d_counts (ping-pong)
    int map_dirty;              // map uploaded from outside since the last fuse pass -> full curvature re-check
    int fuse_tick;              // tick of the last association/merge (a clean at another tick re-checks everything)
    uint32_t cap;               // per shard
    uint32_t *d_counts;         // [2][HRBF_MAX_SHARDS]: live surfel counts of all G shards, ping-pong with the clean pass
    uint32_t *h_count_pinned;   // [HRBF_MAX_SHARDS] async read-back (1-frame lag)
    uint8_t *h_stage[3]; hipEvent_t ev_stage[3]; bool stage_used[3]; uint32_t stage_head;   // pinned input staging ring
    const uint8_t *ride_rgb_src;   // device view of a staged RGB image that the next st_filter uploads (hrbf_process_frame)
    hipEvent_t ev_count; bool ev_pending; uint32_t ub_growth_since;
    bool inputs_lent;   // the last frame read the caller's device buffers (hrbf_process_frame_device): d_rgb / d_depth hold an OLDER frame
    // ownership by spatial hash: the next free global-order id (the same on every rank: seed size, then + Q per clean pass),
    // 1 / cell size, the device word holding the smallest id alive, scratch of the hashed seeding
    int peer_fallback;          // the ranks agreed to exchange packed records because a rank could not map its peers (hrbf_shard_exchange_mode)
    int renumber_failed;        // hash ownership: the id renumbering failed; frames fail fast until the status is cleared
    int hash_mode; uint32_t g_next, g_renumber_at, hash_renumbered; float hash_inv_cell; uint32_t *d_gfirst; uint32_t *d_init_flags2, *d_init_offs2, *d_gtotal;
    RecPlanes rec; int32_t *d_rec_flag; uint32_t *d_rec_best;
    uint32_t *d_init_flags, *d_init_offs;
    uint32_t max_tiles;
    float4 *d_clean_tex;        // packed index-map texels for the clean test (k_resolve)
    float clean_thr; int clean_time;   // the threshold and time baked into them
    DevPose *d_pose;
    int index_submap;           // submap id stamped on new surfels (HRBFFusion::indexSubmap)
    uint8_t *d_submap_active; int n_submap_active;   // KeyFrameIDMap (null = all active)
    float *d_delta; int delta_cap;                   // updateModel matrices
    OdoComm comm;               // row-sharded registration (null comm + virtual_world <= 1: single GPU)
    hrbf_comm_counters cstats;  // hrbf_comm_stats: what the sharded paths issued since the last reset
    unsigned long long ar_count[2];   // limb all-reduces {calls, bytes}, bumped by launch_odometry through comm.ar_count
    int ar_failed;              // launch_odometry: an all-reduce returned non-zero -> HRBF_STATUS_COLLECTIVE
    int fill_flag_fresh;        // DevPose::should_fill_in was computed by the last k_fillin (nothing touched the prediction since)
    OdoBuffers odo;
    // timing
    int timing; hipEvent_t ev[12]; float timings[8];
    // per-frame ring: HIP events bracketing the fuse pass — F2 (k_apply_merges: m0..m1) and F3 (k_clean_flags +
    // k_fuse_stream: e0..e1), nothing else — + its statistics words
    hipEvent_t *ring_e0, *ring_e1, *ring_m0, *ring_m1; uint32_t *d_stats_ring; uint32_t ring_head; uint32_t ring_valid; uint32_t ring_stride; int level0_done;
    uint32_t ring_merge_head;   // ring slot the last merge events went to (a clean without a fuse has no F2 part)
    uint32_t status;            // sticky HRBF_STATUS_* bits folded from the device on blocking calls
    PoseLog *h_pose_log, *d_pose_log_view;   // pinned host ring + its device-side address
    uint32_t frames_enqueued;   // process_frame calls so far = index of the next frame in the pose log
};

static void peer_close(hrbf_context *c);
static void peer_shm_release(hrbf_context *c);

template <typename T>
static int dalloc(T **p, size_t n)
{
    hipError_t e = hipMalloc((void **)p, sizeof(T) * (n ? n : 1));
    if (e != hipSuccess) { hrbf_set_error("hipMalloc(%zu): %s", sizeof(T) * n, hipGetErrorString(e)); return HRBF_ERR_DEVICE; }
    e = hipMemset(*p, 0, sizeof(T) * (n ? n : 1));
    if (e != hipSuccess) { hrbf_set_error("hipMemset: %s", hipGetErrorString(e)); return HRBF_ERR_DEVICE; }
    return HRBF_OK;
}
#define DA(ptr, n) do { int _r = dalloc(&(ptr), (n)); if (_r) { hrbf_destroy(c); return _r; } } while (0)

static int alloc_planes(hrbf_context *c, MapPlanes &m, size_t n)
{
    int r;
    if ((r = dalloc(&m.p0, n)) || (r = dalloc(&m.p1, n)) || (r = dalloc(&m.p2, n)) || (r = dalloc(&m.p3, n)) ||
        (r = dalloc(&m.p4, n)))
        return r;
    (void)c;
    return HRBF_OK;
}

static void free_planes(MapPlanes &m)
{
    hipFree(m.p0); hipFree(m.p1); hipFree(m.p2); hipFree(m.p3); hipFree(m.p4);
    m.p0 = m.p1 = m.p2 = m.p3 = m.p4 = nullptr;
}
// everything a shard owns; the merge slots still have to be filled with 0xFFFFFFFF by the caller
static int alloc_shard(hrbf_context *c, MapShard &sh)
{
    int r = alloc_planes(c, sh.map, c->cap);
    if (!r) r = dalloc(&sh.d_slot, c->cap);
    if (!r) r = dalloc(&sh.d_stats, 8);
    if (!r) r = dalloc(&sh.d_tile_count[0], (size_t)c->max_tiles * fuse_tile_count_stride());
    if (!r) r = dalloc(&sh.d_tile_count[1], (size_t)c->max_tiles * fuse_tile_count_stride());
    sh.tile_dirty[0] = sh.tile_dirty[1] = 0; sh.tile_par = 0; sh.epoch = 0;
    if (!r) r = dalloc(&sh.d_tile_done, c->max_tiles);
    if (!r) r = dalloc(&sh.d_keep_flags, (size_t)c->cap + (size_t)c->Q + 64);
#ifdef CLEAN_CLASS_WORD
    if (!r) r = dalloc(&sh.d_class_word, (size_t)c->cap + 64);
#endif
    if (!r) r = dalloc(&sh.d_merged_part, (size_t)merge_workgroups(c->Q));
    sh.count_ub = 0;
    return r;
}
static void free_shard(MapShard &sh)
{
    free_planes(sh.map);
    void *q[] = {sh.d_slot, sh.d_stats, sh.d_tile_count[0], sh.d_tile_count[1], sh.d_tile_done, sh.d_keep_flags, sh.d_merged_part, sh.d_class_word};
    for (void *p : q) if (p) hipFree(p);
    sh.d_slot = sh.d_stats = sh.d_tile_count[0] = sh.d_tile_count[1] = sh.d_tile_done = nullptr; sh.d_keep_flags = nullptr; sh.d_class_word = nullptr;
    sh.d_merged_part = nullptr;
    void *hq[] = {sh.d_gid, sh.d_own_local, sh.d_rec_lbest, sh.d_zpriv};
    for (void *p : hq) if (p) hipFree(p);
    sh.d_gid = sh.d_own_local = sh.d_rec_lbest = nullptr; sh.d_zpriv = nullptr;
}
static void free_scratch(ShardScratch &x)
{
    void *q[] = {x.zbuf, x.rec_count, x.send_idx, x.recv_idx, x.send_f, x.recv_f};
    for (void *p : q) if (p) hipFree(p);
    if (x.h_counts) hipHostFree(x.h_counts);
    memset(&x, 0, sizeof(x));
}
static float4 *map_plane(const MapPlanes &m, int k)
{
    switch (k) { case 0: return m.p0; case 1: return m.p1; case 2: return m.p2; case 3: return m.p3; default: return m.p4; }
}
static uint32_t *counts_live(hrbf_context *c) { return c->d_counts + (size_t)c->target * HRBF_MAX_SHARDS; }
static uint32_t *counts_next(hrbf_context *c) { return c->d_counts + (size_t)(1 - c->target) * HRBF_MAX_SHARDS; }
static ShardRef shard_ref(hrbf_context *c, int k)
{
    ShardRef r = {counts_live(c), c->shard_first + k, c->G, nullptr, nullptr, nullptr, nullptr};
    if (c->hash_mode) { r.gid = c->sh[k].d_gid; r.g_first = c->d_gfirst; r.own_local = c->sh[k].d_own_local; r.rec_lbest = c->sh[k].d_rec_lbest; }
    return r;
}

extern "C" int hrbf_create(const hrbf_params *p, int device, hrbf_handle *out)
{
    if (!p || !out) { hrbf_set_error("null argument"); return HRBF_ERR_INVALID; }
    *out = nullptr;
    if (p->width <= 0 || p->height <= 0 || p->width % 8 || p->height % 8) {
        hrbf_set_error("width/height must be positive multiples of 8 (3-level pyramid + quarter grid)");
        return HRBF_ERR_INVALID;
    }
    if (p->curv_estimation_window > 3.0f || p->predict_window_multiplier > 3.0f || p->clean_window_multiplier > 8.0f) {
        hrbf_set_error("window multipliers above the reference defaults (3/3/8) are not supported");
        return HRBF_ERR_INVALID;
    }
    {   // every real parameter finite (a NaN multiplier would reach an int conversion on the host), a camera that projects, a depth unit
        const float reals[] = {p->fx, p->fy, p->cx, p->cy, p->depth_scale, p->confidence_threshold, p->depth_cutoff, p->icp_weight,
                               p->max_depth_processed, p->init_radius_multiplier, p->curv_estimation_window, p->curv_valid_threshold,
                               p->normal_estimation_pca, p->conf_eval_epsilon, p->icp_curv_weight_lambda, p->predict_window_multiplier,
                               p->predict_conf_threshold, p->clean_window_multiplier, p->dense_enough_thresh};
        for (float v : reals)
            if (!(v - v == 0.0f)) { hrbf_set_error("a parameter is NaN or infinite"); return HRBF_ERR_INVALID; }
        if (p->fx == 0.0f || p->fy == 0.0f || !(p->depth_scale > 0.0f)) {
            hrbf_set_error("fx and fy must not be 0 (fy may be negative), depth_scale must be positive");
            return HRBF_ERR_INVALID;
        }
        if (p->curv_estimation_window < 0.0f || p->predict_window_multiplier < 0.0f || p->clean_window_multiplier < 0.0f || p->icp_search_radius < 0 ||
            p->max_surfels <= 0) {
            hrbf_set_error("negative window / search radius, or no room for surfels");
            return HRBF_ERR_INVALID;
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        hrbf_set_error("no HIP device visible: libhrbf_mi355 has no CPU fallback");
        return HRBF_ERR_NODEVICE;
    }
    if (device < 0 || device >= ndev) { hrbf_set_error("device %d out of range (%d visible)", device, ndev); return HRBF_ERR_INVALID; }
    HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        hrbf_set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return HRBF_ERR_NODEVICE;
    }
    if (p->max_surfels <= 0) { hrbf_set_error("max_surfels must be positive"); return HRBF_ERR_INVALID; }
    hrbf_context *c = (hrbf_context *)calloc(1, sizeof(hrbf_context));
    if (!c) { hrbf_set_error("out of host memory"); return HRBF_ERR_DEVICE; }
    c->prm = *p; c->device = device;
    c->W = p->width; c->H = p->height; c->P = c->W * c->H; c->Q = (c->W / 2) * (c->H / 2);
    c->tick = 1;
    c->cam.W = c->W; c->cam.H = c->H; c->cam.fx = p->fx; c->cam.fy = p->fy; c->cam.cx = p->cx; c->cam.cy = p->cy;
    c->cam.camz = (float)(1.0 / (double)p->fx); c->cam.camw = (float)(1.0 / (double)p->fy);
    c->cam.max_dist = sqrtf(((float)c->H * 0.5f) * ((float)c->H * 0.5f) + ((float)c->W * 0.5f) * ((float)c->W * 0.5f));
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { hrbf_set_error("hipStreamCreate: %s", hipGetErrorString(e)); free(c); return HRBF_ERR_DEVICE; }
    const size_t P = c->P;
    DA(c->d_rgb, P * 3); DA(c->d_depth, P);
    DA(c->d_depth_filtered, P); DA(c->d_depth_metric, P); DA(c->d_depth_metric_f, P); DA(c->d_radius, P);
    DA(c->d_gradmag, P); DA(c->d_confidence, P);
    DA(c->d_vertex_raw, P); DA(c->d_vertex_filtered, P); DA(c->d_normal, P); DA(c->d_normal_opt, P); DA(c->d_normal_pca, P);
    DA(c->d_curv1, P); DA(c->d_curv2, P);
    DA(c->d_idx, P); DA(c->d_zbuf, P);
    DA(c->d_im_vertconf, P); DA(c->d_im_colortime, P); DA(c->d_im_normrad, P); DA(c->d_im_curvmax, P); DA(c->d_im_curvmin, P);
    DA(c->d_pr_image, P * 4); DA(c->d_fi_image, P * 4);
    DA(c->d_pr_vertex, P); DA(c->d_pr_normal, P); DA(c->d_pr_curv1, P); DA(c->d_pr_curv2, P);
    DA(c->d_fi_vertex, P); DA(c->d_fi_normal, P); DA(c->d_fi_curv1, P); DA(c->d_fi_curv2, P);
    DA(c->d_pr_time, P); DA(c->d_pr_icpw, P); DA(c->d_fi_icpw, P);
    c->cap = (uint32_t)p->max_surfels;
    c->max_tiles = (c->cap + c->Q) / fuse_tile_items() + 2;
    c->nsh = 1; c->G = 1; c->shard_first = 0; c->shard_real = 0;
    { int r; if ((r = alloc_shard(c, c->sh[0])) || (r = alloc_planes(c, c->rec, c->Q))) { hrbf_destroy(c); return r; } }
    DA(c->d_counts, 2 * HRBF_MAX_SHARDS);
    DA(c->d_rec_flag, c->Q); DA(c->d_rec_best, c->Q);
    DA(c->d_init_flags, P); DA(c->d_init_offs, P);
    DA(c->d_pose, 1);
    DA(c->d_clean_tex, clean_tex_elems((int)P));
    e = hipHostMalloc((void **)&c->h_count_pinned, sizeof(uint32_t) * HRBF_MAX_SHARDS, hipHostMallocDefault);
    if (e != hipSuccess) { hrbf_set_error("hipHostMalloc: %s", hipGetErrorString(e)); hrbf_destroy(c); return HRBF_ERR_DEVICE; }
    c->h_count_pinned[0] = 0;
    e = hipHostMalloc((void **)&c->h_pose_log, sizeof(PoseLog), hipHostMallocMapped | hipHostMallocCoherent);   // fine-grained: device stores bypass L2
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&c->d_pose_log_view, c->h_pose_log, 0);
    if (e != hipSuccess) { hrbf_set_error("pose log: %s", hipGetErrorString(e)); hrbf_destroy(c); return HRBF_ERR_DEVICE; }
    c->h_pose_log->completed = 0;
    hipEventCreateWithFlags(&c->ev_count, hipEventDisableTiming);
    for (int i = 0; i < 12; ++i) hipEventCreate(&c->ev[i]);
    c->ring_e0 = (hipEvent_t *)calloc(HRBF_RING, sizeof(hipEvent_t)); c->ring_e1 = (hipEvent_t *)calloc(HRBF_RING, sizeof(hipEvent_t));
    c->ring_m0 = (hipEvent_t *)calloc(HRBF_RING, sizeof(hipEvent_t)); c->ring_m1 = (hipEvent_t *)calloc(HRBF_RING, sizeof(hipEvent_t));
    for (int i = 0; i < HRBF_RING; ++i) {
        hipEventCreate(&c->ring_e0[i]); hipEventCreate(&c->ring_e1[i]); hipEventCreate(&c->ring_m0[i]); hipEventCreate(&c->ring_m1[i]);
    }
    DA(c->d_stats_ring, (size_t)HRBF_RING * 8);
    // odometry buffers
    for (int i = 0; i < HRBF_NUM_PYRS; ++i) {
        OdoLevel &L = c->odo.lv[i];
        L.rows = c->H >> i; L.cols = c->W >> i;
        const size_t n = (size_t)L.rows * L.cols;
        DA(L.vmap_g, 4 * n); DA(L.nmap_g, 4 * n); DA(L.ck1_g, 4 * n); DA(L.ck2_g, 4 * n);
        DA(L.vmap_c, 4 * n); DA(L.nmap_c, 4 * n); DA(L.ck1_c, 4 * n); DA(L.ck2_c, 4 * n);
        DA(L.icpw, n); DA(L.last_depth, n); DA(L.next_depth, n);
        DA(L.last_image, n); DA(L.next_image, n); DA(L.last_next_image, n);
        DA(L.dIdx, n); DA(L.dIdy, n); DA(L.cloud, 3 * n);
        DA(L.icp_cur, 2 * n); DA(L.icp_model, 2 * n); DA(L.rgb_mask, n); DA(L.cloud4, n); DA(L.dIxy, n);
        if (c->prm.use_sparse_icp) DA(L.sparse, 2 * n);
    }
    { uint8_t *st; DA(st, odo_state_bytes()); c->odo.state = (OdoState *)st; }
    DA(c->odo.corres, P * 6); DA(c->odo.corres_diff, P);
    c->odo.max_blocks = (int)((P + 255) / 256);
    { uint8_t *blk; DA(blk, odo_slot_bytes());
      c->odo.icp_part = (long long *)blk; c->odo.rgb_part = c->odo.icp_part + 32 * 87;
      c->odo.res_part = c->odo.rgb_part + 32 * 87; c->odo.so3_part = c->odo.res_part + 64 * 2; }
    DA(c->odo.totals, 256);
    if (predict_upload_tables() != 0) { hrbf_set_error("constant upload failed"); hrbf_destroy(c); return HRBF_ERR_DEVICE; }
    // the zero-fills of dalloc() ran on the null stream and c->stream is non-blocking: drain them before the
    // initialisation kernels touch the same buffers
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { hrbf_set_error("init sync: %s", hipGetErrorString(e)); hrbf_destroy(c); return HRBF_ERR_DEVICE; }
    launch_fill_u32(c->stream, c->sh[0].d_slot, c->cap, 0xFFFFFFFFu);
    launch_zbuf_reset(c->stream, c->d_zbuf, P);
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    launch_pose_set(c->stream, c->d_pose, I, 1);
    { float one = 1.0f; hipMemcpyAsync(&c->d_pose->weighting, &one, sizeof(float), hipMemcpyHostToDevice, c->stream); }
    e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { hrbf_set_error("init sync: %s", hipGetErrorString(e)); hrbf_destroy(c); return HRBF_ERR_DEVICE; }
    *out = c;
    return HRBF_OK;
}


// ------------------------------------------------------------------------------------------ RCCL (row-sharded registration)
// librccl is loaded on first use: a single-GPU process never touches it and the library keeps no link-time dependency.
#include <dlfcn.h>
namespace {
struct RcclId128 { char b[128]; };   // ncclUniqueId: 128 opaque bytes, passed by value
struct RcclApi {
    void *lib;
    int (*GetUniqueId)(void *id128);
    int (*CommInitRank)(void **comm, int nranks, RcclId128 id, int rank);
    int (*CommDestroy)(void *comm);
    int (*AllReduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t s);
    const char *(*GetErrorString)(int);
    int (*AllGather)(const void *send, void *recv, size_t sendcount, int dtype, void *comm, hipStream_t s);
    int (*Send)(const void *send, size_t count, int dtype, int peer, void *comm, hipStream_t s);
    int (*Recv)(void *recv, size_t count, int dtype, int peer, void *comm, hipStream_t s);
    int (*GroupStart)();
    int (*GroupEnd)();
    int (*CommCount)(void *comm, int *count);        // optional (hrbf_comm_stats)
    int (*CommUserRank)(void *comm, int *rank);      // optional
};
RcclApi g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
// ncclDataType_t / ncclRedOp_t values (rccl.h)
const int kNcclUint32 = 3, kNcclInt64 = 4, kNcclUint64 = 5, kNcclSum = 0, kNcclMin = 3;

int rccl_load()
{
    if (g_rccl.lib) return HRBF_OK;
    void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { hrbf_set_error("librccl.so: %s", dlerror()); return HRBF_ERR_COMM; }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(lib, "ncclAllGather");
    g_rccl.Send = (decltype(g_rccl.Send))dlsym(lib, "ncclSend");
    g_rccl.Recv = (decltype(g_rccl.Recv))dlsym(lib, "ncclRecv");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(lib, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(lib, "ncclGroupEnd");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(lib, "ncclCommCount");
    g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))dlsym(lib, "ncclCommUserRank");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.AllGather ||
        !g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd) {
        hrbf_set_error("librccl.so lacks an expected entry point"); dlclose(lib); return HRBF_ERR_COMM;
    }
    g_rccl.lib = lib;
    return HRBF_OK;
}
int rccl_allreduce_i64(void *comm, long long *buf, size_t count, hipStream_t s)
{
    return g_rccl.AllReduce(buf, buf, count, kNcclInt64, kNcclSum, comm, s);   // in place, on the context's stream
}
// the three collectives of a sharded map: z-buffer keys (min), resolved images as integers (sum: one owner per pixel,
// zeros elsewhere -> exact), live counts (all-gather; `own` = row + rank, the in-place form)
int rccl_allreduce_min_u64(void *comm, unsigned long long *buf, size_t count, hipStream_t s)
{
    return g_rccl.AllReduce(buf, buf, count, kNcclUint64, kNcclMin, comm, s);
}
int rccl_allreduce_sum_u32(void *comm, uint32_t *buf, size_t count, hipStream_t s)
{
    return g_rccl.AllReduce(buf, buf, count, kNcclUint32, kNcclSum, comm, s);
}
int rccl_allgather_u32(void *comm, const uint32_t *own, uint32_t *row, size_t count_each, hipStream_t s)
{
    return g_rccl.AllGather(own, row, count_each, kNcclUint32, comm, s);
}
}   // namespace
// hrbf_comm_stats: one exchange step of kind `what` issued, `bytes` contributed by this rank
#define COMM_COUNT(c, what, bytes) do { (c)->cstats.what += 1; (c)->cstats.what##_bytes += (uint64_t)(bytes); } while (0)

extern "C" void hrbf_destroy(hrbf_handle c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    odo_release(c->odo);
    void *ptrs[] = {c->d_rgb, c->d_depth, c->d_depth_filtered, c->d_depth_metric, c->d_depth_metric_f, c->d_radius,
                    c->d_gradmag, c->d_confidence, c->d_vertex_raw, c->d_vertex_filtered, c->d_normal, c->d_normal_opt,
                    c->d_normal_pca, c->d_curv1, c->d_curv2, c->d_idx, c->d_zbuf, c->d_im_vertconf, c->d_im_colortime,
                    c->d_im_normrad, c->d_im_curvmax, c->d_im_curvmin, c->d_pr_image, c->d_fi_image, c->d_pr_vertex,
                    c->d_pr_normal, c->d_pr_curv1, c->d_pr_curv2, c->d_fi_vertex, c->d_fi_normal, c->d_fi_curv1,
                    c->d_fi_curv2, c->d_pr_time, c->d_pr_icpw, c->d_fi_icpw, c->d_counts, c->d_rec_flag, c->d_rec_best,
                    c->d_init_flags, c->d_init_offs, c->d_pose, c->d_clean_tex, c->d_gfirst, c->d_init_flags2, c->d_init_offs2, c->d_gtotal,
                    c->odo.state, c->odo.corres, c->odo.corres_diff, c->odo.icp_part, c->odo.totals};
    for (void *p : ptrs) if (p) hipFree(p);
    for (int k = 0; k < HRBF_MAX_SHARDS; ++k) free_shard(c->sh[k]);
    free_scratch(c->x);
    free_planes(c->rec);
    for (int i = 0; i < HRBF_NUM_PYRS; ++i) {
        OdoLevel &L = c->odo.lv[i];
        void *q[] = {L.vmap_g, L.nmap_g, L.ck1_g, L.ck2_g, L.vmap_c, L.nmap_c, L.ck1_c, L.ck2_c, L.icpw, L.last_depth,
                     L.next_depth, L.last_image, L.next_image, L.last_next_image, L.dIdx, L.dIdy, L.cloud, L.icp_cur, L.icp_model, L.rgb_mask, L.cloud4, L.dIxy, L.sparse};
        for (void *p : q) if (p) hipFree(p);
    }
    if (c->h_count_pinned) hipHostFree(c->h_count_pinned);
    if (c->h_pose_log) hipHostFree(c->h_pose_log);
    for (int k = 0; k < 3; ++k) { if (c->h_stage[k]) hipHostFree(c->h_stage[k]); if (c->ev_stage[k]) hipEventDestroy(c->ev_stage[k]); }
    if (c->ev_count) hipEventDestroy(c->ev_count);
    for (int i = 0; i < 12; ++i) if (c->ev[i]) hipEventDestroy(c->ev[i]);
    hipEvent_t *rings[] = {c->ring_e0, c->ring_e1, c->ring_m0, c->ring_m1};
    for (hipEvent_t *r : rings) {
        if (!r) continue;
        for (int i = 0; i < HRBF_RING; ++i) if (r[i]) hipEventDestroy(r[i]);
        free(r);
    }
    if (c->d_stats_ring) hipFree(c->d_stats_ring);
    if (c->d_fit_curv1) hipFree(c->d_fit_curv1);
    if (c->d_fit_curv2) hipFree(c->d_fit_curv2);
    if (c->d_fit_normal) hipFree(c->d_fit_normal);
    peer_close(c); peer_shm_release(c);
    if (c->comm.comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm.comm);
    if (c->d_comm_scratch) { hipFree(c->d_comm_scratch); c->d_comm_scratch = nullptr; }
    if (c->d_submap_active) hipFree(c->d_submap_active);
    if (c->d_delta) hipFree(c->d_delta);
    if (c->stream) hipStreamDestroy(c->stream);
    free(c);
}

// ------------------------------------------------------------------------------------------ stages
static OdoSources make_sources(hrbf_context *c)
{
    OdoSources s;
    s.pr_vertex = c->d_pr_vertex; s.pr_normal = c->d_pr_normal; s.pr_curv1 = c->d_pr_curv1; s.pr_curv2 = c->d_pr_curv2;
    s.pr_icpw = c->d_pr_icpw; s.pr_image = c->d_pr_image;
    s.fi_vertex = c->d_fi_vertex; s.fi_normal = c->d_fi_normal; s.fi_curv1 = c->d_fi_curv1; s.fi_curv2 = c->d_fi_curv2;
    s.fi_icpw = c->d_fi_icpw; s.fi_image = c->d_fi_image;
    s.vertex_filtered = c->d_vertex_filtered; s.normal = c->d_normal; s.curv1 = c->d_curv1; s.curv2 = c->d_curv2;
    s.rgb = c->d_rgb;
    return s;
}
static OdoConfig make_cfg(hrbf_context *c)
{
    OdoConfig g;
    const hrbf_params &p = c->prm;
    g.fx = p.fx; g.fy = p.fy; g.cx = p.cx; g.cy = p.cy;
    g.rgb_only = p.rgb_only; g.icp_weight = p.icp_weight; g.pyramid = p.pyramid; g.fast_odom = p.fast_odom; g.so3 = p.so3;
    g.frame_to_frame_rgb = p.frame_to_frame_rgb;
    g.use_search = p.icp_use_corr_search; g.search_radius = p.icp_search_radius; g.use_weighted = p.icp_use_weighted;
    g.rgb_use_grad = p.rgb_use_grad_weight; g.curv_thr = p.curv_valid_threshold;
    g.use_sparse = p.use_sparse_icp;
    return g;
}

static void st_filter(hrbf_context *c)
{
    // a pending RGB upload of the host-pointer entry rides in this launch (nothing reads the RGB image before the next kernel but one)
    launch_filter_metric(c->stream, c->cam, c->d_depth, c->d_depth_filtered, c->d_depth_metric, c->d_depth_metric_f,
                         c->prm.depth_scale, c->prm.depth_cutoff, c->prm.use_bilateral, c->ride_rgb_src, c->d_rgb,
                         (size_t)c->P * 3);
    c->ride_rgb_src = nullptr;
}
static void st_vnr(hrbf_context *c)
{
    launch_vertex_normal_radius(c->stream, c->cam, c->d_depth_metric, c->d_depth_metric_f, c->d_vertex_raw,
                                c->d_vertex_filtered, c->d_normal, c->d_normal_pca, c->d_radius,
                                c->prm.init_radius_multiplier, c->prm.normal_estimation_pca > 0.0f);
}
static OdoSources make_sources(hrbf_context *c);
static OdoConfig make_cfg(hrbf_context *c);
// with_level0: the frame path, when a registration follows and the shouldFillIn flag on the device is the current one —
// level 0 of the registration pyramids is written from the tail of the curvature kernel (c->level0_done)
static void st_curv(hrbf_context *c, bool with_level0 = false)
{
    c->level0_done = 0;
    if (with_level0) {
        const OdoConfig cfg = make_cfg(c);
        Level0Args l0;
        l0.L = c->odo.lv[0]; l0.src = make_sources(c); l0.dp = c->d_pose; l0.f2f = cfg.frame_to_frame_rgb; l0.curv_thr = cfg.curv_thr;
        l0.pack = (!cfg.use_search && !cfg.use_sparse) ? 1 : 0;   // the packed-operand registration reads nothing else of level 0's model maps
        launch_curvature_level0(c->stream, c->cam, c->d_vertex_filtered, c->d_normal, c->d_curv1, c->d_curv2, c->d_gradmag,
                                c->d_normal_opt, c->prm.curv_estimation_window, l0);
        c->level0_done = 1 + l0.pack;
    } else
    launch_curvature(c->stream, c->cam, c->d_vertex_filtered, c->d_normal, c->d_curv1, c->d_curv2, c->d_gradmag,
                     c->d_normal_opt, c->prm.curv_estimation_window);
    // updateNormalRad: NORMAL <- NORMAL_OPT (HRBFFusion.cpp:1301-1310); the kernel rewrites every pixel, so the
    // two buffers just trade places
    float4 *t = c->d_normal; c->d_normal = c->d_normal_opt; c->d_normal_opt = t;
}
static void st_conf(hrbf_context *c)
{
    launch_confidence(c->stream, c->cam, c->d_gradmag, c->d_confidence, &c->d_pose->weighting, c->prm.use_conf_eval,
                      c->prm.conf_eval_epsilon);
}
// a test hook's environment variable names THIS rank: a complete non-negative decimal number equal to it (atoi would read "",
// "abc" and "0x1" as rank 0)
static bool env_names_rank(const char *name, int rank)
{
    const char *v = getenv(name);
    if (!v || !*v) return false;
    char *end = nullptr;
    const long n = strtol(v, &end, 10);
    return end != v && *end == 0 && n >= 0 && n == (long)rank;
}
// ---- peer link -----------------------------------------------------------------------------------------------------------
#include <fcntl.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static int peer_barrier_host(hrbf_context *c)
{
    PeerLink &pl = c->peer;
    // a rank in trouble still ARRIVES (its generation counter stays aligned with the shared one) and says so in the segment: every
    // rank leaves this barrier with the same verdict instead of the healthy ones spinning for a minute (round-3 advice)
    const bool bad = hipStreamSynchronize(c->stream) != hipSuccess;
    if (bad) { pl.shm->failed = 1u; __sync_synchronize(); }
    c->cstats.host_barriers += 1;
    const uint32_t target = ++pl.gen * (uint32_t)pl.world;
    __sync_fetch_and_add(&pl.shm->arrived, 1u);
    struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint32_t spins = 0;; ++spins) {
        if ((int32_t)(__atomic_load_n(&pl.shm->arrived, __ATOMIC_ACQUIRE) - target) >= 0) {
            if (bad) { hrbf_set_error("peer barrier: stream error"); return HRBF_ERR_DEVICE; }
            if (pl.shm->failed) { hrbf_set_error("peer barrier: a rank reported a failure"); return HRBF_ERR_COMM; }
            return HRBF_OK;
        }
        if ((spins & 63u) == 63u) {
            struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            if (t1.tv_sec - t0.tv_sec > 60) { hrbf_set_error("peer barrier: a rank did not arrive within 60 s"); c->status |= HRBF_STATUS_INTERNAL_BOUND; return HRBF_ERR_COMM; }
            usleep(20);
        }
    }
}
// The registration's int64 all-reduce on the shared-memory transport (hrbf_comm_init_peer + hrbf_set_row_sharding(h, 1)): the
// sums make a round trip through the host segment — slow, and only there so that the row-sharded registration of REAL ranks
// (strips, fold, all-reduce, stand-alone solve: launch_odometry's sharded path) runs between processes on the one GPU a test
// box has, where RCCL refuses a second rank.  Integer sums: the result does not depend on the order of the ranks.
static int shm_allreduce_i64(void *ctx, long long *buf, size_t n, hipStream_t s)
{
    hrbf_context *c = (hrbf_context *)ctx;
    PeerLink &pl = c->peer;
    if (!pl.shm_mode || n > PEER_RED_WORDS) return -1;
    long long mine[PEER_RED_WORDS] = {0};
    hipMemcpyAsync(mine, buf, sizeof(long long) * n, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) { pl.shm->failed = 1u; __sync_synchronize(); }   // `mine` may be stale: every rank leaves the barrier below with the failure
    const int slot = (int)((pl.gen + 1u) & 1u);
    for (size_t i = 0; i < n; ++i) pl.shm->red[slot][pl.rank][i] = mine[i];
    __sync_synchronize();
    if (peer_barrier_host(c)) { c->status |= HRBF_STATUS_INTERNAL_BOUND; return -1; }   // the caller (launch_odometry) cannot stop a frame half way: the status bit says its pose is not to be trusted
    for (size_t i = 0; i < n; ++i) {
        long long t = 0;
        for (int g = 0; g < pl.world; ++g) t += pl.shm->red[slot][g][i];
        mine[i] = t;
    }
    hipMemcpyAsync(buf, mine, sizeof(long long) * n, hipMemcpyHostToDevice, s);
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;    // `mine` lives on this stack
}
// the point after which every rank's writes into this rank's images are complete
static int peer_meet(hrbf_context *c)
{
    COMM_COUNT(c, word_allreduce, 4);
    if (c->peer.shm_mode) return peer_barrier_host(c);
    return rccl_allreduce_sum_u32(c->comm.comm, c->peer.d_token, 1, c->stream) == 0 ? HRBF_OK : HRBF_ERR_COMM;
}
static void peer_close(hrbf_context *c)
{
    PeerLink &pl = c->peer;
    for (int g = 0; g < HRBF_PEER_MAX; ++g)
        for (int b = 0; b < PEER_BUFS; ++b)
            if (pl.opened[g][b]) { hipIpcCloseMemHandle(pl.opened[g][b]); pl.opened[g][b] = nullptr; }
    pl.d_token = nullptr;   // part of d_comm_scratch, which lives as long as the communicator
    pl.enabled = 0;
}
static void peer_shm_release(hrbf_context *c)
{
    PeerLink &pl = c->peer;
    if (pl.shm) { munmap((void *)pl.shm, sizeof(PeerShm)); pl.shm = nullptr; }
    if (pl.shm_owner && pl.shm_name[0]) shm_unlink(pl.shm_name);
    pl.shm_mode = 0; pl.shm_owner = 0; pl.shm_name[0] = 0;
}
// exchange the IPC handles of this rank's z-buffer and six index-map planes and map everybody else's
static int peer_map_images(hrbf_context *c, int *all_mapped)
{
    *all_mapped = 0;
    PeerLink &pl = c->peer;
    const int G = pl.world, me = pl.rank;
    void *mine[PEER_BUFS] = {c->d_zbuf, c->d_im_vertconf, c->d_im_normrad, c->d_im_colortime, c->d_im_curvmax, c->d_im_curvmin, c->d_clean_tex, c->sh[0].d_gid};
    const int NB = c->sh[0].d_gid ? PEER_BUFS : PEER_BUFS - 1;   // the id plane exists under hash ownership only
    hipIpcMemHandle_t all[HRBF_PEER_MAX][PEER_BUFS];
    memset(all, 0, sizeof(all));
    // A rank that fails locally (no IPC handle, a handle that does not open: another node, another IPC namespace) still goes
    // through EVERY meeting point below, and the outcome is agreed on by all ranks — a rank that left early would make the others
    // issue collectives it never joins (round-3 advice).  *all_mapped = 0: nobody keeps a mapping (the caller falls back to the
    // packed-record exchange on the RCCL transport; the shm transport has no other path and fails on every rank alike).
    int fail = 0;
    for (int b = 0; b < NB; ++b) {
        const hipError_t e = hipIpcGetMemHandle(&all[me][b], mine[b]);
        if (e != hipSuccess) { hrbf_set_error("hipIpcGetMemHandle: %s", hipGetErrorString(e)); fail = 1; memset(&all[me][b], 0, sizeof(all[me][b])); }
    }
    uint32_t *d = nullptr;   // RCCL transport: device staging of the handle bytes (448 B per rank) + the agreement word
    const size_t words = sizeof(all[0]) / 4;
    if (pl.shm_mode) {
        COMM_COUNT(c, allgather, sizeof(all[me]));
        memcpy((void *)pl.shm->handles[me], all[me], sizeof(all[me]));
        if (fail) pl.shm->map_failed = 1u;
        __sync_synchronize();
        int r = peer_barrier_host(c);
        if (r) return r;
        memcpy(all, (const void *)pl.shm->handles, sizeof(all));
    } else {
        d = c->d_comm_scratch;      // allocated with the communicator: a local out-of-memory cannot keep this rank out of the collectives below
        int e = 0;
        if (d) {
            hipMemcpyAsync(d + (size_t)me * words, all[me], sizeof(all[me]), hipMemcpyHostToDevice, c->stream);
            COMM_COUNT(c, allgather, words * 4);
            e = rccl_allgather_u32(c->comm.comm, d + (size_t)me * words, d, words, c->stream);
            hipMemcpyAsync(all, d, sizeof(all[0]) * (size_t)G, hipMemcpyDeviceToHost, c->stream);
        }
        const hipError_t se = hipStreamSynchronize(c->stream);
        if (!d || e != 0 || se != hipSuccess) { hrbf_set_error("peer link: handle all-gather failed"); return HRBF_ERR_COMM; }   // the communicator itself is broken: nothing to agree over
    }
    PeerImages &pi = pl.img;
    memset(&pi, 0, sizeof(pi));
    pi.world = G; pi.me = me;
    for (int g = 0; g < G && !fail; ++g) {
        void *q[PEER_BUFS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        for (int b = 0; b < NB && !fail; ++b) {
            if (g == me) { q[b] = mine[b]; continue; }
            const hipError_t e = hipIpcOpenMemHandle(&q[b], all[g][b], hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { hrbf_set_error("hipIpcOpenMemHandle(rank %d, buffer %d): %s", g, b, hipGetErrorString(e)); fail = 1; break; }
            pl.opened[g][b] = q[b];
        }
        pi.zbuf[g] = (unsigned long long *)q[0]; pi.vertconf[g] = (float4 *)q[1]; pi.normrad[g] = (float4 *)q[2];
        pi.colortime[g] = (float4 *)q[3]; pi.curvmax[g] = (float4 *)q[4]; pi.curvmin[g] = (float4 *)q[5]; pi.clean[g] = (float4 *)q[6]; pi.gid[g] = (uint32_t *)q[7];
    }
    const char *forced = getenv("HRBF_TEST_FAIL_PEER_MAP");   // tests: this rank pretends its mapping failed ("all" or a complete rank number)
    if (forced && (!strcmp(forced, "all") || env_names_rank("HRBF_TEST_FAIL_PEER_MAP", me))) fail = 1;
    uint32_t failed_ranks = 0;
    if (pl.shm_mode) {
        if (fail) { pl.shm->map_failed = 1u; __sync_synchronize(); }
        COMM_COUNT(c, word_allreduce, 4);
        const int r = peer_barrier_host(c);    // nobody touches a peer's memory before everybody has mapped it — or has said it cannot
        if (r) return r;
        failed_ranks = pl.shm->map_failed;
    } else {
        const uint32_t f = (uint32_t)fail;
        uint32_t *w = d + COMM_SCRATCH_VOTE;
        hipMemcpyAsync(w, &f, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
        COMM_COUNT(c, word_allreduce, 4);
        const int e = rccl_allreduce_sum_u32(c->comm.comm, w, 1, c->stream);
        hipMemcpyAsync(&failed_ranks, w, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
        const hipError_t se = hipStreamSynchronize(c->stream);
        if (e != 0 || se != hipSuccess) { hrbf_set_error("peer link: agreement all-reduce failed"); return HRBF_ERR_COMM; }
    }
    if (failed_ranks) {
        for (int g = 0; g < HRBF_PEER_MAX; ++g)
            for (int b = 0; b < PEER_BUFS; ++b)
                if (pl.opened[g][b]) { hipIpcCloseMemHandle(pl.opened[g][b]); pl.opened[g][b] = nullptr; }
        memset(&pi, 0, sizeof(pi));
        *all_mapped = 0;
        return HRBF_OK;
    }
    if (!pl.shm_mode) {      // no allocation after the agreement: the ranks cannot end up on different transports
        pl.d_token = c->d_comm_scratch + COMM_SCRATCH_TOKEN;
        hipMemsetAsync(pl.d_token, 0, sizeof(uint32_t), c->stream);
    }
    pl.enabled = 1;
    *all_mapped = 1;
    return HRBF_OK;
}

static void refresh_count_ub(hrbf_context *c)
{
    // 1-frame-lag read-back of the surfel counts; never blocks in steady state
    if (c->ev_pending && hipEventQuery(c->ev_count) == hipSuccess) {
        c->ev_pending = false;
        for (int k = 0; k < c->nsh; ++k) {
            const int gk = c->shard_first + k;
            const uint32_t ub = c->h_count_pinned[gk] + ((c->hash_mode || gk == c->G - 1) ? c->ub_growth_since : 0u);   // only the last shard grows (hash ownership: all may)
            if (ub < c->sh[k].count_ub) c->sh[k].count_ub = ub;
        }
    }
}
static void request_count(hrbf_context *c)
{
    if (c->ev_pending) return;
    hipMemcpyAsync(c->h_count_pinned, counts_live(c), sizeof(uint32_t) * (size_t)c->G, hipMemcpyDeviceToHost, c->stream);
    hipEventRecord(c->ev_count, c->stream);
    c->ev_pending = true;
    c->ub_growth_since = 0;
}

// ---- collectives of a sharded map.  Real mode (one shard per rank): RCCL on the context's stream, in place.  The
// single-process test mode reduces the private outputs of the virtual shards with local kernels instead (st_indices).
static void shard_allgather_counts(hrbf_context *c, uint32_t *row)
{
    if (c->nsh > 1) COMM_COUNT(c, allgather, 4);      // one process playing all shards: the row is already whole
    if (!c->shard_real) return;
    COMM_COUNT(c, allgather, 4);
    if (c->peer.shm_mode) {   // through the rendezvous segment (double buffered by barrier generation)
        PeerLink &pl = c->peer;
        uint32_t mine = 0;
        hipMemcpyAsync(&mine, row + pl.rank, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
        (void)hipStreamSynchronize(c->stream);      // an error here is reported by the barrier below, which this rank still reaches
        const int slot = (int)((pl.gen + 1u) & 1u);
        pl.shm->counts[slot][pl.rank] = mine;
        __sync_synchronize();
        if (peer_barrier_host(c)) return;
        uint32_t all[HRBF_PEER_MAX];
        for (int g = 0; g < pl.world; ++g) all[g] = pl.shm->counts[slot][g];
        hipMemcpyAsync(row, all, sizeof(uint32_t) * (size_t)pl.world, hipMemcpyHostToDevice, c->stream);
        hipStreamSynchronize(c->stream);   // `all` is a stack buffer
        return;
    }
    if (c->comm.comm) rccl_allgather_u32(c->comm.comm, row + c->comm.rank, row, 1, c->stream);
}
// hash ownership: the smallest global-order id alive, over all shards of all ranks, into c->d_gfirst (device word).  Local
// shards by a one-thread kernel; between ranks the word travels with the counts (shard_allgather_counts).
static void hash_refresh_gfirst(hrbf_context *c, const uint32_t *counts_row)
{
    if (!c->hash_mode) return;
    const uint32_t *gids[HRBF_MAX_SHARDS];
    for (int k = 0; k < c->nsh; ++k) gids[k] = c->sh[k].d_gid;
    launch_gfirst(c->stream, counts_row, c->shard_first, c->nsh, gids, c->d_gfirst, 0);
    if (c->nsh > 1) COMM_COUNT(c, allgather, 4);
    if (!c->shard_real) return;
    COMM_COUNT(c, allgather, 4);
    uint32_t mine = HRBF_NO_SURFEL, all[HRBF_PEER_MAX];
    if (c->peer.shm_mode) {   // through the rendezvous segment, like the counts
        PeerLink &pl = c->peer;
        hipMemcpyAsync(&mine, c->d_gfirst, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
        (void)hipStreamSynchronize(c->stream);      // reported by the barrier below
        const int slot = (int)((pl.gen + 1u) & 1u);
        pl.shm->counts[slot][pl.rank] = mine;
        __sync_synchronize();
        if (peer_barrier_host(c)) return;
        for (int g = 0; g < pl.world; ++g) { all[g] = pl.shm->counts[slot][g]; mine = all[g] < mine ? all[g] : mine; }
        hipMemcpyAsync(c->d_gfirst, &mine, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
        hipStreamSynchronize(c->stream);
    } else if (c->comm.comm) {   // RCCL: all-gather one word per rank into the scratch row, then its minimum — all on the stream, no host round trip
        uint32_t *row = c->x.rec_count;   // HRBF_MAX_SHARDS words, free outside a projection
        hipMemcpyAsync(row + c->comm.rank, c->d_gfirst, sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream);
        rccl_allgather_u32(c->comm.comm, row + c->comm.rank, row, 1, c->stream);
        launch_min_row_u32(c->stream, row, c->comm.world, c->d_gfirst);
    }
}
static void st_init(hrbf_context *c)
{
    if (c->hash_mode) {
        // every shard seeds the surfels whose cell is its own; their place in the seed frame's draw order is their global id
        hipMemsetAsync(counts_live(c), 0, sizeof(uint32_t) * HRBF_MAX_SHARDS, c->stream);
        c->ev_pending = false; c->ub_growth_since = 0;
        for (int k = 0; k < c->nsh; ++k) {
            launch_initialise_hashed(c->stream, c->cam, c->d_pose, c->d_vertex_raw, c->d_normal, c->d_rgb, c->d_curv1, c->d_curv2,
                                     c->d_gradmag, c->prm.use_conf_eval, c->prm.conf_eval_epsilon, c->prm.curv_valid_threshold,
                                     c->d_init_flags, c->d_init_offs, c->d_init_flags2, c->d_init_offs2, c->sh[k].map, c->sh[k].d_gid,
                                     c->cap, counts_live(c) + c->shard_first + k, c->d_gtotal, c->sh[k].d_stats + 7, c->G,
                                     c->shard_first + k, c->hash_inv_cell);
            c->sh[k].count_ub = (uint32_t)c->P < c->cap ? (uint32_t)c->P : c->cap;
        }
        uint32_t total = 0;
        hipMemcpyAsync(&total, c->d_gtotal, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
        hipStreamSynchronize(c->stream);
        c->g_next = total;
        shard_allgather_counts(c, counts_live(c));
        hash_refresh_gfirst(c, counts_live(c));
        launch_odo_first_rgb(c->stream, c->odo, c->d_rgb);
        return;
    }
    // the seed goes to the end of the global order = the last shard; every other shard starts empty
    const int kl = c->G - 1 - c->shard_first;
    if (c->G > 1) hipMemsetAsync(counts_live(c), 0, sizeof(uint32_t) * HRBF_MAX_SHARDS, c->stream);
    for (int k = 0; k < c->nsh; ++k) c->sh[k].count_ub = 0;
    c->ev_pending = false; c->ub_growth_since = 0;   // a count read-back still in flight describes the map that is being replaced
    if (kl >= 0 && kl < c->nsh) {
        launch_initialise(c->stream, c->cam, c->d_pose, c->d_vertex_raw, c->d_normal, c->d_rgb, c->d_curv1, c->d_curv2,
                          c->d_gradmag, c->prm.use_conf_eval, c->prm.conf_eval_epsilon, c->prm.curv_valid_threshold,
                          c->d_init_flags, c->d_init_offs, c->sh[kl].map, c->cap, counts_live(c) + (c->G - 1), c->sh[kl].d_stats + 7);
        c->sh[kl].count_ub = (uint32_t)c->P < c->cap ? (uint32_t)c->P : c->cap;
    }
    shard_allgather_counts(c, counts_live(c));
    launch_odo_first_rgb(c->stream, c->odo, c->d_rgb);
}
// what: which outputs of the projection the next consumer reads (k_resolve); the stage API asks for everything
// classes: the projection also leaves pass A's first decision per surfel in the keep-byte plane (k_project) — only the frame
// path asks for it, where the clean pass follows at once on the same map and pose (st_clean(c, true))
static void st_indices(hrbf_context *c, bool for_clean = true, int what = 7, bool classes = false)
{
    const float maxd = c->prm.max_depth_processed;
    const float cthr = c->prm.confidence_threshold;
#define CLS(k) (classes ? c->sh[k].d_keep_flags : nullptr), cthr, c->sh[k].d_class_word, c->prm.clean_window_multiplier
    float4 *ctex = for_clean ? c->d_clean_tex : nullptr;
    if (for_clean && (what & 4)) { c->clean_thr = c->prm.confidence_threshold; c->clean_time = c->tick; }
    if (c->G == 1 && !c->shard_real) {
        launch_project(c->stream, c->cam, c->d_pose, maxd, c->sh[0].map, shard_ref(c, 0), c->sh[0].count_ub, c->d_zbuf,
                       c->d_submap_active, c->n_submap_active, CLS(0));
        launch_resolve(c->stream, c->cam, c->d_pose, c->sh[0].map, shard_ref(c, 0), c->d_zbuf, c->d_idx, c->d_im_vertconf,
                       c->d_im_colortime, c->d_im_normrad, c->d_im_curvmax, c->d_im_curvmin, ctex, what, 1,
                       c->clean_thr, c->clean_time);
        return;
    }
    // sharded map: every shard projects its own surfels under GLOBAL ids -> min-reduce of the packed keys -> every
    // shard gathers the winners it owns (zeros elsewhere) and packs them as winner records -> the records of the other
    // shards are exchanged (variable length: 4 + 16..80 bytes per HIT pixel instead of dense images) and scattered.
    for (int k = 0; k < c->nsh; ++k) {
        if (c->hash_mode) {   // two-level z-test: private {depth, local index}, then {depth, gid} into the buffer that is reduced
            launch_project(c->stream, c->cam, c->d_pose, maxd, c->sh[k].map, shard_ref(c, k), c->sh[k].count_ub, c->sh[k].d_zpriv,
                           c->d_submap_active, c->n_submap_active, CLS(k));
            launch_keys_global(c->stream, c->sh[k].d_zpriv, c->sh[k].d_gid, c->d_zbuf, c->P, k > 0 ? 1 : 0);
            continue;
        }
        unsigned long long *zb = k == 0 ? c->d_zbuf : c->x.zbuf;
        launch_project(c->stream, c->cam, c->d_pose, maxd, c->sh[k].map, shard_ref(c, k), c->sh[k].count_ub, zb,
                       c->d_submap_active, c->n_submap_active, CLS(k));
#undef CLS
        if (k > 0) launch_zbuf_min_merge(c->stream, c->d_zbuf, c->x.zbuf, c->P);
    }
    if (c->shard_real || c->nsh > 1) COMM_COUNT(c, key_min_reduce, 8 * (size_t)c->P);   // RCCL below; shm: barrier + launch_zbuf_min_peers; virtual: launch_zbuf_min_merge / launch_keys_global above
    if (c->shard_real && c->comm.comm) rccl_allreduce_min_u64(c->comm.comm, c->d_zbuf, (size_t)c->P, c->stream);
    const uint32_t cap = (uint32_t)c->P;
    float4 *ctx_clean = for_clean ? c->d_clean_tex : nullptr;
    if (!c->shard_real) {
        // one process plays all shards in turn: shard 0 writes the images (zeros where it owns nothing), every further
        // shard packs its winners and they are scattered right away — the data flow of the exchange without a wire
        for (int k = 0; k < c->nsh; ++k) {
            uint32_t *cnt = c->x.rec_count + k;
            if (k > 0) hipMemsetAsync(cnt, 0, sizeof(uint32_t), c->stream);
            launch_resolve(c->stream, c->cam, c->d_pose, c->sh[k].map, shard_ref(c, k), c->d_zbuf, c->d_idx, c->d_im_vertconf,
                           c->d_im_colortime, c->d_im_normrad, c->d_im_curvmax, c->d_im_curvmin, ctx_clean, what,
                           k == c->nsh - 1 ? 1 : 0, c->clean_thr, c->clean_time, k > 0 ? cnt : nullptr,
                           k > 0 ? c->x.send_idx : nullptr, c->x.send_f, cap, k == 0 ? 1 : 0, c->sh[k].d_zpriv);
            if (k > 0)
                launch_winner_unpack(c->stream, c->P, c->cam.W, cnt, 0, cap, c->x.send_idx, c->x.send_f, cap, what, c->d_im_vertconf,
                                     c->d_im_colortime, c->d_im_normrad, c->d_im_curvmax, c->d_im_curvmin, ctx_clean);
        }
        return;
    }
    if (c->peer.enabled) {
        // one shard per rank, images peer-mapped: the ranks meet once the keys are reduced and once the owners have written
        PeerLink &pl = c->peer;
        const unsigned long long *zred = c->d_zbuf;
        if (pl.shm_mode) {
            if (peer_barrier_host(c)) return;                              // everybody has projected
            launch_zbuf_min_peers(c->stream, pl.img, c->x.zbuf, c->P);       // min over the ranks' private z-buffers
            zred = c->x.zbuf;
        }   // RCCL transport: the all-reduce above left the reduced keys in d_zbuf
        launch_resolve_scatter(c->stream, c->cam, c->d_pose, c->sh[0].map, shard_ref(c, 0), zred, c->d_idx, pl.img, what, for_clean,
                               c->clean_thr, c->clean_time, c->sh[0].d_zpriv);
        if (peer_meet(c)) return;                                           // every owner has written into every rank's images
        launch_zbuf_reset(c->stream, c->d_zbuf, c->P);                       // re-arm the private z-buffer: nobody reads it any more
        return;
    }
    // one shard per rank
    const int me = c->comm.rank, G = c->G;
    uint32_t *cnt = c->x.rec_count + me;
    hipMemsetAsync(cnt, 0, sizeof(uint32_t), c->stream);
    launch_resolve(c->stream, c->cam, c->d_pose, c->sh[0].map, shard_ref(c, 0), c->d_zbuf, c->d_idx, c->d_im_vertconf,
                   c->d_im_colortime, c->d_im_normrad, c->d_im_curvmax, c->d_im_curvmin, ctx_clean, what, 1, c->clean_thr,
                   c->clean_time, cnt, c->x.send_idx, c->x.send_f, cap, 1, c->sh[0].d_zpriv);
    if (!c->comm.comm || G == 1) return;
    // sizes of the variable-length exchange: all-gather of the record counts, read back (the one host round trip of the pass)
    COMM_COUNT(c, allgather, 4);
    rccl_allgather_u32(c->comm.comm, cnt, c->x.rec_count, 1, c->stream);
    hipMemcpyAsync(c->x.h_counts, c->x.rec_count, sizeof(uint32_t) * (size_t)G, hipMemcpyDeviceToHost, c->stream);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return;
    const uint32_t n_me = c->x.h_counts[me] < cap ? c->x.h_counts[me] : cap;
    int planes[6], np = 0;
    if (what & 1) { planes[np++] = 0; planes[np++] = 1; }
    if (what & 2) { planes[np++] = 2; planes[np++] = 3; planes[np++] = 4; }
    if ((what & 4) && for_clean) planes[np++] = 5;
    g_rccl.GroupStart();
    uint32_t off = 0;
    for (int p = 0; p < G; ++p) {
        if (p == me) continue;
        const uint32_t n_p = c->x.h_counts[p];
        if (off + n_p > cap) break;   // cannot happen: a pixel has one owner, so the counts of all ranks add up to <= P
        if (n_me) { COMM_COUNT(c, send, 4 * (size_t)n_me); g_rccl.Send(c->x.send_idx, n_me, kNcclUint32, p, c->comm.comm, c->stream); }
        if (n_p) { COMM_COUNT(c, recv, 4 * (size_t)n_p); g_rccl.Recv(c->x.recv_idx + off, n_p, kNcclUint32, p, c->comm.comm, c->stream); }
        for (int t = 0; t < np; ++t) {
            if (n_me) { COMM_COUNT(c, send, 16 * (size_t)n_me); g_rccl.Send(c->x.send_f + (size_t)planes[t] * cap, (size_t)n_me * 4, kNcclUint32, p, c->comm.comm, c->stream); }
            if (n_p) { COMM_COUNT(c, recv, 16 * (size_t)n_p); g_rccl.Recv(c->x.recv_f + (size_t)planes[t] * cap + off, (size_t)n_p * 4, kNcclUint32, p, c->comm.comm, c->stream); }
        }
        off += n_p;
    }
    g_rccl.GroupEnd();
    if (off) {
        launch_winner_unpack(c->stream, c->P, c->cam.W, nullptr, 0, off, c->x.recv_idx, c->x.recv_f, cap, what,
                             c->d_im_vertconf, c->d_im_colortime, c->d_im_normrad, c->d_im_curvmax, c->d_im_curvmin, ctx_clean);
    }
}
// whether this frame's fuse pass is bracketed by the ring's events: every record costs the stream ~4.6 us (a barrier
// packet), four per frame plus the statistics copy are 2 % of a frame — a caller that only wants an average can sample
static inline bool ring_frame(const hrbf_context *c)
{
    return c->timing && c->G == 1 && (c->ring_stride <= 1 || c->tick % c->ring_stride == 0);
}
static void st_fuse(hrbf_context *c)
{
    // association is replicated (it reads images only); each shard keeps the merge slots of its own surfels and applies
    // the merges that land on them
    const bool ring = ring_frame(c);
    for (int k = 0; k < c->nsh; ++k)
        launch_fuse(c->stream, c->cam, c->d_pose, c->tick, c->prm.max_depth_processed, c->index_submap, c->d_depth_metric,
                    c->d_normal_pca, c->d_curv1, c->d_curv2, c->d_confidence, c->d_rgb, c->d_idx, c->d_im_vertconf,
                    c->d_im_normrad, c->rec, c->d_rec_flag, c->d_rec_best, c->sh[k].d_slot, c->sh[k].map, shard_ref(c, k),
                    c->sh[k].d_stats, c->prm.curv_valid_threshold, ring ? c->ring_m0[c->ring_head % HRBF_RING] : nullptr,
                    ring ? c->ring_m1[c->ring_head % HRBF_RING] : nullptr, c->sh[k].d_merged_part,
                    RecNormalSrc{c->d_depth_metric_f, c->prm.init_radius_multiplier, c->prm.normal_estimation_pca > 0.0f ? 1 : 0});
    c->fuse_tick = c->tick; c->ring_merge_head = c->ring_head;
}
// hash ownership: ids are never renumbered by the frame path and grow by Q per clean pass (55 000 frames of VGA fill 32 bits).
// When the next pass could run out, every surfel's id becomes its RANK in the global order (the number of ids below it over all
// shards: a sum of lower bounds over the shards' ascending id planes — the peers' planes are IPC-mapped like their images), and
// g_next restarts at the surfel count.  Order, hence every result, is unchanged; only the names in the index image change.
static int read_counts(hrbf_context *c, uint32_t out[HRBF_MAX_SHARDS]);
// One verdict on every rank of a sharded map: > 0 when ANY rank says `fail` (or, on the shared-memory transport, ever said so: the
// segment's flag is sticky), 0 when none does, < 0 when the transport itself is broken.  Every rank must call it, whatever
// happened to it locally — that is the point.  One process playing all shards: the local answer.
static int peer_vote(hrbf_context *c, int fail)
{
    if (!c->shard_real) return fail ? 1 : 0;
    COMM_COUNT(c, word_allreduce, 4);
    if (c->peer.shm_mode) {
        if (fail) { c->peer.shm->failed = 1u; __sync_synchronize(); }
        return peer_barrier_host(c) != HRBF_OK ? 1 : 0;
    }
    if (!c->comm.comm || !c->d_comm_scratch) return -1;
    uint32_t f = fail ? 1u : 0u, any = 0;
    uint32_t *w = c->d_comm_scratch + COMM_SCRATCH_VOTE;
    hipMemcpyAsync(w, &f, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    const int e = rccl_allreduce_sum_u32(c->comm.comm, w, 1, c->stream);
    hipMemcpyAsync(&any, w, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e != 0 || hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    return any ? 1 : 0;
}
// All ranks commit the new ids or none does (round-4 advice).  Phase A is local (counts, every allocation) and ends in a vote: a
// rank that failed there has not touched a collective yet and its peers learn it BEFORE they issue one it could not join (the
// size of the all-gather depends on counts the failing rank may not have).  Phase B (all-gather / rank kernels / copy-back, with
// the two meeting points) ends in a second vote on the stream's and the communicator's health; only then g_next is reset.
// With one shard per rank a failure is final for the map: the ranks could not line their meeting points up again in a retry
// that only some of them need (hrbf_get_status(clear) re-arms a retry for a single process only).
static int hash_renumber(hrbf_context *c)
{
    uint32_t cnt[HRBF_MAX_SHARDS] = {0};
    int rc = HRBF_OK;
    // tests: this rank pretends an allocation failed.  The value must be a complete decimal rank number and the map one shard
    // per rank: "" / "yes" / "0x1" inject nothing, and a single process (whose rank is always 0) is never hit by a stray "0"
    if (c->shard_real && env_names_rank("HRBF_TEST_FAIL_RENUMBER", c->comm.rank)) { hrbf_set_error("hash ownership: id renumbering: injected failure"); rc = HRBF_ERR_DEVICE; }
    if (rc == HRBF_OK && read_counts(c, cnt)) { hrbf_set_error("hash ownership: id renumbering: the counts could not be read"); rc = HRBF_ERR_DEVICE; }
    uint64_t total = 0;
    for (int g = 0; g < c->G; ++g) total += cnt[g];
    const bool mapped = c->shard_real && c->peer.enabled && c->peer.img.gid[c->peer.rank];
    const bool records = c->shard_real && !mapped;            // no peer-mapped id planes: they are all-gathered
    if (records && !c->comm.comm) { hrbf_set_error("hash ownership: id renumbering without a communicator"); return HRBF_ERR_INVALID; }   // the same on every rank
    const uint32_t *ptrs[HRBF_MAX_SHARDS] = {nullptr};
    if (!c->shard_real) { for (int k = 0; k < c->nsh; ++k) ptrs[k] = c->sh[k].d_gid; }
    else if (mapped) { for (int g = 0; g < c->G; ++g) ptrs[g] = c->peer.img.gid[g]; }
    uint32_t *gathered = nullptr, *tmp[HRBF_MAX_SHARDS] = {nullptr};
    uint32_t maxc = 1;      // the planes are padded to the longest (every plane has room for c->cap ids; the padding is never read)
    for (int g = 0; g < c->G; ++g) maxc = cnt[g] > maxc ? cnt[g] : maxc;
    // ---- phase A: everything that can fail locally
    if (rc == HRBF_OK && records && hipMalloc((void **)&gathered, sizeof(uint32_t) * (size_t)maxc * (size_t)c->G) != hipSuccess) {
        gathered = nullptr; hrbf_set_error("hash ownership: id renumbering: out of device memory"); rc = HRBF_ERR_DEVICE;
    }
    for (int k = 0; k < c->nsh && rc == HRBF_OK; ++k) {
        const uint32_t nk = cnt[c->shard_first + k];
        if (hipMalloc((void **)&tmp[k], sizeof(uint32_t) * (size_t)(nk ? nk : 1)) != hipSuccess) {
            tmp[k] = nullptr; hrbf_set_error("hash ownership: id renumbering: out of device memory"); rc = HRBF_ERR_DEVICE;
        }
    }
    auto release = [&]() { for (int k = 0; k < c->nsh; ++k) if (tmp[k]) hipFree(tmp[k]); if (gathered) hipFree(gathered); };
    char why[160]; snprintf(why, sizeof(why), "%s", rc != HRBF_OK ? hrbf_last_error() : "");   // the vote's transport may set its own text
    const int v1 = peer_vote(c, rc != HRBF_OK);
    if (v1 != 0) {
        release();
        if (rc == HRBF_OK) { hrbf_set_error(v1 < 0 ? "hash ownership: id renumbering: the ranks could not vote" : "hash ownership: id renumbering failed on another rank"); rc = HRBF_ERR_COMM; }
        else hrbf_set_error("%s", why);
        return rc;      // nobody has issued a collective of phase B, nobody has changed an id
    }
    // ---- phase B
    if (records) {
        hipMemcpyAsync(gathered + (size_t)c->comm.rank * maxc, c->sh[0].d_gid, sizeof(uint32_t) * (size_t)maxc, hipMemcpyDeviceToDevice, c->stream);
        COMM_COUNT(c, allgather, 4 * (size_t)maxc);
        if (rccl_allgather_u32(c->comm.comm, gathered + (size_t)c->comm.rank * maxc, gathered, maxc, c->stream) != 0) rc = HRBF_ERR_COMM;
        for (int g = 0; g < c->G; ++g) ptrs[g] = gathered + (size_t)g * maxc;
    }
    for (int k = 0; k < c->nsh; ++k)
        launch_gid_rank(c->stream, ptrs, counts_live(c), c->G, c->sh[k].d_gid, cnt[c->shard_first + k], tmp[k]);
    if (mapped && peer_meet(c)) rc = HRBF_ERR_COMM;          // every rank has read every plane
    int committed_copy = 0;
    if (rc == HRBF_OK) {
        for (int k = 0; k < c->nsh; ++k) {
            const uint32_t nk = cnt[c->shard_first + k];
            if (nk) hipMemcpyAsync(c->sh[k].d_gid, tmp[k], sizeof(uint32_t) * (size_t)nk, hipMemcpyDeviceToDevice, c->stream);
        }
        committed_copy = 1;
    }
    if (mapped && peer_meet(c)) rc = HRBF_ERR_COMM;          // nobody reads a half-written plane in the next pass
    if (hipStreamSynchronize(c->stream) != hipSuccess) rc = HRBF_ERR_DEVICE;
    release();
    // the second verdict: a failure in here is a broken stream or communicator — the ids of this map are no longer trustworthy on
    // any rank (committed_copy may differ between them), and every rank says so
    const int v2 = peer_vote(c, rc != HRBF_OK);
    if (v2 != 0 || rc != HRBF_OK) {
        if (rc == HRBF_OK) rc = HRBF_ERR_COMM;
        hrbf_set_error("hash ownership: id renumbering broke down between the ranks (stream or communicator error%s)", committed_copy ? "; ids partly rewritten" : "");
        return rc;
    }
    c->g_next = (uint32_t)total;
    ++c->hash_renumbered;
    return HRBF_OK;
}
static int st_clean(hrbf_context *c, bool have_class = false)
{
    // the clean texels carry the confidence threshold and the time they were resolved with; a caller that changed
    // either since (stage API / named operators) gets a fresh projection instead of a stale test
    if (c->clean_thr != c->prm.confidence_threshold || c->clean_time != c->tick) { st_indices(c, true, 7); have_class = false; }
    if (c->hash_mode && c->g_next > c->g_renumber_at) {
        // ids must never run past g_renumber_at: a pass that cannot renumber does not run (the map stays as the merge left it), the
        // condition is sticky (HRBF_STATUS_ID_SPACE) and every later frame fails at once instead of retrying a host-synchronous,
        // allocating operation per frame (round-3 advice) — until the caller clears the status
        const int rr = c->renumber_failed ? HRBF_ERR_DEVICE : hash_renumber(c);
        if (rr != HRBF_OK) { c->renumber_failed = 1; c->status |= HRBF_STATUS_ID_SPACE; return rr; }
    }
    // (the same decision on every rank: g_next is a pure function of the frames processed)
    const bool ring = ring_frame(c);
    for (int k = 0; k < c->nsh; ++k) {
        const int gk = c->shard_first + k;
        const bool last = c->hash_mode || gk == c->G - 1;   // new surfels are appended at the end of the global order (hash ownership: by every shard, its own)
        MapShard &sh = c->sh[k];
        const int cur = sh.tile_par;
        uint32_t dirty[2] = {sh.tile_dirty[cur], sh.tile_dirty[1 - cur]};
        launch_clean(c->stream, c->cam, c->d_pose, c->prm.max_depth_processed, c->prm.confidence_threshold,
                     c->prm.curv_valid_threshold, c->tick, c->prm.clean_window_multiplier,
                     (c->map_dirty || c->fuse_tick != c->tick) ? 1 : 0, sh.map, c->rec, c->d_rec_flag, counts_live(c) + gk,
                     counts_next(c) + gk, sh.count_ub, sh.d_stats, c->cap, c->d_clean_tex, sh.d_keep_flags,
                     sh.d_tile_count[cur], sh.d_tile_count[1 - cur], dirty, sh.d_tile_done, ++sh.epoch,
                     c->max_tiles, ring ? c->ring_e0[c->ring_head % HRBF_RING] : nullptr,
                     ring ? c->ring_e1[c->ring_head % HRBF_RING] : nullptr, c->d_submap_active, c->n_submap_active,
                     last ? c->Q : 0, (c->shard_real || k == c->nsh - 1) ? 1 : 0,
                     ring ? c->d_stats_ring + (size_t)(c->ring_head % HRBF_RING) * 8 : nullptr, sh.d_merged_part,
                     c->hash_mode ? sh.d_gid : nullptr, c->g_next, c->G, gk, c->hash_inv_cell, have_class ? 1 : 0, sh.d_class_word);
        sh.tile_dirty[cur] = dirty[0]; sh.tile_dirty[1 - cur] = dirty[1]; sh.tile_par = 1 - cur;
        if (last) {
            const uint64_t ub = (uint64_t)sh.count_ub + (uint64_t)c->Q;
            sh.count_ub = ub > c->cap ? c->cap : (uint32_t)ub;
        }
    }
    shard_allgather_counts(c, counts_next(c));
    if (c->hash_mode) { c->g_next += (uint32_t)c->Q; hash_refresh_gfirst(c, counts_next(c)); }
    if (ring) {   // the statistics are parked in the ring slot behind the pass
        if (c->ring_merge_head != c->ring_head || c->fuse_tick != c->tick) {   // no merge this frame: an empty F2 interval
            hipEventRecord(c->ring_m0[c->ring_head % HRBF_RING], c->stream);
            hipEventRecord(c->ring_m1[c->ring_head % HRBF_RING], c->stream);
        }
        c->ring_head++;
        if (c->ring_valid < HRBF_RING) c->ring_valid++;
    }
    c->target = 1 - c->target;
    c->map_dirty = 0;
    c->ub_growth_since += (uint32_t)c->Q;
    return HRBF_OK;
}
// with_fill_in: the frame path — the ray-cast kernel fills in from the live frame as well (FILL_* images), so what is left
// of FillIn is the end-of-frame bookkeeping (st_end_of_frame)
static void st_predict(hrbf_context *c, bool with_fill_in = false)
{
    c->fill_flag_fresh = 0;
    FillIn f;
    f.thr = c->prm.curv_valid_threshold; f.frame_to_frame_rgb = c->prm.frame_to_frame_rgb;
    f.vertex_filtered = c->d_vertex_filtered; f.normal = c->d_normal; f.curv1 = c->d_curv1; f.curv2 = c->d_curv2;
    f.confidence = c->d_confidence; f.rgb = c->d_rgb;
    f.fi_vertex = c->d_fi_vertex; f.fi_normal = c->d_fi_normal; f.fi_curv1 = c->d_fi_curv1; f.fi_curv2 = c->d_fi_curv2;
    f.fi_icpw = c->d_fi_icpw; f.fi_image = c->d_fi_image;
    launch_predict_hrbf(c->stream, c->cam, c->d_im_vertconf, c->d_im_normrad, c->d_im_colortime, c->d_im_curvmax,
                        c->d_im_curvmin, (int)c->prm.predict_window_multiplier, c->prm.predict_min_neighbors,
                        c->prm.predict_max_neighbors, c->prm.predict_conf_threshold, c->prm.icp_curv_weight_lambda,
                        c->d_pr_image, c->d_pr_vertex, c->d_pr_normal, c->d_pr_curv1, c->d_pr_curv2, c->d_pr_time,
                        c->d_pr_icpw, with_fill_in ? &f : nullptr);
}
static void st_end_of_frame(hrbf_context *c, bool log_frame)
{
    launch_end_of_frame(c->stream, c->cam, c->d_pr_vertex, c->prm.dense_enough_thresh, c->d_pose,
                        log_frame ? c->d_pose_log_view : nullptr, c->frames_enqueued);
    c->fill_flag_fresh = 1;
}
static void st_fillin(hrbf_context *c, bool end_of_frame = false, bool log_frame = false)
{
    launch_fillin(c->stream, c->P, c->prm.curv_valid_threshold, c->prm.icp_curv_weight_lambda, c->prm.frame_to_frame_rgb,
                  c->d_pr_vertex, c->d_pr_normal, c->d_pr_curv1, c->d_pr_curv2, c->d_pr_icpw, c->d_pr_image,
                  c->d_vertex_filtered, c->d_normal, c->d_curv1, c->d_curv2, c->d_confidence, c->d_rgb, c->d_fi_vertex,
                  c->d_fi_normal, c->d_fi_curv1, c->d_fi_curv2, c->d_fi_icpw, c->d_fi_image, c->cam,
                  c->prm.dense_enough_thresh, end_of_frame ? c->d_pose : nullptr, log_frame ? c->d_pose_log_view : nullptr,
                  c->frames_enqueued);
    c->fill_flag_fresh = end_of_frame ? 1 : 0;
}
static void st_odometry(hrbf_context *c, float weight_multiplier = -1.0f)
{
    if (!c->fill_flag_fresh)
        launch_should_fill_in(c->stream, c->cam, c->d_pr_vertex, c->prm.dense_enough_thresh, &c->d_pose->should_fill_in);
    OdoSources src = make_sources(c);
    OdoConfig cfg = make_cfg(c);
    const bool sharded = (c->comm.allreduce_i64 != nullptr || c->comm.virtual_world > 1) && !c->rows_replicated;
    launch_odometry(c->stream, c->odo, src, cfg, c->d_pose, sharded ? &c->comm : nullptr, weight_multiplier, c->level0_done);
    if (c->ar_failed) { c->status |= HRBF_STATUS_COLLECTIVE; c->ar_failed = 0; }   // either transport: the solve ran on unreduced sums
    c->level0_done = 0;
}

#define TIMER(i) do { if (c->timing & 1) hipEventRecord(c->ev[i], c->stream); } while (0)

static int process_frame_resident(hrbf_context *c, float wmul)
{
    hipSetDevice(c->device);
    c->cstats.frames += 1;
    if (c->renumber_failed) { hrbf_set_error("hash ownership: the id space is exhausted and renumbering failed (HRBF_STATUS_ID_SPACE)%s", c->shard_real ? "; final for a map shared by ranks" : "; clear the status to retry"); return HRBF_ERR_DEVICE; }
    int frame_rc = HRBF_OK;
    char frame_err[200] = "";
    refresh_count_ub(c);
    TIMER(0);
    st_filter(c); st_vnr(c);
    st_curv(c, c->tick > 1 && !c->prm.load_trajectory && c->fill_flag_fresh && !c->fit_in_frame);
    if (c->fit_in_frame)   // EXTENSION, off by default (hrbf_set_hrbf_fit): PRINCIPAL_CURV1 / 2 of the live frame from the true Hermite-RBF fit
    {   // default ridge 0.1: an exact interpolant of noisy normals amplifies the noise.  The frame's results are then NOT the reference's: sticky status bit
        launch_hrbf_fit(c->stream, c->cam, c->d_vertex_filtered, c->d_normal, c->fit_window, c->fit_support, c->fit_ridge, c->fit_jump, c->d_curv1, c->d_curv2, c->d_fit_normal);
        c->status |= HRBF_STATUS_EXTENSION;
    }
    TIMER(1);
    if (c->tick == 1) {
        st_init(c);
        TIMER(2); TIMER(3); TIMER(4); TIMER(5); TIMER(6);
    } else {
        const bool fused_weighting = !c->prm.load_trajectory && wmul >= 0.0f;   // the last solve also sets the weighting
        if (!c->prm.load_trajectory) st_odometry(c, fused_weighting ? wmul : -1.0f);
        if (!fused_weighting) launch_frame_epilogue(c->stream, c->d_pose, wmul, 1);
        TIMER(2);
        st_conf(c);
        if (!c->prm.rgb_only) {
            st_indices(c, false, 1);     // association reads the index, vertex/conf and normal/radius images
            TIMER(3);
            st_fuse(c);
            TIMER(4);
            st_indices(c, true, 4, true);      // the clean test reads the packed texels only; + pass A's classes
            TIMER(5);
            frame_rc = st_clean(c, true);   // nothing between the projection and the pass: the classes describe this map and pose
            if (frame_rc) snprintf(frame_err, sizeof(frame_err), "%s", hrbf_last_error());   // later stages (peer barriers) may set their own text
            TIMER(6);
        } else { TIMER(3); TIMER(4); TIMER(5); TIMER(6); }
    }
    st_indices(c, false, 3);             // prediction + the images a caller can fetch
    TIMER(7);
    st_predict(c, true);        // ray cast + fill-in from the live frame
    TIMER(8);
    st_end_of_frame(c, true);   // shouldFillIn of the next frame + lastPose <- currPose + the pose log entry
    TIMER(9);
    request_count(c);
    c->tick++; c->frames_enqueued++;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { hrbf_set_error("launch: %s", hipGetErrorString(e)); return HRBF_ERR_DEVICE; }
    if (frame_rc) hrbf_set_error("%s", frame_err);
    return frame_rc;   // the frame ran, but its clean pass did not (id renumbering failed): the caller must not go on
}

extern "C" int hrbf_upload_frame(hrbf_handle c, const uint8_t *rgb, const uint16_t *depth)
{
    if (!c || !rgb || !depth) { hrbf_set_error("null argument"); return HRBF_ERR_INVALID; }
    hipSetDevice(c->device);
    HIP_CHECK(hipMemcpyAsync(c->d_rgb, rgb, (size_t)c->P * 3, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(c->d_depth, depth, (size_t)c->P * 2, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));   // host buffers are borrowed only for the call
    c->inputs_lent = false;
    return HRBF_OK;
}

extern "C" int hrbf_process_frame(hrbf_handle c, const uint8_t *rgb, const uint16_t *depth, int64_t ts, float wmul)
{
    (void)ts;
    if (!c || !rgb || !depth) { hrbf_set_error("null argument"); return HRBF_ERR_INVALID; }
    hipSetDevice(c->device);
    // The caller's buffers are borrowed for the call only: they are copied into a pinned staging slot (ring of 3,
    // allocated on first use) and uploaded from there asynchronously, so the call never waits for the GPU unless the
    // host runs three frames ahead.
    const size_t nrgb = (size_t)c->P * 3, ndep = (size_t)c->P * 2;
    const int slot = (int)(c->stage_head % 3u);
    if (!c->h_stage[slot]) {
        HIP_CHECK(hipHostMalloc((void **)&c->h_stage[slot], nrgb + ndep + 16, hipHostMallocMapped));
        HIP_CHECK(hipEventCreateWithFlags(&c->ev_stage[slot], hipEventDisableTiming));
    }
    if (c->stage_used[slot]) HIP_CHECK(hipEventSynchronize(c->ev_stage[slot]));
    memcpy(c->h_stage[slot], rgb, nrgb);
    memcpy(c->h_stage[slot] + ((nrgb + 15) & ~(size_t)15), depth, ndep);
    // uploaded by a kernel that reads the pinned slot over PCIe: no copy-engine hand-over on the compute stream
    uint8_t *dev_view = nullptr;
    HIP_CHECK(hipHostGetDevicePointer((void **)&dev_view, c->h_stage[slot], 0));
    const size_t dep_off = (nrgb + 15) & ~(size_t)15;   // keeps the depth half 16-byte aligned
    // only the depth image heads the frame; the RGB image is copied by filler workgroups of the first kernel (st_filter)
    launch_copy_inputs(c->stream, dev_view, 0, dev_view + dep_off, ndep, c->d_rgb, (uint8_t *)c->d_depth);
    c->ride_rgb_src = dev_view;
    c->inputs_lent = false;
    const int r = process_frame_resident(c, wmul);
    HIP_CHECK(hipEventRecord(c->ev_stage[slot], c->stream));   // the slot is free once its readers are through
    c->stage_used[slot] = true;
    c->stage_head++;
    return r;
}

extern "C" int hrbf_process_frame_device(hrbf_handle c, const void *d_rgb, const void *d_depth, int64_t ts, float wmul)
{
    (void)ts;
    if (!c || !d_rgb || !d_depth) { hrbf_set_error("null argument"); return HRBF_ERR_INVALID; }
    hipSetDevice(c->device);
    // no staging copy: every launch of this frame reads the caller's buffers directly (see the header for the
    // lifetime contract); the context's own input buffers serve the host-pointer entry and the stage API
    uint8_t *own_rgb = c->d_rgb; uint16_t *own_depth = c->d_depth;
    c->d_rgb = (uint8_t *)d_rgb; c->d_depth = (uint16_t *)d_depth;
    const int r = process_frame_resident(c, wmul);
    c->d_rgb = own_rgb; c->d_depth = own_depth;
    c->inputs_lent = true;   // the stage seams that read the raw frame would find an older one in the context's own buffers
    return r;
}

extern "C" int hrbf_bootstrap(hrbf_handle c, const uint8_t *rgb, const uint16_t *depth)
{
    int r = hrbf_upload_frame(c, rgb, depth);
    if (r) return r;
    st_filter(c); st_vnr(c); st_curv(c);
    launch_odo_first_rgb(c->stream, c->odo, c->d_rgb);
    st_conf(c);
    st_indices(c); st_predict(c); st_fillin(c, true);
    c->tick = 2;
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return HRBF_OK;
}

extern "C" int hrbf_synchronize(hrbf_handle c)
{
    if (!c) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return HRBF_OK;
}

extern "C" int hrbf_run_stage(hrbf_handle c, int stage)
{
    if (!c) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    if (c->inputs_lent && (stage == HRBF_STAGE_FILTER_DEPTH || stage == HRBF_STAGE_METRICISE || stage == HRBF_STAGE_INITIALISE ||
                           stage == HRBF_STAGE_FUSE || stage == HRBF_STAGE_FILLIN)) {
        // found by tests/gpu_fuzz_api.py: the stage silently ran on the raw images of the last HOST-pointer frame
        hrbf_set_error("run_stage: the last frame read the caller's device buffers (hrbf_process_frame_device), which the context does not keep; "
                       "hrbf_upload_frame() the frame this stage is to run on");
        return HRBF_ERR_INVALID;
    }
    switch (stage) {
        case HRBF_STAGE_FILTER_DEPTH: st_filter(c); break;    // also writes both metric images
        case HRBF_STAGE_METRICISE: st_filter(c); break;
        case HRBF_STAGE_VERTEX_NORMAL_RADIUS: st_vnr(c); break;
        case HRBF_STAGE_CURVATURE: st_curv(c); break;
        case HRBF_STAGE_CONFIDENCE: st_conf(c); break;
        case HRBF_STAGE_INITIALISE: st_init(c); break;
        case HRBF_STAGE_PREDICT_INDICES: st_indices(c); break;
        case HRBF_STAGE_FUSE: st_fuse(c); break;
        case HRBF_STAGE_CLEAN: { const int r = st_clean(c); if (r) return r; } break;
        case HRBF_STAGE_PREDICT_HRBF: st_predict(c); break;
        case HRBF_STAGE_FILLIN: st_fillin(c); break;
        case HRBF_STAGE_ODOMETRY: st_odometry(c); break;
        default: hrbf_set_error("unknown stage %d", stage); return HRBF_ERR_INVALID;
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));
    HIP_CHECK(hipGetLastError());
    return HRBF_OK;
}

// ------------------------------------------------------------------------------------------ getters / setters
extern "C" int hrbf_get_pose(hrbf_handle c, float out[16])
{
    if (!c || !out) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    Rigid r;
    HIP_CHECK(hipMemcpyAsync(&r, &c->d_pose->pose, sizeof(Rigid), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 3; ++i) { for (int k = 0; k < 3; ++k) out[k * 4 + i] = r.r[i * 3 + k]; out[12 + i] = r.t[i]; out[i * 4 + 3] = 0.0f; }
    out[15] = 1.0f;
    return HRBF_OK;
}
// ---- trajectory without blocking: the pose of every processed frame sits in a pinned ring the device appends to
extern "C" uint32_t hrbf_frames_enqueued(hrbf_handle c) { return c ? c->frames_enqueued : 0u; }
extern "C" uint32_t hrbf_frames_completed(hrbf_handle c)
{
    if (!c || !c->h_pose_log) return 0u;
    return __atomic_load_n(&c->h_pose_log->completed, __ATOMIC_ACQUIRE);
}
extern "C" int hrbf_get_pose_log(hrbf_handle c, uint32_t first_frame, uint32_t count, float *out16, int wait)
{
    if (!c || (count && !out16)) return HRBF_ERR_INVALID;
    if (wait) { hipSetDevice(c->device); HIP_CHECK(hipStreamSynchronize(c->stream)); }
    const uint32_t done = hrbf_frames_completed(c);
    if (first_frame >= done) return 0;
    if (done - first_frame > POSE_LOG_CAP || done > first_frame + POSE_LOG_CAP) {
        const uint32_t oldest = done > POSE_LOG_CAP ? done - POSE_LOG_CAP : 0u;
        if (first_frame < oldest) { hrbf_set_error("pose log: frame %u was overwritten (ring of %u)", first_frame, POSE_LOG_CAP); return HRBF_ERR_INVALID; }
    }
    uint32_t n = done - first_frame; if (n > count) n = count;
    for (uint32_t k = 0; k < n; ++k) {
        const volatile PoseRecord *rec = &c->h_pose_log->poses[(first_frame + k) % POSE_LOG_CAP];
        // the tag is written after the pose; `completed` after the tag.  A record whose tag has not landed although
        // `completed` says so would mean the device's posted writes were reordered: wait for it (bounded), never return junk
        int spins = 0;
        while (__atomic_load_n(&rec->tag, __ATOMIC_ACQUIRE) != first_frame + k + 1u) {
            if (++spins > 1000000) { hrbf_set_error("pose log: record %u never landed", first_frame + k); return HRBF_ERR_DEVICE; }
        }
        Rigid r;
        for (int i = 0; i < 9; ++i) r.r[i] = rec->pose.r[i];
        for (int i = 0; i < 3; ++i) r.t[i] = rec->pose.t[i];
        float *out = out16 + (size_t)k * 16;
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) out[j * 4 + i] = r.r[i * 3 + j]; out[12 + i] = r.t[i]; out[i * 4 + 3] = 0.0f; }
        out[15] = 1.0f;
    }
    return (int)n;
}

extern "C" int hrbf_set_pose(hrbf_handle c, const float in[16])
{
    if (!c || !in) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    launch_pose_set(c->stream, c->d_pose, in, c->tick == 1 ? 1 : 0);
    return HRBF_OK;
}
extern "C" int hrbf_get_tick(hrbf_handle c) { return c ? c->tick : -1; }
extern "C" int hrbf_set_tick(hrbf_handle c, int t) { if (!c) return HRBF_ERR_INVALID; c->tick = t; return HRBF_OK; }
extern "C" int hrbf_set_weighting(hrbf_handle c, float w)
{
    if (!c) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    HIP_CHECK(hipMemcpyAsync(&c->d_pose->weighting, &w, sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return HRBF_OK;
}
extern "C" int hrbf_last_weighting(hrbf_handle c, float *w)
{
    if (!c || !w) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    HIP_CHECK(hipMemcpyAsync(w, &c->d_pose->weighting, sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return HRBF_OK;
}
static int read_counts(hrbf_context *c, uint32_t out[HRBF_MAX_SHARDS])
{
    if (hipMemcpyAsync(out, counts_live(c), sizeof(uint32_t) * HRBF_MAX_SHARDS, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
    return hipStreamSynchronize(c->stream) == hipSuccess ? 0 : -1;
}
// the GLOBAL count (all shards; on a rank of a sharded map the other ranks' counts are the all-gathered ones)
extern "C" uint32_t hrbf_surfel_count(hrbf_handle c)
{
    if (!c) return 0;
    hipSetDevice(c->device);
    uint32_t n[HRBF_MAX_SHARDS];
    if (read_counts(c, n)) return 0;
    uint64_t tot = 0;
    for (int g = 0; g < c->G; ++g) tot += n[g];
    for (int k = 0; k < c->nsh; ++k)   // exact knowledge tightens the launch bounds
        if (n[c->shard_first + k] < c->sh[k].count_ub) c->sh[k].count_ub = n[c->shard_first + k];
    return (uint32_t)tot;
}
extern "C" int hrbf_last_icp(hrbf_handle c, float *err, float *cnt)
{
    if (!c || !err || !cnt) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    float v[2];
    HIP_CHECK(hipMemcpyAsync(v, &c->d_pose->last_icp_error, sizeof(float) * 2, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    *err = v[0]; *cnt = v[1];
    return HRBF_OK;
}

// AoS <-> SoA transposes for the map boundary
__global__ void k_map_to_aos(MapPlanes m, uint32_t n, float4 *__restrict__ out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[(size_t)i * 5] = m.p0[i]; out[(size_t)i * 5 + 1] = m.p1[i]; out[(size_t)i * 5 + 2] = m.p2[i];
    out[(size_t)i * 5 + 3] = m.p3[i]; out[(size_t)i * 5 + 4] = m.p4[i];
}
__global__ void k_map_from_aos(MapPlanes m, uint32_t n, const float4 *__restrict__ in)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    m.p0[i] = in[(size_t)i * 5]; m.p1[i] = in[(size_t)i * 5 + 1]; m.p2[i] = in[(size_t)i * 5 + 2];
    m.p3[i] = in[(size_t)i * 5 + 3]; m.p4[i] = in[(size_t)i * 5 + 4];
}

// The local shards in global order: the whole map for a single-GPU context and in the single-process sharded mode;
// a rank of a sharded map returns its own range (its size: hrbf_local_surfel_count).
extern "C" int hrbf_comm_stats(hrbf_handle c, hrbf_comm_counters *out, int reset)
{
    if (!c || !out) return HRBF_ERR_INVALID;
    *out = c->cstats;
    out->limb_allreduce = c->ar_count[0]; out->limb_allreduce_bytes = c->ar_count[1];
    out->transport = HRBF_TRANSPORT_NONE; out->world = 1; out->rank = 0;
    if (c->comm.comm) {
        // the library's own communicator, asked directly: what RCCL says, not what hrbf_comm_init was told
        out->transport = HRBF_TRANSPORT_RCCL; out->world = -1; out->rank = -1;
        int n = 0, r = 0;
        if (g_rccl.CommCount && g_rccl.CommCount(c->comm.comm, &n) == 0) out->world = n;
        if (g_rccl.CommUserRank && g_rccl.CommUserRank(c->comm.comm, &r) == 0) out->rank = r;
    } else if (c->peer.shm_mode) {
        out->transport = HRBF_TRANSPORT_SHM; out->world = c->peer.world; out->rank = c->peer.rank;
    } else if (c->comm.virtual_world > 1 || c->nsh > 1) {
        out->transport = HRBF_TRANSPORT_VIRTUAL; out->world = c->comm.virtual_world > 1 ? c->comm.virtual_world : c->nsh; out->rank = 0;
    }
    if (reset) { memset(&c->cstats, 0, sizeof(c->cstats)); c->ar_count[0] = c->ar_count[1] = 0; }
    return HRBF_OK;
}
extern "C" uint32_t hrbf_local_surfel_count(hrbf_handle c)
{
    if (!c) return 0;
    hipSetDevice(c->device);
    uint32_t n[HRBF_MAX_SHARDS];
    if (read_counts(c, n)) return 0;
    uint64_t tot = 0;
    for (int k = 0; k < c->nsh; ++k) tot += n[c->shard_first + k];
    return (uint32_t)tot;
}
extern "C" int hrbf_download_map(hrbf_handle c, float *out, size_t cap_surfels)
{
    if (!c || !out) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    uint32_t cnt[HRBF_MAX_SHARDS];
    if (read_counts(c, cnt)) { hrbf_set_error("download_map: count read-back failed"); return HRBF_ERR_DEVICE; }
    size_t n = 0;
    for (int k = 0; k < c->nsh; ++k) n += cnt[c->shard_first + k];
    if (cap_surfels < n) { hrbf_set_error("download_map: buffer holds %zu surfels, map has %zu", cap_surfels, n); return HRBF_ERR_CAPACITY; }
    if (n == 0) return HRBF_OK;
    if (c->hash_mode && c->nsh > 1) {
        // the local shards merged by global-order id: the order of the single map (one process playing all shards: the whole map)
        std::vector<std::vector<float>> rows(c->nsh); std::vector<std::vector<uint32_t>> ids(c->nsh);
        for (int k = 0; k < c->nsh; ++k) {
            const uint32_t nk = cnt[c->shard_first + k];
            rows[k].resize((size_t)nk * 20); ids[k].resize(nk);
            if (!nk) continue;
            float4 *tmp = nullptr;
            HIP_CHECK(hipMalloc((void **)&tmp, sizeof(float4) * 5 * (size_t)nk));
            hipLaunchKernelGGL(k_map_to_aos, dim3((nk + 255) / 256), dim3(256), 0, c->stream, c->sh[k].map, nk, tmp);
            hipError_t e = hipMemcpyAsync(rows[k].data(), tmp, sizeof(float4) * 5 * (size_t)nk, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(ids[k].data(), c->sh[k].d_gid, sizeof(uint32_t) * (size_t)nk, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            hipFree(tmp);
            if (e != hipSuccess) { hrbf_set_error("download_map: %s", hipGetErrorString(e)); return HRBF_ERR_DEVICE; }
        }
        std::vector<size_t> at(c->nsh, 0);
        for (size_t o = 0; o < n; ++o) {
            int best = -1;
            for (int k = 0; k < c->nsh; ++k)
                if (at[k] < ids[k].size() && (best < 0 || ids[k][at[k]] < ids[best][at[best]])) best = k;
            memcpy(out + o * 20, &rows[best][at[best] * 20], 80);
            ++at[best];
        }
        return HRBF_OK;
    }
    float4 *tmp = nullptr;
    HIP_CHECK(hipMalloc((void **)&tmp, sizeof(float4) * 5 * n));
    size_t at = 0;
    for (int k = 0; k < c->nsh; ++k) {
        const uint32_t nk = cnt[c->shard_first + k];
        if (nk) hipLaunchKernelGGL(k_map_to_aos, dim3((nk + 255) / 256), dim3(256), 0, c->stream, c->sh[k].map, nk, tmp + at * 5);
        at += nk;
    }
    hipError_t e = hipMemcpyAsync(out, tmp, sizeof(float4) * 5 * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(tmp);
    if (e != hipSuccess) { hrbf_set_error("download_map: %s", hipGetErrorString(e)); return HRBF_ERR_DEVICE; }
    return HRBF_OK;
}
// hash ownership: the global-order ids of the local shards' surfels, in the order hrbf_download_map returns them for ONE local
// shard (a rank of a real sharded map) — the caller merges the ranks' maps by id to obtain the order of the single map
extern "C" int hrbf_download_gids(hrbf_handle c, uint32_t *out, size_t cap_surfels)
{
    if (!c || !out) return HRBF_ERR_INVALID;
    if (!c->hash_mode || c->nsh != 1) { hrbf_set_error("download_gids: a rank of a hash-owned map only"); return HRBF_ERR_INVALID; }
    hipSetDevice(c->device);
    uint32_t cnt[HRBF_MAX_SHARDS];
    if (read_counts(c, cnt)) { hrbf_set_error("download_gids: count read-back failed"); return HRBF_ERR_DEVICE; }
    const size_t n = cnt[c->shard_first];
    if (cap_surfels < n) return HRBF_ERR_CAPACITY;
    if (n) HIP_CHECK(hipMemcpyAsync(out, c->sh[0].d_gid, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return HRBF_OK;
}
// how a sharded map's ranks exchange the index map: 0 not sharded over ranks, 1 peer-mapped images (owner-side scatter),
// 2 packed records because HRBF_SHARD_EXCHANGE=records asked for them, 3 packed records because the ranks AGREED that somebody
// cannot map its peers (another node / IPC namespace)
extern "C" int hrbf_shard_exchange_mode(hrbf_handle c)
{
    if (!c) return HRBF_ERR_INVALID;
    if (!c->shard_real) return 0;
    return c->peer.enabled ? 1 : (c->peer_fallback ? 3 : 2);
}
// hash ownership: how often the ids were renumbered (hash_renumber: when the next pass could exhaust 32 bits; HRBF_HASH_RENUMBER_AT)
// extension: true Hermite-RBF fit per pixel on the matrix core (k_fit.hip); see include/hrbf_mi355.h
static int fit_alloc(hrbf_context *c)
{
    if (c->d_fit_curv1) return HRBF_OK;
    float4 **q[3] = {&c->d_fit_curv1, &c->d_fit_curv2, &c->d_fit_normal};
    for (int i = 0; i < 3; ++i)
        if (hipMalloc((void **)q[i], sizeof(float4) * (size_t)c->P) != hipSuccess) { *q[i] = nullptr; hrbf_set_error("hrbf fit: out of device memory"); return HRBF_ERR_DEVICE; }
    return HRBF_OK;
}
extern "C" int hrbf_set_hrbf_fit(hrbf_handle c, int enable)
{
    if (!c) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    if (enable) { const int r = fit_alloc(c); if (r) return r; }
    c->fit_in_frame = enable ? 1 : 0;
    if (c->fit_window == 0) { c->fit_window = 2; c->fit_support = 1.25f; c->fit_ridge = 0.1f; c->fit_jump = 3.0f; }
    return HRBF_OK;
}
extern "C" int hrbf_set_hrbf_fit_params(hrbf_handle c, int window, float support, float ridge, float jump)
{
    if (!c || window < 1 || window > 2 || !(support > 1.0f) || !(ridge >= 0.0f) || !(jump > 0.0f)) { hrbf_set_error("hrbf_set_hrbf_fit_params: window 1..2, support > 1, ridge >= 0, jump > 0"); return HRBF_ERR_INVALID; }
    c->fit_window = window; c->fit_support = support; c->fit_ridge = ridge; c->fit_jump = jump;
    return HRBF_OK;
}
extern "C" int hrbf_get_hrbf_fit(hrbf_handle c, int *enabled, int *window, float *support, float *ridge, float *jump)
{
    if (!c) return HRBF_ERR_INVALID;
    if (enabled) *enabled = c->fit_in_frame;
    if (window) *window = c->fit_window ? c->fit_window : 2;
    if (support) *support = c->fit_window ? c->fit_support : 1.25f;
    if (ridge) *ridge = c->fit_window ? c->fit_ridge : 0.1f;
    if (jump) *jump = c->fit_window ? c->fit_jump : 3.0f;
    return HRBF_OK;
}
extern "C" int hrbf_fit_curvature(hrbf_handle c, int window, float support, float ridge, float jump, float *ms)
{
    if (!c || window < 1 || window > 2 || !(support > 1.0f) || !(ridge >= 0.0f) || !(jump > 0.0f)) { hrbf_set_error("fit_curvature: window 1..2, support > 1, ridge >= 0, jump > 0"); return HRBF_ERR_INVALID; }
    hipSetDevice(c->device);
    { const int r = fit_alloc(c); if (r) return r; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ms) { HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); hipEventRecord(e0, c->stream); }
    launch_hrbf_fit(c->stream, c->cam, c->d_vertex_filtered, c->d_normal, window, support, ridge, jump, c->d_fit_curv1, c->d_fit_curv2, c->d_fit_normal);
    if (ms) {
        hipEventRecord(e1, c->stream);
        const hipError_t e = hipEventSynchronize(e1);
        if (e == hipSuccess) hipEventElapsedTime(ms, e0, e1);
        hipEventDestroy(e0); hipEventDestroy(e1);
        HIP_CHECK(e);
    }
    HIP_CHECK(hipGetLastError());
    return HRBF_OK;
}
extern "C" int hrbf_gn_graph_captures(hrbf_handle c) { return c ? (int)c->odo.gn_graph_captures : HRBF_ERR_INVALID; }
extern "C" int hrbf_hash_renumber_count(hrbf_handle c) { return c ? (int)c->hash_renumbered : HRBF_ERR_INVALID; }
// the shard a surfel at (x, y, z) is inserted into under hash ownership (host code: no device needed)
extern "C" int hrbf_hash_owner(float x, float y, float z, float cell_metres, int n_shards)
{
    if (!(cell_metres > 0.0f) || n_shards < 1) return HRBF_ERR_INVALID;
    return (int)hash_owner(x, y, z, 1.0f / cell_metres, n_shards);
}
// the live surfel counts of all G shards (0 for k >= G); returns the partition: 0 one map, 1 contiguous ranges, 2 spatial hash
extern "C" int hrbf_shard_counts(hrbf_handle c, uint32_t out[8])
{
    if (!c || !out) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    uint32_t cnt[HRBF_MAX_SHARDS];
    if (read_counts(c, cnt)) return HRBF_ERR_DEVICE;
    for (int k = 0; k < 8; ++k) out[k] = k < c->G ? cnt[k] : 0u;
    return c->G <= 1 ? 0 : (c->hash_mode ? 2 : 1);
}
// equal split of n items over G shards: shard g gets [lo, hi)
static void shard_slice(size_t n, int G, int g, size_t *lo, size_t *hi)
{
    const size_t base = n / (size_t)G, rem = n % (size_t)G;
    *lo = (size_t)g * base + ((size_t)g < rem ? (size_t)g : rem);
    *hi = *lo + base + ((size_t)g < rem ? 1 : 0);
}
// `in` is always the WHOLE map (n surfels in global order); a sharded context keeps the slices of its local shards
extern "C" int hrbf_upload_map(hrbf_handle c, const float *in, size_t n)
{
    if (!c || (!in && n)) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    uint32_t cnt[HRBF_MAX_SHARDS] = {0};
    if (c->hash_mode) {
        // every surfel goes to the shard its cell hashes to, keeping the order of `in`; its row number is its global-order id
        std::vector<uint8_t> owner(n);
        for (size_t i = 0; i < n; ++i) {
            owner[i] = (uint8_t)hash_owner(in[i * 20], in[i * 20 + 1], in[i * 20 + 2], c->hash_inv_cell, c->G);
            ++cnt[owner[i]];
        }
        for (int g = 0; g < c->G; ++g)
            if (cnt[g] > c->cap) { hrbf_set_error("upload_map: %u surfels of shard %d exceed the shard capacity %u", cnt[g], g, c->cap); return HRBF_ERR_CAPACITY; }
        for (int k = 0; k < c->nsh; ++k) {
            const int gk = c->shard_first + k;
            const uint32_t nk = cnt[gk];
            if (nk) {
                std::vector<float> rows((size_t)nk * 20); std::vector<uint32_t> ids(nk);
                size_t at = 0;
                for (size_t i = 0; i < n; ++i)
                    if (owner[i] == gk) { memcpy(&rows[at * 20], in + i * 20, 80); ids[at++] = (uint32_t)i; }
                float4 *tmp = nullptr;
                HIP_CHECK(hipMalloc((void **)&tmp, sizeof(float4) * 5 * (size_t)nk));
                hipError_t e = hipMemcpyAsync(tmp, rows.data(), sizeof(float4) * 5 * (size_t)nk, hipMemcpyHostToDevice, c->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(c->sh[k].d_gid, ids.data(), sizeof(uint32_t) * (size_t)nk, hipMemcpyHostToDevice, c->stream);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(k_map_from_aos, dim3((nk + 255) / 256), dim3(256), 0, c->stream, c->sh[k].map, nk, tmp);
                    e = hipStreamSynchronize(c->stream);
                }
                hipFree(tmp);
                if (e != hipSuccess) { hrbf_set_error("upload_map: %s", hipGetErrorString(e)); return HRBF_ERR_DEVICE; }
            }
            c->sh[k].count_ub = nk;
        }
        HIP_CHECK(hipMemcpyAsync(counts_live(c), cnt, sizeof(cnt), hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        c->g_next = (uint32_t)n;
        hash_refresh_gfirst(c, counts_live(c));
        c->map_dirty = 1;
        c->ev_pending = false; c->ub_growth_since = 0;
        return HRBF_OK;
    }
    for (int g = 0; g < c->G; ++g) {
        size_t lo, hi; shard_slice(n, c->G, g, &lo, &hi);
        if (hi - lo > c->cap) { hrbf_set_error("upload_map: %zu surfels exceed the shard capacity %u", hi - lo, c->cap); return HRBF_ERR_CAPACITY; }
        cnt[g] = (uint32_t)(hi - lo);
    }
    for (int k = 0; k < c->nsh; ++k) {
        size_t lo, hi; shard_slice(n, c->G, c->shard_first + k, &lo, &hi);
        const uint32_t nk = (uint32_t)(hi - lo);
        if (nk) {
            float4 *tmp = nullptr;
            HIP_CHECK(hipMalloc((void **)&tmp, sizeof(float4) * 5 * (size_t)nk));
            hipError_t e = hipMemcpyAsync(tmp, in + lo * 20, sizeof(float4) * 5 * (size_t)nk, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_map_from_aos, dim3((nk + 255) / 256), dim3(256), 0, c->stream, c->sh[k].map, nk, tmp);
                e = hipStreamSynchronize(c->stream);
            }
            hipFree(tmp);
            if (e != hipSuccess) { hrbf_set_error("upload_map: %s", hipGetErrorString(e)); return HRBF_ERR_DEVICE; }
        }
        c->sh[k].count_ub = nk;
    }
    HIP_CHECK(hipMemcpyAsync(counts_live(c), cnt, sizeof(cnt), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->map_dirty = 1;
    // a count read-back armed before the upload describes the OLD map: folding it into count_ub later would size the
    // next fuse pass for fewer surfels than the device holds
    c->ev_pending = false; c->ub_growth_since = 0;
    return HRBF_OK;
}

#define SETTER(name, field, type)                                                         \
    extern "C" int name(hrbf_handle c, type v) { if (!c) return HRBF_ERR_INVALID; c->prm.field = v; return HRBF_OK; }
SETTER(hrbf_set_rgb_only, rgb_only, int)
SETTER(hrbf_set_icp_weight, icp_weight, float)
SETTER(hrbf_set_pyramid, pyramid, int)
SETTER(hrbf_set_fast_odom, fast_odom, int)
SETTER(hrbf_set_so3, so3, int)
SETTER(hrbf_set_frame_to_frame_rgb, frame_to_frame_rgb, int)
SETTER(hrbf_set_confidence_threshold, confidence_threshold, float)
SETTER(hrbf_set_depth_cutoff, depth_cutoff, float)

static void *img_ptr(hrbf_context *c, int which, size_t *bytes)
{
    const size_t P = (size_t)c->P;
    switch (which) {
#define I1(ID, F) case ID: *bytes = P * 4; return c->F;
#define I4(ID, F) case ID: *bytes = P * 16; return c->F;
        I1(HRBF_IMG_DEPTH_FILTERED, d_depth_filtered) I1(HRBF_IMG_DEPTH_METRIC, d_depth_metric)
        I1(HRBF_IMG_DEPTH_METRIC_FILTERED, d_depth_metric_f)
        I4(HRBF_IMG_VERTEX_RAW, d_vertex_raw) I4(HRBF_IMG_VERTEX_FILTERED, d_vertex_filtered)
        I4(HRBF_IMG_NORMAL, d_normal) I4(HRBF_IMG_NORMAL_PCA, d_normal_pca) I1(HRBF_IMG_RADIUS, d_radius)
        I4(HRBF_IMG_CURV1, d_curv1) I4(HRBF_IMG_CURV2, d_curv2) I1(HRBF_IMG_GRADIENT_MAG, d_gradmag)
        I1(HRBF_IMG_CONFIDENCE, d_confidence) I1(HRBF_IMG_INDEX, d_idx)
        I4(HRBF_IMG_INDEX_VERTCONF, d_im_vertconf) I4(HRBF_IMG_INDEX_COLORTIME, d_im_colortime)
        I4(HRBF_IMG_INDEX_NORMRAD, d_im_normrad) I4(HRBF_IMG_INDEX_CURVMAX, d_im_curvmax)
        I4(HRBF_IMG_INDEX_CURVMIN, d_im_curvmin)
        I1(HRBF_IMG_PRED_IMAGE, d_pr_image) I4(HRBF_IMG_PRED_VERTEX, d_pr_vertex) I4(HRBF_IMG_PRED_NORMAL, d_pr_normal)
        I4(HRBF_IMG_PRED_CURV1, d_pr_curv1) I4(HRBF_IMG_PRED_CURV2, d_pr_curv2) I1(HRBF_IMG_PRED_TIME, d_pr_time)
        I1(HRBF_IMG_PRED_ICPWEIGHT, d_pr_icpw)
        I1(HRBF_IMG_FILL_IMAGE, d_fi_image) I4(HRBF_IMG_FILL_VERTEX, d_fi_vertex) I4(HRBF_IMG_FILL_NORMAL, d_fi_normal)
        I4(HRBF_IMG_FILL_CURV1, d_fi_curv1) I4(HRBF_IMG_FILL_CURV2, d_fi_curv2) I1(HRBF_IMG_FILL_ICPWEIGHT, d_fi_icpw)
        I4(HRBF_IMG_FIT_CURV1, d_fit_curv1) I4(HRBF_IMG_FIT_CURV2, d_fit_curv2) I4(HRBF_IMG_FIT_NORMAL, d_fit_normal)
#undef I1
#undef I4
        default: *bytes = 0; return nullptr;
    }
}
extern "C" size_t hrbf_image_bytes(hrbf_handle c, int which) { size_t b = 0; if (c) img_ptr(c, which, &b); return b; }
extern "C" int hrbf_get_image(hrbf_handle c, int which, void *out, size_t bytes)
{
    if (!c || !out) return HRBF_ERR_INVALID;
    size_t b; void *p = img_ptr(c, which, &b);
    if (!p || bytes < b) { hrbf_set_error("get_image(%d): bad id or buffer too small", which); return HRBF_ERR_INVALID; }
    hipSetDevice(c->device);
    HIP_CHECK(hipMemcpyAsync(out, p, b, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return HRBF_OK;
}
extern "C" int hrbf_set_image(hrbf_handle c, int which, const void *in, size_t bytes)
{
    if (!c || !in) return HRBF_ERR_INVALID;
    size_t b; void *p = img_ptr(c, which, &b);
    if (!p || bytes < b) { hrbf_set_error("set_image(%d): bad id or buffer too small", which); return HRBF_ERR_INVALID; }
    hipSetDevice(c->device);
    c->fill_flag_fresh = 0;
    HIP_CHECK(hipMemcpyAsync(p, in, b, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return HRBF_OK;
}

// on: 0 = off, 1 = region events + fuse ring, 2 = fuse ring only (two events per frame instead of twelve)
extern "C" int hrbf_enable_timing(hrbf_handle c, int on)
{
    if (!c) return HRBF_ERR_INVALID;
    c->timing = on == 2 ? 2 : (on ? 3 : 0);
    return HRBF_OK;
}
extern "C" int hrbf_set_fuse_ring_stride(hrbf_handle c, int every_nth_frame)
{
    if (!c || every_nth_frame < 1) return HRBF_ERR_INVALID;
    c->ring_stride = (uint32_t)every_nth_frame;
    return HRBF_OK;
}
extern "C" int hrbf_get_timings(hrbf_handle c, float out[8])
{
    if (!c || !out) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 8; ++i) out[i] = 0.0f;
    if (!(c->timing & 1)) return HRBF_OK;
    float ms;
    if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) out[0] = ms;   // Initialization
    if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) out[1] = ms;   // Registration
    if (hipEventElapsedTime(&ms, c->ev[3], c->ev[4]) == hipSuccess) out[2] = ms;   // Integration (fuse)
    if (hipEventElapsedTime(&ms, c->ev[7], c->ev[8]) == hipSuccess) out[3] = ms;   // Prediction
    if (hipEventElapsedTime(&ms, c->ev[5], c->ev[6]) == hipSuccess) out[4] = ms;   // clean/compact/append stream pass
    if (hipEventElapsedTime(&ms, c->ev[0], c->ev[9]) == hipSuccess) out[5] = ms;   // whole frame
    if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) out[6] = ms;   // first projection (+confidence)
    // a region whose events no frame has recorded yet (timing switched on, nothing run since) fails the query: tolerated above, but the
    // runtime keeps the error for the next hipGetLastError() — the next launch check would report it as its own (tests/gpu_fuzz_api.py)
    (void)hipGetLastError();
    return HRBF_OK;
}
static int fuse_ring_read(hrbf_context *c, int max_frames, float *merge_ms, float *stream_ms, uint32_t *stats, int nstat)
{
    hipSetDevice(c->device);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    int n = (int)c->ring_valid; if (n > max_frames) n = max_frames;
    uint32_t *h = (uint32_t *)malloc(sizeof(uint32_t) * 8 * HRBF_RING);
    if (hipMemcpy(h, c->d_stats_ring, sizeof(uint32_t) * 8 * HRBF_RING, hipMemcpyDeviceToHost) != hipSuccess) { free(h); return -1; }
    for (int i = 0; i < n; ++i) {   // oldest first among the last n
        uint32_t slot = (c->ring_head - (uint32_t)n + (uint32_t)i) % HRBF_RING;
        float ms = 0.0f, mm = 0.0f;
        if (hipEventElapsedTime(&ms, c->ring_e0[slot], c->ring_e1[slot]) != hipSuccess) ms = -1.0f;
        if (hipEventElapsedTime(&mm, c->ring_m0[slot], c->ring_m1[slot]) != hipSuccess) mm = -1.0f;
        stream_ms[i] = ms; merge_ms[i] = mm;
        memcpy(&stats[i * nstat], &h[slot * 8], sizeof(uint32_t) * (size_t)nstat);
    }
    (void)hipGetLastError();   // a slot without recorded events reads as -1 above; do not leave the query's error to the next launch check
    free(h);
    return n;
}
extern "C" int hrbf_get_fuse_ring(hrbf_handle c, int max_frames, float *kernel_ms, uint32_t *stats4)
{
    if (!c || !kernel_ms || !stats4 || max_frames < 0) return -1;
    float *mm = (float *)malloc(sizeof(float) * (size_t)(max_frames + 1));
    const int n = fuse_ring_read(c, max_frames, mm, kernel_ms, stats4, 4);
    for (int i = 0; i < n; ++i) kernel_ms[i] = (kernel_ms[i] < 0.0f || mm[i] < 0.0f) ? -1.0f : kernel_ms[i] + mm[i];   // F2 + F3
    free(mm);
    return n;
}
extern "C" int hrbf_get_fuse_ring_parts(hrbf_handle c, int max_frames, float *merge_ms, float *stream_ms, uint32_t *stats8)
{
    if (!c || !merge_ms || !stream_ms || !stats8 || max_frames < 0) return -1;
    return fuse_ring_read(c, max_frames, merge_ms, stream_ms, stats8, 8);
}
extern "C" int hrbf_reset_fuse_ring(hrbf_handle c) { if (!c) return -1; c->ring_head = 0; c->ring_valid = 0; return 0; }
extern "C" int hrbf_set_load_trajectory(hrbf_handle c, int v) { if (!c) return HRBF_ERR_INVALID; c->prm.load_trajectory = v; return HRBF_OK; }

extern "C" int hrbf_so3_step(hrbf_handle c, const uint8_t *last_image, const uint8_t *next_image, int rows, int cols,
                             const float image_basis[9], const float kinv[9], const float krlr[9], double A_out[9],
                             double b_out[3], double residual_out[2])
{
    if (!c || !last_image || !next_image || rows <= 0 || cols <= 0 || !image_basis || !kinv || !krlr || !A_out || !b_out || !residual_out) {
        hrbf_set_error("so3_step: null argument or empty image"); return HRBF_ERR_INVALID;
    }
    hipSetDevice(c->device);
    return run_so3_step(c->stream, last_image, next_image, rows, cols, image_basis, kinv, krlr, A_out, b_out, residual_out);
}
extern "C" int hrbf_rgb_residual(hrbf_handle c, float min_scale, const int16_t *dIdx, const int16_t *dIdy,
                                 const float *last_depth, const float *next_depth, const uint8_t *last_image,
                                 const uint8_t *next_image, int rows, int cols, const float kt[3], const float krkinv[9],
                                 int16_t *corres_out, float *diff_out, long long *count, long long *sigma)
{
    if (!c || !corres_out || !diff_out || !count || !sigma || rows <= 0 || cols <= 0 || !dIdx || !dIdy || !last_depth || !next_depth ||
        !last_image || !next_image || !kt || !krkinv) {
        hrbf_set_error("rgb_residual: null argument or empty image"); return HRBF_ERR_INVALID;
    }
    hipSetDevice(c->device);
    return run_rgb_residual(c->stream, min_scale, dIdx, dIdy, last_depth, next_depth, last_image, next_image, rows, cols, kt,
                            krkinv, corres_out, diff_out, count, sigma);
}
extern "C" int hrbf_rgb_step(hrbf_handle c, const int16_t *corres, const float *corres_diff, float sigma,
                             const float *cloud, float fx, float fy, const int16_t *dIdx, const int16_t *dIdy,
                             int use_grad_weight, int rows, int cols, double A_out[36], double b_out[6],
                             double residual_out[2])
{
    if (!c || !corres || !corres_diff || !cloud || rows <= 0 || cols <= 0 || !dIdx || !dIdy || !A_out || !b_out || !residual_out) {
        hrbf_set_error("rgb_step: null argument or empty image"); return HRBF_ERR_INVALID;
    }
    hipSetDevice(c->device);
    return run_rgb_step(c->stream, corres, corres_diff, sigma, cloud, fx, fy, dIdx, dIdy, use_grad_weight, rows, cols, A_out,
                        b_out, residual_out);
}

// ---- the callers' side of the path: submap bookkeeping and the rigid map correction
extern "C" int hrbf_set_index_submap(hrbf_handle c, int index)
{
    if (!c || index < 0 || index > (1 << 24)) return HRBF_ERR_INVALID;   // stored in a float32 attribute of the surfel (Vertex.cpp:21-44)
    c->index_submap = index;
    return HRBF_OK;
}
extern "C" int hrbf_set_active_submaps(hrbf_handle c, const uint8_t *active, int n)
{
    if (!c || n < 0 || (n > 0 && !active)) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->d_submap_active) { hipFree(c->d_submap_active); c->d_submap_active = nullptr; }
    c->n_submap_active = 0;
    if (n > 0) {
        HIP_CHECK(hipMalloc((void **)&c->d_submap_active, (size_t)n));
        HIP_CHECK(hipMemcpy(c->d_submap_active, active, (size_t)n, hipMemcpyHostToDevice));
        c->n_submap_active = n;
    }
    return HRBF_OK;
}
extern "C" int hrbf_update_model(hrbf_handle c, const float *delta16_colmajor, int n)
{
    // n: one matrix per submap id; the reference keeps them in a 19 200-float texture, 1 200 matrices (GlobalModel.cpp:690-767) — a count
    // beyond that is a caller's slip (and n x 64 bytes are about to be read from the pointer)
    if (!c || n < 0 || n > 1200 || (n > 0 && !delta16_colmajor)) { hrbf_set_error("update_model: bad arguments (n = %d; 0 <= n <= 1200)", n); return HRBF_ERR_INVALID; }
    if (n == 0) return HRBF_OK;
    hipSetDevice(c->device);
    if (n > c->delta_cap) {
        HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->d_delta) hipFree(c->d_delta);
        c->d_delta = nullptr; c->delta_cap = 0;
        HIP_CHECK(hipMalloc((void **)&c->d_delta, sizeof(float) * 16 * (size_t)n));
        c->delta_cap = n;
    }
    HIP_CHECK(hipMemcpyAsync(c->d_delta, delta16_colmajor, sizeof(float) * 16 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));   // host matrices are borrowed only for the call
    for (int k = 0; k < c->nsh; ++k)
        launch_update_model(c->stream, c->sh[k].map, counts_live(c) + c->shard_first + k, c->sh[k].count_ub, c->d_delta, n);
    c->map_dirty = 1;   // positions moved: the next clean re-checks everything
    HIP_CHECK(hipGetLastError());
    return HRBF_OK;
}

// sticky status bits (include/hrbf_mi355.h HRBF_STATUS_*): folded from the device by this call (it synchronises)
extern "C" int hrbf_get_status(hrbf_handle c, uint32_t *flags, int clear)
{
    if (!c || !flags) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    for (int k = 0; k < c->nsh; ++k) {
        uint32_t v = 0;
        HIP_CHECK(hipMemcpyAsync(&v, c->sh[k].d_stats + 7, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        if (v & 1u) c->status |= HRBF_STATUS_INTERNAL_BOUND;
        if (v & 4u) c->status |= HRBF_STATUS_FUSE_TIMEOUT;
        if (v & 2u) c->status |= HRBF_STATUS_CAPACITY;
    }
    {
        const int so3 = odo_read_timeouts(c->stream, c->odo.state, clear);
        if (so3 < 0) { hrbf_set_error("get_status: read-back failed"); return HRBF_ERR_DEVICE; }
        if (so3 > 0) c->status |= HRBF_STATUS_SO3_TIMEOUT;
    }
    *flags = c->status;
    if (clear && !c->shard_real) c->renumber_failed = 0;   // a single process may retry the renumbering with the next frame; a rank of a sharded
                                                            // map may not (the retry is a collective only some ranks would enter): the map has to be rebuilt
    if (clear) {
        c->status = 0;
        for (int k = 0; k < c->nsh; ++k) HIP_CHECK(hipMemsetAsync(c->sh[k].d_stats + 7, 0, sizeof(uint32_t), c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return HRBF_OK;
}

// measurement probe: one Gauss-Newton iteration's pixel work of pyramid `level` in ONE workgroup
extern "C" int hrbf_probe_single_workgroup_iteration(hrbf_handle c, int level, int iters, float *ms_out)
{
    if (!c) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    OdoConfig cfg = make_cfg(c);
    return odo_probe_single_wg(c->stream, c->odo, cfg, level, iters, ms_out);
}

// test probe: k_predict_hrbf takes v_sqrt_f32 plus two residual tests for a correctly rounded root; this runs the check
// of that shortcut over every non-negative finite float on the device it will run on
extern "C" int hrbf_probe_sqrt_rounding(hrbf_handle c, uint64_t out[6])
{
    if (!c || !out) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    unsigned long long *d = nullptr;
    if (hipMalloc(&d, 6 * sizeof(unsigned long long)) != hipSuccess) return HRBF_ERR_DEVICE;
    int rc = predict_probe_sqrt(c->stream, d) == 0 ? HRBF_OK : HRBF_ERR_DEVICE;
    if (rc == HRBF_OK && hipMemcpyAsync(out, d, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = HRBF_ERR_DEVICE;
    if (hipStreamSynchronize(c->stream) != hipSuccess) rc = HRBF_ERR_DEVICE;
    hipFree(d);
    return rc;
}

// test probe: the bilateral filter scales its exp polynomial by 2^k with v_ldexp_f32 where hd_expf multiplies twice
extern "C" int hrbf_probe_exp_scaling(hrbf_handle c, uint64_t out[2])
{
    if (!c || !out) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    unsigned long long *d = nullptr;
    if (hipMalloc(&d, 2 * sizeof(unsigned long long)) != hipSuccess) return HRBF_ERR_DEVICE;
    int rc = pre_probe_exp_scaling(c->stream, d) == 0 ? HRBF_OK : HRBF_ERR_DEVICE;
    if (rc == HRBF_OK && hipMemcpyAsync(out, d, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = HRBF_ERR_DEVICE;
    if (hipStreamSynchronize(c->stream) != hipSuccess) rc = HRBF_ERR_DEVICE;
    hipFree(d);
    return rc;
}

// test probe: k_curvature divides without the range scaling of the compiler's expansion on tame tiles
extern "C" int hrbf_probe_division(hrbf_handle c, uint64_t out[2])
{
    if (!c || !out) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    unsigned long long *d = nullptr;
    if (hipMalloc(&d, 2 * sizeof(unsigned long long)) != hipSuccess) return HRBF_ERR_DEVICE;
    int rc = pre_probe_division(c->stream, d) == 0 ? HRBF_OK : HRBF_ERR_DEVICE;
    if (rc == HRBF_OK && hipMemcpyAsync(out, d, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = HRBF_ERR_DEVICE;
    if (hipStreamSynchronize(c->stream) != hipSuccess) rc = HRBF_ERR_DEVICE;
    hipFree(d);
    return rc;
}

extern "C" int hrbf_get_fuse_stats(hrbf_handle c, uint32_t out[4])
{
    if (!c || !out) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    // summed over the local shards (every shard sees the whole record set, so `in`/`out` add up and merged/appended too)
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int k = 0; k < c->nsh; ++k) {
        uint32_t v[4];
        const uint32_t nparts = merge_workgroups(c->Q);
        std::vector<uint32_t> parts(nparts);   // merged = the per-workgroup words of k_apply_merges, added up here
        HIP_CHECK(hipMemcpyAsync(v, c->sh[k].d_stats, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipMemcpyAsync(parts.data(), c->sh[k].d_merged_part, sizeof(uint32_t) * nparts, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        v[1] = 0;
        for (uint32_t t : parts) v[1] += t;
        for (int t = 0; t < 4; ++t) out[t] += v[t];
    }
    return HRBF_OK;
}

extern "C" int hrbf_icp_step(hrbf_handle c, const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                             const float *nmap_curr, const float *ck1_curr, const float *ck2_curr,
                             const float Rprev_inv[9], const float tprev[3], float fx, float fy, float cx, float cy,
                             const float *vmap_g_prev, const float *nmap_g_prev, const float *ck1_g_prev,
                             const float *ck2_g_prev, const float *icp_weight_prev, int rows, int cols,
                             float dist_thresh, float angle_thresh, int use_weight, double A_out[36], double b_out[6],
                             double residual_out[2])
{
    // (found by tests/gpu_probe_abi_zero_args.py: this seam read its host matrices without looking at them first)
    if (!c || !Rcurr || !tcurr || !vmap_curr || !nmap_curr || !ck1_curr || !ck2_curr || !Rprev_inv || !tprev || !vmap_g_prev || !nmap_g_prev ||
        !ck1_g_prev || !ck2_g_prev || !icp_weight_prev || rows <= 0 || cols <= 0 || !A_out || !b_out || !residual_out) {
        hrbf_set_error("icp_step: null argument or empty image"); return HRBF_ERR_INVALID;
    }
    hipSetDevice(c->device);
    return run_icp_step(c->stream, Rcurr, tcurr, vmap_curr, nmap_curr, ck1_curr, ck2_curr, Rprev_inv, tprev, fx, fy, cx,
                        cy, vmap_g_prev, nmap_g_prev, ck1_g_prev, ck2_g_prev, icp_weight_prev, rows, cols, dist_thresh,
                        angle_thresh, use_weight, A_out, b_out, residual_out, nullptr, nullptr, nullptr);
}

// icpStep with useSparse = true (reduce.cu:302-315,455-492): lambda_map in, z_map (z_thrinkMap) and corres (corresICP)
// out — device images, 3 interleaved floats / 2 int32 per pixel — and updateLambdaMap (cudafuncs.cu:1030-1111)
extern "C" int hrbf_icp_step_sparse(hrbf_handle c, const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                                    const float *nmap_curr, const float *ck1_curr, const float *ck2_curr,
                                    const float Rprev_inv[9], const float tprev[3], float fx, float fy, float cx, float cy,
                                    const float *vmap_g_prev, const float *nmap_g_prev, const float *ck1_g_prev,
                                    const float *ck2_g_prev, const float *icp_weight_prev, int rows, int cols,
                                    float dist_thresh, float angle_thresh, int use_weight, const float *lambda_map,
                                    float *z_map_out, int32_t *corres_out, double A_out[36], double b_out[6],
                                    double residual_out[2])
{
    if (!c || !lambda_map || !z_map_out || !corres_out || !Rcurr || !tcurr || !vmap_curr || !nmap_curr || !ck1_curr || !ck2_curr || !Rprev_inv ||
        !tprev || !vmap_g_prev || !nmap_g_prev || !ck1_g_prev || !ck2_g_prev || !icp_weight_prev || rows <= 0 || cols <= 0 || !A_out || !b_out ||
        !residual_out) {
        hrbf_set_error("icp_step_sparse: null argument or empty image"); return HRBF_ERR_INVALID;
    }
    hipSetDevice(c->device);
    return run_icp_step(c->stream, Rcurr, tcurr, vmap_curr, nmap_curr, ck1_curr, ck2_curr, Rprev_inv, tprev, fx, fy, cx,
                        cy, vmap_g_prev, nmap_g_prev, ck1_g_prev, ck2_g_prev, icp_weight_prev, rows, cols, dist_thresh,
                        angle_thresh, use_weight, A_out, b_out, residual_out, lambda_map, z_map_out, corres_out);
}
extern "C" int hrbf_update_lambda_map(hrbf_handle c, const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                                      const float Rprev_inv[9], const float tprev[3], const float *vmap_g_prev,
                                      const int32_t *corres, const float *z_map, float *lambda_map, int rows, int cols)
{
    if (!c || !corres || !z_map || !lambda_map || !Rcurr || !tcurr || !vmap_curr || !Rprev_inv || !tprev || !vmap_g_prev || rows <= 0 || cols <= 0) {
        hrbf_set_error("update_lambda_map: null argument or empty image"); return HRBF_ERR_INVALID;
    }
    hipSetDevice(c->device);
    return run_update_lambda_map(c->stream, Rcurr, tcurr, vmap_curr, Rprev_inv, tprev, vmap_g_prev, corres, z_map, lambda_map,
                                 rows, cols);
}

// ------------------------------------------------------------------------------------------ named operators
// GlobalModel::{initialise,fuse,clean} (GlobalModel.h:50-107) and IndexMap::{predictIndices,predictHRBF}
// (IndexMap.h:43-68) with the arguments the reference passes explicitly — pose, time, cut-offs — on the context's map
// and images (the reference's GPUTexture arguments; fill them with hrbf_set_image / hrbf_upload_frame + stages).
// The given pose / time / thresholds become the context's current ones, exactly as after the reference's call.
static void op_set_pose_time(hrbf_context *c, const float pose16[16], int time)
{
    hipSetDevice(c->device);
    if (pose16) launch_pose_set(c->stream, c->d_pose, pose16, 0);
    if (time > 0) c->tick = time;
}
extern "C" int hrbf_initialise(hrbf_handle c, const float init_pose16[16])
{
    if (!c) return HRBF_ERR_INVALID;
    op_set_pose_time(c, init_pose16, 0);
    st_init(c);
    HIP_CHECK(hipGetLastError());
    return HRBF_OK;
}
extern "C" int hrbf_predict_indices(hrbf_handle c, const float pose16[16], int time, float depth_cutoff, int index_submap)
{
    if (!c) return HRBF_ERR_INVALID;
    op_set_pose_time(c, pose16, time);
    if (depth_cutoff > 0.0f) c->prm.max_depth_processed = depth_cutoff;
    if (index_submap >= 0) c->index_submap = index_submap;
    st_indices(c);
    HIP_CHECK(hipGetLastError());
    return HRBF_OK;
}
extern "C" int hrbf_fuse(hrbf_handle c, const float pose16[16], int time, float depth_cutoff, int index_submap)
{
    if (!c) return HRBF_ERR_INVALID;
    op_set_pose_time(c, pose16, time);
    if (depth_cutoff > 0.0f) c->prm.max_depth_processed = depth_cutoff;
    if (index_submap >= 0) c->index_submap = index_submap;
    st_fuse(c);
    HIP_CHECK(hipGetLastError());
    return HRBF_OK;
}
extern "C" int hrbf_clean(hrbf_handle c, const float pose16[16], int time, float conf_threshold, float max_depth)
{
    if (!c) return HRBF_ERR_INVALID;
    op_set_pose_time(c, pose16, time);
    if (conf_threshold >= 0.0f) c->prm.confidence_threshold = conf_threshold;
    if (max_depth > 0.0f) c->prm.max_depth_processed = max_depth;
    const int r = st_clean(c);
    if (r) return r;
    HIP_CHECK(hipGetLastError());
    return HRBF_OK;
}
extern "C" int hrbf_predict_hrbf(hrbf_handle c)
{
    if (!c) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    st_predict(c);
    HIP_CHECK(hipGetLastError());
    return HRBF_OK;
}

extern "C" int hrbf_dense_enough(hrbf_handle c, int *dense)
{
    if (!c || !dense) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    launch_should_fill_in(c->stream, c->cam, c->d_pr_vertex, c->prm.dense_enough_thresh, &c->d_pose->should_fill_in);
    c->fill_flag_fresh = 1;
    int fill = 0;
    HIP_CHECK(hipMemcpyAsync(&fill, &c->d_pose->should_fill_in, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    *dense = !fill;
    return HRBF_OK;
}

extern "C" int hrbf_comm_unique_id(uint8_t out128[128])
{
    if (!out128) return HRBF_ERR_INVALID;
    int r = rccl_load();
    if (r) return r;
    const int e = g_rccl.GetUniqueId(out128);
    if (e != 0) { hrbf_set_error("ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"); return HRBF_ERR_COMM; }
    return HRBF_OK;
}

// rank >= 0: join the communicator `id128` as `rank` of `world` (one process per GPU).
// rank < 0 (test hook, no RCCL): this single process plays `world` virtual ranks in turn — checks the strip
// arithmetic of the sharded path on one GPU.  world <= 1 with rank < 0 returns to the single-GPU path.
extern "C" int hrbf_comm_init(hrbf_handle c, int rank, int world, const uint8_t id128[128])
{
    if (!c || world < 1) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->comm.comm) { g_rccl.CommDestroy(c->comm.comm); c->comm.comm = nullptr; }
    if (c->d_comm_scratch) { hipFree(c->d_comm_scratch); c->d_comm_scratch = nullptr; }
    peer_close(c); peer_shm_release(c);
    c->comm.rank = 0; c->comm.world = 1; c->comm.virtual_world = 0; c->comm.allreduce_i64 = nullptr; c->comm.ar_count = c->ar_count; c->comm.ar_failed = &c->ar_failed;
    memset(&c->cstats, 0, sizeof(c->cstats)); c->ar_count[0] = c->ar_count[1] = 0;
    if (rank < 0) { c->comm.virtual_world = world > 1 ? world : 0; return HRBF_OK; }
    if (rank >= world || !id128) return HRBF_ERR_INVALID;
    int r = rccl_load();
    if (r) return r;
    RcclId128 id;
    memcpy(id.b, id128, 128);
    // everything a later collective needs on the device is allocated here; a rank that cannot still joins the communicator
    // (its peers are inside ncclCommInitRank) and leaves it again
    const bool have = hipMalloc((void **)&c->d_comm_scratch, sizeof(uint32_t) * COMM_SCRATCH_WORDS) == hipSuccess;
    if (!have) c->d_comm_scratch = nullptr; else hipMemset(c->d_comm_scratch, 0, sizeof(uint32_t) * COMM_SCRATCH_WORDS);
    void *comm = nullptr;
    const int e = g_rccl.CommInitRank(&comm, world, id, rank);
    if (e != 0 || !comm) { if (have) hipFree(c->d_comm_scratch); c->d_comm_scratch = nullptr; hrbf_set_error("ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"); return HRBF_ERR_COMM; }
    if (!have) { g_rccl.CommDestroy(comm); hrbf_set_error("hrbf_comm_init: out of device memory"); return HRBF_ERR_DEVICE; }
    c->comm.comm = comm; c->comm.rank = rank; c->comm.world = world; c->comm.allreduce_i64 = rccl_allreduce_i64; c->comm.ar_ctx = comm;
    return HRBF_OK;
}

// Rendezvous of the peer link without RCCL: a POSIX shared-memory segment whose name is the id.  Works for ranks on different
// GPUs of one node and for several ranks on ONE GPU (which RCCL refuses) — the single-device test of the sharded map.
extern "C" int hrbf_peer_unique_id(uint8_t out128[128])
{
    if (!out128) return HRBF_ERR_INVALID;
    memset(out128, 0, 128);
    struct timespec t; clock_gettime(CLOCK_REALTIME, &t);
    snprintf((char *)out128, 64, "/hrbf_peer_%d_%lx", (int)getpid(), (unsigned long)t.tv_nsec);
    const int fd = shm_open((const char *)out128, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) { hrbf_set_error("shm_open(%s) failed", (const char *)out128); return HRBF_ERR_COMM; }
    const int r = ftruncate(fd, sizeof(PeerShm));
    close(fd);
    if (r != 0) { shm_unlink((const char *)out128); hrbf_set_error("ftruncate of the rendezvous segment failed"); return HRBF_ERR_COMM; }
    return HRBF_OK;   // zero-filled by the kernel; rank 0 of hrbf_comm_init_peer unlinks it when its context goes away
}
// a rendezvous that is abandoned before rank 0 joined it (rank 0's context unlinks the segment when it goes away): remove the name
extern "C" int hrbf_peer_release_id(const uint8_t id128[128])
{
    if (!id128 || id128[0] != '/') return HRBF_ERR_INVALID;
    char name[64]; memcpy(name, id128, 63); name[63] = 0;
    return shm_unlink(name) == 0 ? HRBF_OK : HRBF_ERR_COMM;
}
extern "C" int hrbf_comm_init_peer(hrbf_handle c, int rank, int world, const uint8_t id128[128])
{
    if (!c || !id128 || world < 2 || world > HRBF_PEER_MAX || rank < 0 || rank >= world || id128[0] != '/') return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->comm.comm) { g_rccl.CommDestroy(c->comm.comm); c->comm.comm = nullptr; }
    if (c->d_comm_scratch) { hipFree(c->d_comm_scratch); c->d_comm_scratch = nullptr; }
    peer_close(c); peer_shm_release(c);
    c->comm.rank = rank; c->comm.world = world; c->comm.virtual_world = 0; c->comm.allreduce_i64 = nullptr; c->comm.ar_count = c->ar_count; c->comm.ar_failed = &c->ar_failed;
    memset(&c->cstats, 0, sizeof(c->cstats)); c->ar_count[0] = c->ar_count[1] = 0;
    PeerLink &pl = c->peer;
    memcpy(pl.shm_name, id128, 63); pl.shm_name[63] = 0;
    const int fd = shm_open(pl.shm_name, O_RDWR, 0600);
    if (fd < 0) { hrbf_set_error("shm_open(%s) failed", pl.shm_name); return HRBF_ERR_COMM; }
    void *m = mmap(nullptr, sizeof(PeerShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { hrbf_set_error("mmap of the rendezvous segment failed"); return HRBF_ERR_COMM; }
    pl.shm = (PeerShm *)m; pl.shm_mode = 1; pl.shm_owner = rank == 0; pl.rank = rank; pl.world = world; pl.gen = 0;
    // the barrier counter of a segment is never reset: a second set of contexts on the same id would walk through every barrier
    if (__sync_fetch_and_add(&pl.shm->attached, 1u) >= (uint32_t)world) {
        pl.shm_owner = 0; peer_shm_release(c);
        hrbf_set_error("comm_init_peer: this rendezvous id has already served its %d ranks; take a new one from hrbf_peer_unique_id", world);
        return HRBF_ERR_INVALID;
    }
    // the registration sums: every rank reduces the whole image unless hrbf_set_row_sharding(h, 1) asks for the strips + the
    // (host round trip) all-reduce of this transport
    c->comm.allreduce_i64 = shm_allreduce_i64; c->comm.ar_ctx = c;
    c->rows_replicated = 1;
    return peer_barrier_host(c);
}

// ------------------------------------------------------------------------------------------ sharded surfel map
// SURVEY §8e sharding 2.  The global surfel order (the order a single GPU would hold) is cut into G contiguous ranges,
// one per rank; ids in the index map stay GLOBAL, so z-test ties, the association, the merge rule and the order of the
// map are exactly the single-GPU ones.  New surfels are appended at the end of the order, i.e. on the last shard;
// hrbf_map_rebalance() re-cuts the ranges evenly.  Uses the communicator of hrbf_comm_init: a real one (one shard per
// rank) or the virtual one (this process plays all G shards in turn, reductions done by local kernels).
extern "C" int hrbf_map_shard_init(hrbf_handle c, int enable)
{
    if (!c) return HRBF_ERR_INVALID;
    hipSetDevice(c->device);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (hrbf_surfel_count(c) != 0) { hrbf_set_error("map_shard_init: the map must be empty (upload it afterwards)"); return HRBF_ERR_INVALID; }
    int G = 1, nsh = 1, first = 0, real = 0;
    if (enable) {
        if (c->comm.comm || c->peer.shm_mode) { G = c->comm.world; nsh = 1; first = c->comm.rank; real = 1; }
        else if (c->comm.virtual_world > 1) { G = c->comm.virtual_world; nsh = G; }
        else { hrbf_set_error("map_shard_init: call hrbf_comm_init first"); return HRBF_ERR_INVALID; }
        if (G > HRBF_MAX_SHARDS) { hrbf_set_error("map_shard_init: at most %d shards", HRBF_MAX_SHARDS); return HRBF_ERR_INVALID; }
    }
    for (int k = nsh; k < HRBF_MAX_SHARDS; ++k) free_shard(c->sh[k]);
    for (int k = 1; k < nsh; ++k)
        if (!c->sh[k].map.p0) {
            int r = alloc_shard(c, c->sh[k]);
            if (r) return r;
            HIP_CHECK(hipDeviceSynchronize());   // dalloc zero-fills on the null stream
            launch_fill_u32(c->stream, c->sh[k].d_slot, c->cap, 0xFFFFFFFFu);
        }
    free_scratch(c->x);
    if (nsh > 1 || real) {
        const size_t P = (size_t)c->P;
        int r = dalloc(&c->x.rec_count, HRBF_MAX_SHARDS);
        if (!r) r = dalloc(&c->x.send_idx, P);
        if (!r) r = dalloc(&c->x.send_f, 6 * P);
        if (!r && (nsh > 1 || c->peer.shm_mode)) r = dalloc(&c->x.zbuf, P);   // virtual shards' scratch / the reduced keys of the shm transport
        if (!r && real) r = dalloc(&c->x.recv_idx, P);
        if (!r && real) r = dalloc(&c->x.recv_f, 6 * P);
        if (r) return r;
        if (real) HIP_CHECK(hipHostMalloc((void **)&c->x.h_counts, sizeof(uint32_t) * HRBF_MAX_SHARDS, hipHostMallocDefault));
        HIP_CHECK(hipDeviceSynchronize());
        if (c->x.zbuf) launch_zbuf_reset(c->stream, c->x.zbuf, c->P);
    }
    c->G = G; c->nsh = nsh; c->shard_first = first; c->shard_real = real;
    // enable == 2: ownership by spatial hash of the surfel's cell instead of contiguous ranges of the global order
    c->hash_mode = 0;
    if (enable == 2 && (G > 1 || real)) {   // world size 1 with a real communicator: the same code path, every collective issued
        const char *cs = getenv("HRBF_HASH_CELL");   // cell edge in metres
        const float cell = cs ? (float)atof(cs) : 0.25f;
        c->hash_inv_cell = 1.0f / (cell > 0.0f ? cell : 0.25f);
        c->g_next = 0; c->hash_renumbered = 0;
        const char *ra = getenv("HRBF_HASH_RENUMBER_AT");   // tests: renumber long before 32 bits run out
        c->g_renumber_at = ra ? (uint32_t)strtoul(ra, nullptr, 10) : 0xFFFFFFFFu - 4u * (uint32_t)c->Q;
        int r = 0;
        if (!c->d_gfirst) { r = dalloc(&c->d_gfirst, 1); if (!r) r = dalloc(&c->d_gtotal, 1); if (!r) r = dalloc(&c->d_init_flags2, (size_t)c->P); if (!r) r = dalloc(&c->d_init_offs2, (size_t)c->P); }
        for (int k = 0; k < nsh && !r; ++k) {
            MapShard &sh = c->sh[k];
            if (sh.d_gid) continue;
            r = dalloc(&sh.d_gid, (size_t)c->cap);
            if (!r) r = dalloc(&sh.d_own_local, (size_t)c->P);
            if (!r) r = dalloc(&sh.d_rec_lbest, (size_t)c->Q);
            if (!r) r = dalloc(&sh.d_zpriv, (size_t)c->P);
        }
        if (r) return r;
        HIP_CHECK(hipDeviceSynchronize());
        for (int k = 0; k < nsh; ++k) launch_zbuf_reset(c->stream, c->sh[k].d_zpriv, c->P);
        const uint32_t none = HRBF_NO_SURFEL;
        HIP_CHECK(hipMemcpyAsync(c->d_gfirst, &none, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        c->hash_mode = 1;
    }
    peer_close(c);
    c->peer_fallback = 0; c->renumber_failed = 0;
    if (real) {
        // the index-map images of all ranks are mapped into every rank (the owner of a winner writes them, st_indices).
        // HRBF_SHARD_EXCHANGE=records keeps the packed-record exchange over ncclSend / ncclRecv instead (RCCL transport only).
        const char *ex = getenv("HRBF_SHARD_EXCHANGE");
        if (c->peer.shm_mode || !(ex && !strcmp(ex, "records"))) {
            if (!c->peer.shm_mode) { c->peer.rank = c->comm.rank; c->peer.world = c->comm.world; }
            int all_mapped = 0;
            const int r = peer_map_images(c, &all_mapped);
            if (r) return r;
            if (!all_mapped) {
                // agreed by every rank: somebody cannot map its peers (ranks on different nodes, no shared IPC namespace).  The RCCL
                // transport then exchanges packed winner records over ncclSend / ncclRecv on EVERY rank (HRBF_SHARD_EXCHANGE=records,
                // single-node restriction lifted at the price of the pack / unpack kernels); the shm transport has nothing else.
                if (c->peer.shm_mode) { hrbf_set_error("map_shard_init: a rank could not map its peers' images (shared-memory transport: no fallback)"); return HRBF_ERR_COMM; }
                c->peer_fallback = 1;
            }
        }
    }
    for (int k = 0; k < nsh; ++k) c->sh[k].count_ub = 0;
    HIP_CHECK(hipMemsetAsync(c->d_counts, 0, sizeof(uint32_t) * 2 * HRBF_MAX_SHARDS, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->ev_pending = false;
    return HRBF_OK;
}

// with a communicator present, choose whether the registration reductions are row-sharded + all-reduced (default) or
// every rank reduces the whole image itself (no registration collectives; the map may still be sharded)
extern "C" int hrbf_set_row_sharding(hrbf_handle c, int enable)
{
    if (!c) return HRBF_ERR_INVALID;
    c->rows_replicated = enable ? 0 : 1;
    return HRBF_OK;
}

// Pure host logic (no device needed): the even re-cut of G contiguous ranges.  counts[g] -> new_counts[g], and the
// list of moves {src shard, dst shard, offset in src, offset in dst, length} that realises it (at most 2G - 1).
extern "C" int hrbf_rebalance_plan(const uint32_t *counts, int G, uint32_t *new_counts, uint32_t *moves5, int *n_moves)
{
    if (!counts || !new_counts || !moves5 || !n_moves || G < 1 || G > HRBF_MAX_SHARDS) return HRBF_ERR_INVALID;
    uint64_t N = 0;
    for (int g = 0; g < G; ++g) N += counts[g];
    uint64_t oa = 0;
    int nm = 0;
    for (int a = 0; a < G; ++a) {            // old range of shard a: [oa, oa + counts[a])
        uint64_t tb = 0;
        for (int b = 0; b < G; ++b) {        // new range of shard b: [tb, tb + m_b)
            size_t lo_, hi_; shard_slice((size_t)N, G, b, &lo_, &hi_);
            const uint64_t mb = hi_ - lo_;
            if (a == 0) new_counts[b] = (uint32_t)mb;
            const uint64_t lo = oa > tb ? oa : tb, hi = (oa + counts[a]) < (tb + mb) ? (oa + counts[a]) : (tb + mb);
            if (hi > lo) {
                uint32_t *m = moves5 + 5 * nm++;
                m[0] = (uint32_t)a; m[1] = (uint32_t)b; m[2] = (uint32_t)(lo - oa); m[3] = (uint32_t)(lo - tb); m[4] = (uint32_t)(hi - lo);
            }
            tb += mb;
        }
        oa += counts[a];
    }
    *n_moves = nm;
    return HRBF_OK;
}

// Re-cut the shards evenly (synchronous; call it every few hundred frames or after a bulk upload).  Pieces that change
// owner travel with ncclSend / ncclRecv (real mode) or device copies (single-process mode) into a second set of planes,
// which then becomes the shard.
extern "C" int hrbf_map_rebalance(hrbf_handle c)
{
    if (!c) return HRBF_ERR_INVALID;
    if (c->G == 1) return HRBF_OK;
    if (c->hash_mode) return HRBF_OK;   // ownership by cell: the hash balances the shards, there is no range to re-cut
    if (c->peer.shm_mode) { hrbf_set_error("map_rebalance: the re-cut moves surfels with ncclSend / ncclRecv; not available on the shared-memory transport"); return HRBF_ERR_INVALID; }
    hipSetDevice(c->device);
    uint32_t cnt[HRBF_MAX_SHARDS], ncnt[HRBF_MAX_SHARDS] = {0}, moves[5 * 2 * HRBF_MAX_SHARDS];
    if (read_counts(c, cnt)) { hrbf_set_error("map_rebalance: count read-back failed"); return HRBF_ERR_DEVICE; }
    int nm = 0;
    int r = hrbf_rebalance_plan(cnt, c->G, ncnt, moves, &nm);
    if (r) return r;
    for (int g = 0; g < c->G; ++g)
        if (ncnt[g] > c->cap) { hrbf_set_error("map_rebalance: %u surfels per shard exceed the capacity %u", ncnt[g], c->cap); return HRBF_ERR_CAPACITY; }
    MapPlanes tmp[HRBF_MAX_SHARDS];
    memset(tmp, 0, sizeof(tmp));
    for (int k = 0; k < c->nsh; ++k)
        if ((r = alloc_planes(c, tmp[k], c->cap))) { for (int t = 0; t <= k; ++t) free_planes(tmp[t]); return r; }
    HIP_CHECK(hipDeviceSynchronize());
    const int first = c->shard_first;
    if (c->shard_real) g_rccl.GroupStart();
    for (int i = 0; i < nm; ++i) {
        const int a = (int)moves[5 * i], b = (int)moves[5 * i + 1];
        const size_t so = moves[5 * i + 2], d_o = moves[5 * i + 3], len = moves[5 * i + 4];
        const bool src_local = a >= first && a < first + c->nsh, dst_local = b >= first && b < first + c->nsh;
        for (int pl = 0; pl < 5; ++pl) {
            float4 *sp = src_local ? map_plane(c->sh[a - first].map, pl) + so : nullptr;
            float4 *dp = dst_local ? map_plane(tmp[b - first], pl) + d_o : nullptr;
            if (src_local && dst_local) hipMemcpyAsync(dp, sp, sizeof(float4) * len, hipMemcpyDeviceToDevice, c->stream);
            else if (src_local) { COMM_COUNT(c, send, 16 * len); g_rccl.Send(sp, len * 4, kNcclUint32, b, c->comm.comm, c->stream); }
            else if (dst_local) { COMM_COUNT(c, recv, 16 * len); g_rccl.Recv(dp, len * 4, kNcclUint32, a, c->comm.comm, c->stream); }
        }
    }
    if (c->shard_real) g_rccl.GroupEnd();
    hipError_t e = hipMemcpyAsync(counts_live(c), ncnt, sizeof(ncnt), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    for (int k = 0; k < c->nsh; ++k) {
        MapPlanes old = c->sh[k].map;
        c->sh[k].map = tmp[k];
        free_planes(old);
        c->sh[k].count_ub = ncnt[first + k];
    }
    c->ev_pending = false;
    if (e != hipSuccess) { hrbf_set_error("map_rebalance: %s", hipGetErrorString(e)); return HRBF_ERR_DEVICE; }
    return HRBF_OK;
}

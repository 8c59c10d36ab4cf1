// kernels.h — host-callable launchers of the gfx950 kernels (internal to libhrbf_mi355).
#pragma once
#include "common.h"

struct MapPlanes { float4 *p0, *p1, *p2, *p3, *p4; };
typedef MapPlanes RecPlanes;

// Map sharding (SURVEY §8e sharding 2).  The global surfel order is the concatenation of the G shards: shard k holds
// the global ids [off_k, off_k + n_k) with off_k = n_0 + ... + n_(k-1).  `counts` is the device array of the G live
// counts (all-gathered after every clean); a single-GPU context is G = 1, k = 0, off = 0.
//
// Ownership by spatial hash ("sharded by spatial hash of surfel position ... fixed at insertion"): `gid` non-null.
// The global order is then not a concatenation: every surfel carries its place in it (gid: unique, ascending inside a shard,
// never renumbered — survivors keep theirs, new surfels take g_next + record index), a shard owns the surfels whose cell
// hashes to it (hash_owner), and the z-test runs in two levels: a private z-buffer keyed {depth, LOCAL index} per shard,
// then {depth, gid of the local winner} min-reduced over the shards — the winner of the single map's {depth, index} test,
// because local index order is gid order.  A shard owns a pixel iff its private winner's global key equals the reduced one.
struct ShardRef {
    const uint32_t *counts; int k, G;
    const uint32_t *gid;       // nullable (contiguous ranges)
    const uint32_t *g_first;   // device word: smallest gid alive over all shards = "surfel 0" of the reference's `current > 0U` gates
    uint32_t *own_local;       // P words of this shard: local index of the pixel's winner if this shard owns it, else HRBF_NO_SURFEL
    uint32_t *rec_lbest;       // Q words of this shard: local index of the record's matched surfel if owned, else HRBF_NO_SURFEL
};
#define HRBF_NO_SURFEL 0xFFFFFFFFu
// cell -> shard.  Host (upload) and device (appends, seeding) must agree: one fp32 multiply and floorf per axis, integer mixing
__host__ __device__ inline uint32_t hash_owner(float x, float y, float z, float inv_cell, int G)
{
    const float fx = floorf(x * inv_cell), fy = floorf(y * inv_cell), fz = floorf(z * inv_cell);
    const bool ok = fx > -1.0e9f && fx < 1.0e9f && fy > -1.0e9f && fy < 1.0e9f && fz > -1.0e9f && fz < 1.0e9f;   // NaN / far away: cell 0
    const uint32_t ix = ok ? (uint32_t)(int)fx : 0u, iy = ok ? (uint32_t)(int)fy : 0u, iz = ok ? (uint32_t)(int)fz : 0u;
    uint32_t h = ix * 73856093u ^ iy * 19349663u ^ iz * 83492791u;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h % (uint32_t)G;
}

// device-resident pose block: written by the odometry epilogue, read by every map kernel, so a
// frame needs no host round trip (the reference syncs ~40x per frame)
struct DevPose {
    Rigid pose;        // T_wc
    Rigid tinv;        // cofactor inverse
    Rigid prev;        // pose at the end of the previous frame (lastPose)
    float weighting;   // velocity weighting (HRBFFusion.cpp:1112-1123)
    float last_icp_error, last_icp_count;
    int should_fill_in;
    float frame_wmul;  // processFrame's weightMultiplier of the frame being registered: a device word, so that the captured
                       // Gauss-Newton graph does not depend on a value the caller changes per frame (GUI/src/HRBF_fusion.cpp:225)
};

// trajectory log in pinned, device-mapped host memory: the end-of-frame kernel appends the frame's pose and publishes
// the frame count with a system-scope release, so a caller can follow the trajectory without ever blocking the stream
#define POSE_LOG_CAP 65536u
struct PoseRecord { Rigid pose; uint32_t tag; uint32_t pad[3]; };   // tag = frame index + 1, written after the pose
struct PoseLog {
    uint32_t completed;                 // frames whose pose is in the ring (monotonic)
    uint32_t pad[15];
    PoseRecord poses[POSE_LOG_CAP];     // frame f at poses[f % POSE_LOG_CAP]
};

// ---- k_fit.hip (optional extension: true Hermite-RBF fit on the matrix core; not part of processFrame)
void launch_hrbf_fit(hipStream_t s, const Cam &cam, const float4 *vertex, const float4 *normal, int w, float support, float ridge,
                     float jump, float4 *out_c1, float4 *out_c2, float4 *out_n);

// ---- k_pre.hip
void launch_filter_metric(hipStream_t s, const Cam &cam, const uint16_t *raw, float *filtered, float *metric,
                          float *metric_f, float depthFactor, float maxD, int bilateral, const uint8_t *ride_src = nullptr,
                          uint8_t *ride_dst = nullptr, size_t ride_bytes = 0);   // ride: a copy hidden behind the kernel (see k_filter_metric)
void launch_vertex_normal_radius(hipStream_t s, const Cam &cam, const float *dm, const float *dmf, float4 *vr,
                                 float4 *vf, float4 *n, float4 *npca, float *radius, float radius_mult, int use_pca);
void launch_curvature(hipStream_t s, const Cam &cam, const float4 *vf, const float4 *normal_in, float4 *c1, float4 *c2,
                      float *gradmag, float4 *normal_out, float win);
void launch_confidence(hipStream_t s, const Cam &cam, const float *gradmag, float *conf, const float *weighting,
                       int use_conf_eval, float eps);

// ---- k_map.hip
void launch_initialise(hipStream_t s, const Cam &cam, const DevPose *dp, const float4 *vertex_raw, const float4 *normal,
                       const uint8_t *rgb, const float4 *curv1, const float4 *curv2, const float *gradmag,
                       int use_conf_eval, float eps, float thr, uint32_t *flags, uint32_t *offs, MapPlanes out,
                       uint32_t cap, uint32_t *count, uint32_t *status /* nullable: |= 2 when the seed frame exceeds cap */);
void launch_initialise_hashed(hipStream_t s, const Cam &cam, const DevPose *dp, const float4 *vertex_raw, const float4 *normal,
                              const uint8_t *rgb, const float4 *curv1, const float4 *curv2, const float *gradmag,
                              int use_conf_eval, float eps, float thr, uint32_t *flags, uint32_t *offs, uint32_t *flags2,
                              uint32_t *offs2, MapPlanes out, uint32_t *gid_out, uint32_t cap, uint32_t *count, uint32_t *total,
                              uint32_t *status, int G, int me, float inv_cell);
void launch_gfirst(hipStream_t s, const uint32_t *counts, int first, int nsh, const uint32_t *const *gids, uint32_t *out, int merge);
void launch_iota_u32(hipStream_t s, uint32_t *p, uint32_t n, uint32_t base);
void launch_min_row_u32(hipStream_t s, const uint32_t *row, int n, uint32_t *out);
// out[i] = number of ids, over the G ascending planes ptrs[g][0 .. counts[g]), that are smaller than mine[i]
void launch_gid_rank(hipStream_t s, const uint32_t *const *ptrs, const uint32_t *counts, int G, const uint32_t *mine, uint32_t n, uint32_t *out);
// projection = launch_project (z-buffer of packed keys) + launch_resolve (winner gather).  With a sharded map the
// z-buffers are min-reduced between the two; every shard then resolves the winners it owns (zeros elsewhere), packs
// them as compact winner records and the records of the other shards are scattered into the images
// (launch_winner_unpack); `rearm` lets the last local shard leave the z-buffer empty for the next pass.
void launch_project(hipStream_t s, const Cam &cam, const DevPose *dp, float maxDepth, MapPlanes m, ShardRef sh,
                    uint32_t count_ub, unsigned long long *zbuf, const uint8_t *submap_active /* nullable: KeyFrameIDMap */,
                    int n_active,
                    uint8_t *item_class = nullptr /* nullable: one byte per surfel for pass A of the clean pass (k_project) */,
                    float confThr = 0.0f,
                    uint32_t *item_word = nullptr /* CLEAN_CLASS_WORD builds: {class, first window texel} per surfel instead of the byte */,
                    float clean_window_multiplier = 0.0f);
void launch_resolve(hipStream_t s, const Cam &cam, const DevPose *dp, MapPlanes m, ShardRef sh, unsigned long long *zbuf,
                    uint32_t *idx, float4 *vertconf, float4 *colortime, float4 *normrad, float4 *curvmax, float4 *curvmin,
                    float4 *clean_tex /* nullable: packed texels for the clean test (clean_tex_elems) */,
                    int what /* 1 geometry images | 2 attribute images | 4 clean texels */, int rearm,
                    float clean_conf_thr, int clean_time /* baked into the clean texels */,
                    uint32_t *rec_count = nullptr, uint32_t *rec_idx = nullptr /* nullable: pack the owned winners */,
                    float4 *rec_f = nullptr /* 6 planes of rec_cap float4 */, uint32_t rec_cap = 0,
                    int dense = 1 /* 0: records only (a virtual shard behind the first) */,
                    unsigned long long *zpriv = nullptr /* hash ownership: this shard's private z-buffer {depth, local index}; zbuf then holds the reduced {depth, gid} keys */);
void launch_keys_global(hipStream_t s, const unsigned long long *zpriv, const uint32_t *gid, unsigned long long *out, int P, int merge);
// sharded map: scatter `*count` (or, with count == null, n_ub) winner records, starting at record `first`, into the dense images
void launch_winner_unpack(hipStream_t s, int P, int W, const uint32_t *count, uint32_t first, uint32_t n_ub, const uint32_t *ridx,
                          const float4 *rf, uint32_t cap, int what, float4 *vertconf, float4 *colortime, float4 *normrad,
                          float4 *curvmax, float4 *curvmin, float4 *clean_tex);
size_t clean_tex_elems(int P);   // float4 elements of the clean-texel buffer (4 x 2 blocked, sign of z = updated)
void launch_zbuf_min_merge(hipStream_t s, unsigned long long *dst, unsigned long long *src_reset, int P);   // local stand-in for allReduce(min)
// sharded map over peer-mapped images: the index-map images (and the private z-buffer) of every rank of the node, as mapped
// into this process (hipIpcOpenMemHandle; entry `me` = the local buffers)
#define HRBF_PEER_MAX 8
struct PeerImages {
    int world, me;
    unsigned long long *zbuf[HRBF_PEER_MAX];
    float4 *vertconf[HRBF_PEER_MAX], *normrad[HRBF_PEER_MAX], *colortime[HRBF_PEER_MAX], *curvmax[HRBF_PEER_MAX], *curvmin[HRBF_PEER_MAX],
        *clean[HRBF_PEER_MAX];
    uint32_t *gid[HRBF_PEER_MAX];   // hash ownership: the ranks' id planes (renumbering reads them)
};
void launch_zbuf_min_peers(hipStream_t s, const PeerImages &pi, unsigned long long *zred, int P);
void launch_resolve_scatter(hipStream_t s, const Cam &cam, const DevPose *dp, MapPlanes m, ShardRef sh, const unsigned long long *zred,
                            uint32_t *idx, const PeerImages &pi, int what, bool for_clean, float clean_conf_thr, int clean_time,
                            unsigned long long *zpriv = nullptr);
// what data.vert recomputes a new point's normal and radius from (data.vert:83-96)
struct RecNormalSrc { const float *depth_metric_f; float radius_mult; int use_pca; };
void launch_fuse(hipStream_t s, const Cam &cam, const DevPose *dp, int tick, float maxDepth, int index_submap,
                 const float *depth_metric, const float4 *normal_pca, const float4 *curv1, const float4 *curv2,
                 const float *confidence, const uint8_t *rgb, const uint32_t *idx, const float4 *vertconf,
                 const float4 *normrad, RecPlanes rec, int32_t *rec_flag, uint32_t *rec_best, uint32_t *slot,
                 MapPlanes m, ShardRef sh, uint32_t *stats, float curvThr,
                 hipEvent_t m0, hipEvent_t m1 /* nullable: bracket the merge kernel (F2) */,
                 uint32_t *merged_part /* merge_workgroups(Q) words: the merged count, one word per workgroup of k_apply_merges */,
                 RecNormalSrc rn);
void launch_clean(hipStream_t s, const Cam &cam, const DevPose *dp, float maxDepth, float confThr, float curvThr,
                  int time, float clean_window_multiplier, int full_check, MapPlanes m, RecPlanes rec, int32_t *rec_flag,
                  const uint32_t *count_in, uint32_t *count_out, uint32_t count_ub, uint32_t *stats, uint32_t cap,
                  const float4 *clean_tex, uint8_t *keep_flags,
                  uint32_t *tile_count /* this pass's counters (zero on entry) */,
                  uint32_t *tile_count_next /* the other buffer: zeroed by this pass for the next one */,
                  uint32_t *tile_dirty /* [2] host bookkeeping: entries {this, other} buffer may hold; updated */,
                  uint32_t *tile_done, uint32_t epoch /* per-shard pass counter, > 0 */,
                  uint32_t max_tiles, hipEvent_t e0, hipEvent_t e1, const uint8_t *submap_active, int n_active,
                  int n_records /* Q on the shard that takes the appends (the last one), else 0 */,
                  int zero_records /* re-arm the record flags at the end */,
                  uint32_t *stats_ring_slot /* nullable (timing ring): the pass's 8 statistics words are copied here afterwards */,
                  const uint32_t *merged_part /* as launch_fuse: summed into word 1 of the ring slot */,
                  uint32_t *gid /* nullable; hash ownership: the shard's global-order ids, moved along with the planes */,
                  uint32_t g_base /* id of record 0 if it is appended (record q gets g_base + q) */,
                  int hash_G, int hash_me, float hash_inv_cell /* only records whose cell hashes to hash_me are appended */,
                  int have_class = 0 /* keep_flags[0..count) holds the classes launch_project(item_class) wrote for THIS map and pose */,
                  const uint32_t *class_word = nullptr /* CLEAN_CLASS_WORD builds: what launch_project(item_word) wrote */);
void launch_update_model(hipStream_t s, MapPlanes m, const uint32_t *count, uint32_t count_ub, const float *delta16, int n);
void launch_fill_u32(hipStream_t s, uint32_t *p, size_t n, uint32_t v);
void launch_zbuf_reset(hipStream_t s, unsigned long long *zbuf, int P);
void launch_copy_inputs(hipStream_t s, const uint8_t *src_rgb, size_t nrgb, const uint8_t *src_dep, size_t ndep,
                        uint8_t *d_rgb, uint8_t *d_dep);   // src: device-visible pinned host memory
uint32_t fuse_tile_items();
uint32_t merge_workgroups(int Q);
uint32_t fuse_tile_count_stride();

// ---- k_predict.hip
int predict_upload_tables();
// the live-frame side of FillIn::{vertex,normal,curvature,image}: inputs, parameters that are not the prediction's, outputs
struct FillIn {
    float thr; int frame_to_frame_rgb;
    const float4 *vertex_filtered, *normal, *curv1, *curv2; const float *confidence; const uint8_t *rgb;
    float4 *fi_vertex, *fi_normal, *fi_curv1, *fi_curv2; float *fi_icpw; uint8_t *fi_image;
};
// fill != null: the kernel also fills in from the live frame (the FILL_* images), pixel by pixel, from the prediction it
// still holds in registers — the frame path; the stage API keeps prediction and fill-in apart (launch_fillin)
void launch_predict_hrbf(hipStream_t s, const Cam &cam, const float4 *vertconf, const float4 *normrad,
                         const float4 *colortime, const float4 *curvmax, const float4 *curvmin, int win, int minn,
                         int maxn, float cthr, float lambda, uint8_t *pr_image, float4 *pr_vertex, float4 *pr_normal,
                         float4 *pr_curv1, float4 *pr_curv2, uint32_t *pr_time, float *pr_icpw, const FillIn *fill);
// end-of-frame bookkeeping alone (shouldFillIn of the next frame, lastPose <- currPose, pose ring entry)
void launch_end_of_frame(hipStream_t s, const Cam &cam, const float4 *pr_vertex, float dense_thresh, DevPose *dp,
                         PoseLog *pose_log, uint32_t frame_idx);
void launch_fillin(hipStream_t s, int P, float thr, float lambda, int f2f, const float4 *pr_vertex,
                   const float4 *pr_normal, const float4 *pr_curv1, const float4 *pr_curv2, const float *pr_icpw,
                   const uint8_t *pr_image, const float4 *vertex_filtered, const float4 *normal, const float4 *curv1,
                   const float4 *curv2, const float *confidence, const uint8_t *rgb, float4 *fi_vertex,
                   float4 *fi_normal, float4 *fi_curv1, float4 *fi_curv2, float *fi_icpw, uint8_t *fi_image,
                   const Cam &cam, float dense_thresh, DevPose *dp_end_of_frame /* nullable */,
                   PoseLog *pose_log /* nullable, used with dp_end_of_frame */, uint32_t frame_idx);
void launch_should_fill_in(hipStream_t s, const Cam &cam, const float4 *pr_vertex, float thresh, int *flag);

// ---- k_odo.hip
struct OdoLevel {
    int rows, cols;
    float *vmap_g, *nmap_g, *ck1_g, *ck2_g;   // model maps, planar 4 x rows x cols, global frame
    float *vmap_c, *nmap_c, *ck1_c, *ck2_c;   // live-frame maps
    float *icpw;
    float *last_depth, *next_depth;
    uint8_t *last_image, *next_image, *last_next_image;
    int16_t *dIdx, *dIdy;
    float *cloud;                             // rows*cols*3 (x,y,z interleaved)
    float4 *icp_cur, *icp_model;              // 2 x float4 per pixel: packed operands of the ICP kernel (k_odo_prepare)
    uint8_t *rgb_mask;                        // iteration-invariant part of the RGB residual's pixel test (k_odo_prepare)
    float4 *cloud4;                           // back-projected cloud as one 16-B texel per pixel (xyz, -), in-frame RGB step
    int32_t *dIxy;                            // Sobel gradients packed: dIdx in the low, dIdy in the high 16 bits
    float4 *sparse;                           // sparse ICP only: {lambda.xyz, corres} {z.xyz, -} per pixel, else null
};

struct OdoConfig {
    float fx, fy, cx, cy;
    int rgb_only; float icp_weight; int pyramid, fast_odom, so3, frame_to_frame_rgb;
    int use_search, search_radius, use_weighted, rgb_use_grad;
    float curv_thr;
    int use_sparse;
};

struct OdoState;   // device-resident Gauss-Newton state, defined in k_odo.hip

struct OdoBuffers {
    OdoLevel lv[HRBF_NUM_PYRS];
    OdoState *state;
    int16_t *corres;        // P*6
    float *corres_diff;     // P
    long long *icp_part;    // 32 slot rows x 87 limbs   } one allocation, in this order
    long long *rgb_part;    // 32 x 87                   }
    long long *res_part;    // 64 x 2 (count, sigma)     }
    long long *so3_part;    // 32 x 33                   }
    long long *totals;      // 87 + 87 + 2 + 33 (all-reduce buffer)
    int max_blocks;
    // the 57 launches of the Gauss-Newton loop replayed as one hipGraph (one instance per image-pointer parity)
    void *gn_graph_exec[2]; OdoConfig gn_graph_cfg[2]; int gn_graph_weighting[2]; void *gn_graph_dp[2]; int swap_parity;
    unsigned int gn_graph_captures;   // how often a graph was (re-)captured: 2 in a steady run, one per parity
};

struct OdoSources {   // images the odometry is initialised from (selected on device by should_fill_in)
    const float4 *pr_vertex, *pr_normal, *pr_curv1, *pr_curv2; const float *pr_icpw; const uint8_t *pr_image;
    const float4 *fi_vertex, *fi_normal, *fi_curv1, *fi_curv2; const float *fi_icpw; const uint8_t *fi_image;
    const float4 *vertex_filtered, *normal, *curv1, *curv2; const uint8_t *rgb;
};

// ---- level 0 of the registration pyramids, one pixel (shared by k_odo_level0 and the tail of k_curvature) ----
__device__ __forceinline__ uint8_t intensity_u8(int r, int g, int b)
{
    float v = (float)r * 0.114f;
    v = v + (float)g * 0.299f;
    v = v + (float)b * 0.587f;
    return (uint8_t)(int)v;
}
// v / n / k1 / k2: the live frame's filtered vertex, normal (after updateNormalRad) and curvature texels of pixel i.
// pack != null (packed-operand registration only): the level-0 part of k_odo_prepare's transform + pack_icp_texels is done
// here as well, from registers — the model texel goes into the global frame with the pose *pack and both packed texels
// are written; the planar level-0 model maps then stay in the camera frame (only the pyramid's next level reads them).
__device__ __forceinline__ void odo_level0_pixel(int i, int P, const OdoLevel &L, const OdoSources &src, int fill, int f2f,
                                                 float curv_thr, const float4 v_live, const float4 n_live,
                                                 const float4 k1_live, const float4 k2_live, const Rigid *pack = nullptr)
{
    const float4 *vtex = fill ? src.fi_vertex : src.pr_vertex;
    const float4 *ntex = fill ? src.fi_normal : src.pr_normal;
    const uint8_t *img = (fill || f2f) ? src.fi_image : src.pr_image;
    const float4 *k1t = fill ? src.fi_curv1 : src.pr_curv1;
    const float4 *k2t = fill ? src.fi_curv2 : src.pr_curv2;
    const float *iwt = fill ? src.fi_icpw : src.pr_icpw;
    const float qn = hd_nanf();
    const size_t PP = (size_t)P;
    // model
    {
        float4 v = vtex[i], n = ntex[i];
        float4 vo = make_float4(qn, qn, qn, qn), no = vo;
        if (!(v.z == 0.0f) && n.w > 0.0f) { vo = v; no = n; }
        L.vmap_g[i] = vo.x; L.vmap_g[PP + i] = vo.y; L.vmap_g[2 * PP + i] = vo.z; L.vmap_g[3 * PP + i] = vo.w;
        L.nmap_g[i] = no.x; L.nmap_g[PP + i] = no.y; L.nmap_g[2 * PP + i] = no.z; L.nmap_g[3 * PP + i] = no.w;
        L.last_depth[i] = (v.z > 6.0f || v.z <= 0.0f) ? qn : v.z;
        L.last_image[i] = intensity_u8(img[i * 4], img[i * 4 + 1], img[i * 4 + 2]);
        float4 s = k1t[i], o = make_float4(qn, qn, qn, qn);
        if (s.w < curv_thr && s.w > -curv_thr && !hd_isnanf(s.w)) o = s;
        L.ck1_g[i] = o.x; L.ck1_g[PP + i] = o.y; L.ck1_g[2 * PP + i] = o.z; L.ck1_g[3 * PP + i] = o.w;
        const float k1w = o.w;
        s = k2t[i]; o = make_float4(qn, qn, qn, qn);
        if (s.w < curv_thr && s.w > -curv_thr && !hd_isnanf(s.w)) o = s;
        L.ck2_g[i] = o.x; L.ck2_g[PP + i] = o.y; L.ck2_g[2 * PP + i] = o.z; L.ck2_g[3 * PP + i] = o.w;
        float w = iwt[i];
        w = w > 0.0f ? w : qn;
        L.icpw[i] = w;
        if (pack) {   // tranformMapsKernel / tranformCurvMapsKernel on this pixel (a NaN x marks an empty texel), then the pack
            const Rigid &T = *pack;
            f3 vg = mk3(vo.x, vo.y, vo.z), ng = mk3(no.x, no.y, no.z);
            if (!hd_isnanf(vo.x)) { vg = rot_mul(T, vg); vg = mk3(vg.x + T.t[0], vg.y + T.t[1], vg.z + T.t[2]); }
            if (!hd_isnanf(no.x)) ng = rot_mul(T, ng);
            // the curvature directions are rotated too, but only their w (the curvature, untouched) enters the validity
            const bool ok = !(hd_isnanf(vg.x) || hd_isnanf(ng.x) || hd_isnanf(k1w) || hd_isnanf(o.w));
            L.icp_model[2 * i] = make_float4(vg.x, vg.y, vg.z, w);
            L.icp_model[2 * i + 1] = make_float4(ng.x, ng.y, ng.z, ok ? 1.0f : 0.0f);
        }
    }
    // live frame
    {
        const float4 v = v_live, n = n_live;
        float4 vo = make_float4(qn, qn, qn, qn), no = vo;
        if (!(v.z == 0.0f) && n.w > 0.0f) { vo = v; no = n; }
        L.vmap_c[i] = vo.x; L.vmap_c[PP + i] = vo.y; L.vmap_c[2 * PP + i] = vo.z; L.vmap_c[3 * PP + i] = vo.w;
        L.nmap_c[i] = no.x; L.nmap_c[PP + i] = no.y; L.nmap_c[2 * PP + i] = no.z; L.nmap_c[3 * PP + i] = no.w;
        L.next_depth[i] = (v.z > 6.0f || v.z <= 0.0f) ? qn : v.z;
        L.next_image[i] = intensity_u8(src.rgb[i * 3], src.rgb[i * 3 + 1], src.rgb[i * 3 + 2]);
        float4 s = k1_live, o = make_float4(qn, qn, qn, qn);
        if (s.w < curv_thr && s.w > -curv_thr && !hd_isnanf(s.w)) o = s;
        L.ck1_c[i] = o.x; L.ck1_c[PP + i] = o.y; L.ck1_c[2 * PP + i] = o.z; L.ck1_c[3 * PP + i] = o.w;
        const float k1w = o.w;
        s = k2_live; o = make_float4(qn, qn, qn, qn);
        if (s.w < curv_thr && s.w > -curv_thr && !hd_isnanf(s.w)) o = s;
        L.ck2_c[i] = o.x; L.ck2_c[PP + i] = o.y; L.ck2_c[2 * PP + i] = o.z; L.ck2_c[3 * PP + i] = o.w;
        if (pack) {
            const bool ok = !(hd_isnanf(vo.x) || hd_isnanf(no.x) || hd_isnanf(k1w) || hd_isnanf(o.w));
            L.icp_cur[2 * i] = make_float4(vo.x, vo.y, vo.z, ok ? 1.0f : 0.0f);
            L.icp_cur[2 * i + 1] = make_float4(no.x, no.y, no.z, 0.0f);
        }
    }
}


size_t odo_state_bytes();
size_t odo_slot_bytes();
void odo_release(OdoBuffers &ob);   // destroys the cached graphs
int odo_read_timeouts(hipStream_t s, OdoState *st, int clear);
int pre_probe_exp_scaling(hipStream_t s, unsigned long long *d_out2);   // exhaustive check of the exp scaling of the bilateral filter
int pre_probe_division(hipStream_t s, unsigned long long *d_out2);      // unscaled FMA division of k_curvature against the compiler's
int predict_probe_sqrt(hipStream_t s, unsigned long long *d_out6);   // exhaustive check of the sqrt shortcut of k_predict_hrbf
int odo_probe_single_wg(hipStream_t s, OdoBuffers &ob, const OdoConfig &cfg, int level, int iters, float *ms_out);   // measurement probe   // frames whose SO3 kernel hit its poll bound (sticky)
void launch_odo_first_rgb(hipStream_t s, const OdoBuffers &ob, const uint8_t *rgb);
// full registration: pyramids + SO3 pre-alignment + 3-level Gauss-Newton; updates *dp (pose, weighting inputs)
// Row-sharded registration (SURVEY §8e sharding 1): every rank holds the full pyramids and reduces the image rows
// [rank, rank+1) * rows / world of each level; the int64 limb sums are all-reduced (SUM) and every rank takes the same
// step.  allreduce_i64 is ncclAllReduce(ncclInt64, ncclSum) bound by abi.hip; `virtual_world` > 1 is a test hook: one
// process plays all ranks in turn into the same slot rows (no collective), which checks the strip arithmetic on a
// single GPU.
struct OdoComm {
    void *comm;
    int rank, world;
    int virtual_world;
    int (*allreduce_i64)(void *ar_ctx, long long *buf, size_t count, hipStream_t s);   // in place, on the stream (or synchronously)
    void *ar_ctx;       // its first argument: the RCCL communicator, or the context itself on the shared-memory transport
    int *ar_failed;                 // nullable: set to 1 when an all-reduce returns non-zero (the frame cannot stop half way: the caller raises a status bit)
    unsigned long long *ar_count;   // nullable: {calls, bytes} of the limb all-reduces the sharded path ISSUES (counted whether a wire
                                    // carries them or one process plays the ranks) — hrbf_comm_stats
};
// weight_multiplier >= 0: the velocity weighting of the frame epilogue is computed by the last solve as well.
// level0_done: launch_curvature_level0 has written level 0 of the pyramids for this frame already (1), and the packed
// ICP operands of level 0 in the global frame as well (2)
void launch_odometry(hipStream_t s, OdoBuffers &ob, const OdoSources &src, const OdoConfig &cfg, DevPose *dp,
                     const OdoComm *oc /* nullable */, float weight_multiplier, int level0_done = 0);
// the curvature pass with level 0 of the registration pyramids written from its tail (odo_level0_pixel): the live values
// are in registers there and the kernel is ALU-bound, so the 100 MB of level-0 traffic hide behind it.  Needs a valid
// should_fill_in flag in *dp (set at the end of the previous frame).
struct Level0Args { OdoLevel L; OdoSources src; const DevPose *dp; int f2f; float curv_thr; int pack; };
void launch_curvature_level0(hipStream_t s, const Cam &cam, const float4 *vf, const float4 *normal_in, float4 *c1, float4 *c2,
                             float *gradmag, float4 *normal_out, float win, const Level0Args &l0);
// pose bookkeeping
void launch_pose_set(hipStream_t s, DevPose *dp, const float pose16_colmajor[16], int also_prev);
void launch_frame_epilogue(hipStream_t s, DevPose *dp, float weight_multiplier, int tracked);
void launch_pose_commit_prev(hipStream_t s, DevPose *dp);
// standalone icpStep seam
int run_icp_step(hipStream_t s, const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                 const float *nmap_curr, const float *ck1_curr, const float *ck2_curr, const float Rprev_inv[9],
                 const float tprev[3], float fx, float fy, float cx, float cy, const float *vmap_g_prev,
                 const float *nmap_g_prev, const float *ck1_g_prev, const float *ck2_g_prev, const float *icpw, int rows,
                 int cols, float dist_thresh, float angle_thresh, int use_weight, double A_out[36], double b_out[6],
                 double residual_out[2], const float *lambda_map /* nullable: sparse variant */, float *z_map_out,
                 int32_t *corres_out);
int run_update_lambda_map(hipStream_t s, const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                          const float Rprev_inv[9], const float tprev[3], const float *vmap_g_prev, const int32_t *corres,
                          const float *z_map, float *lambda_map, int rows, int cols);

// standalone so3Step / computeRgbResidual / rgbStep seams (device images, host matrices)
int run_so3_step(hipStream_t s, const uint8_t *last_image, const uint8_t *next_image, int rows, int cols,
                 const float basis[9], const float kinv[9], const float krlr[9], double A_out[9], double b_out[3],
                 double residual_out[2]);
int run_rgb_residual(hipStream_t s, float min_scale, const int16_t *dIdx, const int16_t *dIdy, const float *last_depth,
                     const float *next_depth, const uint8_t *last_image, const uint8_t *next_image, int rows, int cols,
                     const float kt[3], const float krkinv[9], int16_t *corres_out, float *diff_out, long long *count,
                     long long *sigma);
int run_rgb_step(hipStream_t s, const int16_t *corres, const float *corres_diff, float sigma, const float *cloud, float fx,
                 float fy, const int16_t *dIdx, const int16_t *dIdy, int use_grad_weight, int rows, int cols,
                 double A_out[36], double b_out[6], double residual_out[2]);

// k_map.hip — surfel-map kernels for gfx950: seeding, index-map projection, data association,
// sparse merge and the streaming clean + stable-compaction + append pass ("the fuse kernel").
//
// Replaces GlobalModel::{initialise,fuse,clean} (Core/src/GlobalModel.cpp:214-288,355-688) and
// IndexMap::predictIndices (Core/src/IndexMap.cpp:193-267) with their GLSL programs
// (init_unstableTex.*, index_map.*, data.*, update.vert, copy_unstable.*).
//
// Map layout in HBM: five float4 planes of `cap` entries (pos_conf, color_time, norm_rad, curv1,
// curv2), two ping-pong copies.  A wavefront reading plane p touches 64 x 16 B = 1 KiB contiguous.
//
// Design notes (MI355X-first, not a translation of the GL pipeline):
//  * the reference re-writes the whole map in fuse stage 2 (update.vert, 160 B/surfel) and clears a
//    1.69 GB scatter target every frame; here only the <= W*H/4 merged surfels are touched, in place
//    (k_apply_merges), winner per surfel = lowest draw-order record via atomicMin on a per-surfel slot.
//  * projection = u64 atomicMin z-buffer (depth bits << 32 | id) + winner gather.
//  * clean + append = ONE streaming pass with a decoupled look-back scan (single read + single write
//    of every surviving surfel = 160 B/surfel), order preserving like GL transform feedback.
#include "common.h"
#include "kernels.h"
#include "pca_normal.h"

// ------------------------------------------------------------------------------------------
// F4: seeding (init_unstableTex.vert:51-98).  Column-major order like the reference's draw;
// compaction by a single-block-per-column-chunk scan is overkill for a once-per-run pass:
// flags -> exclusive scan (k_scan_small) -> scatter.
__global__ void k_init_flags(Cam cam, const DevPose *__restrict__ dp, const float4 *__restrict__ normal,
                             const float4 *__restrict__ curv1, const float4 *__restrict__ curv2, float thr,
                             uint32_t *__restrict__ flags)
{
    int o = blockIdx.x * blockDim.x + threadIdx.x;   // column-major order index
    int P = cam.W * cam.H;
    if (o >= P) return;
    const Rigid pose = dp->pose;
    int px = o / cam.H, py = o - px * cam.H;
    int i = py * cam.W + px;
    float4 nl = normal[i], k1 = curv1[i], k2 = curv2[i];
    f3 ng = rot_mul(pose, xyz(nl));
    flags[o] = (len3(ng) > 0.5f && k1.w > -thr && k1.w < thr && k2.w > -thr && k2.w < thr) ? 1u : 0u;
}

// generic exclusive scan of n uint32 (n <= a few 100k): one block, sequential chunks.
__global__ __launch_bounds__(1024) void k_scan_small(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                     int n, uint32_t *__restrict__ total)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024) {
        int i = base + threadIdx.x;
        uint32_t v = i < n ? in[i] : 0u;
        uint32_t incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        uint32_t c = carry;
        if (i < n) out[i] = c + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ void k_init_scatter(Cam cam, const DevPose *__restrict__ dp, const float4 *__restrict__ vertex_raw,
                               const float4 *__restrict__ normal, const uint8_t *__restrict__ rgb,
                               const float4 *__restrict__ curv1, const float4 *__restrict__ curv2,
                               const float *__restrict__ gradmag, int use_conf_eval, float eps,
                               const uint32_t *__restrict__ flags, const uint32_t *__restrict__ offs, MapPlanes out,
                               uint32_t cap, const uint32_t *__restrict__ gid_src /* nullable */, uint32_t *__restrict__ gid_out)
{
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    int P = cam.W * cam.H;
    if (o >= P || !flags[o]) return;
    const Rigid pose = dp->pose;
    uint32_t n = offs[o];
    if (n >= cap) return;
    int px = o / cam.H, py = o - px * cam.H;
    int i = py * cam.W + px;
    float4 vl = vertex_raw[i], nl = normal[i];
    f3 pg = xform(pose, xyz(vl));
    float conf = radial_confidence(hd_px_attribute(px, cam.W), hd_px_attribute(py, cam.H), cam.cx, cam.cy, cam.max_dist, 1.0f);   // x = texcoord.x * cols of the uv attribute
    if (use_conf_eval > 0) conf = conf * hd_expf(-eps / hd_sqrtf(gradmag[i]));
    f3 ng = rot_mul(pose, xyz(nl));
    out.p0[n] = make_float4(pg.x, pg.y, pg.z, conf);
    out.p1[n] = make_float4(encode_color_bytes(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]), 0.0f, 1.0f, 1.0f);
    out.p2[n] = make_float4(ng.x, ng.y, ng.z, nl.w);
    out.p3[n] = curv1[i];
    out.p4[n] = curv2[i];
    if (gid_src) gid_out[n] = gid_src[o];   // hash ownership: place in the seed frame's global (column-major draw) order
}

// hash ownership: of the seed frame's surfels this shard keeps those whose cell is its own
__global__ void k_init_owner(Cam cam, const DevPose *__restrict__ dp, const float4 *__restrict__ vertex_raw,
                             const uint32_t *__restrict__ flags, uint32_t *__restrict__ flags_mine, int G, int me, float inv_cell)
{
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    int P = cam.W * cam.H;
    if (o >= P) return;
    int px = o / cam.H, py = o - px * cam.H;
    const f3 pg = xform(dp->pose, xyz(vertex_raw[py * cam.W + px]));
    flags_mine[o] = (flags[o] && hash_owner(pg.x, pg.y, pg.z, inv_cell, G) == (uint32_t)me) ? 1u : 0u;
}
// smallest global-order id alive over the given shards (HRBF_NO_SURFEL if all are empty); merge: min with what *out holds
struct GidPtrs { const uint32_t *p[8]; };
__global__ void k_gfirst(const uint32_t *__restrict__ counts, int first, int nsh, GidPtrs g, uint32_t *__restrict__ out, int merge)
{
    uint32_t m = merge ? *out : HRBF_NO_SURFEL;
    for (int k = 0; k < nsh; ++k)
        if (counts[first + k] > 0u) { const uint32_t v = g.p[k][0]; m = v < m ? v : m; }
    *out = m;
}
void launch_gfirst(hipStream_t s, const uint32_t *counts, int first, int nsh, const uint32_t *const *gids, uint32_t *out, int merge)
{
    GidPtrs g;
    for (int k = 0; k < 8; ++k) g.p[k] = k < nsh ? gids[k] : nullptr;
    hipLaunchKernelGGL(k_gfirst, dim3(1), dim3(1), 0, s, counts, first, nsh, g, out, merge);
}
__global__ void k_iota_u32(uint32_t *p, uint32_t n, uint32_t base)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = base + i;
}

__global__ void k_clamp_count(uint32_t *count, uint32_t cap, uint32_t *status)
{
    if (*count > cap) { *count = cap; if (status) atomicOr(status, 2u); }   // seed frame larger than the map: HRBF_STATUS_CAPACITY
}

// ------------------------------------------------------------------------------------------
// M1: projection (index_map.vert:34-66 + GL point raster + GL_LESS z-test)
#define ZB_EMPTY 0xFFFFFFFFFFFFFFFFull

__global__ void k_fill_u64(unsigned long long *p, int n, unsigned long long v)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__device__ __forceinline__ uint32_t shard_offset(const ShardRef &sh)
{
    uint32_t o = 0;
    for (int j = 0; j < sh.k; ++j) o += sh.counts[j];
    return o;
}

// What the projection in front of the clean pass leaves per surfel for pass A (k_clean_flags), in the keep-byte plane pass A is
// about to overwrite: a STABLE surfel outside the frustum is kept whatever else it holds (copy_unstable.vert:62-166 tests nothing
// on it), so pass A reads one byte for it instead of its 16-byte position — on a map that is mostly out of view the position
// stream of pass A shrinks to the surfels that need a test.  The projection has the position in a register anyway.
// ONE statement of "the surfel projects into the frustum" (copy_unstable.vert:77-83), used by the projection's class and by the
// clean test itself: the class is trusted by pass A, so the two must never drift apart (round-5 advice)
__device__ __forceinline__ bool project_in_view(const Cam &cam, f3 h, float maxDepth, float &u, float &v)
{
    u = ((cam.fx * h.x) / h.z) + cam.cx;
    v = ((cam.fy * h.y) / h.z) + cam.cy;
    return h.z < maxDepth && h.z > 0.0f && u > 0.0f && v > 0.0f && u < (float)cam.W && v < (float)cam.H;
}
#define CLEAN_CLASS_OUT_STABLE 0
#define CLEAN_CLASS_OUT_UNSTABLE 1
#define CLEAN_CLASS_IN_VIEW 2
__global__ __launch_bounds__(256) void k_project(Cam cam, const DevPose *__restrict__ dp, float maxDepth, const float4 *__restrict__ pos,
                                                 ShardRef sh,
                                                 unsigned long long *__restrict__ zbuf,
                                                 const float4 *__restrict__ color_time /* read only with a mask */,
                                                 const uint8_t *__restrict__ submap_active /* nullable */, int n_active,
                                                 uint8_t *__restrict__ item_class /* nullable: see CLEAN_CLASS_* */, float confThr
#ifdef CLEAN_CLASS_WORD
                                                 , uint32_t *__restrict__ item_word /* nullable: class | first window texel, see below */, float wm
#endif
                                                 )
{
    // ids in the keys are GLOBAL for contiguous ranges; LOCAL in the private z-buffer of a hash-owned shard (kernels.h)
    const uint32_t n = sh.counts[sh.k], off = sh.gid ? 0u : shard_offset(sh);
    const Rigid tinv = dp->tinv;
    // PROJECT_UNROLL position loads in flight per lane.  Measured at 4.3 M surfels (69.5 MB stream): 1 / 2 / 4 / 8 loads in
    // flight = 25.3 / 26.1 / 25.4 / 28.3 us — the kernel is not latency-bound, more loads in flight buy nothing
#ifndef PROJECT_UNROLL
#define PROJECT_UNROLL 1
#endif
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t s0 = blockIdx.x * blockDim.x + threadIdx.x; s0 < n; s0 += PROJECT_UNROLL * stride) {
        float4 pk[PROJECT_UNROLL];
#pragma unroll
        for (int k = 0; k < PROJECT_UNROLL; ++k) {
            const uint32_t s = s0 + (uint32_t)k * stride;
            pk[k] = s < n ? pos[s] : make_float4(0.0f, 0.0f, -1.0f, 0.0f);   // z < 0: culled below
        }
#pragma unroll
        for (int k = 0; k < PROJECT_UNROLL; ++k) {
            const uint32_t s = s0 + (uint32_t)k * stride;
            if (s >= n) continue;
            const float4 p = pk[k];
            f3 h = xform(tinv, xyz(p));
            float u, v;
            const bool inv = project_in_view(cam, h, maxDepth, u, v);
            if (item_class) {   // the projection in front of the clean pass: this lane holds what pass A's first decision needs
                const uint32_t cl = inv ? CLEAN_CLASS_IN_VIEW : (p.w < confThr ? CLEAN_CLASS_OUT_UNSTABLE : CLEAN_CLASS_OUT_STABLE);
#ifdef CLEAN_CLASS_WORD
                // ... and where its window starts: the first texel of the half-pixel walk of either axis (the walk visits that texel and
                // the next two, hd_halfpixel_walk), so pass A can request the window's texels TOGETHER with the item's planes instead of
                // behind the projected position it would first have to compute from them (the last dependent round trip of the pass)
                if (item_word) {
                    uint32_t word = cl;
                    if (inv) {
                        const hd_walk wx = hd_halfpixel_walk(u, cam.W, wm), wy = hd_halfpixel_walk(v, cam.H, wm);
                        word |= ((uint32_t)hd_window_texel(wx.lo, cam.W) << 2) | ((uint32_t)hd_window_texel(wy.lo, cam.H) << 15);
                    }
                    item_word[s] = word;
                } else
#endif
                item_class[s] = (uint8_t)cl;
            }
            if (submap_active) {   // index_map.vert:41-45: surfels of inactive submaps are not drawn
                const uint32_t sm = hd_cvt_u32(color_time[s].y);
                if (sm >= (uint32_t)n_active || submap_active[sm] == 0) continue;
            }
            if (h.z > maxDepth || h.z < 0.0f) continue;
            // viewport transform + snap to the 1/256-pixel grid, as the GL rasteriser places the point (hrbf_detmath.h)
            int clip_u, clip_v;
            u = hd_gl_point_window_coord(u, (float)cam.W, &clip_u);
            v = hd_gl_point_window_coord(v, (float)cam.H, &clip_v);
            if (clip_u || clip_v || !(u >= 0.0f && u < (float)cam.W && v >= 0.0f && v < (float)cam.H)) continue;
            int ix = (int)hd_floorf(u), iy = (int)hd_floorf(v);
            unsigned long long key = ((unsigned long long)hd_f2u(h.z) << 32) | (unsigned long long)(off + s);
            unsigned long long *cell = &zbuf[iy * cam.W + ix];
            // cheap pre-filter: keys only ever decrease, so a stale larger-or-equal read is conclusive
            if (key < __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(cell, key);
        }
    }
}

// hash ownership, second level of the z-test: {depth, gid of the private winner} per pixel; merge != 0 takes the minimum with
// what `out` already holds (one process playing several shards: the reduction RCCL / the peers perform between ranks)
__global__ __launch_bounds__(256) void k_keys_global(const unsigned long long *__restrict__ zpriv, const uint32_t *__restrict__ gid,
                                                     unsigned long long *__restrict__ out, int P, int merge)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const unsigned long long kp = zpriv[i];
    unsigned long long k = ZB_EMPTY;
    if (kp != ZB_EMPTY) k = (kp & 0xFFFFFFFF00000000ull) | (unsigned long long)gid[(uint32_t)(kp & 0xFFFFFFFFull)];
    if (merge) { const unsigned long long o = out[i]; k = o < k ? o : k; }
    out[i] = k;
}
void launch_keys_global(hipStream_t s, const unsigned long long *zpriv, const uint32_t *gid, unsigned long long *out, int P, int merge)
{
    hipLaunchKernelGGL(k_keys_global, dim3((P + 255) / 256), dim3(256), 0, s, zpriv, gid, out, P, merge);
}
// the pixel's winner as this shard sees it: {owned, local index, id the index image shows}
struct PixelWinner { bool owned; uint32_t s, id; };
__device__ __forceinline__ PixelWinner pixel_winner(const ShardRef &sh, unsigned long long key, unsigned long long *__restrict__ zpriv, int i)
{
    PixelWinner w; w.owned = false; w.s = 0u;
    const uint32_t sg = (uint32_t)(key & 0xFFFFFFFFull);
    w.id = sg;
    if (sh.gid) {
        const unsigned long long kp = zpriv[i];
        if (kp != ZB_EMPTY) {
            zpriv[i] = ZB_EMPTY;   // the private z-buffer is left clean for the next projection
            const uint32_t sl = (uint32_t)(kp & 0xFFFFFFFFull);
            if (((kp & 0xFFFFFFFF00000000ull) | (unsigned long long)sh.gid[sl]) == key) { w.owned = true; w.s = sl; }
        }
        if (key != ZB_EMPTY && sg == *sh.g_first) w.id = 0u;   // the first surfel of the global order is the reference's id 0
        if (sh.own_local) sh.own_local[i] = w.owned ? w.s : HRBF_NO_SURFEL;
    } else if (key != ZB_EMPTY) {
        w.s = sg - shard_offset(sh);
        w.owned = w.s < sh.counts[sh.k];
    }
    return w;
}

// `what`: 1 = vertconf + normrad (association), 2 = colortime + curvature images (prediction), 4 = packed clean
// texels (clean test).  A frame projects three times and each consumer reads a different subset, so each pass
// gathers and writes only its own (84 -> 36 / 36 / 84 bytes per pixel); the index image is always written.
#define RESOLVE_GEOM 1
#define RESOLVE_ATTR 2
#define RESOLVE_CLEAN 4
// The clean test's view of the index map (pass A of the fuse): ONE 16-byte texel per pixel {winner position in the
// camera frame, winner's init time} — zero when there is no winner, when it is surfel 0 (the reference's
// `current > 0U` gate, copy_unstable.vert:111) or when its confidence is not above the threshold (both predicates of
// the test need it).  A zero texel has z = 0 and fails every `vcf.z > lp.z` of the test, so validity costs no extra
// flag; and a winner's z is never negative (the projection culls h.z < 0), so the SIGN of z carries the test's third
// input, "the winner was updated this frame" (it was a separate 1-bit-per-pixel mask: two 4-byte gathers per window row).
// Texels are stored in 4 x 2 blocks — eight texels = one 128-byte line — so the 2..3 x 2..3 distinct texels of a window
// lie in 1..4 lines instead of 2..3 rows x 1..2 lines (W % 4 == 0, H % 2 == 0: both are multiples of 8).
__host__ __device__ inline size_t clean_tex_float4s(int P) { return (size_t)P + 8; }
__device__ __forceinline__ uint32_t clean_tex_slot(int x, int y, int W)
{
    return ((((uint32_t)y >> 1) * ((uint32_t)W >> 2) + ((uint32_t)x >> 2)) << 3) + (((uint32_t)y & 1u) << 2) + ((uint32_t)x & 3u);
}
__device__ __forceinline__ uint32_t clean_tex_slot_of_pixel(uint32_t i, int W) { return clean_tex_slot((int)(i % (uint32_t)W), (int)(i / (uint32_t)W), W); }
__device__ __forceinline__ float4 clean_texel_pack(float4 t, bool updated)
{
    if (updated) t.z = hd_u2f(hd_f2u(t.z) | 0x80000000u);
    return t;
}
__device__ __forceinline__ bool clean_texel_unpack(float4 &t)   // returns `updated`, leaves the plain texel
{
    const uint32_t zb = hd_f2u(t.z);
    t.z = hd_u2f(zb & 0x7FFFFFFFu);
    return (zb >> 31) != 0u;
}

// Sharded map: instead of reducing dense images between the ranks, a rank packs ONE record per pixel whose
// winner it owns — the pixel index and the winner's attributes the
// next consumer reads — into compact arrays that are exchanged (all-gather-v) and scattered by k_winner_unpack.
// rec.f holds six planes of `cap` float4: 0 vertconf, 1 normrad, 2 colortime, 3 curvmax, 4 curvmin, 5 clean texel.
struct WinnerRecords { uint32_t *count; uint32_t *idx; float4 *f; uint32_t cap; };

// what the index map holds for a pixel whose z-test winner is surfel `s` of this shard (global id `sg`)
struct WinnerTexels {
    float4 vc = {0, 0, 0, 0}, nr = {0, 0, 0, 0}, ct = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, clean = {0, 0, 0, 0};
};
__device__ __forceinline__ bool resolve_winner(const MapPlanes &m, uint32_t s, uint32_t sg, const Rigid &tinv, int what,
                                               float clean_conf_thr, int clean_time, WinnerTexels &o)
{
    bool updated = false;
    const float4 p = m.p0[s];
    const f3 h = xform(tinv, xyz(p));
    if (what & RESOLVE_GEOM) {
        const float4 nr = m.p2[s];
        const f3 n = normalize3(rot_mul(tinv, xyz(nr)));
        o.vc = make_float4(h.x, h.y, h.z, p.w);
        o.nr = make_float4(n.x, n.y, n.z, nr.w);
    }
    if (what & (RESOLVE_ATTR | RESOLVE_CLEAN)) {
        const float4 ct = m.p1[s];
        if (what & RESOLVE_ATTR) { o.ct = ct; o.c1 = m.p3[s]; o.c2 = m.p4[s]; }
        if ((what & RESOLVE_CLEAN) && sg > 0u && p.w > clean_conf_thr) {
            o.clean = make_float4(h.x, h.y, h.z, ct.z);
            updated = ct.w == (float)clean_time;
        }
    }
    return updated;
}

__global__ __launch_bounds__(256) void k_resolve(Cam cam, const DevPose *__restrict__ dp, MapPlanes m, ShardRef sh, int rearm,
                                                 unsigned long long *__restrict__ zbuf,
                                                 uint32_t *__restrict__ idx, float4 *__restrict__ vertconf,
                                                 float4 *__restrict__ colortime, float4 *__restrict__ normrad,
                                                 float4 *__restrict__ curvmax, float4 *__restrict__ curvmin,
                                                 float4 *__restrict__ clean_tex, int what, float clean_conf_thr,
                                                 int clean_time, WinnerRecords rec, int dense,
                                                 unsigned long long *__restrict__ zpriv /* hash ownership: this shard's private z-buffer */)
{
    const int P = cam.W * cam.H;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    // P is a multiple of 64 (W, H multiples of 8) but not always of 256: a wave past the end is dead as a whole.  It must not
    // return — the record packing below has workgroup barriers and sums one count per wave
    const bool live = i < P;
    unsigned long long key = live ? zbuf[i] : ZB_EMPTY;
    const Rigid tinv = dp->tinv;
    const float4 z4 = make_float4(0, 0, 0, 0);
    uint32_t sg = 0u, s = 0u;
    bool owned = false;
    if (key != ZB_EMPTY && rearm) zbuf[i] = ZB_EMPTY;   // leave the depth buffer clean for the next projection (no separate clear pass)
    if (live && (key != ZB_EMPTY || sh.gid)) {
        const PixelWinner w = pixel_winner(sh, key, zpriv, i);   // not owned: the winner lives on another shard, contribute zeros
        owned = w.owned; s = w.s; sg = key != ZB_EMPTY ? w.id : 0u;
    }
    if (dense && live) idx[i] = sg;
    WinnerTexels o;
    bool updated = false;
    if (owned) updated = resolve_winner(m, s, sg, tinv, what, clean_conf_thr, clean_time, o);
    const float4 o_vc = o.vc, o_nr = o.nr, o_ct = o.ct, o_c1 = o.c1, o_c2 = o.c2, o_clean = o.clean;
    if (dense && live) {
        if (what & RESOLVE_GEOM) { vertconf[i] = o_vc; normrad[i] = o_nr; }
        if (what & RESOLVE_ATTR) { colortime[i] = o_ct; curvmax[i] = o_c1; curvmin[i] = o_c2; }
        if (what & RESOLVE_CLEAN) clean_tex[clean_tex_slot_of_pixel((uint32_t)i, cam.W)] = clean_texel_pack(o_clean, updated);
    }
    if (rec.idx) {   // pack the owned winners: one position per workgroup from a single atomic (same-address atomics serialise)
        __shared__ uint32_t s_n[4], s_base;
        const unsigned long long bal = __ballot(owned);
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        if (lane == 0) s_n[wid] = (uint32_t)__popcll(bal);
        __syncthreads();
        if (threadIdx.x == 0) { const uint32_t n = s_n[0] + s_n[1] + s_n[2] + s_n[3]; s_base = n ? atomicAdd(rec.count, n) : 0u; }
        __syncthreads();
        if (owned) {
            uint32_t pos = s_base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            for (int w = 0; w < wid; ++w) pos += s_n[w];
            if (pos < rec.cap) {
                rec.idx[pos] = (uint32_t)i;
                const size_t cap = rec.cap;
                if (what & RESOLVE_GEOM) { rec.f[pos] = o_vc; rec.f[cap + pos] = o_nr; }
                if (what & RESOLVE_ATTR) { rec.f[2 * cap + pos] = o_ct; rec.f[3 * cap + pos] = o_c1; rec.f[4 * cap + pos] = o_c2; }
                if (what & RESOLVE_CLEAN) rec.f[5 * cap + pos] = clean_texel_pack(o_clean, updated);
            }
        }
    }
}

// scatter the winner records of other shards into the dense images (the pixels they cover were written as zeros by this
// rank's own k_resolve: one owner per pixel)
__global__ __launch_bounds__(256) void k_winner_unpack(int P, int W, const uint32_t *__restrict__ count /* nullable */, uint32_t n_fixed, uint32_t first, const uint32_t *__restrict__ ridx,
                                                       const float4 *__restrict__ rf, uint32_t cap, int what,
                                                       float4 *__restrict__ vertconf, float4 *__restrict__ colortime,
                                                       float4 *__restrict__ normrad, float4 *__restrict__ curvmax,
                                                       float4 *__restrict__ curvmin, float4 *__restrict__ clean_tex)
{
    const uint32_t n = count ? *count : n_fixed;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const uint32_t pos = first + r;
        const uint32_t i = ridx[pos];
        if (i >= (uint32_t)P) continue;
        const size_t c = cap;
        if (what & RESOLVE_GEOM) { vertconf[i] = rf[pos]; normrad[i] = rf[c + pos]; }
        if (what & RESOLVE_ATTR) { colortime[i] = rf[2 * c + pos]; curvmax[i] = rf[3 * c + pos]; curvmin[i] = rf[4 * c + pos]; }
        if (what & RESOLVE_CLEAN) clean_tex[clean_tex_slot_of_pixel(i, W)] = rf[5 * c + pos];   // packed by the owner (clean_texel_pack)
    }
}

// workgroups (4 waves) needed to give every 8x8 tile of the quarter grid one wave (k_associate, record part of k_clean_flags)
__host__ __device__ inline uint32_t quarter_tile_blocks(int W, int H)
{
    const uint32_t tiles = (((uint32_t)W / 2u + 7u) / 8u) * (((uint32_t)H / 2u + 7u) / 8u);
    return (tiles + 3u) / 4u;
}

// Which of the texels {p-1, p, p+1} (clamped to the image) one axis of data.vert's half-pixel walk visits (data.vert:108-138):
//     step = (1 / (cols * scale)) * 0.5;  for (i = tc - scale * step * 2; i < tc + scale * step * 2; i += step)  texel floor(fl(i * cols))
// accumulated in fp32 from the uv attribute tc.  Exact arithmetic visits all three; in fp32 the sample at p + 1/2 lands in texel p
// at about a fifth of the columns of a 640-wide image and, unless the loop then takes a fifth sample, texel p+1 is never looked at.
// bit a = texel clamp(p + a - 1) is visited.
__device__ __forceinline__ int assoc_axis_mask(float tc, int n, int p)
{
    const float step = (1.0f / ((float)n * 1.0f)) * 0.5f;
    const float hi = tc + (1.0f * step * 2.0f);
    int m = 0;
    for (float i = tc - (1.0f * step * 2.0f); i < hi; i += step) {
        const int t = hd_window_texel(i, n);
#pragma unroll
        for (int a = 0; a < 3; ++a) m |= (t == clampi(p + a - 1, 0, n - 1)) << a;
    }
    return m;
}

// ------------------------------------------------------------------------------------------
// F1: data association (data.vert:63-198) over the quarter grid; record q = (px/2)*(H/2) + py/2
// preserves the reference's column-major draw order among active pixels.
__global__ __launch_bounds__(256) void k_associate(Cam cam, const DevPose *__restrict__ dp, int tick, float maxDepth, int index_submap,
                                                   const float *__restrict__ depth_metric,
                                                   const float4 *__restrict__ normal_pca,
                                                   const float4 *__restrict__ curv1, const float4 *__restrict__ curv2,
                                                   const float *__restrict__ confidence,
                                                   const uint8_t *__restrict__ rgb, const uint32_t *__restrict__ idx,
                                                   const float4 *__restrict__ vertconf,
                                                   const float4 *__restrict__ normrad, RecPlanes rec,
                                                   int32_t *__restrict__ rec_flag, uint32_t *__restrict__ rec_best,
                                                   uint32_t *__restrict__ slot, ShardRef sh, uint32_t *__restrict__ stats,
                                                   RecNormalSrc rn)
{
    // the pass's statistics words start from zero ([0..3] counts, [4] force-full-check, [5] ticket, [6] moved; [7] is the
    // sticky status): nothing in this kernel reads them and the next kernel (k_apply_merges) is ordered behind it
    if (blockIdx.x == 0 && threadIdx.x < 7) stats[threadIdx.x] = 0u;
    const int QW = cam.W / 2, QH = cam.H / 2;
    // One wave = one 8x8 tile of the quarter grid (a 16x16 pixel block: the per-pixel loads and the index-map samples
    // share cache lines); 8 consecutive lanes walk down a column, so they own 8 consecutive record indices q — the
    // reference's COLUMN-major draw order (data.vert), which decides "first primitive wins" and the append order — and
    // the five record stores are 128-byte runs (row-major threads made them 64 separate lines each).
    const int tiles_x = (QW + 7) / 8;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int qx = (tile % tiles_x) * 8 + (lane >> 3), qy = (tile / tiles_x) * 8 + (lane & 7);
    const int tpar = tick % 2;
    // The filtered depth under the wave's 16 x 16 pixel block (+ the PCA window's reach of 3) is staged in LDS when any of its
    // pixels needs data.vert's own normal (see below): 22 x 22 floats per wave, one barrier for the workgroup — before any exit.
    constexpr int RT = 22;
    __shared__ float s_zf[4][RT * RT];
    const int bx0 = (tile % tiles_x) * 16 - 3, by0 = (tile / tiles_x) * 16 - 3;   // image coordinates of tile texel (0, 0)
    {
        const int ppx = qx * 2 + tpar, ppy = qy * 2 + tpar;
        const bool need = rn.use_pca && qx < QW && qy < QH && ppx < cam.W && ppy < cam.H &&
                          (hd_uv_attribute(ppx, cam.W) != hd_uv_fragment(ppx, cam.W) || hd_uv_attribute(ppy, cam.H) != hd_uv_fragment(ppy, cam.H));
        if (__ballot(need) != 0ull) {
            float *t = s_zf[threadIdx.x >> 6];
            // the lane's eight texels are requested together (as a loop this was load, wait, LDS store eight times over: eight
            // dependent round trips at the head of a kernel that runs one wave per SIMD)
            constexpr int NST = (RT * RT + 63) / 64;
            float zv[NST];
#pragma unroll
            for (int r = 0; r < NST; ++r) {
                const int e = lane + r * 64, ee = e < RT * RT ? e : RT * RT - 1;
                const int gx = clampi(bx0 + ee % RT, 0, cam.W - 1), gy = clampi(by0 + ee / RT, 0, cam.H - 1);
                zv[r] = rn.depth_metric_f[gy * cam.W + gx];
            }
            static_assert(NST == 8, "the pin below lists eight values");
            asm volatile("" : "+v"(zv[0]), "+v"(zv[1]), "+v"(zv[2]), "+v"(zv[3]), "+v"(zv[4]), "+v"(zv[5]), "+v"(zv[6]), "+v"(zv[7]));
#pragma unroll
            for (int r = 0; r < NST; ++r) {
                const int e = lane + r * 64;
                if (e < RT * RT) t[e] = zv[r];
            }
        }
    }
    __syncthreads();
    if (qx >= QW || qy >= QH) return;
    const int q = qx * QH + qy;
    const Rigid pose = dp->pose;
    const int px = qx * 2 + tpar, py = qy * 2 + tpar;
    int flag = 0;
    uint32_t best = 0, lbest = HRBF_NO_SURFEL;
    if (px < cam.W && py < cam.H) {
        const int i = py * cam.W + px;
        const float x = hd_px_attribute(px, cam.W), y = hd_px_attribute(py, cam.H);   // data.vert:66-67: texcoord (the uv attribute) * cols, rows
        const float zr = depth_metric[i];
        f3 vl = mk3((x - cam.cx) * zr * cam.camz, (y - cam.cy) * zr * cam.camw, zr);
        // the 9 + 9 + 9 index-map texels of the association are requested here, before the record's normal is (re)computed:
        // they do not depend on it, and the arithmetic below runs while they are on their way
        uint32_t cur[9]; float4 vcf9[9], nr9[9];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int sx = clampi(px + a - 1, 0, cam.W - 1);
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int sy = clampi(py + b - 1, 0, cam.H - 1);
                const int si = sy * cam.W + sx;
                cur[a * 3 + b] = idx[si]; vcf9[a * 3 + b] = vertconf[si]; nr9[a * 3 + b] = normrad[si];
            }
        }
        float4 npca = normal_pca[i];
        {
            // data.vert RECOMPUTES the new point's normal and radius from the filtered depth (data.vert:83-96) with ITS texcoord —
            // the host-computed uv attribute — and ITS x, y (floats).  The image holds the fragment shader's result; where the
            // inputs differ (an ulp of texcoord at a third of the columns of a 640 x 480 image; always for central differences,
            // which then run on half-pixel coordinates) the record's are recomputed here.  Never at power-of-two sizes with PCA.
            const float tax = hd_uv_attribute(px, cam.W), tay = hd_uv_attribute(py, cam.H);
            const float zf = rn.depth_metric_f[i];
            if (rn.use_pca) {
                f3 nr = xyz(npca);
                if (tax != hd_uv_fragment(px, cam.W) || tay != hd_uv_fragment(py, cam.H))
                    nr = pca_normal_tile(s_zf[threadIdx.x >> 6], RT, bx0, by0, cam.W, cam.H, tax, tay, zf, cam.cx, cam.cy, cam.camz, cam.camw);
                npca = make_float4(nr.x, nr.y, nr.z, rn.radius_mult * get_radius(zf, nr.z, cam.camz, cam.camw));   // data.vert:96
            } else {
                const int W = cam.W, H = cam.H;
                f3 nr = mk3(0.0f, 0.0f, 0.0f);
                const bool ok = depth_metric[py * W + clampi(px - 1, 0, W - 1)] != 0.0f &&
                                depth_metric[clampi(py - 1, 0, H - 1) * W + px] != 0.0f &&
                                depth_metric[py * W + clampi(px + 1, 0, W - 1)] != 0.0f &&
                                depth_metric[clampi(py + 1, 0, H - 1) * W + px] != 0.0f;
                if (ok) {
                    const float cx = cam.cx, cy = cam.cy, camz = cam.camz, camw = cam.camw;
                    const f3 va = mk3((x - cx) * zf * camz, (y - cy) * zf * camw, zf);
                    float z;
                    z = rn.depth_metric_f[py * W + clampi(px + 1, 0, W - 1)]; f3 vxf = mk3(((x + 1.0f) - cx) * z * camz, (y - cy) * z * camw, z);
                    z = rn.depth_metric_f[py * W + clampi(px - 1, 0, W - 1)]; f3 vxb = mk3(((x - 1.0f) - cx) * z * camz, (y - cy) * z * camw, z);
                    z = rn.depth_metric_f[clampi(py + 1, 0, H - 1) * W + px]; f3 vyf = mk3((x - cx) * z * camz, ((y + 1.0f) - cy) * z * camw, z);
                    z = rn.depth_metric_f[clampi(py - 1, 0, H - 1) * W + px]; f3 vyb = mk3((x - cx) * z * camz, ((y - 1.0f) - cy) * z * camw, z);
                    f3 del_x = sub3(scale3(add3(vxb, va), 0.5f), scale3(add3(vxf, va), 0.5f));
                    f3 del_y = sub3(scale3(add3(vyb, va), 0.5f), scale3(add3(vyf, va), 0.5f));
                    nr = normalize3(cross3(del_x, del_y));
                }
                npca = make_float4(nr.x, nr.y, nr.z, rn.radius_mult * get_radius(zf, nr.z, cam.camz, cam.camw));
            }
        }
        f3 nl = xyz(npca);
        float4 k1 = curv1[i], k2 = curv2[i];
        if (len3(nl) > 0.8f && vl.z > 0.3f && vl.z <= maxDepth && k1.w > -300.0f && k1.w < 300.0f &&
            k2.w > -300.0f && k2.w < 300.0f) {
            float bestDist = 1000.0f;
            int counter = 0, best_t = 0;
            float xl = (x - cam.cx) * cam.camz, yl = (y - cam.cy) * cam.camw;
            float lambda = hd_sqrtf((xl * xl + yl * yl) + 1.0f);
            f3 ray = mk3(xl, yl, 1.0f);
            float lray = len3(ray);
            float lnl = len3(nl);
            // The reference walks a half-pixel grid (data.vert:108-160) whose samples fall in the texels {-1, 0, +1} of each axis —
            // those the fp32 walk actually reaches (assoc_axis_mask).  A texel met a second time offers the same distance and
            // `dist < bestDist` is strict, so revisits (and the extra ones border clamping creates) never change anything: the
            // visited texels in first-visit order (x outer, y inner) give the same result.  All 9 + 9 + 9 loads are issued before
            // the first test.
            const int mx = assoc_axis_mask(hd_uv_attribute(px, cam.W), cam.W, px), my = assoc_axis_mask(hd_uv_attribute(py, cam.H), cam.H, py);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const uint32_t current = ((mx >> (t / 3)) & (my >> (t % 3)) & 1) ? cur[t] : 0u;
                if (current > 0u) {
                    const float4 vcf = vcf9[t];
                    if (hd_fabsf((vcf.z * lambda) - (vl.z * lambda)) < 0.05f) {
                        float dist = len3(cross3(ray, xyz(vcf))) / lray;
                        const float4 nr = nr9[t];
                        bool ok = hd_fabsf(nr.z) < 0.75f;
                        if (!ok) {
                            float ang = hd_acosf(dot3(xyz(nr), nl) / (len3(xyz(nr)) * lnl));
                            ok = hd_fabsf(ang) < 0.5f;
                        }
                        if (dist < bestDist && ok) { counter++; bestDist = dist; best = current; best_t = t; }
                    }
                }
            }
            f3 pg = xform(pose, vl);
            f3 ng = rot_mul(pose, nl);
            rec.p0[q] = make_float4(pg.x, pg.y, pg.z, confidence[i]);
            rec.p1[q] = make_float4(encode_color_bytes(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]), (float)index_submap,
                                    (float)tick, counter > 0 ? -1.0f : -2.0f);
            rec.p2[q] = make_float4(ng.x, ng.y, ng.z, npca.w);
            rec.p3[q] = k1;
            rec.p4[q] = k2;
            flag = counter > 0 ? 1 : 2;
            if (counter > 0) {   // first primitive in draw order wins; the slot lives with the surfel's owner
                if (sh.gid) {   // hash ownership: the owner knows the local index of the winner of that pixel (k_resolve)
                    const int sx = clampi(px + best_t / 3 - 1, 0, cam.W - 1), sy = clampi(py + best_t % 3 - 1, 0, cam.H - 1);
                    lbest = sh.own_local[sy * cam.W + sx];
                    if (lbest != HRBF_NO_SURFEL) atomicMin(&slot[lbest], (uint32_t)q);
                } else {
                    const uint32_t l = best - shard_offset(sh);
                    if (l < sh.counts[sh.k]) atomicMin(&slot[l], (uint32_t)q);
                }
            }
        }
    }
    rec_flag[q] = flag;
    rec_best[q] = best;
    if (sh.rec_lbest) sh.rec_lbest[q] = lbest;
}

// F2: sparse in-place merge (update.vert:51-115): only the winning record of each surfel applies.
// 512 threads per workgroup: 150 instead of 300 same-address atomics for the merged count (15.9 -> 14.3 us inside the timed
// pass, A/B twice on the headline and the worst-case leg; 1024 threads: 17-18 us, too few workgroups in flight)
#ifndef MERGE_THREADS
#define MERGE_THREADS 512
#endif
__global__ __launch_bounds__(MERGE_THREADS) void k_apply_merges(int Q, int tick, RecPlanes rec,
                                                      const int32_t *__restrict__ rec_flag,
                                                      const uint32_t *__restrict__ rec_best, uint32_t *__restrict__ slot,
                                                      MapPlanes m, ShardRef sh, uint32_t *__restrict__ merged, float curvThr,
                                                      uint32_t *__restrict__ merged_part)
{
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    bool act = q < Q && rec_flag[q] == 1;
    uint32_t l = 0u;
    if (sh.gid) { l = act ? sh.rec_lbest[q] : HRBF_NO_SURFEL; act = act && l != HRBF_NO_SURFEL; }
    else { l = act ? rec_best[q] - shard_offset(sh) : 0u; act = act && l < sh.counts[sh.k]; }   // only the owner of the matched surfel applies the merge
    const uint32_t s = act ? l : 0u;
    // everything the merge may need is requested in one round (record planes, the slot word and the target surfel's
    // planes) instead of slot -> surfel in two dependent rounds; surfel 0 stands in for inactive lanes
    const uint32_t winner = slot[s];
    const int qs = q < Q ? q : 0;
    const float4 r0 = rec.p0[qs], r1 = rec.p1[qs], r2 = rec.p2[qs], r3 = rec.p3[qs], r4 = rec.p4[qs];
    const float4 vp = m.p0[s], vc = m.p1[s], vn = m.p2[s], c1 = m.p3[s], c2 = m.p4[s];
    act = act && winner == (uint32_t)q;
    {   // merged count: one plain store per workgroup into its own word; whoever wants the number adds the words up
        // (k_copy_stats for the timing ring, hrbf_get_fuse_stats on the host).  One atomic per workgroup on one address —
        // 300, then 150 of them — had cost the kernel 1.5 us each way: same-address atomics serialise at the memory side
        __shared__ uint32_t s_m[MERGE_THREADS / 64];
        const unsigned long long bal = __ballot(act);
        if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = (uint32_t)__popcll(bal);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
#pragma unroll
            for (int w = 0; w < MERGE_THREADS / 64; ++w) t += s_m[w];
            merged_part[blockIdx.x] = t;
        }
    }
    if (!act) return;
    slot[s] = 0xFFFFFFFFu;   // re-arm for the next frame (only touched entries are reset)
    const float c_k = vp.w, a = r0.w, sum = c_k + a;
    if (r2.w < (1.0f + 0.5f) * vn.w) {
        m.p0[s] = make_float4(((c_k * vp.x) + (a * r0.x)) / sum, ((c_k * vp.y) + (a * r0.y)) / sum,
                              ((c_k * vp.z) + (a * r0.z)) / sum, sum);
        f3 oc = decode_color(vc.x), nc = decode_color(r1.x);
        f3 avg = mk3(((c_k * oc.x) + (a * nc.x)) / sum, ((c_k * oc.y) + (a * nc.y)) / sum,
                     ((c_k * oc.z) + (a * nc.z)) / sum);
        m.p1[s] = make_float4(encode_color(avg), vc.y, vc.z, (float)tick);
        f3 nn = normalize3(mk3(((c_k * vn.x) + (a * r2.x)) / sum, ((c_k * vn.y) + (a * r2.y)) / sum,
                               ((c_k * vn.z) + (a * r2.z)) / sum));
        m.p2[s] = make_float4(nn.x, nn.y, nn.z, ((c_k * vn.w) + (a * r2.w)) / sum);
        const float nk1 = ((c_k * c1.w) + (a * r3.w)) / sum, nk2 = ((c_k * c2.w) + (a * r4.w)) / sum;
        m.p3[s] = make_float4(((c_k * c1.x) + (a * r3.x)) / sum, ((c_k * c1.y) + (a * r3.y)) / sum,
                              ((c_k * c1.z) + (a * r3.z)) / sum, nk1);
        m.p4[s] = make_float4(((c_k * c2.x) + (a * r4.x)) / sum, ((c_k * c2.y) + (a * r4.y)) / sum,
                              ((c_k * c2.z) + (a * r4.z)) / sum, nk2);
        // The clean pass skips the curvature re-check of surfels it does not otherwise have to read.  A convex
        // combination of two valid curvatures can leave [-thr, thr] only through fp32 rounding at the very edge;
        // if it ever does, ask this frame's clean pass for a full check (merged[3] = force flag).
        if (!(nk1 >= -curvThr && nk1 <= curvThr && nk2 >= -curvThr && nk2 <= curvThr)) merged[3] = 1u;
    } else {
        m.p0[s] = make_float4(vp.x, vp.y, vp.z, sum);
        m.p1[s] = make_float4(vc.x, vc.y, vc.z, (float)tick);
    }
}

// ------------------------------------------------------------------------------------------
// F3: the fuse kernel — clean test (copy_unstable.vert:62-166) + order-preserving compaction of the
// whole map + append of the new-surfel records, in ONE streaming pass, IN PLACE.
//
// Items 0..N-1 are surfels, N..N+Q-1 are association records (draw order).  Tiles of FUSE_TILE
// items are claimed through an atomic ticket so that a tile's predecessors have always started
// (forward progress for the decoupled look-back).  Tile status words are single 8-byte granules
// {flag:2 | value:32} written/read with relaxed agent-scope atomics (the datum is the flag).
//
// Traffic design (the reference streams 400 B/surfel through update.vert + copy_unstable.vert):
//  * the test of a surfel that is out of view needs only pos_conf + color_time (32 B); norm_rad is
//    fetched for in-view surfels only; the curvature validity of a surfel that was not merged this
//    frame was established when it last passed this kernel, so curv planes are read only for
//    surfels merged this frame (lastTime == time) or after an external map upload (full_check);
//  * compaction is stable and leftwards (out <= in), so it is done in place: a surfel whose
//    output slot equals its input slot is not moved at all.  Until the first removal of a frame
//    nothing is written; after it, survivors are re-read (all 5 planes) and written `shift` slots
//    to the left.  Worst case (removal at index 0) = 80 B read + 80 B written per surfel, the
//    classic out-of-place cost.
//  * in-place safety: tile t writes into the source range of tiles t' <= t only.  Every tile raises
//    tile_done[t'] once all the loads it will ever issue have returned (s_waitcnt vmcnt(0)); a
//    writer polls tile_done of the tiles its output range overlaps.  tile_done never waits on
//    another tile's writes, so there is no serial chain.
#ifndef FUSE_THREADS
#define FUSE_THREADS 1024
#endif
#ifndef FUSE_WAVES_PER_EU   // register budget of pass B (second __launch_bounds__ argument of hipcc = waves per SIMD)
#define FUSE_WAVES_PER_EU 4
#endif
// workgroups of pass B per CU: while one waits for its loads another one stores
#define FUSE_WG_PER_CU ((FUSE_WAVES_PER_EU * 4) / (FUSE_THREADS / 64) > 0 ? (FUSE_WAVES_PER_EU * 4) / (FUSE_THREADS / 64) : 1)
#ifndef FUSE_IPT
#define FUSE_IPT 1   // items per thread and tile
#endif
#define FUSE_TILE (FUSE_THREADS * FUSE_IPT)
#define TC_STRIDE 32   // one tile counter per 128-byte line: wave atomics of different tiles never share a line

struct CleanParams {
    Cam cam;
    const DevPose *dp;
    float maxDepth, confThr, curvThr;
    int time;
    int nw;        // samples per axis = ceil(2 * clean_window_multiplier) (copy_unstable.vert:106-108, half-pixel steps)
    float w0;      // clean_window_multiplier * 0.5
    float wm;      // clean_window_multiplier: the walk itself is taken literally in fp32 (hd_halfpixel_walk)
    int full_check;
    const uint8_t *submap_active;   // nullable: KeyFrameIDMap (copy_unstable.vert:98-101)
    int n_active;
    int hash_G, hash_me; float hash_inv_cell;   // hash ownership (hash_G > 1): a shard appends the new surfels whose cell is its own
    const uint32_t *class_word;     // CLEAN_CLASS_WORD builds: k_project's {class, first window texel} per surfel (with have_class)
};

#ifdef CLEAN_DIAG
// diagnostic build: [0] in-view items that reach the window test, [1] of them those with NO visited texel behind
// them (`vcf.z > lp.z` false everywhere: a max-winner-depth image would settle them with one 4-byte gather), [2] dropped
__device__ unsigned long long g_clean_diag[4];
extern "C" int hrbf_probe_clean_diag(unsigned long long out[4], int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clean_diag), sizeof(unsigned long long) * 4) != hipSuccess) return -1;
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_clean_diag), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
// CLEAN_CLASS_WORD == 4: only the 2 x 2 texels every walk visits are requested ahead (the third row / column, visited by about half
// of the walks per axis, stays a dependent load): 16 instead of 36 registers of texels in flight
#if defined(CLEAN_CLASS_WORD) && (CLEAN_CLASS_WORD + 0) == 4
#define CLEAN_PRE(jx, jy) ((jx) < 2 && (jy) < 2)
#else
#define CLEAN_PRE(jx, jy) true
#endif
struct Tex9 { float4 t[9]; };   // by value / reference with constant indices only: stays in registers (a nullable pointer to it went to scratch)
// window part of the test; returns false when the surfel must be dropped.
// The reference walks a 4x4 half-pixel grid (copy_unstable.vert:104-141) = 2..3 DISTINCT texels per axis,
// some visited twice.  Each distinct texel is fetched once from the packed clean texture (2 x float4,
// written by k_resolve) and its two predicates are counted with the multiplicity of the visit pattern.
__device__ __forceinline__ bool clean_window(const CleanParams &cp, const Rigid &tinv, f3 lp, float x, float y,
                                             float init_time, float submap, float4 vn,
                                             const float4 *__restrict__ clean_tex,
                                             const Tex9 &pre_tex = Tex9() /* 9 texels requested ahead from (pre_sx0, pre_sy0), [jx * 3 + jy] */,
                                             int pre_sx0 = -1, int pre_sy0 = -1)
{
    const Cam &cam = cp.cam;
    int count = 0, zCount = 0;
    f3 ln = normalize3(rot_mul(tinv, xyz(vn)));
    bool own_active = true;
    if (cp.submap_active) {
        const uint32_t sm = hd_cvt_u32(submap);
        own_active = sm < (uint32_t)cp.n_active && cp.submap_active[sm] != 0;
    }
    const bool nz_ok = hd_fabsf(ln.z) > 0.85f && own_active;
    const float rad14 = vn.w * 1.4f;
    const float ftime = (float)cp.time;
    (void)ftime;
    if (cp.nw == 4) {
        // the 4 samples of an axis — 5 where the fp32-accumulated coordinate ends an ulp below the bound (hd_halfpixel_walk) — are
        // non-decreasing with steps <= 1 over [x - 1, x + 1]: texels s0, s0+1, s0+2 with multiplicities.
        // All (<= 9) distinct texels and the three mask rows are requested in one batch, then evaluated.
        const hd_walk wkx = hd_halfpixel_walk(x, cam.W, cp.wm), wky = hd_halfpixel_walk(y, cam.H, cp.wm);
        int mx[3] = {0, 0, 0}, my[3] = {0, 0, 0}, sx0 = -1, sy0 = -1;
        for (float fi = wkx.lo; fi < wkx.hi; fi += wkx.step) {
            const int t = hd_window_texel(fi, cam.W);
            if (sx0 < 0) sx0 = t;
            const int d = t - sx0;
            mx[0] += d == 0; mx[1] += d == 1; mx[2] += d == 2;
        }
        for (float fj = wky.lo; fj < wky.hi; fj += wky.step) {
            const int t = hd_window_texel(fj, cam.H);
            if (sy0 < 0) sy0 = t;
            const int d = t - sy0;
            my[0] += d == 0; my[1] += d == 1; my[2] += d == 2;
        }
        if (sx0 < 0 || sy0 < 0) return true;   // an empty walk (cannot happen for a surfel in view)
        const int sxk[1] = {sx0}, syk[1] = {sy0};
        float4 ta[9];
#pragma unroll
        for (int jy = 0; jy < 3; ++jy)
#pragma unroll
            for (int jx = 0; jx < 3; ++jx) {   // texels outside the visit pattern may lie outside the image: not read
                ta[jx * 3 + jy] = make_float4(0, 0, 0, 0);
                if (CLEAN_PRE(jx, jy) && pre_sx0 == sx0 && pre_sy0 == sy0) { if (mx[jx] * my[jy] > 0) ta[jx * 3 + jy] = pre_tex.t[jx * 3 + jy]; }   // the same texels, requested a round earlier
                else if (mx[jx] * my[jy] > 0) ta[jx * 3 + jy] = clean_tex[clean_tex_slot(sxk[0] + jx, syk[0] + jy, cam.W)];
            }
#pragma unroll
        for (int jx = 0; jx < 3; ++jx)
#pragma unroll
            for (int jy = 0; jy < 3; ++jy) {
                const int wgt = mx[jx] * my[jy];
                float4 vcf = ta[jx * 3 + jy];   // {winner xyz, winner init time}, sign of z = updated this frame; all zero = no usable winner
                const bool upd = clean_texel_unpack(vcf);
                ta[jx * 3 + jy] = vcf;
                if (wgt > 0 && vcf.z > lp.z) {
                    float dx = vcf.x - lp.x, dy = vcf.y - lp.y;
                    if (vcf.w < init_time && vcf.z - lp.z < 0.01f && hd_sqrtf(dx * dx + dy * dy) < rad14) count += wgt;
                    if (upd && vcf.z - lp.z > 0.01f && nz_ok) zCount += wgt;
                }
            }
#ifdef CLEAN_DIAG
        {
            bool behind = false;
#pragma unroll
            for (int t = 0; t < 9; ++t) behind |= (mx[t / 3] * my[t % 3] > 0) && ta[t].z > lp.z;
            atomicAdd(&g_clean_diag[0], 1ull);
            if (!behind) atomicAdd(&g_clean_diag[1], 1ull);
            if (count > 8 || zCount > 4) atomicAdd(&g_clean_diag[2], 1ull);
        }
#endif
        return !(count > 8 || zCount > 4);
    }
    const hd_walk wkx = hd_halfpixel_walk(x, cam.W, cp.wm), wky = hd_halfpixel_walk(y, cam.H, cp.wm);
    int prev_sx = -1, colc = 0, colz = 0;
    for (float fi = wkx.lo; fi < wkx.hi; fi += wkx.step) {
        const int sx = hd_window_texel(fi, cam.W);
        if (sx != prev_sx) {
            prev_sx = sx; colc = 0; colz = 0;
            int prev_sy = -1, c1 = 0, z1 = 0;
            for (float fj = wky.lo; fj < wky.hi; fj += wky.step) {
                const int sy = hd_window_texel(fj, cam.H);
                if (sy != prev_sy) {
                    prev_sy = sy; c1 = 0; z1 = 0;
                    float4 vcf = clean_tex[clean_tex_slot(sx, sy, cam.W)];
                    const bool upd = clean_texel_unpack(vcf);
                    if (vcf.z > lp.z) {
                        float dx = vcf.x - lp.x, dy = vcf.y - lp.y;
                        if (vcf.w < init_time && vcf.z - lp.z < 0.01f && hd_sqrtf(dx * dx + dy * dy) < rad14) c1 = 1;
                        if (upd && vcf.z - lp.z > 0.01f && nz_ok) z1 = 1;
                    }
                }
                colc += c1; colz += z1;
            }
        }
        count += colc; zCount += colz;
    }
    return !(count > 8 || zCount > 4);
}

__device__ __forceinline__ bool in_view(const CleanParams &cp, const Rigid &tinv, float4 vp, f3 &lp, float &x, float &y)
{
    const Cam &cam = cp.cam;
    lp = xform(tinv, xyz(vp));
    return project_in_view(cam, lp, cp.maxDepth, x, y);
}

// the clean test of one item: surfel `idx` of the map or association record `idx`
__device__ __forceinline__ bool clean_item(const CleanParams &cp, const Rigid &tinv, float ftime, const MapPlanes &m,
                                           const RecPlanes &rec, bool is_surf, uint32_t idx,
                                           const float4 *__restrict__ clean_tex, const float4 vp /* pos_conf of the item */,
                                           const bool pre = false /* the caller knew the item's class and requested its planes together */,
                                           const float4 pre_vc = {0, 0, 0, 0}, const float4 pre_vn = {0, 0, 0, 0},
                                           const Tex9 &pre_tex = Tex9(), int pre_sx0 = -1, int pre_sy0 = -1)
{
    bool keep = true;
    f3 lp; float x, y;
    const bool inv = in_view(cp, tinv, vp, lp, x, y);
    // color_time is needed for in-view items (init time, merged-this-frame), for unstable surfels (stale
    // test) and for records; a stable surfel outside the frustum is decided by pos_conf alone (16 B)
    const bool need_ct = inv || !is_surf || vp.w < cp.confThr || cp.full_check;
    float4 vc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (need_ct) vc = pre ? pre_vc : (is_surf ? m.p1[idx] : rec.p1[idx]);
    if (inv) {
        const float4 vn = pre ? pre_vn : (is_surf ? m.p2[idx] : rec.p2[idx]);
        keep = clean_window(cp, tinv, lp, x, y, vc.z, vc.y, vn, clean_tex, pre_tex, pre_sx0, pre_sy0);
    }
    if (!is_surf || cp.full_check || (need_ct && vc.w == ftime)) {
        const float k1 = is_surf ? m.p3[idx].w : rec.p3[idx].w;
        const float k2 = is_surf ? m.p4[idx].w : rec.p4[idx].w;
        if (k1 < -cp.curvThr || k1 > cp.curvThr || k2 < -cp.curvThr || k2 > cp.curvThr) keep = false;
    }
    if (need_ct) {
        float lastw = vc.w;
        if (lastw == -2.0f) lastw = ftime;
        if (lastw == -1.0f || ((ftime - lastw) > 200.0f && vp.w < cp.confThr)) keep = false;
    }
    return keep;
}

// Pass A: the clean test of every surfel and record -> one keep byte per item.  Embarrassingly
// parallel, so the in-view minority (16 index-map samples each) is spread over the whole chip by the
// hardware scheduler instead of stalling the scan tiles that happen to contain it.
// The first ceil(Q / 256) workgroups take the records: every record is in view (18 texel gathers each), so they are
// the first work the chip starts, and a wave takes an 8x8 tile of the quarter grid — its record loads are eight
// 128-byte runs and its window gathers fall into ~18 image rows x a few lines, instead of 64 rows of one column,
// which is what 64 consecutive record indices q (column-major draw order) would give.  Their keep counts reach the tile counters through an LDS histogram.
#define CLEAN_HIST 256
__global__ __launch_bounds__(256) void k_clean_flags(CleanParams cp, MapPlanes m, RecPlanes rec,
                                                     const int32_t *__restrict__ rec_flag, int Q,
                                                     const uint32_t *__restrict__ count_in,
                                                     const float4 *__restrict__ clean_tex,
                                                     uint8_t *__restrict__ keep_flags,
                                                     uint32_t *__restrict__ tile_count, uint32_t *__restrict__ stats,
                                                     int have_class /* keep_flags[0..N) holds k_project's CLEAN_CLASS_* of this very map and pose */)
{
    const uint32_t N = *count_in;
    const Rigid tinv = cp.dp->tinv;
    const float ftime = (float)cp.time;
    cp.full_check |= (int)stats[4];   // raised by k_apply_merges (see there)
    if (blockIdx.x == 0 && threadIdx.x == 0) { stats[2] = 0; stats[5] = 0; stats[6] = 0; }   // appended, tile ticket, moved: pass B
    const uint32_t nrb = Q > 0 ? quarter_tile_blocks(cp.cam.W, cp.cam.H) : 0u;
    if (blockIdx.x < nrb) {
        __shared__ uint32_t s_hist[CLEAN_HIST];
        const uint32_t base_tile = N / FUSE_TILE;
        const uint32_t ntile = (N + (uint32_t)Q - 1u) / FUSE_TILE - base_tile + 1u;
        const bool use_hist = ntile <= CLEAN_HIST;
        s_hist[threadIdx.x] = 0u;
        __syncthreads();
        // one wave = one 8x8 tile of the quarter grid; 8 consecutive lanes walk down a column = 8 consecutive q
        const uint32_t QW = (uint32_t)cp.cam.W / 2u, QH = (uint32_t)cp.cam.H / 2u;
        const uint32_t tiles_x = (QW + 7u) / 8u;
        const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
        const uint32_t qx = (tile % tiles_x) * 8u + (lane >> 3), qy = (tile / tiles_x) * 8u + (lane & 7u);
        if (qx < QW && qy < QH) {
            const uint32_t q = qx * QH + qy;
            const float4 rp = rec.p0[q];
            const bool mine = cp.hash_G <= 1 || hash_owner(rp.x, rp.y, rp.z, cp.hash_inv_cell, cp.hash_G) == (uint32_t)cp.hash_me;
            // flag 1 = a merge mark (colour word w = -1): copy_unstable.vert:159-162 drops it whatever the window says — its test
            // (a normal load and <= 12 gathers for 61 000 of the headline stream's 63 600 records) is not run; flag 2 = a new surfel
#ifndef CLEAN_TEST_MERGE_MARKS
            const bool keep = rec_flag[q] == 2 && mine && clean_item(cp, tinv, ftime, m, rec, false, q, clean_tex, rp);
#else
            const bool keep = rec_flag[q] != 0 && mine && clean_item(cp, tinv, ftime, m, rec, false, q, clean_tex, rp);
#endif
            keep_flags[N + q] = keep ? 1 : 0;
            if (keep) {
                const uint32_t tl = (N + q) / FUSE_TILE;
                if (use_hist) atomicAdd(&s_hist[tl - base_tile], 1u);
                else atomicAdd(&tile_count[(size_t)tl * TC_STRIDE], 1u);
            }
        }
        __syncthreads();
        if (use_hist && threadIdx.x < ntile && s_hist[threadIdx.x])
            atomicAdd(&tile_count[(size_t)(base_tile + threadIdx.x) * TC_STRIDE], s_hist[threadIdx.x]);
        return;
    }
    const uint32_t sb = blockIdx.x - nrb, sgrid = gridDim.x - nrb;
    const uint32_t N64 = (N + 63u) & ~63u;
    // Plain round-robin grid-stride on purpose: the in-view minority (the expensive items) is clustered in the
    // array, and an XCD-contiguous chunking (one eighth of the array per XCD, better L2 locality for the clean
    // texels) measured 2x SLOWER because one or two XCDs then own all the heavy work (profiles/r01 notes).
    // CLEAN_UNROLL position loads per lane in flight before the first item is tested.  Measured at 4.3 M surfels:
    // 1 / 2 / 4 / 8 = 36 / 38 / 38 / 43 us (and neither the keep-byte stores nor the tile-count atomics show up when
    // compiled out): the pass is bound by the in-view minority's window gathers, not by the stream's latency
#ifndef CLEAN_UNROLL
#define CLEAN_UNROLL 1
#endif
    const uint32_t stride = sgrid * blockDim.x;
    for (uint32_t it0 = sb * blockDim.x + threadIdx.x; it0 < N64; it0 += CLEAN_UNROLL * stride) {
        float4 vp[CLEAN_UNROLL], vc[CLEAN_UNROLL], vn[CLEAN_UNROLL];
#ifdef CLEAN_CLASS_WORD
        Tex9 tex[CLEAN_UNROLL]; int psx[CLEAN_UNROLL], psy[CLEAN_UNROLL];
#endif
        bool settled[CLEAN_UNROLL];   // a stable surfel outside the frustum, classified by the projection: kept, nothing read
        const bool cls = have_class && !cp.full_check;
#pragma unroll
        for (int k = 0; k < CLEAN_UNROLL; ++k) {
            const uint32_t it = it0 + (uint32_t)k * stride;
            const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            vc[k] = z4; vn[k] = z4;
#ifdef CLEAN_CLASS_WORD
            if (cls && cp.class_word) {   // one word: the class AND where the window starts — planes and window texels in ONE round
                const uint32_t wd = it < N ? cp.class_word[it] : CLEAN_CLASS_OUT_STABLE;
                const uint32_t cl = wd & 3u;
                settled[k] = cl == CLEAN_CLASS_OUT_STABLE;
                vp[k] = settled[k] ? z4 : m.p0[it];
                if (!settled[k]) vc[k] = m.p1[it];
                psx[k] = psy[k] = -1;
                if (cl == CLEAN_CLASS_IN_VIEW) {
                    vn[k] = m.p2[it];
                    if (cp.nw == 4) {
                        psx[k] = (int)((wd >> 2) & 0x1FFFu); psy[k] = (int)(wd >> 15);
#pragma unroll
                        for (int jx = 0; jx < 3; ++jx)
#pragma unroll
                            for (int jy = 0; jy < 3; ++jy) {   // the walk visits texels first .. first + 2 at most; clamped: an address inside the image
                                if (!CLEAN_PRE(jx, jy)) continue;
                                const int tx = psx[k] + jx < cp.cam.W ? psx[k] + jx : cp.cam.W - 1, ty = psy[k] + jy < cp.cam.H ? psy[k] + jy : cp.cam.H - 1;
                                tex[k].t[jx * 3 + jy] = clean_tex[clean_tex_slot(tx, ty, cp.cam.W)];
                            }
                    }
                }
            } else
#endif
            if (cls) {   // one byte decides what the item needs, and what it needs is requested in ONE round (position -> colour/time,
                         // normal/radius are two dependent rounds without the class)
                const uint32_t cl = it < N ? keep_flags[it] : CLEAN_CLASS_OUT_STABLE;
                settled[k] = cl == CLEAN_CLASS_OUT_STABLE;
                vp[k] = settled[k] ? z4 : m.p0[it];
                if (!settled[k]) vc[k] = m.p1[it];
                if (cl == CLEAN_CLASS_IN_VIEW) vn[k] = m.p2[it];
            } else {
                settled[k] = false;
                vp[k] = it < N ? m.p0[it] : z4;
            }
        }
#pragma unroll
        for (int k = 0; k < CLEAN_UNROLL; ++k) {
            const uint32_t it = it0 + (uint32_t)k * stride;
            if (it >= N64) break;          // wave-uniform: N64 and the strides are multiples of 64
#ifdef CLEAN_CLASS_WORD
            const bool havew = cls && cp.class_word;
            const bool keep = it < N && (settled[k] || clean_item(cp, tinv, ftime, m, rec, true, it, clean_tex, vp[k], cls, vc[k], vn[k],
                                                                  tex[k], havew ? psx[k] : -1, havew ? psy[k] : -1));
#else
            const bool keep = it < N && (settled[k] || clean_item(cp, tinv, ftime, m, rec, true, it, clean_tex, vp[k], cls, vc[k], vn[k]));
#endif
#ifndef CLEAN_NO_STORE
            if (it < N) keep_flags[it] = keep ? 1 : 0;
#endif
            // 64 consecutive items share a tile (FUSE_TILE % 64 == 0): one atomic per wave feeds the tile count
            const unsigned long long bal = __ballot(keep);
            if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&tile_count[(size_t)(it / FUSE_TILE) * TC_STRIDE], (uint32_t)__popcll(bal));
        }
    }
}

// Pass B: in-place leftward move of the survivors + append of the records (see header comment).
// A tile's output offset = sum of the keep counts of all earlier tiles, summed directly by the tile
// (<= a few thousand L2-resident words staged in LDS): no scan kernel, no look-back chain.
// Forward progress does NOT depend on co-residency: tiles are handed out by an atomic ticket in increasing order, so
// a tile a writer waits for (tile_done of a tile its output range overlaps, always a LOWER tile) was claimed earlier
// by a workgroup that is already running, and raising tile_done needs no wait at all (it follows the tile's own
// loads) -> no wait can cycle, whatever else occupies the device (other contexts, RCCL kernels, a smaller grid).
// tile_done holds the EPOCH of the pass that raised it (a per-shard launch counter), the tile counters are double
// buffered (this pass zeroes the buffer the next pass accumulates into) and the record flags are re-armed by the
// lanes that move the records: the pass needs no re-arm kernel.
// Tiles before the first one whose survivors change place (first tile that is not completely kept, or the tile that
// holds the end of the map) are never touched and never written into: the ticket starts there.
struct MoveSlot { float4 a, b, c, d, e; uint32_t it, o, g; bool keep; };
__device__ __forceinline__ void move_load(MoveSlot &sl, const MapPlanes &m, const RecPlanes &rec, uint32_t N, float ftime,
                                          const uint32_t *__restrict__ gid, uint32_t g_base)
{
    if (!sl.keep) return;
    if (gid) sl.g = sl.it < N ? gid[sl.it] : g_base + (sl.it - N);   // hash ownership: the surfel's place in the global order moves along
    // (non-temporal loads / stores measured: 195 vs 187 us for the whole pass at 4.3 M surfels — plain accesses stay)
    if (sl.it < N) { sl.a = m.p0[sl.it]; sl.b = m.p1[sl.it]; sl.c = m.p2[sl.it]; sl.d = m.p3[sl.it]; sl.e = m.p4[sl.it]; }
    else {
        const uint32_t q = sl.it - N;
        const float4 ct = rec.p1[q];
        sl.a = rec.p0[q]; sl.c = rec.p2[q]; sl.d = rec.p3[q]; sl.e = rec.p4[q];
        sl.b = make_float4(ct.x, ct.y, ct.z, ct.w == -2.0f ? ftime : ct.w);     // copy_unstable.vert:155-158
    }
}
__device__ __forceinline__ uint32_t move_store(const MoveSlot &sl, const MapPlanes &m, uint32_t N, uint32_t cap, uint32_t *__restrict__ gid)
{
    if (!sl.keep || sl.o >= cap) return 0u;
    if (sl.o != sl.it || sl.it >= N) {
        m.p0[sl.o] = sl.a; m.p1[sl.o] = sl.b; m.p2[sl.o] = sl.c; m.p3[sl.o] = sl.d; m.p4[sl.o] = sl.e;
        if (gid) gid[sl.o] = sl.g;
    }
    return sl.it >= N ? 1u : 0u;
}

__global__ __launch_bounds__(FUSE_THREADS, FUSE_WAVES_PER_EU) void k_fuse_stream(int time, MapPlanes m, RecPlanes rec, int Q,
                                                              const uint8_t *__restrict__ keep_flags,
                                                              const uint32_t *__restrict__ tile_count,
                                                              const uint32_t *__restrict__ count_in,
                                                              uint32_t *__restrict__ count_out,
                                                              uint32_t *__restrict__ stats, uint32_t cap,
                                                              uint32_t *__restrict__ tile_done, uint32_t epoch,
                                                              uint32_t lds_tiles, uint32_t *__restrict__ tile_count_next,
                                                              uint32_t nzero_next, int32_t *__restrict__ rec_flag_rearm,
                                                              uint32_t *__restrict__ gid, uint32_t g_base)
{
    constexpr int NWAVE = FUSE_THREADS / 64;
    __shared__ uint32_t s_wcnt[FUSE_IPT][NWAVE];
    __shared__ uint32_t s_psum[NWAVE];
    __shared__ uint32_t s_first[NWAVE];
    __shared__ uint32_t s_ticket;
    const uint32_t N = *count_in;
    const uint32_t total = N + (uint32_t)Q;
    const uint32_t num_tiles = (total + FUSE_TILE - 1) / FUSE_TILE;
    const uint32_t surfel_tiles = (N + FUSE_TILE - 1) / FUSE_TILE;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float ftime = (float)time;
    // the other tile-counter buffer is the one the NEXT pass accumulates into: zero what its last use left behind
    for (uint32_t t = blockIdx.x * FUSE_THREADS + threadIdx.x; t < nzero_next; t += gridDim.x * FUSE_THREADS)
        tile_count_next[(size_t)t * TC_STRIDE] = 0u;
    if (num_tiles == 0) {   // an empty shard that takes no appends this frame
        if (blockIdx.x == 0 && threadIdx.x == 0) { *count_out = 0; stats[0] = 0; stats[3] = 0; }
        return;
    }
    if (num_tiles > lds_tiles) {
        // the host's bound on the item count was too small (cannot happen unless a count read-back is mis-tracked):
        // refuse loudly instead of corrupting — the map is left as it is for this frame, the status bit is raised, and
        // the counters pass A accumulated beyond what the host will re-arm are cleared here
        for (uint32_t t = blockIdx.x * FUSE_THREADS + threadIdx.x; t < num_tiles; t += gridDim.x * FUSE_THREADS)
            const_cast<uint32_t *>(tile_count)[(size_t)t * TC_STRIDE] = 0u;
        if (blockIdx.x == 0 && threadIdx.x == 0) { *count_out = N < cap ? N : cap; stats[0] = N; stats[3] = N; atomicOr(&stats[7], 1u); }
        return;
    }

    // every tile count is staged in LDS once (one round of global loads per workgroup); the per-tile prefixes are
    // then summed out of LDS
    extern __shared__ uint32_t s_cnt[];
    uint32_t first = num_tiles - 1;   // the last tile always "moves" (it holds the end of the map and/or the records)
    for (uint32_t t = threadIdx.x; t < num_tiles; t += FUSE_THREADS) {
        const uint32_t c = tile_count[(size_t)t * TC_STRIDE];
        s_cnt[t] = c;
        if ((c != FUSE_TILE || (t + 1u) * FUSE_TILE > N) && t < first) first = t;
    }
    for (int d = 32; d > 0; d >>= 1) { const uint32_t o = __shfl_down(first, d); first = o < first ? o : first; }
    if (lane == 0) s_first[wid] = first;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) first = s_first[w] < first ? s_first[w] : first;
    first = __builtin_amdgcn_readfirstlane(first);
    // Only as many workgroups as there are tiles to process draw tickets (the ticket word is one address: every draw is
    // serialised at the memory side); the others leave.  Those that stay still take tiles dynamically, so whichever of
    // them runs makes progress — no assumption about which workgroups are resident.
    if (blockIdx.x >= num_tiles - first) return;

    uint32_t done_upto = first, prefix = first * FUSE_TILE;   // every tile before `first` is full: prefix(first) = first * TILE
    // statistics are accumulated per lane over all tiles of the workgroup and added ONCE at the end: same-address
    // atomics are serialised at the memory side (~8 ns each) and one per wave and tile made the pass atomic-bound
    uint32_t acc_appended = 0, acc_moved = 0;
    for (;;) {
        // (the next ticket drawn one tile ahead, so that the draw travels under this tile's loads, measured slower twice: with 4 items
        //  per lane in round 2, with 1 item per lane in round 3 — 34.5 vs 32.2 us headline, 176 vs 172 us at 4.3 M surfels)
        if (threadIdx.x == 0) s_ticket = atomicAdd(&stats[5], 1u);
        __syncthreads();
        const uint32_t tile = first + s_ticket;
        if (tile >= num_tiles) break;
        const uint32_t base = tile * FUSE_TILE;
        {   // prefix of this tile = prefix of the previous tile this workgroup took + the counts in between (tickets grow)
            uint32_t psum = 0;
            for (uint32_t t = done_upto + threadIdx.x; t < tile; t += FUSE_THREADS) psum += s_cnt[t];
            for (int d = 32; d > 0; d >>= 1) psum += __shfl_down(psum, d);
            if (lane == 0) s_psum[wid] = psum;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) prefix += s_psum[w];
            done_upto = tile;
        }
        const uint32_t tile_total = s_cnt[tile];
        if (tile == num_tiles - 1 && threadIdx.x == 0) {
            const uint32_t tot = prefix + tile_total;
            *count_out = tot > cap ? cap : tot;
            stats[0] = N; stats[3] = tot > cap ? cap : tot;
            if (tot > cap) atomicOr(&stats[7], 2u);   // capacity reached: surfels were dropped (HRBF_ERR_CAPACITY on the next blocking call)
        }
        const bool tile_has_surfels = base < N;
        const uint32_t n_surf_here = N > base ? (N - base < FUSE_TILE ? N - base : FUSE_TILE) : 0u;
        const bool moves = (prefix != base) || (tile_total != n_surf_here) || (base + FUSE_TILE > N);
        if (!moves) {   // nothing in this tile changes place: no load, no store
            if (threadIdx.x == 0)
                __hip_atomic_store(&tile_done[tile], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();   // s_ticket / s_psum reuse
            continue;
        }
        // FUSE_IPT slots, written out by a macro list: an array of MoveSlot went to scratch (64 B / lane) and cost 30 us at 4.3 M surfels
#if FUSE_IPT == 1
#define FUSE_SLOTS(X) X(0)
#elif FUSE_IPT == 2
#define FUSE_SLOTS(X) X(0) X(1)
#elif FUSE_IPT == 4
#define FUSE_SLOTS(X) X(0) X(1) X(2) X(3)
#else
#error "FUSE_IPT must be 1, 2 or 4"
#endif
        const unsigned long long lt = (1ull << lane) - 1ull;
#define X(k) MoveSlot s##k; s##k.it = base + (uint32_t)(k) * FUSE_THREADS + threadIdx.x; s##k.keep = s##k.it < total && keep_flags[s##k.it] != 0; \
             const unsigned long long b##k = __ballot(s##k.keep); if (lane == 0) s_wcnt[k][wid] = (uint32_t)__popcll(b##k);
        FUSE_SLOTS(X)
#undef X
        __syncthreads();
        {   // item order inside the tile: k-major, then wave, then lane
            uint32_t run = prefix;
#define X(k) { uint32_t o = 0; _Pragma("unroll") for (int w = 0; w < NWAVE; ++w) { if (w == wid) o = run; run += s_wcnt[k][w]; } s##k.o = o + (uint32_t)__popcll(b##k & lt); }
            FUSE_SLOTS(X)
#undef X
        }
#define X(k) move_load(s##k, m, rec, N, ftime, gid, g_base);
        FUSE_SLOTS(X)
#undef X
        // every load this tile will ever issue on the map has returned -> publish tile_done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile_has_surfels && threadIdx.x == 0)
            __hip_atomic_store(&tile_done[tile], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // wait until the source tiles our output range overlaps have finished reading (all of them were claimed before
        // this one; tiles before `first` are never claimed and never written into: prefix >= first * TILE)
        if (tile_total > 0 && threadIdx.x < 64) {
            const uint32_t t_lo = prefix / FUSE_TILE;
            const uint32_t t_hi = (prefix + tile_total - 1) / FUSE_TILE;
            for (uint32_t t = t_lo + (uint32_t)lane; t <= t_hi && t < surfel_tiles && t < tile; t += 64) {
                // every spin is bounded (MI355X_MICROARCH.md, correctness boundaries): the wait cannot cycle by
                // construction, but a protocol bug must surface as a status bit, never as a hung device
                uint32_t spins = 0;
                while (__hip_atomic_load(&tile_done[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 24)) { atomicOr(&stats[7], 4u); break; }
                }
            }
        }
        __syncthreads();
#define X(k) acc_appended += move_store(s##k, m, N, cap, gid);
        FUSE_SLOTS(X)
#undef X
        if (base + FUSE_TILE > N && rec_flag_rearm) {   // the lanes that handled records re-arm the record flags
#define X(k) if (s##k.it >= N && s##k.it < total) rec_flag_rearm[s##k.it - N] = 0;
            FUSE_SLOTS(X)
#undef X
        }
        // statistics: surfels that really changed slot (real traffic = 160 B each)
#define X(k) acc_moved += (uint32_t)(s##k.keep && s##k.it < N && s##k.o != s##k.it);
        FUSE_SLOTS(X)
#undef X
        __syncthreads();   // s_wcnt / s_psum / s_ticket reuse
    }
    // one atomic per workgroup and statistic (s_psum / s_first are free again: every path above ended with a barrier)
    for (int d = 32; d > 0; d >>= 1) { acc_appended += __shfl_down(acc_appended, d); acc_moved += __shfl_down(acc_moved, d); }
    if (lane == 0) { s_psum[wid] = acc_appended; s_first[wid] = acc_moved; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, mv = 0;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) { a += s_psum[w]; mv += s_first[w]; }
        if (a) atomicAdd(&stats[2], a);
        if (mv) atomicAdd(&stats[6], mv);
    }
}

// instrumentation only (timing ring on): park the pass's item statistics {in, merged, appended, out, -, -, moved, status}
__global__ void k_copy_stats(const uint32_t *__restrict__ stats, uint32_t *__restrict__ slot,
                             const uint32_t *__restrict__ merged_part, int nparts)
{
    if (threadIdx.x < 8 && threadIdx.x != 1) slot[threadIdx.x] = stats[threadIdx.x];
    uint32_t t = 0;   // merged = sum of k_apply_merges' per-workgroup words (one wave)
    for (int i = threadIdx.x; i < nparts; i += 64) t += merged_part[i];
    for (int d = 32; d > 0; d >>= 1) t += __shfl_down(t, d);
    if (threadIdx.x == 0) slot[1] = t;
}
__global__ void k_zero_i32(int32_t *p, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
__global__ void k_fill_u32(uint32_t *p, size_t n, uint32_t v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------ launchers
void launch_initialise(hipStream_t s, const Cam &cam, const DevPose *dp, const float4 *vertex_raw, const float4 *normal,
                       const uint8_t *rgb, const float4 *curv1, const float4 *curv2, const float *gradmag,
                       int use_conf_eval, float eps, float thr, uint32_t *flags, uint32_t *offs, MapPlanes out,
                       uint32_t cap, uint32_t *count, uint32_t *status)
{
    int P = cam.W * cam.H;
    hipLaunchKernelGGL(k_init_flags, dim3((P + 255) / 256), dim3(256), 0, s, cam, dp, normal, curv1, curv2, thr, flags);
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, s, flags, offs, P, count);
    hipLaunchKernelGGL(k_init_scatter, dim3((P + 255) / 256), dim3(256), 0, s, cam, dp, vertex_raw, normal, rgb, curv1,
                       curv2, gradmag, use_conf_eval, eps, flags, offs, out, cap, (const uint32_t *)nullptr, (uint32_t *)nullptr);
    hipLaunchKernelGGL(k_clamp_count, dim3(1), dim3(1), 0, s, count, cap, status);
}
// hash ownership: flags / offs rank the seed frame's surfels in the global order (the same on every shard), flags2 / offs2 the
// ones this shard keeps; *total receives the number of surfels of the whole seed (the next free global id)
void launch_initialise_hashed(hipStream_t s, const Cam &cam, const DevPose *dp, const float4 *vertex_raw, const float4 *normal,
                              const uint8_t *rgb, const float4 *curv1, const float4 *curv2, const float *gradmag,
                              int use_conf_eval, float eps, float thr, uint32_t *flags, uint32_t *offs, uint32_t *flags2,
                              uint32_t *offs2, MapPlanes out, uint32_t *gid_out, uint32_t cap, uint32_t *count, uint32_t *total,
                              uint32_t *status, int G, int me, float inv_cell)
{
    int P = cam.W * cam.H;
    hipLaunchKernelGGL(k_init_flags, dim3((P + 255) / 256), dim3(256), 0, s, cam, dp, normal, curv1, curv2, thr, flags);
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, s, flags, offs, P, total);
    hipLaunchKernelGGL(k_init_owner, dim3((P + 255) / 256), dim3(256), 0, s, cam, dp, vertex_raw, flags, flags2, G, me, inv_cell);
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, s, flags2, offs2, P, count);
    hipLaunchKernelGGL(k_init_scatter, dim3((P + 255) / 256), dim3(256), 0, s, cam, dp, vertex_raw, normal, rgb, curv1,
                       curv2, gradmag, use_conf_eval, eps, flags2, offs2, out, cap, (const uint32_t *)offs, gid_out);
    hipLaunchKernelGGL(k_clamp_count, dim3(1), dim3(1), 0, s, count, cap, status);
}
__global__ __launch_bounds__(256) void k_gid_rank(GidPtrs g, const uint32_t *__restrict__ counts, int G, const uint32_t *__restrict__ mine,
                                                  uint32_t n, uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = mine[i];
    uint32_t r = 0;
    for (int k = 0; k < G; ++k) {
        uint32_t lo = 0, hi = counts[k];
        const uint32_t *p = g.p[k];
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (p[mid] < v) lo = mid + 1; else hi = mid; }
        r += lo;
    }
    out[i] = r;
}
void launch_gid_rank(hipStream_t s, const uint32_t *const *ptrs, const uint32_t *counts, int G, const uint32_t *mine, uint32_t n, uint32_t *out)
{
    if (!n) return;
    GidPtrs g;
    for (int k = 0; k < 8; ++k) g.p[k] = k < G ? ptrs[k] : nullptr;
    hipLaunchKernelGGL(k_gid_rank, dim3((n + 255) / 256), dim3(256), 0, s, g, counts, G, mine, n, out);
}
__global__ void k_min_row_u32(const uint32_t *__restrict__ row, int n, uint32_t *__restrict__ out)
{
    uint32_t m = HRBF_NO_SURFEL;
    for (int k = 0; k < n; ++k) m = row[k] < m ? row[k] : m;
    *out = m;
}
void launch_min_row_u32(hipStream_t s, const uint32_t *row, int n, uint32_t *out)
{
    hipLaunchKernelGGL(k_min_row_u32, dim3(1), dim3(1), 0, s, row, n, out);
}
void launch_iota_u32(hipStream_t s, uint32_t *p, uint32_t n, uint32_t base)
{
    if (n) hipLaunchKernelGGL(k_iota_u32, dim3((n + 255) / 256), dim3(256), 0, s, p, n, base);
}

// ---- sharded map over peer-mapped images (SURVEY §8e sharding 2): the OWNER of a pixel's winner writes the
// winner's attributes straight into every rank's index-map images (hipIpcMemHandle-mapped; xGMI peers on one node) — no
// packing, no count exchange, no host read-back.  A pixel has exactly one owner, so the writes never collide; a pixel nobody
// hit is zeroed by every rank locally.  The clean texel is written as every other path writes it (clean_texel_pack, 4 x 2 blocks).
__global__ __launch_bounds__(256) void k_zbuf_min_peers(PeerImages pi, unsigned long long *__restrict__ zred, int P)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    unsigned long long k = ZB_EMPTY;
    for (int g = 0; g < pi.world; ++g) { const unsigned long long v = pi.zbuf[g][i]; k = v < k ? v : k; }
    zred[i] = k;
}
__global__ __launch_bounds__(256) void k_resolve_scatter(Cam cam, const DevPose *__restrict__ dp, MapPlanes m, ShardRef sh,
                                                         const unsigned long long *__restrict__ zred, uint32_t *__restrict__ idx,
                                                         PeerImages pi, int what, float clean_conf_thr, int clean_time,
                                                         unsigned long long *__restrict__ zpriv)
{
    const int P = cam.W * cam.H;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const unsigned long long key = zred[i];
    const float4 z4 = make_float4(0, 0, 0, 0);
    if (key == ZB_EMPTY) {   // nobody's surfel: every rank clears its own texels
        idx[i] = 0u;
        if (sh.gid) (void)pixel_winner(sh, key, zpriv, i);   // re-arms the private z-buffer, clears own_local
        const int g = pi.me;
        if (what & RESOLVE_GEOM) { pi.vertconf[g][i] = z4; pi.normrad[g][i] = z4; }
        if (what & RESOLVE_ATTR) { pi.colortime[g][i] = z4; pi.curvmax[g][i] = z4; pi.curvmin[g][i] = z4; }
        if (what & RESOLVE_CLEAN) pi.clean[g][clean_tex_slot_of_pixel((uint32_t)i, cam.W)] = z4;
        return;
    }
    const PixelWinner w = pixel_winner(sh, key, zpriv, i);
    const uint32_t sg = w.id, s = w.s;
    idx[i] = sg;
    if (!w.owned) return;   // another rank owns the winner and writes this pixel
    WinnerTexels o;
    const bool updated = resolve_winner(m, s, sg, dp->tinv, what, clean_conf_thr, clean_time, o);
    o.clean = clean_texel_pack(o.clean, updated);
    const uint32_t ci = clean_tex_slot_of_pixel((uint32_t)i, cam.W);
    for (int g = 0; g < pi.world; ++g) {
        if (what & RESOLVE_GEOM) { pi.vertconf[g][i] = o.vc; pi.normrad[g][i] = o.nr; }
        if (what & RESOLVE_ATTR) { pi.colortime[g][i] = o.ct; pi.curvmax[g][i] = o.c1; pi.curvmin[g][i] = o.c2; }
        if (what & RESOLVE_CLEAN) pi.clean[g][ci] = o.clean;
    }
}
void launch_zbuf_min_peers(hipStream_t s, const PeerImages &pi, unsigned long long *zred, int P)
{
    hipLaunchKernelGGL(k_zbuf_min_peers, dim3((P + 255) / 256), dim3(256), 0, s, pi, zred, P);
}
void launch_resolve_scatter(hipStream_t s, const Cam &cam, const DevPose *dp, MapPlanes m, ShardRef sh, const unsigned long long *zred,
                            uint32_t *idx, const PeerImages &pi, int what, bool for_clean, float clean_conf_thr, int clean_time,
                            unsigned long long *zpriv)
{
    const int P = cam.W * cam.H;
    if (!for_clean) what &= ~RESOLVE_CLEAN;
    hipLaunchKernelGGL(k_resolve_scatter, dim3((P + 255) / 256), dim3(256), 0, s, cam, dp, m, sh, zred, idx, pi, what, clean_conf_thr, clean_time, zpriv);
}

void launch_project(hipStream_t s, const Cam &cam, const DevPose *dp, float maxDepth, MapPlanes m, ShardRef sh,
                    uint32_t count_ub, unsigned long long *zbuf, const uint8_t *submap_active, int n_active,
                    uint8_t *item_class, float confThr, uint32_t *item_word, float wm)
{
    uint32_t blocks = (count_ub + 255) / 256;   // zbuf is ZB_EMPTY on entry: launch_zbuf_reset once, k_resolve afterwards
    if (blocks > 256 * 8) blocks = 256 * 8;   // 8 blocks per CU, grid-stride the rest
    if (blocks == 0) blocks = 1;
#ifdef CLEAN_CLASS_WORD
    hipLaunchKernelGGL(k_project, dim3(blocks), dim3(256), 0, s, cam, dp, maxDepth, m.p0, sh, zbuf, m.p1, submap_active,
                       n_active, item_class, confThr, item_class ? item_word : nullptr, wm);
#else
    (void)item_word; (void)wm;
    hipLaunchKernelGGL(k_project, dim3(blocks), dim3(256), 0, s, cam, dp, maxDepth, m.p0, sh, zbuf, m.p1, submap_active,
                       n_active, item_class, confThr);
#endif
}
void launch_resolve(hipStream_t s, const Cam &cam, const DevPose *dp, MapPlanes m, ShardRef sh, unsigned long long *zbuf,
                    uint32_t *idx, float4 *vertconf, float4 *colortime, float4 *normrad, float4 *curvmax, float4 *curvmin,
                    float4 *clean_tex, int what, int rearm, float clean_conf_thr, int clean_time, uint32_t *rec_count,
                    uint32_t *rec_idx, float4 *rec_f, uint32_t rec_cap, int dense, unsigned long long *zpriv)
{
    const int P = cam.W * cam.H;
    if (!clean_tex) what &= ~RESOLVE_CLEAN;
    WinnerRecords rec = {rec_count, rec_idx, rec_f, rec_cap};
    hipLaunchKernelGGL(k_resolve, dim3((P + 255) / 256), dim3(256), 0, s, cam, dp, m, sh, rearm, zbuf, idx, vertconf,
                       colortime, normrad, curvmax, curvmin, clean_tex, what, clean_conf_thr, clean_time, rec, dense, zpriv);
}
void launch_winner_unpack(hipStream_t s, int P, int W, const uint32_t *count, uint32_t first, uint32_t n_ub, const uint32_t *ridx,
                          const float4 *rf, uint32_t cap, int what, float4 *vertconf, float4 *colortime, float4 *normrad,
                          float4 *curvmax, float4 *curvmin, float4 *clean_tex)
{
    if (!clean_tex) what &= ~RESOLVE_CLEAN;
    uint32_t blocks = (n_ub + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_winner_unpack, dim3(blocks), dim3(256), 0, s, P, W, count, n_ub, first, ridx, rf, cap, what, vertconf, colortime,
                       normrad, curvmax, curvmin, clean_tex);
}

// local stand-ins for the two collectives of a sharded projection (one process playing several shards): the same
// reductions RCCL performs between ranks, so the zero-filling / ownership logic is exercised on one GPU
__global__ void k_zbuf_min_merge(unsigned long long *__restrict__ dst, unsigned long long *__restrict__ src, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long a = dst[i], b = src[i];
    if (b < a) dst[i] = b;
    if (b != ZB_EMPTY) src[i] = ZB_EMPTY;
}
void launch_zbuf_min_merge(hipStream_t s, unsigned long long *dst, unsigned long long *src_reset, int P)
{
    hipLaunchKernelGGL(k_zbuf_min_merge, dim3((P + 255) / 256), dim3(256), 0, s, dst, src_reset, P);
}

void launch_fuse(hipStream_t s, const Cam &cam, const DevPose *dp, int tick, float maxDepth, int index_submap,
                 const float *depth_metric, const float4 *normal_pca, const float4 *curv1, const float4 *curv2,
                 const float *confidence, const uint8_t *rgb, const uint32_t *idx, const float4 *vertconf,
                 const float4 *normrad, RecPlanes rec, int32_t *rec_flag, uint32_t *rec_best, uint32_t *slot,
                 MapPlanes m, ShardRef sh, uint32_t *stats, float curvThr, hipEvent_t m0, hipEvent_t m1, uint32_t *merged_part,
                 RecNormalSrc rn)
{
    int Q = (cam.W / 2) * (cam.H / 2);
    hipLaunchKernelGGL(k_associate, dim3(quarter_tile_blocks(cam.W, cam.H)), dim3(256), 0, s, cam, dp, tick, maxDepth, index_submap,
                       depth_metric, normal_pca, curv1, curv2, confidence, rgb, idx, vertconf, normrad, rec, rec_flag,
                       rec_best, slot, sh, stats, rn);
    if (m0) hipEventRecord(m0, s);   // F2 (update.vert) is part of the roofline-timed fuse: SURVEY §8d "F2+F3"
    hipLaunchKernelGGL(k_apply_merges, dim3(merge_workgroups(Q)), dim3(MERGE_THREADS), 0, s, Q, tick, rec, rec_flag, rec_best, slot, m,
                       sh, stats + 1, curvThr, merged_part);
    if (m1) hipEventRecord(m1, s);
}

void launch_clean(hipStream_t s, const Cam &cam, const DevPose *dp, float maxDepth, float confThr, float curvThr,
                  int time, float clean_window_multiplier, int full_check, MapPlanes m, RecPlanes rec, int32_t *rec_flag,
                  const uint32_t *count_in, uint32_t *count_out, uint32_t count_ub, uint32_t *stats, uint32_t cap,
                  const float4 *clean_tex, uint8_t *keep_flags, uint32_t *tile_count, uint32_t *tile_count_next,
                  uint32_t *tile_dirty /* [2]: entries this / the other buffer may hold */, uint32_t *tile_done, uint32_t epoch,
                  uint32_t max_tiles, hipEvent_t e0, hipEvent_t e1, const uint8_t *submap_active, int n_active,
                  int n_records, int zero_records, uint32_t *stats_ring_slot, const uint32_t *merged_part,
                  uint32_t *gid, uint32_t g_base, int hash_G, int hash_me, float hash_inv_cell, int have_class, const uint32_t *class_word)
{
    const int Qfull = (cam.W / 2) * (cam.H / 2);
    const int Q = n_records;   // records are appended by one shard only (the end of the global order)
    CleanParams cp;
    cp.cam = cam; cp.dp = dp; cp.maxDepth = maxDepth; cp.confThr = confThr; cp.curvThr = curvThr; cp.time = time;
    cp.nw = (int)ceilf(2.0f * clean_window_multiplier); cp.w0 = clean_window_multiplier * 0.5f; cp.wm = clean_window_multiplier; cp.full_check = full_check;
    cp.submap_active = submap_active; cp.n_active = n_active;
    cp.hash_G = gid ? hash_G : 1; cp.hash_me = hash_me; cp.hash_inv_cell = hash_inv_cell;
    cp.class_word = have_class ? class_word : nullptr;
    const uint32_t items_ub = count_ub + (uint32_t)Q;
    uint32_t tiles = (items_ub + FUSE_TILE - 1) / FUSE_TILE;
    if (tiles > max_tiles) tiles = max_tiles;
    if (e0) hipEventRecord(e0, s);
    uint32_t fblocks = (count_ub + 255) / 256;            // surfel workgroups (grid-stride) ...
    if (fblocks > 256u * 16u) fblocks = 256u * 16u;
    if (fblocks == 0) fblocks = 1;
    if (Q > 0) fblocks += quarter_tile_blocks(cam.W, cam.H);   // ... behind the record workgroups
    hipLaunchKernelGGL(k_clean_flags, dim3(fblocks), dim3(256), 0, s, cp, m, rec, rec_flag, Q, count_in, clean_tex,
                       keep_flags, tile_count, stats, have_class);
    // any grid size is safe (ticketed tiles); one FUSE_THREADS-thread workgroup per CU, one item per lane (more loads in
    // flight per lane cost bandwidth on this part)
    uint32_t blocks = tiles < 256u * FUSE_WG_PER_CU ? tiles : 256u * FUSE_WG_PER_CU;
    if (blocks == 0) blocks = 1;
    const size_t lds = sizeof(uint32_t) * (size_t)(tiles ? tiles : 1);
    // maps beyond ~12 M surfels per shard need more than the default 48 KB of dynamic LDS (one word per 1024-item tile; the part allows a
    // workgroup all 160 KiB: ~41 M surfels per shard, beyond which the launch fails and the frame reports HRBF_ERR_DEVICE).  The attribute belongs to the current
    // DEVICE's copy of the function, so it is set whenever it is needed (a process-wide "already raised" flag skipped it for a
    // second context on another device, whose launch then failed): a host-side call per frame, only for such maps
    if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_fuse_stream), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_fuse_stream, dim3(blocks), dim3(FUSE_THREADS), lds, s, time, m, rec, Q, keep_flags, tile_count,
                       count_in, count_out, stats, cap, tile_done, epoch, tiles, tile_count_next, tile_dirty[1],
                       (zero_records && Q > 0) ? rec_flag : nullptr, gid, g_base);
    if (e1) hipEventRecord(e1, s);
    tile_dirty[0] = tiles; tile_dirty[1] = 0;   // this buffer now holds `tiles` counts, the other one is clean
    if (zero_records && Q == 0)   // a rank of a sharded map that takes no appends still re-arms its (replicated) record flags
        hipLaunchKernelGGL(k_zero_i32, dim3((Qfull + 255) / 256), dim3(256), 0, s, rec_flag, Qfull);
    if (stats_ring_slot)
        hipLaunchKernelGGL(k_copy_stats, dim3(1), dim3(64), 0, s, stats, stats_ring_slot, merged_part, (int)merge_workgroups(Qfull));
}

void launch_zbuf_reset(hipStream_t s, unsigned long long *zbuf, int P)
{
    hipLaunchKernelGGL(k_fill_u64, dim3((P + 255) / 256), dim3(256), 0, s, zbuf, P, ZB_EMPTY);
}
void launch_fill_u32(hipStream_t s, uint32_t *p, size_t n, uint32_t v)
{
    hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}

uint32_t fuse_tile_items() { return FUSE_TILE; }
uint32_t merge_workgroups(int Q) { return (uint32_t)((Q + MERGE_THREADS - 1) / MERGE_THREADS); }
size_t clean_tex_elems(int P) { return clean_tex_float4s(P); }
uint32_t fuse_tile_count_stride() { return TC_STRIDE; }

// ------------------------------------------------------------------------------------------------
// GlobalModel::updateModel (GlobalModel.cpp:690-767 -> update_delta_trans.vert:41-104): every surfel is moved by the
// rigid correction of its submap.  In place on the SoA planes: the reference streams the whole 80-byte record through
// transform feedback into the other buffer (160 B/surfel); here colour/time is read for the submap id and only the two
// planes that change are rewritten (48 B read + 32 B written per surfel).  The <= 1200 matrices sit in L2 / K$.
__global__ __launch_bounds__(256) void k_update_model(MapPlanes m, const uint32_t *__restrict__ count,
                                                      const float *__restrict__ delta /* n x 16, column-major */, int n)
{
    const uint32_t N = *count;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < N; s += gridDim.x * blockDim.x) {
        const uint32_t sm = hd_cvt_u32(m.p1[s].y);
        if (sm >= (uint32_t)n) continue;   // texels the reference never uploaded: left alone (oracle/orc_map.c)
        const float *T = delta + (size_t)sm * 16;
        const float4 p = m.p0[s], nr = m.p2[s];
        float4 po, no;
        po.x = ((T[0] * p.x + T[4] * p.y) + T[8] * p.z) + T[12] * 1.0f;
        po.y = ((T[1] * p.x + T[5] * p.y) + T[9] * p.z) + T[13] * 1.0f;
        po.z = ((T[2] * p.x + T[6] * p.y) + T[10] * p.z) + T[14] * 1.0f;
        po.w = p.w;
        no.x = (T[0] * nr.x + T[4] * nr.y) + T[8] * nr.z;
        no.y = (T[1] * nr.x + T[5] * nr.y) + T[9] * nr.z;
        no.z = (T[2] * nr.x + T[6] * nr.y) + T[10] * nr.z;
        no.w = nr.w;
        m.p0[s] = po; m.p2[s] = no;
    }
}

void launch_update_model(hipStream_t s, MapPlanes m, const uint32_t *count, uint32_t count_ub, const float *delta, int n)
{
    uint32_t blocks = (count_ub + 255) / 256;
    if (blocks > 256u * 16u) blocks = 256u * 16u;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_update_model, dim3(blocks), dim3(256), 0, s, m, count, delta, n);
}

// Input upload as a kernel: reads the pinned staging slot over PCIe and writes the two device input images.  A copy
// engine transfer (hipMemcpyAsync) on the compute stream costs two cross-engine hand-overs per copy (~50-100 us each
// measured); a kernel on the same queue costs none.
__global__ void k_copy_inputs(const uint8_t *__restrict__ src_rgb, size_t nrgb, const uint8_t *__restrict__ src_dep,
                              size_t ndep, uint8_t *__restrict__ d_rgb, uint8_t *__restrict__ d_dep)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    const size_t v_rgb = nrgb / 16, v_dep = ndep / 16;
    for (size_t i = t; i < v_rgb; i += nt) reinterpret_cast<uint4 *>(d_rgb)[i] = reinterpret_cast<const uint4 *>(src_rgb)[i];
    for (size_t i = t; i < v_dep; i += nt) reinterpret_cast<uint4 *>(d_dep)[i] = reinterpret_cast<const uint4 *>(src_dep)[i];
    for (size_t i = v_rgb * 16 + t; i < nrgb; i += nt) d_rgb[i] = src_rgb[i];
    for (size_t i = v_dep * 16 + t; i < ndep; i += nt) d_dep[i] = src_dep[i];
}
void launch_copy_inputs(hipStream_t s, const uint8_t *src_rgb, size_t nrgb, const uint8_t *src_dep, size_t ndep,
                        uint8_t *d_rgb, uint8_t *d_dep)
{
    hipLaunchKernelGGL(k_copy_inputs, dim3(512), dim3(256), 0, s, src_rgb, nrgb, src_dep, ndep, d_rgb, d_dep);
}

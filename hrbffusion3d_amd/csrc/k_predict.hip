// k_predict.hip — HRBF ray-cast surface prediction + fill-in for gfx950.
//
// Replaces IndexMap::predictHRBF (Core/src/IndexMap.cpp:413-518 -> predict_hrbf.frag:40-311,
// hrbfbase.glsl:20-34,126-166) and FillIn::{vertex,normal,curvature,image}
// (Core/src/Shaders/FillIn.cpp:93-297 -> fill_*.frag), Resize::vertex + denseEnough
// (HRBFFusion.cpp:974-988,1069-1070).
//
// One thread per pixel, 16x16 tiles.  The GLSL gathers up to 49 neighbours into five private
// vec4[100] arrays; here
//   * the per-texel part of the implicit (centre, support radius^2 and its reciprocal, 10 n) is staged ONCE
//     per tile in LDS (tile + 3-texel halo, 22x22 x 32 B = 15.5 KB), together with one validity bit per
//     texel (the acceptance test of predict_hrbf.frag:85-92 depends on the texel only);
//   * each thread walks the window in the shader's ring order with compile-time offsets and writes the LDS
//     addresses of its accepted neighbours to a private column of an LDS list (<= 49 x 2 B per thread), so
//     the <= 46 implicit evaluations per pixel are a plain counted loop over LDS: one 2-byte and two 16-byte
//     reads per neighbour, no table look-ups, no divisions.
#include "common.h"
#include "kernels.h"

#define TB 16
#define PR 3
#define PTW (TB + 2 * PR)
#define PNT (TB * TB)

struct alignas(16) PTexel { float px, py, pz, T2, sx, sy, sz, invT2; };

// ring visiting order of predict_hrbf.frag:75-80: rings i = 0..3, x offset outer, y offset inner,
// ring-border texels only
struct RingEntry { int dx, dy; bool newcol; };
constexpr RingEntry ring_entry(int slot)
{
    int n = 0;
    for (int i = 0; i <= 3; ++i)
        for (int dj = -i; dj <= i; ++dj) {
            bool first = true;
            for (int dk = -i; dk <= i; ++dk) {
                if (!(dj == -i || dk == -i || dj == i || dk == i)) continue;
                if (n == slot) return RingEntry{dj, dk, first};   // `first`: the inner (k) loop restarts here
                first = false;
                n++;
            }
        }
    return RingEntry{0, 0, false};
}

int predict_upload_tables() { return 0; }   // the ring table is a compile-time constant

// neighbour gathering with the reference's "break only the innermost loop" behaviour
template <int SLOT>
__device__ __forceinline__ void gather_ring(const uint32_t (&wrow)[7], int nslots, int maxn, uint32_t lbase_bytes,
                                            uint16_t *__restrict__ list, int &n, bool &skip)
{
    if constexpr (SLOT < 49) {
        constexpr RingEntry e = ring_entry(SLOT);
        if (SLOT < nslots) {
            if (e.newcol) skip = false;
            const bool acc = !skip && ((wrow[e.dy + 3] >> (e.dx + 3)) & 1u);
            if (acc) {
                list[n * PNT] = (uint16_t)(lbase_bytes + (uint32_t)((e.dy * PTW + e.dx) * (int)sizeof(PTexel)));
                n++;
                if (n > maxn) skip = true;
            }
        }
        gather_ring<SLOT + 1>(wrow, nslots, maxn, lbase_bytes, list, n, skip);
    }
}

__device__ __forceinline__ float4 lds4(const PTexel *tile, uint32_t byte_off)
{
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(tile) + byte_off);
}

// one neighbour of hrbfvalue (hrbfbase.glsl:126-145) with getWeightD (:20-34): branch-free, the contribution of a
// neighbour whose support does not reach p is computed and discarded
__device__ __forceinline__ void hrbf_value_term(const float4 a, const float4 b, const f3 p, bool live, float &value, int &ns)
{
    const float vx = p.x - a.x, vy = p.y - a.y, vz = p.z - a.z;
    const float d2 = (vx * vx + vy * vy) + vz * vz;
    const bool in = live && !(a.w < d2);
    const float r = hd_sqrtf(d2 * b.w);
    const float s = 1.0f - r;
    const float s3 = s * s * s;
    // getWeightD returns the zero vector at d2 == 0 (hrbfbase.glsl:24-27).  There v = p - centre is exactly (+0, +0, +0)
    // (x - x is +0 under round-to-nearest), so ONE select on the scalar factor gives the same three +0 products as
    // three selects on the products — and keeps the T = 0 case (factor = -inf) away from 0 * inf
    const float tt = d2 != 0.0f ? -20.0f * s3 * b.w : 0.0f;
    const float gx = vx * tt, gy = vy * tt, gz = vz * tt;
    const float c = (gx * b.x + gy * b.y) + gz * b.z;
    value = in ? value - c : value;
    ns += in ? 1 : 0;
}

// hrbfvalue over the gathered list, two neighbours per trip; the offsets of the next pair are fetched while the
// current pair is evaluated.  list[n .. n+2] hold a valid dummy offset.
__device__ __forceinline__ float hrbf_value(const PTexel *__restrict__ tile, const uint16_t *__restrict__ list, int n,
                                            f3 p, int &nsup)
{
    float value = 0.0f;
    int ns = 0;
    uint32_t o0 = list[0], o1 = list[PNT];
    for (int k = 0; k < n; k += 2) {
        const float4 a0 = lds4(tile, o0), b0 = lds4(tile, o0 + 16), a1 = lds4(tile, o1), b1 = lds4(tile, o1 + 16);
        o0 = list[(k + 2) * PNT]; o1 = list[(k + 3) * PNT];
        hrbf_value_term(a0, b0, p, true, value, ns);
        hrbf_value_term(a1, b1, p, k + 1 < n, value, ns);
    }
    nsup = ns;
    return value;
}

// hrbfgradient (hrbfbase.glsl:147-166) with getWeightH (:37-69)
__device__ __forceinline__ f3 hrbf_gradient(const PTexel *__restrict__ tile, const uint16_t *__restrict__ list, int n, f3 p)
{
    float grx = 0.0f, gry = 0.0f, grz = 0.0f;
    for (int k = 0; k < n; ++k) {
        const uint32_t o = list[k * PNT];
        const float4 a = lds4(tile, o), b = lds4(tile, o + 16);
        const float sx = b.x, sy = b.y, sz = b.z;
        const float vx = p.x - a.x, vy = p.y - a.y, vz = p.z - a.z;
        const float d2 = (vx * vx + vy * vy) + vz * vz;
        const float T2 = a.w;
        float h0, h1, h2, h4, h5, h8;
        if (d2 > T2) { h0 = h1 = h2 = h4 = h5 = h8 = 0.0f; }
        else if (d2 == 0.0f) { h0 = h4 = h8 = -20.0f / T2; h1 = h2 = h5 = 0.0f; }
        else {
            float r = hd_sqrtf(d2 / T2);
            float s = 1.0f - r;
            float s2 = s * s;
            float t1 = 20.0f * s2 / (T2 * T2 * r);
            float t2 = -r * s * T2;
            h0 = t1 * (3.0f * (vx * vx) + t2);
            h1 = t1 * 3.0f * vx * vy;
            h2 = t1 * 3.0f * vx * vz;
            h4 = t1 * (3.0f * (vy * vy) + t2);
            h5 = t1 * 3.0f * vy * vz;
            h8 = t1 * (3.0f * (vz * vz) + t2);
        }
        grx -= (sx * h0 + sy * h1) + sz * h2;
        gry -= (sx * h1 + sy * h4) + sz * h5;
        grz -= (sx * h2 + sy * h5) + sz * h8;
    }
    return mk3(grx, gry, grz);
}

// capacity of the per-thread neighbour list: once n exceeds maxn every later window column adds at most one
// entry (15 columns can still follow), plus three dummy entries behind the list
__host__ __device__ inline int predict_list_cap(int maxn)
{
    int c = maxn + 1 + 15;
    if (c > 49) c = 49;
    if (c < 1) c = 1;
    return c + 3;
}

__global__ __launch_bounds__(256) void k_predict_hrbf(Cam cam, const float4 *__restrict__ vertconf,
                                                      const float4 *__restrict__ normrad,
                                                      const float4 *__restrict__ colortime,
                                                      const float4 *__restrict__ curvmax,
                                                      const float4 *__restrict__ curvmin, int win, int minn, int maxn,
                                                      float cthr, float lambda, uint8_t *__restrict__ pr_image,
                                                      float4 *__restrict__ pr_vertex, float4 *__restrict__ pr_normal,
                                                      float4 *__restrict__ pr_curv1, float4 *__restrict__ pr_curv2,
                                                      uint32_t *__restrict__ pr_time, float *__restrict__ pr_icpw)
{
    __shared__ PTexel tile[PTW * PTW];
    __shared__ uint32_t s_rowbits[PTW];
    extern __shared__ uint16_t s_list[];   // predict_list_cap(maxn) x 256 entries
    const int W = cam.W, H = cam.H;
    const int bx = blockIdx.x * TB, by = blockIdx.y * TB;
    const int tid = threadIdx.y * TB + threadIdx.x;
    if (tid < PTW) s_rowbits[tid] = 0u;
    __syncthreads();
    for (int i = tid; i < PTW * PTW; i += PNT) {
        int tx = i % PTW, ty = i / PTW;
        int gx = bx + tx - PR, gy = by + ty - PR;
        PTexel t;
        t.px = t.py = t.pz = t.T2 = t.sx = t.sy = t.sz = t.invT2 = 0.0f;
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) {   // outside the image: rejected like `j < 0 || j > 1` (:85)
            float4 v = vertconf[gy * W + gx], n = normrad[gy * W + gx];
            t.px = v.x; t.py = v.y; t.pz = v.z;
            t.T2 = n.w * n.w; t.invT2 = 1.0f / t.T2;
            t.sx = 10.0f * n.x; t.sy = 10.0f * n.y; t.sz = 10.0f * n.z;
            if (!(v.z < 0.1f || len3(mk3(n.x, n.y, n.z)) < 0.1f || v.w < cthr || n.z < 0.0f))
                atomicOr(&s_rowbits[ty], 1u << tx);
        }
        tile[i] = t;
    }
    __syncthreads();
    const int px = bx + threadIdx.x, py = by + threadIdx.y;
    if (px >= W || py >= H) return;
    const int pi = py * W + px;
    const uint32_t lbase_bytes = (uint32_t)(((threadIdx.y + PR) * PTW + threadIdx.x + PR) * (int)sizeof(PTexel));
    uint16_t *list = s_list + tid;

    int n = 0;
    {
        uint32_t wrow[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) wrow[r] = s_rowbits[threadIdx.y + r] >> threadIdx.x;
        bool skip = false;
        gather_ring<0>(wrow, (2 * win + 1) * (2 * win + 1), maxn, lbase_bytes, list, n, skip);
        list[n * PNT] = list[(n + 1) * PNT] = list[(n + 2) * PNT] = (uint16_t)lbase_bytes;   // dummies
    }

    const float x = (float)px + 0.5f, y = (float)py + 0.5f;
    const float xl = (x - cam.cx) * cam.camz, yl = (y - cam.cy) * cam.camw;
    const f3 ray = normalize3(mk3(xl, yl, 1.0f));

    f3 closest = mk3(0, 0, 0);
    {
        float pmin = 1000000.0f;
        for (int k = 0; k < n; ++k) {
            const float4 a = lds4(tile, list[k * PNT]);
            float pj = hd_fabsf(dot3(mk3(a.x, a.y, a.z), ray));
            if (pj < pmin) { closest = scale3(ray, pj); pmin = pj; }
        }
    }

    // Ray march + bisection of predict_hrbf.frag:150-260 as one loop with a single evaluation site: every trip
    // each live lane evaluates the implicit at the next sample of ITS OWN phase, so a wave needs
    // max_lane(total samples) trips instead of sum_phase(max_lane(samples of the phase)).
    //   phase 0: value at `closest`            phase 1: 25 steps of 4 mm away from it until the sign flips
    //   phase 2: 10 steps of 0.4 mm back       phase 3: <= 10 bisections        phase 4: finished
    // The first coarse sample (i = 0) is `closest` itself: its value is v0, which cannot flip the sign, so the
    // march starts at i = 1.
    bool found = false;
    f3 sp = mk3(0, 0, 0), ep = mk3(0, 0, 0), p_temp = mk3(0, 0, 0), q = closest;
    int phase = n > minn ? 0 : 4, it = 0;
    bool pos = false;   // sign class of v0
    while (phase != 4) {
        int nsup;
        const float v = hrbf_value(tile, list, n, q, nsup);
        if (phase == 0) {
            if (nsup > minn) {
                pos = v > 0.0f;
                if (pos) ep = closest; else sp = closest;
                phase = 1; it = 1;
                q = pos ? sub3(ep, scale3(ray, 0.004f * (float)it)) : add3(sp, scale3(ray, 0.004f * (float)it));
            } else phase = 4;
        } else if (phase == 1) {
            if (pos ? v < 0.0f : v > 0.0f) {
                if (pos) sp = q; else ep = q;
                phase = 2; it = 1;
                q = pos ? add3(sp, scale3(ray, 0.0004f * (float)it)) : sub3(ep, scale3(ray, 0.0004f * (float)it));
            } else if (++it < 25) {
                q = pos ? sub3(ep, scale3(ray, 0.004f * (float)it)) : add3(sp, scale3(ray, 0.004f * (float)it));
            } else phase = 4;
        } else if (phase == 2) {
            if (pos ? v > 0.0f : v < 0.0f) {
                if (pos) ep = q; else sp = q;
                phase = 3; it = 0;
            } else if (++it < 11) {
                q = pos ? add3(sp, scale3(ray, 0.0004f * (float)it)) : sub3(ep, scale3(ray, 0.0004f * (float)it));
            } else phase = 4;
        } else {   // phase 3: v is the value at p_temp == q
            if (hd_fabsf(v) < 0.00001f) { found = true; phase = 4; }
            else { if (v < 0.0f) sp = p_temp; else ep = p_temp; ++it; }
        }
        if (phase == 3) {   // next bisection sample, or termination without a further evaluation
            if (it >= 10) phase = 4;
            else {
                const f3 step = sub3(ep, sp);
                if (len3(step) < 0.00001f) { found = true; phase = 4; }
                else { p_temp = add3(sp, scale3(step, 0.5f)); q = p_temp; }
            }
        }
    }

    uchar4 img = make_uchar4(0, 0, 0, 0);
    f3 p_surface = mk3(0, 0, 0), p_normal = mk3(0, 0, 0);
    float4 cmx = make_float4(0, 0, 0, 1000.0f), cmn = make_float4(0, 0, 0, 1000.0f);
    float icpw = 0.0f, confidence = 0.0f, radius = 0.0f;
    uint32_t tm = 0;
    if (found) {
        const f3 ntemp = hrbf_gradient(tile, list, n, p_temp);
        p_surface = p_temp;
        p_normal = normalize3(ntemp);
        float dsm = 1000000.0f;
        int best = -1;
        for (int k = 0; k < n; ++k) {
            const uint32_t off = list[k * PNT];
            const float4 a = lds4(tile, off);
            float dx = p_surface.x - a.x, dy = p_surface.y - a.y, dz = p_surface.z - a.z;
            float dist = hd_sqrtf((dx * dx + dy * dy) + dz * dz);
            if (dist < dsm) { dsm = dist; best = (int)off; }
        }
        if (best >= 0) {
            const int ti = best / (int)sizeof(PTexel);
            const int gi = (by + ti / PTW - PR) * W + (bx + ti % PTW - PR);
            confidence = vertconf[gi].w; radius = normrad[gi].w;
            cmx = curvmax[gi]; cmn = curvmin[gi];
            float4 ct = colortime[gi];
            int ci = (int)ct.x;
            img = make_uchar4((unsigned char)((ci >> 16) & 0xFF), (unsigned char)((ci >> 8) & 0xFF),
                              (unsigned char)(ci & 0xFF), 255);
            tm = (uint32_t)ct.z;
        }
        float a1 = hd_fabsf(cmx.w), a2 = hd_fabsf(cmn.w);
        float cm = a1 > a2 ? a1 : a2;
        icpw = (1.0f / (p_surface.z * p_surface.z)) *
               (confidence / 256.0f + hd_expf(-0.5f * (lambda * lambda) / (cm * cm)));
    }
    reinterpret_cast<uchar4 *>(pr_image)[pi] = img;
    pr_vertex[pi] = make_float4(p_surface.x, p_surface.y, p_surface.z, confidence);
    pr_normal[pi] = make_float4(p_normal.x, p_normal.y, p_normal.z, radius);
    pr_curv1[pi] = cmx; pr_curv2[pi] = cmn;
    pr_time[pi] = tm;
    pr_icpw[pi] = icpw;
}

// Resize::vertex + denseEnough: *flag = 1 when the 20x20-cell thumbnail of the predicted vertex map
// is NOT dense enough (=> shouldFillIn, HRBFFusion.cpp:1069-1070).  Stays on the device.  One workgroup.
__device__ __forceinline__ void should_fill_in_block(const Cam &cam, const float4 *__restrict__ pr_vertex, float thresh,
                                                     int *flag)
{
    const int cs = 20;
    const int w = cam.W / cs, h = cam.H / cs;
    int sum = 0;
    for (int c = threadIdx.x; c < w * h; c += blockDim.x) {
        int i = c % w, j = c / w;
        int sx = (int)hd_floorf(((float)i + 0.5f) * (float)cam.W / (float)w);
        int sy = (int)hd_floorf(((float)j + 0.5f) * (float)cam.H / (float)h);
        sum += pr_vertex[sy * cam.W + sx].z > 0.0f ? 1 : 0;
    }
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d);
    __shared__ int ws[16];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) tot += ws[k];
        float per = (float)tot / (float)(w * h);
        *flag = (per > thresh) ? 0 : 1;
    }
}

// fill_vertex.frag:43-72, fill_normal.frag:36-49, fill_curvature.frag:35-51, fill_rgb.frag:29-37
__global__ void k_fillin(int P, float thr, float lambda, int frame_to_frame_rgb, const float4 *__restrict__ pr_vertex,
                         const float4 *__restrict__ pr_normal, const float4 *__restrict__ pr_curv1,
                         const float4 *__restrict__ pr_curv2, const float *__restrict__ pr_icpw,
                         const uint8_t *__restrict__ pr_image, const float4 *__restrict__ vertex_filtered,
                         const float4 *__restrict__ normal, const float4 *__restrict__ curv1,
                         const float4 *__restrict__ curv2, const float *__restrict__ confidence,
                         const uint8_t *__restrict__ rgb, float4 *__restrict__ fi_vertex,
                         float4 *__restrict__ fi_normal, float4 *__restrict__ fi_curv1, float4 *__restrict__ fi_curv2,
                         float *__restrict__ fi_icpw, uint8_t *__restrict__ fi_image, Cam cam, float dense_thresh,
                         DevPose *dp /* nullable: end-of-frame bookkeeping rides along */,
                         PoseLog *pose_log /* nullable: pinned host ring the frame's pose is appended to */, uint32_t frame_idx)
{
    // end of frame: the next registration's shouldFillIn flag (Resize::vertex + denseEnough on the prediction
    // this kernel reads anyway) and lastPose <- currPose; one workgroup, no separate launches
    if (dp && blockIdx.x == 0) {
        should_fill_in_block(cam, pr_vertex, dense_thresh, &dp->should_fill_in);
        if (threadIdx.x == 0) {
            dp->prev = dp->pose;
            if (pose_log) {   // trajectory without a host round trip: the caller reads the pinned ring whenever it wants
                // The ring is fine-grained (uncached) host memory: the stores go straight to the PCIe write queue, in
                // order.  NO system-scope fence here — a release at system scope writes back the whole L2 (this frame's
                // images) and cost 110 us per frame; the record carries its own frame tag for the reader to check.
                PoseRecord *rec = &pose_log->poses[frame_idx % POSE_LOG_CAP];
                rec->pose = dp->pose;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&rec->tag, frame_idx + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&pose_log->completed, frame_idx + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float4 s = pr_vertex[i];
    if (s.z == 0.0f) {
        float4 fv = vertex_filtered[i], r1 = curv1[i], r2 = curv2[i];
        float4 outv = make_float4(0, 0, 0, 0);
        float outw = 0.0f;
        if (r1.w > -thr && r1.w < thr && r2.w > -thr && r2.w < thr) {
            float vConf = confidence[i];
            float a1 = hd_fabsf(r1.w), a2 = hd_fabsf(r2.w);
            float cm = a1 > a2 ? a1 : a2;
            outw = (1.0f / (fv.z * fv.z)) * (vConf / 256.0f + hd_expf(-0.5f * (lambda * lambda) / (cm * cm)));
            outv = make_float4(fv.x, fv.y, fv.z, vConf);
        }
        fi_vertex[i] = outv; fi_icpw[i] = outw;
    } else { fi_vertex[i] = s; fi_icpw[i] = pr_icpw[i]; }
    float4 n = pr_normal[i];
    fi_normal[i] = (len3(xyz(n)) < 0.8f) ? normal[i] : n;
    float4 k1 = pr_curv1[i], k2 = pr_curv2[i];
    if (k1.w > 300.0f || k2.w > 300.0f) { fi_curv1[i] = curv1[i]; fi_curv2[i] = curv2[i]; }
    else { fi_curv1[i] = k1; fi_curv2[i] = k2; }
    uchar4 e = reinterpret_cast<const uchar4 *>(pr_image)[i];
    if ((int)e.x + (int)e.y + (int)e.z == 0 || frame_to_frame_rgb)
        e = make_uchar4(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2], 255);
    reinterpret_cast<uchar4 *>(fi_image)[i] = e;
}

__global__ void k_should_fill_in(Cam cam, const float4 *__restrict__ pr_vertex, float thresh, int *flag)
{
    should_fill_in_block(cam, pr_vertex, thresh, flag);
}

void launch_predict_hrbf(hipStream_t s, const Cam &cam, const float4 *vertconf, const float4 *normrad,
                         const float4 *colortime, const float4 *curvmax, const float4 *curvmin, int win, int minn,
                         int maxn, float cthr, float lambda, uint8_t *pr_image, float4 *pr_vertex, float4 *pr_normal,
                         float4 *pr_curv1, float4 *pr_curv2, uint32_t *pr_time, float *pr_icpw)
{
    dim3 g((cam.W + TB - 1) / TB, (cam.H + TB - 1) / TB);
    const size_t list_bytes = (size_t)predict_list_cap(maxn) * PNT * sizeof(uint16_t);
    hipLaunchKernelGGL(k_predict_hrbf, g, dim3(TB, TB), list_bytes, s, cam, vertconf, normrad, colortime, curvmax, curvmin, win,
                       minn, maxn, cthr, lambda, pr_image, pr_vertex, pr_normal, pr_curv1, pr_curv2, pr_time, pr_icpw);
}

void launch_fillin(hipStream_t s, int P, float thr, float lambda, int f2f, const float4 *pr_vertex,
                   const float4 *pr_normal, const float4 *pr_curv1, const float4 *pr_curv2, const float *pr_icpw,
                   const uint8_t *pr_image, const float4 *vertex_filtered, const float4 *normal, const float4 *curv1,
                   const float4 *curv2, const float *confidence, const uint8_t *rgb, float4 *fi_vertex,
                   float4 *fi_normal, float4 *fi_curv1, float4 *fi_curv2, float *fi_icpw, uint8_t *fi_image,
                   const Cam &cam, float dense_thresh, DevPose *dp_end_of_frame, PoseLog *pose_log, uint32_t frame_idx)
{
    hipLaunchKernelGGL(k_fillin, dim3((P + 255) / 256), dim3(256), 0, s, P, thr, lambda, f2f, pr_vertex, pr_normal,
                       pr_curv1, pr_curv2, pr_icpw, pr_image, vertex_filtered, normal, curv1, curv2, confidence, rgb,
                       fi_vertex, fi_normal, fi_curv1, fi_curv2, fi_icpw, fi_image, cam, dense_thresh, dp_end_of_frame,
                       dp_end_of_frame ? pose_log : (PoseLog *)nullptr, frame_idx);
}

void launch_should_fill_in(hipStream_t s, const Cam &cam, const float4 *pr_vertex, float thresh, int *flag)
{
    hipLaunchKernelGGL(k_should_fill_in, dim3(1), dim3(256), 0, s, cam, pr_vertex, thresh, flag);
}

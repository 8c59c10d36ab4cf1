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
//     per tile in LDS (tile + 3-texel halo, two planes of 22x22 x 16 B = 15.5 KB), together with one validity bit per
//     texel (the acceptance test of predict_hrbf.frag:85-92 depends on the texel only);
//   * each thread walks the window in the shader's ring order with compile-time offsets and writes the LDS
//     addresses of its accepted neighbours to a private column of an LDS list (<= 49 x 2 B per thread), so
//     the <= 46 implicit evaluations per pixel are a plain counted loop over LDS: one 2-byte and two 16-byte
//     reads per neighbour, no table look-ups, no divisions.
#include "common.h"
#include "kernels.h"

#ifndef PREDICT_TBX
#define PREDICT_TBX 16
#endif
#ifndef PREDICT_TBY
#define PREDICT_TBY 16
#endif
#define TBX PREDICT_TBX
#define TBY PREDICT_TBY
#define PR 3
#define PTW (TBX + 2 * PR)
#define PTH (TBY + 2 * PR)
#define PNT (TBX * TBY)

// per-texel part of the implicit in LDS, two planes of 16 B per texel (a wave's reads of neighbouring texels are then
// contiguous — 32-byte records made every 16-byte read a two-way bank conflict, which showed once the arithmetic was packed):
//   plane A: centre xyz, support radius^2        plane B (PLANE_B bytes behind): 10 n, 1 / radius^2
#define PTEXEL_BYTES 16
#define DUMMY_TEXEL (PTW * PTH)                       // support radius^2 = -1: in reach of nothing
#define PLANE_B ((PTW * PTH + 1) * PTEXEL_BYTES)

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

// The per-thread neighbour list: entry k of thread t is the 16-bit half (k & 1) of the 32-bit word [k / 2][t], so one 4-byte
// read fetches the LDS offsets of the two neighbours a trip of hrbf_value evaluates.  `list` points at the thread's word 0.
__device__ __forceinline__ int list_slot(int k) { return (k >> 1) * (2 * PNT) + (k & 1); }

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
                list[list_slot(n)] = (uint16_t)(lbase_bytes + (uint32_t)((e.dy * PTW + e.dx) * PTEXEL_BYTES));
                n++;
                if (n > maxn) skip = true;
            }
        }
        gather_ring<SLOT + 1>(wrow, nslots, maxn, lbase_bytes, list, n, skip);
    }
}

__device__ __forceinline__ float4 lds4(const float4 *tile, uint32_t byte_off)
{
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(tile) + byte_off);
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

// Keeps a scalar result scalar: without it the instruction selector turns a pair built from two scalar adds into a packed add
// of two shuffled pairs (three v_mov per pair).  No instruction is emitted.
__device__ __forceinline__ float opaque(float x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// Correctly rounded sqrt of two arguments that are zero or normal numbers: v_sqrt_f32 is within one ulp, and the two residual
// tests (the ones the compiler's own expansion of sqrtf makes) move to the neighbouring float when that one is closer.  The
// compiler's expansion also rescales arguments below 2^-96 and re-checks for 0 / inf; neither is needed here: the consumer is
// 1 - r, which is exactly 1 for every r < 2^-25 whatever its last bit, a zero argument falls through both tests unchanged,
// and arguments above 1 belong to neighbours whose term is discarded.
__device__ __forceinline__ v2f sqrt_pair_for_unit_complement(const v2f x)
{
    v2f s;
    s.x = __builtin_amdgcn_sqrtf(x.x); s.y = __builtin_amdgcn_sqrtf(x.y);
    const v2i si = __builtin_bit_cast(v2i, s);
    const v2f dn = __builtin_bit_cast(v2f, si - 1), up = __builtin_bit_cast(v2f, si + 1);
    const v2f ed = __builtin_elementwise_fma(-dn, s, x), eu = __builtin_elementwise_fma(-up, s, x);
    s.x = ed.x <= 0.0f ? dn.x : s.x; s.y = ed.y <= 0.0f ? dn.y : s.y;
    s.x = eu.x > 0.0f ? up.x : s.x; s.y = eu.y > 0.0f ? up.y : s.y;
    return s;
}

// hrbfvalue (hrbfbase.glsl:126-145) with getWeightD (:20-34) over the gathered list, TWO neighbours per trip with the
// arithmetic written on float pairs (v_pk_add / v_pk_mul / v_pk_fma_f32: the same IEEE operations, half the issue slots —
// this loop is VALU-issue bound).  x / y of one neighbour share a pair (they are adjacent in the LDS quad), the z
// column and everything scalar per neighbour (d2, r, s, the factor) pair up ACROSS the two neighbours.  Branch-free: the
// term of a neighbour whose support does not reach p is computed and discarded.  One 4-byte read fetches the offsets of the
// next pair while the current one is evaluated; list entries n .. n+3 hold the offset of the dummy texel (support -1:
// never reached).
//
// getWeightD returns the zero vector at d2 == 0 (hrbfbase.glsl:24-27).  There v = p - centre is exactly (+0, +0, +0)
// (x - x is +0 under round-to-nearest), so ONE select on the scalar factor gives the same three +0 products as three selects
// on the products — and keeps the T = 0 case (factor = -inf or NaN) away from 0 * inf.
//
// SAFE = false (every staged texel of the tile is finite and tame, tile_is_tame below) drops that select and folds the
// support test into one select on the factor instead of one on the running sum:
//   * d2 == 0 with a finite factor: the products are (+0) * factor = -0 or +0, their sum c is a zero, and value - (+-0)
//     == value because the running sum is never -0 (it starts at +0 and x - x is +0);
//   * a neighbour out of reach gets factor 0: v and 10 n are finite, so c is a zero again.
// COUNT: the number of neighbours in reach is only consumed for the first sample of a ray (predict_hrbf.frag:139-141).
#ifdef PREDICT_TRIP_STATS
// measurement build: [0] (sample, centre) pairs evaluated, [1] of them with the centre's support reaching the
// sample (hrbfbase.glsl:137-138), [2] samples, [3] samples no centre reaches, [4] list entries over all rays, [5] entries whose
// support never reaches the ray's line at all, [6] entries whose support does not reach the marched stretch (+-10 cm of `closest`)
__device__ unsigned long long g_support_stats[8];
extern "C" int hrbf_probe_predict_support(unsigned long long out[8], int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_support_stats), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_support_stats), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
template <bool SAFE, bool COUNT>
__device__ __forceinline__ float hrbf_value(const float4 *__restrict__ tile, const uint16_t *__restrict__ list, int n,
                                            f3 p, int &nsup)
{
    float value = 0.0f;
    int ns = 0;
#ifdef PREDICT_TRIP_STATS
    int ns_stat = 0;
#endif
    const v2f pxy = {p.x, p.y};
    const uint32_t *pairs = reinterpret_cast<const uint32_t *>(list);
    uint32_t oo = pairs[0];
    for (int k = 0; k < n; k += 2) {
        const uint32_t o0 = oo & 0xffffu, o1 = oo >> 16;
        const float4 a0 = lds4(tile, o0), b0 = lds4(tile, o0 + PLANE_B), a1 = lds4(tile, o1), b1 = lds4(tile, o1 + PLANE_B);
        pairs += PNT;
        oo = pairs[0];
        const v2f v0 = pxy - (v2f){a0.x, a0.y}, v1 = pxy - (v2f){a1.x, a1.y};
        const v2f vz = {opaque(p.z - a0.z), opaque(p.z - a1.z)};
        const v2f q0 = v0 * v0, q1 = v1 * v1;
        const v2f d2 = (v2f){opaque(q0.x + q0.y), opaque(q1.x + q1.y)} + vz * vz;
        const bool in0 = !(a0.w < d2.x), in1 = (!SAFE || k + 1 < n) && !(a1.w < d2.y);
        const v2f inv = {b0.w, b1.w};
        const v2f s = 1.0f - sqrt_pair_for_unit_complement(d2 * inv);
        v2f tt = -20.0f * (s * s * s) * inv;
        if (SAFE) { tt.x = d2.x != 0.0f ? tt.x : 0.0f; tt.y = d2.y != 0.0f ? tt.y : 0.0f; }
        else { tt.x = in0 ? tt.x : 0.0f; tt.y = in1 ? tt.y : 0.0f; }
        const v2f c0 = (v0 * tt.x) * (v2f){b0.x, b0.y}, c1 = (v1 * tt.y) * (v2f){b1.x, b1.y};
        const v2f gz = vz * tt;
        const v2f c = (v2f){opaque(c0.x + c0.y), opaque(c1.x + c1.y)} + (v2f){opaque(gz.x * b0.z), opaque(gz.y * b1.z)};
        if (SAFE) {
            value = in0 ? value - c.x : value;
            value = in1 ? value - c.y : value;
        } else {
            value = value - c.x;
            value = value - c.y;
        }
        if (COUNT) ns += (in0 ? 1 : 0) + (in1 ? 1 : 0);
#ifdef PREDICT_TRIP_STATS
        ns_stat += (in0 ? 1 : 0) + ((k + 1 < n && !(a1.w < d2.y)) ? 1 : 0);
#endif
    }
#ifdef PREDICT_TRIP_STATS
    atomicAdd(&g_support_stats[0], (unsigned long long)n); atomicAdd(&g_support_stats[1], (unsigned long long)ns_stat);
    atomicAdd(&g_support_stats[2], 1ull); if (ns_stat == 0) atomicAdd(&g_support_stats[3], 1ull);
#endif
    nsup = ns;
    return value;
}

// A texel the shortcuts of hrbf_value<false> are exact for: centre below 1e15 in magnitude (squares and the march stay
// finite), 10 n finite, 1 / T^2 below 1e30 (20 / T^2 finite).  NaNs fail every comparison.
__device__ __forceinline__ bool texel_is_tame(const float4 ta, const float4 tb)
{
    return hd_fabsf(ta.x) < 1e15f && hd_fabsf(ta.y) < 1e15f && hd_fabsf(ta.z) < 1e15f && hd_fabsf(tb.x) < 1e30f &&
           hd_fabsf(tb.y) < 1e30f && hd_fabsf(tb.z) < 1e30f && hd_fabsf(tb.w) < 1e30f;
}

// Exhaustive check of what sqrt_pair_for_unit_complement relies on, over every non-negative finite float
// (tests/test_parity_gpu.py::test_sqrt_shortcut_is_exhaustively_exact):
//   out[0..3]  v_sqrt_f32 against the correctly rounded root: equal / one ulp low / one ulp high / anything else
//   out[4]     arguments >= 2^-96 where the shortcut differs from the correctly rounded root
//   out[5]     arguments <  2^-96 where 1 - shortcut differs from 1 - root (the only way the loop consumes it)
__global__ void k_probe_sqrt(unsigned long long *out)
{
    unsigned long long c[6] = {0, 0, 0, 0, 0, 0};
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t b = blockIdx.x * blockDim.x + threadIdx.x; b < 0x7f800000ull; b += stride) {
        const float x = __builtin_bit_cast(float, (uint32_t)b);
        const float exact = hd_sqrtf(x), raw = __builtin_amdgcn_sqrtf(x);
        const int d = (int)(__builtin_bit_cast(uint32_t, raw) - __builtin_bit_cast(uint32_t, exact));
        c[d == 0 ? 0 : d == -1 ? 1 : d == 1 ? 2 : 3]++;
        const v2f r = sqrt_pair_for_unit_complement((v2f){x, x});
        if (x >= 0x1p-96f) c[4] += (__builtin_bit_cast(uint32_t, r.x) != __builtin_bit_cast(uint32_t, exact)) ||
                                   (__builtin_bit_cast(uint32_t, r.y) != __builtin_bit_cast(uint32_t, exact));
        else c[5] += (1.0f - r.x != 1.0f - exact) || (1.0f - r.y != 1.0f - exact);
    }
    for (int i = 0; i < 6; ++i)
        if (c[i]) atomicAdd(&out[i], c[i]);
}

int predict_probe_sqrt(hipStream_t s, unsigned long long *d_out6)
{
    if (hipMemsetAsync(d_out6, 0, 6 * sizeof(unsigned long long), s) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_probe_sqrt, dim3(4096), dim3(256), 0, s, d_out6);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// hrbfgradient (hrbfbase.glsl:147-166) with getWeightH (:37-69)
__device__ __forceinline__ f3 hrbf_gradient(const float4 *__restrict__ tile, const uint16_t *__restrict__ list, int n, f3 p)
{
    float grx = 0.0f, gry = 0.0f, grz = 0.0f;
    for (int k = 0; k < n; ++k) {
        const uint32_t o = list[list_slot(k)];
        const float4 a = lds4(tile, o), b = lds4(tile, o + PLANE_B);
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

// Ray march + bisection of predict_hrbf.frag:150-260 as one loop with a single evaluation site: every trip
// each live lane evaluates the implicit at the next sample of ITS OWN phase, so a wave needs
// max_lane(total samples) trips instead of sum_phase(max_lane(samples of the phase)).
//   first: value at `closest`               phase 1: 25 steps of 4 mm away from it until the sign flips
//   phase 2: 10 steps of 0.4 mm back        phase 3: <= 10 bisections        phase 4: finished
// The first coarse sample (i = 0) is `closest` itself: its value is v0, which cannot flip the sign, so the
// march starts at i = 1.  Returns whether a surface point was found (in p_temp).
#ifdef PREDICT_TRIP_STATS
struct TripStats { float v0, v1; int k1, k2, k3; };   // measurement build: first two values, samples per phase
#define TS(x) x
#else
#define TS(x)
#endif
template <bool SAFE>
__device__ __forceinline__ bool ray_march(const float4 *__restrict__ tile, const uint16_t *__restrict__ list, int n, int minn,
                                          const f3 closest, const f3 ray, f3 &p_temp, int &trips
#ifdef PREDICT_TRIP_STATS
                                          , TripStats &ts
#endif
                                          )
{
    bool found = false;
    trips = 0;
    TS(ts.v0 = 0.0f; ts.v1 = 0.0f; ts.k1 = ts.k2 = ts.k3 = 0;)
    f3 sp = mk3(0, 0, 0), ep = mk3(0, 0, 0), q = closest;
    int phase = 4, it = 1;
    bool pos = false;   // sign class of v0
    if (n > minn) {
        int nsup;
        const float v0 = hrbf_value<SAFE, true>(tile, list, n, closest, nsup);
        trips = 1;
        TS(ts.v0 = v0;)
        if (nsup > minn) {
            pos = v0 > 0.0f;
            if (pos) ep = closest; else sp = closest;
            phase = 1;
            q = pos ? sub3(ep, scale3(ray, 0.004f * (float)it)) : add3(sp, scale3(ray, 0.004f * (float)it));
        }
    }
    while (phase != 4) {
        int unused;
        const float v = hrbf_value<SAFE, false>(tile, list, n, q, unused);
        ++trips;
        TS(if (trips == 2) ts.v1 = v; if (phase == 1) ++ts.k1; else if (phase == 2) ++ts.k2; else ++ts.k3;)
        if (phase == 1) {
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
    return found;
}

// capacity of the per-thread neighbour list: once n exceeds maxn every later window column adds at most one
// entry (15 columns can still follow), plus four dummy entries behind the list (hrbf_value reads one pair ahead); even
__host__ __device__ inline int predict_list_cap(int maxn)
{
    int c = maxn + 1 + 15;
    if (c > 49) c = 49;
    if (c < 1) c = 1;
    return (c + 4 + 1) & ~1;
}

// one pixel of FillIn::{vertex,normal,curvature,image} (fill_vertex.frag:43-72, fill_normal.frag:36-49,
// fill_curvature.frag:35-51, fill_rgb.frag:29-37) from the pixel's predicted values: shared by k_fillin (the stage) and
// the tail of k_predict_hrbf<true> (the frame path: the prediction is still in registers, the live images are read while
// the kernel's ALU work hides them)
__device__ __forceinline__ void fill_pixel(int i, float thr, float lambda, int frame_to_frame_rgb, const float4 s,
                                           const float4 n, const float4 k1, const float4 k2, const float icpw, uchar4 e,
                                           const FillIn &f)
{
    if (s.z == 0.0f) {
        float4 fv = f.vertex_filtered[i], r1 = f.curv1[i], r2 = f.curv2[i];
        float4 outv = make_float4(0, 0, 0, 0);
        float outw = 0.0f;
        if (r1.w > -thr && r1.w < thr && r2.w > -thr && r2.w < thr) {
            float vConf = f.confidence[i];
            float a1 = hd_fabsf(r1.w), a2 = hd_fabsf(r2.w);
            float cm = a1 > a2 ? a1 : a2;
            outw = (1.0f / (fv.z * fv.z)) * (vConf / 256.0f + hd_expf(-0.5f * (lambda * lambda) / (cm * cm)));
            outv = make_float4(fv.x, fv.y, fv.z, vConf);
        }
        f.fi_vertex[i] = outv; f.fi_icpw[i] = outw;
    } else { f.fi_vertex[i] = s; f.fi_icpw[i] = icpw; }
    // not `cond ? f.normal[i] : n`: the ?: on HIP_vector_type objects left a (never read) 16-byte copy in scratch memory, and
    // with it a scratch segment for every wave of k_predict_hrbf<true> and k_fillin (ScratchSize 32 -> 0)
    if (len3(xyz(n)) < 0.8f) f.fi_normal[i] = f.normal[i];
    else f.fi_normal[i] = n;
    if (k1.w > 300.0f || k2.w > 300.0f) { f.fi_curv1[i] = f.curv1[i]; f.fi_curv2[i] = f.curv2[i]; }
    else { f.fi_curv1[i] = k1; f.fi_curv2[i] = k2; }
    if ((int)e.x + (int)e.y + (int)e.z == 0 || frame_to_frame_rgb)
        e = make_uchar4(f.rgb[i * 3], f.rgb[i * 3 + 1], f.rgb[i * 3 + 2], 255);
    reinterpret_cast<uchar4 *>(f.fi_image)[i] = e;
}

template <bool FILL /* also fill in from the live frame (process_frame path) */>
__global__ __launch_bounds__(PNT) void k_predict_hrbf(Cam cam, const float4 *__restrict__ vertconf,
                                                      const float4 *__restrict__ normrad,
                                                      const float4 *__restrict__ colortime,
                                                      const float4 *__restrict__ curvmax,
                                                      const float4 *__restrict__ curvmin, int win, int minn, int maxn,
                                                      float cthr, float lambda, uint8_t *__restrict__ pr_image,
                                                      float4 *__restrict__ pr_vertex, float4 *__restrict__ pr_normal,
                                                      float4 *__restrict__ pr_curv1, float4 *__restrict__ pr_curv2,
                                                      uint32_t *__restrict__ pr_time, float *__restrict__ pr_icpw, FillIn fill)
{
    __shared__ float4 tile[2 * (PTW * PTH + 1)];
    __shared__ uint32_t s_rowbits[PTH];
    __shared__ uint32_t s_untame;
    extern __shared__ uint16_t s_list[];   // predict_list_cap(maxn) x 256 entries
    const int W = cam.W, H = cam.H;
    const int bx = blockIdx.x * TBX, by = blockIdx.y * TBY;
    const int tid = threadIdx.y * TBX + threadIdx.x;
    if (tid < PTH) s_rowbits[tid] = 0u;
    if (tid == 0) {
        s_untame = 0u;
        tile[DUMMY_TEXEL] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
        tile[PTW * PTH + 1 + DUMMY_TEXEL] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    __syncthreads();
    for (int i = tid; i < PTW * PTH; i += PNT) {
        int tx = i % PTW, ty = i / PTW;
        int gx = bx + tx - PR, gy = by + ty - PR;
        float4 ta = make_float4(0.0f, 0.0f, 0.0f, 0.0f), tb = ta;
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) {   // outside the image: rejected like `j < 0 || j > 1` (:85)
            float4 v = vertconf[gy * W + gx], n = normrad[gy * W + gx];
            const float T2 = n.w * n.w;
            ta = make_float4(v.x, v.y, v.z, T2);
            tb = make_float4(10.0f * n.x, 10.0f * n.y, 10.0f * n.z, 1.0f / T2);
            if (!(v.z < 0.1f || len3(mk3(n.x, n.y, n.z)) < 0.1f || v.w < cthr || n.z < 0.0f)) {
                atomicOr(&s_rowbits[ty], 1u << tx);
                if (!texel_is_tame(ta, tb)) s_untame = 1u;
            }
        }
        tile[i] = ta; tile[PTW * PTH + 1 + i] = tb;
    }
    __syncthreads();
    // a wave marches an 8 x 8 block of the tile, not a 16 x 4 strip: the number of samples a ray takes and the length of its
    // neighbour list vary smoothly over the image, and the wave pays for its slowest lane (PREDICT_WAVE_8X8=0: the strip)
#ifndef PREDICT_WAVE_8X8
#define PREDICT_WAVE_8X8 1
#endif
#if PREDICT_WAVE_8X8 && PREDICT_TBX == 16 && PREDICT_TBY == 16
    const int lxx = (tid & 7) + ((tid >> 6) & 1) * 8, lyy = ((tid >> 3) & 7) + (tid >> 7) * 8;
#else
    const int lxx = (int)threadIdx.x, lyy = (int)threadIdx.y;
#endif
    const int px = bx + lxx, py = by + lyy;
    if (px >= W || py >= H) return;
    const int pi = py * W + px;
    const uint32_t lbase_bytes = (uint32_t)(((lyy + PR) * PTW + lxx + PR) * PTEXEL_BYTES);
    uint16_t *list = s_list + 2 * tid;

    int n = 0;
    {
        uint32_t wrow[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) wrow[r] = s_rowbits[lyy + r] >> lxx;
        bool skip = false;
        gather_ring<0>(wrow, (2 * win + 1) * (2 * win + 1), maxn, lbase_bytes, list, n, skip);
#pragma unroll
        for (int d = 0; d < 4; ++d) list[list_slot(n + d)] = (uint16_t)(DUMMY_TEXEL * PTEXEL_BYTES);
    }

    const float x = hd_px_fragment(px, cam.W), y = hd_px_fragment(py, cam.H);   // predict_hrbf.frag:42-43: texcoord * cols, rows
    const float xl = (x - cam.cx) * cam.camz, yl = (y - cam.cy) * cam.camw;
    const f3 ray = normalize3(mk3(xl, yl, 1.0f));

    f3 closest = mk3(0, 0, 0);
    {
        float pmin = 1000000.0f;
        for (int k = 0; k < n; ++k) {
            const float4 a = lds4(tile, list[list_slot(k)]);
            float pj = hd_fabsf(dot3(mk3(a.x, a.y, a.z), ray));
            if (pj < pmin) { closest = scale3(ray, pj); pmin = pj; }
        }
    }

    f3 p_temp = mk3(0, 0, 0);
    int trips;   // samples of the implicit this ray took (statistics builds only; dead otherwise)
#ifdef PREDICT_TRIP_STATS
    if (n > minn) {   // which list entries could a ray-parameter interval test discard for the WHOLE march?
        const float rr = dot3(ray, ray), tc = dot3(closest, ray) / rr;
        int never = 0, not_in_stretch = 0;
        for (int k = 0; k < n; ++k) {
            const float4 a = lds4(tile, list[list_slot(k)]);
            const f3 c = mk3(a.x, a.y, a.z);
            const float ta = dot3(c, ray) / rr;                       // parameter of the centre's foot point on the line
            const float perp2 = dot3(c, c) - ta * ta * rr;            // squared distance of the centre from the line
            if (perp2 > a.w) { ++never; ++not_in_stretch; continue; }
            const float half = hd_sqrtf((a.w - perp2) / rr);          // the line is inside the support for ta - half .. ta + half
            const float reach = 0.1f / hd_sqrtf(rr);                  // 25 coarse steps of 4 mm either way
            if (ta + half < tc - reach || ta - half > tc + reach) ++not_in_stretch;
        }
        atomicAdd(&g_support_stats[4], (unsigned long long)n); atomicAdd(&g_support_stats[5], (unsigned long long)never);
        atomicAdd(&g_support_stats[6], (unsigned long long)not_in_stretch);
    }
    TripStats ts;
    const bool found = s_untame ? ray_march<true>(tile, list, n, minn, closest, ray, p_temp, trips, ts)
                                : ray_march<false>(tile, list, n, minn, closest, ray, p_temp, trips, ts);
#else
    const bool found = s_untame ? ray_march<true>(tile, list, n, minn, closest, ray, p_temp, trips)
                                : ray_march<false>(tile, list, n, minn, closest, ray, p_temp, trips);
#endif

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
            const uint32_t off = list[list_slot(k)];
            const float4 a = lds4(tile, off);
            float dx = p_surface.x - a.x, dy = p_surface.y - a.y, dz = p_surface.z - a.z;
            float dist = hd_sqrtf((dx * dx + dy * dy) + dz * dz);
            if (dist < dsm) { dsm = dist; best = (int)off; }
        }
        if (best >= 0) {
            const int ti = best / PTEXEL_BYTES;
            const int gi = (by + ti / PTW - PR) * W + (bx + ti % PTW - PR);
            confidence = vertconf[gi].w; radius = normrad[gi].w;
            cmx = curvmax[gi]; cmn = curvmin[gi];
            float4 ct = colortime[gi];
            int ci = hd_cvt_i32(ct.x);
            img = make_uchar4((unsigned char)((ci >> 16) & 0xFF), (unsigned char)((ci >> 8) & 0xFF),
                              (unsigned char)(ci & 0xFF), 255);
            tm = hd_cvt_u32(ct.z);
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
#ifdef PREDICT_TRIP_STATS   // measurement build: samples | neighbours << 8 | found << 16 | k1 << 17 | k2 << 22 | k3 << 26 instead of the
    // time stamp; the first two values of the implicit in the curvature image's direction words (tests/gpu_probe_predict_trips.py)
    tm = (uint32_t)trips | ((uint32_t)n << 8) | (found ? 1u << 16 : 0u) | ((uint32_t)ts.k1 << 17) | ((uint32_t)ts.k2 << 22) | ((uint32_t)ts.k3 << 26);
    pr_curv1[pi] = make_float4(ts.v0, ts.v1, 0.0f, cmx.w);
#endif
    pr_time[pi] = tm;
    pr_icpw[pi] = icpw;
    if (FILL)
        fill_pixel(pi, fill.thr, lambda, fill.frame_to_frame_rgb, make_float4(p_surface.x, p_surface.y, p_surface.z, confidence),
                   make_float4(p_normal.x, p_normal.y, p_normal.z, radius), cmx, cmn, icpw, img, fill);
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

// End of frame, one workgroup: the next registration's shouldFillIn flag (Resize::vertex + denseEnough on the finished
// prediction), lastPose <- currPose, and the frame's pose appended to the pinned trajectory ring.
__device__ __forceinline__ void end_of_frame_block(const Cam &cam, const float4 *__restrict__ pr_vertex, float dense_thresh,
                                                   DevPose *dp, PoseLog *pose_log, uint32_t frame_idx)
{
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

// the same bookkeeping alone, behind k_predict_hrbf<true> (which has filled in already)
__global__ void k_end_of_frame(Cam cam, const float4 *__restrict__ pr_vertex, float dense_thresh, DevPose *dp, PoseLog *pose_log,
                               uint32_t frame_idx)
{
    end_of_frame_block(cam, pr_vertex, dense_thresh, dp, pose_log, frame_idx);
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
    if (dp && blockIdx.x == 0) end_of_frame_block(cam, pr_vertex, dense_thresh, dp, pose_log, frame_idx);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    FillIn f;
    f.thr = thr; f.frame_to_frame_rgb = frame_to_frame_rgb;
    f.vertex_filtered = vertex_filtered; f.normal = normal; f.curv1 = curv1; f.curv2 = curv2; f.confidence = confidence; f.rgb = rgb;
    f.fi_vertex = fi_vertex; f.fi_normal = fi_normal; f.fi_curv1 = fi_curv1; f.fi_curv2 = fi_curv2; f.fi_icpw = fi_icpw; f.fi_image = fi_image;
    fill_pixel(i, thr, lambda, frame_to_frame_rgb, pr_vertex[i], pr_normal[i], pr_curv1[i], pr_curv2[i], pr_icpw[i],
               reinterpret_cast<const uchar4 *>(pr_image)[i], f);
}

__global__ void k_should_fill_in(Cam cam, const float4 *__restrict__ pr_vertex, float thresh, int *flag)
{
    should_fill_in_block(cam, pr_vertex, thresh, flag);
}

void launch_predict_hrbf(hipStream_t s, const Cam &cam, const float4 *vertconf, const float4 *normrad,
                         const float4 *colortime, const float4 *curvmax, const float4 *curvmin, int win, int minn,
                         int maxn, float cthr, float lambda, uint8_t *pr_image, float4 *pr_vertex, float4 *pr_normal,
                         float4 *pr_curv1, float4 *pr_curv2, uint32_t *pr_time, float *pr_icpw, const FillIn *fill)
{
    dim3 g((cam.W + TBX - 1) / TBX, (cam.H + TBY - 1) / TBY);
    const size_t list_bytes = (size_t)predict_list_cap(maxn) * PNT * sizeof(uint16_t);
    if (fill)
        hipLaunchKernelGGL(k_predict_hrbf<true>, g, dim3(TBX, TBY), list_bytes, s, cam, vertconf, normrad, colortime, curvmax,
                           curvmin, win, minn, maxn, cthr, lambda, pr_image, pr_vertex, pr_normal, pr_curv1, pr_curv2, pr_time,
                           pr_icpw, *fill);
    else
        hipLaunchKernelGGL(k_predict_hrbf<false>, g, dim3(TBX, TBY), list_bytes, s, cam, vertconf, normrad, colortime, curvmax,
                           curvmin, win, minn, maxn, cthr, lambda, pr_image, pr_vertex, pr_normal, pr_curv1, pr_curv2, pr_time,
                           pr_icpw, FillIn{});
}

void launch_end_of_frame(hipStream_t s, const Cam &cam, const float4 *pr_vertex, float dense_thresh, DevPose *dp,
                         PoseLog *pose_log, uint32_t frame_idx)
{
    hipLaunchKernelGGL(k_end_of_frame, dim3(1), dim3(256), 0, s, cam, pr_vertex, dense_thresh, dp, pose_log, frame_idx);
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

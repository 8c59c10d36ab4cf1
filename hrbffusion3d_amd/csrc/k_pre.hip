// k_pre.hip — per-frame pre-processing kernels for gfx950.
//
// Replaces the GLSL passes driven by HRBFFusion::filterDepth / metriciseDepth /
// computeVertexNormalRadius / computeCurvatureGradient / updateNormalRad / VertexConfidence
// (Core/src/HRBFFusion.cpp:1263-1345; shaders depth_bilateral.frag, depth_metric_*.frag,
// depth_vertex_normal_radius.frag, depth_curvature_gradient.frag, depth_confidence_evaluation.frag).
// Compute-bound window kernels: 16x16 pixel tiles (4 wavefronts), window halo staged in LDS.
#include "common.h"
#include "kernels.h"
#include "pca_normal.h"

#define TB 16   // tile edge; 256 threads = 4 wave64

typedef float v2f __attribute__((ext_vector_type(2)));

// (p * 2^k1) * 2^k2 with k1 = k / 2, k2 = k - k1 — hd_expf's two-step scaling — is one correctly rounded scaling by 2^k:
// the first product is exact (no underflow for k >= -160), the second rounds once.  That is what v_ldexp_f32 computes;
// checked bit for bit on the device for every p in [0.5, 2) and k in [-160, 0] by hrbf_probe_exp_scaling.
__device__ __forceinline__ float exp_scale(float p, int k) { return __builtin_ldexpf(p, k); }

// hd_expf_nonpos on two arguments at once (v_pk_mul / v_pk_fma_f32: the same IEEE operations per half)
__device__ __forceinline__ v2f expf_nonpos_pair(const v2f x0)
{
    const bool u0 = x0.x < -103.9f, u1 = x0.y < -103.9f;
    const v2f x = {u0 ? -103.9f : x0.x, u1 ? -103.9f : x0.y};
    const v2f t = x * 1.44269504088896341f;
    const v2f kf = {hd_rintf(t.x), hd_rintf(t.y)};
    v2f r = __builtin_elementwise_fma(kf, (v2f)(-0.693359375f), x);
    r = __builtin_elementwise_fma(kf, (v2f)(2.12194440e-4f), r);
    v2f p = (v2f)(1.3981999507e-3f);
    p = __builtin_elementwise_fma(p, r, (v2f)(8.3334519073e-3f));
    p = __builtin_elementwise_fma(p, r, (v2f)(4.1665795894e-2f));
    p = __builtin_elementwise_fma(p, r, (v2f)(1.6666665459e-1f));
    p = __builtin_elementwise_fma(p, r, (v2f)(5.0000001201e-1f));
    const v2f r2 = r * r;
    p = __builtin_elementwise_fma(p, r2, r);
    p = p + 1.0f;
    v2f w;
    w.x = u0 ? 0.0f : exp_scale(p.x, (int)kf.x);
    w.y = u1 ? 0.0f : exp_scale(p.y, (int)kf.y);
    return w;
}

// the scalar twin (odd tap at the end of a row)
__device__ __forceinline__ float expf_nonpos_dev(const float x0)
{
    const bool under = x0 < -103.9f;
    const float x = under ? -103.9f : x0;
    const float kf = hd_rintf(x * 1.44269504088896341f);
    float r = hd_fmaf(kf, -0.693359375f, x);
    r = hd_fmaf(kf, 2.12194440e-4f, r);
    float p = 1.3981999507e-3f;
    p = hd_fmaf(p, r, 8.3334519073e-3f);
    p = hd_fmaf(p, r, 4.1665795894e-2f);
    p = hd_fmaf(p, r, 1.6666665459e-1f);
    p = hd_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    p = hd_fmaf(p, r2, r);
    p = p + 1.0f;
    return under ? 0.0f : exp_scale(p, (int)kf);
}

// test probe: exp_scale against hd_expf's two multiplications, every p in [0.5, 2) x every k in [-160, 0]
__global__ void k_probe_exp_scaling(unsigned long long *out)
{
    unsigned long long bad = 0, total = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t b = 0x3f000000u + blockIdx.x * blockDim.x + threadIdx.x; b < 0x40000000u; b += stride) {
        const float p = hd_u2f(b);
        for (int k = -160; k <= 0; ++k) {
            const int k1 = k / 2, k2 = k - k1;
            const float ref = (p * hd_u2f((uint32_t)(k1 + 127) << 23)) * hd_u2f((uint32_t)(k2 + 127) << 23);
            bad += hd_f2u(ref) != hd_f2u(exp_scale(p, k));
            ++total;
        }
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicAdd(&out[1], total);
}

int pre_probe_exp_scaling(hipStream_t s, unsigned long long *d_out2)
{
    if (hipMemsetAsync(d_out2, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_probe_exp_scaling, dim3(4096), dim3(256), 0, s, d_out2);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------
// P1 + P2: bilateral (13x13) or gated Gaussian (9x9) on raw depth, plus both metric images.
// LDS: (16+12)^2 raw depths as float-mm (value / adj precomputed once per texel).
// ride: workgroup rows beyond the tile grid copy `ride.bytes` bytes from `ride.src` to `ride.dst` and leave — the RGB half of
// the host-pointer entry's upload (0.9 MB read over PCIe, 23 us), which nothing needs before the curvature kernel, hidden
// behind this kernel's arithmetic instead of heading the frame (the depth half must be there first and stays in front)
struct RideCopy { const uint8_t *src; uint8_t *dst; size_t bytes; };
template <bool BILATERAL>
__global__ __launch_bounds__(256) void k_filter_metric(Cam cam, const uint16_t *__restrict__ raw,
                                                       float *__restrict__ filtered, float *__restrict__ metric,
                                                       float *__restrict__ metric_f, float depthFactor, float maxD,
                                                       RideCopy ride)
{
    const int tiles_y = (cam.H + TB - 1) / TB;
    if ((int)blockIdx.y >= tiles_y) {
        const size_t t = ((size_t)(blockIdx.y - tiles_y) * gridDim.x + blockIdx.x) * (TB * TB) + threadIdx.y * TB + threadIdx.x;
        const size_t nt = (size_t)(gridDim.y - tiles_y) * gridDim.x * (TB * TB), nv = ride.bytes / 16;
        for (size_t i = t; i < nv; i += nt) reinterpret_cast<uint4 *>(ride.dst)[i] = reinterpret_cast<const uint4 *>(ride.src)[i];
        for (size_t i = nv * 16 + t; i < ride.bytes; i += nt) ride.dst[i] = ride.src[i];
        return;
    }
    constexpr int R = BILATERAL ? 6 : 4;
    constexpr int TW = TB + 2 * R;
    __shared__ float tile[TW * TW];
    const int W = cam.W, H = cam.H;
    const int bx = blockIdx.x * TB, by = blockIdx.y * TB;
    const float adj = 1.0f / (depthFactor * 1000.0f);
    for (int i = threadIdx.y * TB + threadIdx.x; i < TW * TW; i += TB * TB) {
        int tx = i % TW, ty = i / TW;
        int gx = bx + tx - R, gy = by + ty - R;
        float v = 0.0f;
        // a tap's coordinate float(c) / n sits ON the texel's edge; NEAREST reads texel floor(fl(fl(c / n) * n)) (hd_tap_texel:
        // c - 1 at 7 rows of a 480-high image, c everywhere at 640) — the tile holds what the taps read
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) v = (float)raw[hd_tap_texel(gy, H) * W + hd_tap_texel(gx, W)] / adj;
        tile[i] = v;
    }
    __syncthreads();
    const int x = bx + threadIdx.x, y = by + threadIdx.y;
    if (x >= W || y >= H) return;
    const float value = (float)raw[y * W + x] / adj;   // the centre is read at the pixel's own texcoord (depth_bilateral.frag:20)
    float out;
    if (value > maxD * 1000.0f || value < 300.0f) out = 0.0f;
    else {
        int x0 = x - R > 0 ? x - R : 0, x1 = x + R + 1 < W ? x + R + 1 : W;
        int y0 = y - R > 0 ? y - R : 0, y1 = y + R + 1 < H ? y + R + 1 : H;
        float sum1 = 0.0f, sum2 = 0.0f;
        if (BILATERAL) {
            // Two taps per trip on float pairs (the loop is VALU-issue bound: 169 taps x ~37 instructions per pixel before).
            // Same operations in the same order per tap; the sums take the taps left to right as before.
            // Measured and rejected earlier: the 169 taps fully unrolled with compile-time offsets (62.6 us) and rows
            // looped / columns unrolled (70.4 us) against the plain clamped loop (57.4 us; 50.3 with hd_expf_nonpos).
            for (int cy = y0; cy < y1; ++cy) {
                const float dy = (float)y - (float)cy;
                const float dy2 = dy * dy;
                const float *row = &tile[(cy - by + R) * TW + (R - bx)];
                // dx = (float)x - (float)cx is a small integer: stepping it by 2 gives the same values as converting cx
                const float *q = row + x0;
                v2f dx = {(float)(x - x0), (float)(x - x0 - 1)};
                int left = x1 - x0;
                for (; left >= 2; left -= 2, q += 2, dx -= 2.0f) {
                    const v2f tmp = {q[0], q[1]};
                    const v2f space2 = dx * dx + dy2;
                    const v2f dv = value - tmp;
                    const v2f color2 = dv * dv;
                    const v2f w = expf_nonpos_pair(-(space2 * 0.024691358f + color2 * 0.000555556f));
                    const v2f tw = tmp * w;
                    sum1 += tw.x; sum2 += w.x;
                    sum1 += tw.y; sum2 += w.y;
                }
                if (left) {
                    const float tmp = q[0];
                    const float space2 = dx.x * dx.x + dy2;
                    const float dv = value - tmp;
                    const float color2 = dv * dv;
                    const float weight = expf_nonpos_dev(-(space2 * 0.024691358f + color2 * 0.000555556f));
                    sum1 += tmp * weight;
                    sum2 += weight;
                }
            }
        } else {
            for (int cy = y0; cy < y1; ++cy) {
                const float dy = (float)y - (float)cy;
                const float *row = &tile[(cy - by + R) * TW + (R - bx)];
                for (int cx = x0; cx < x1; ++cx) {
                    float tmp = row[cx];
                    float dx = (float)x - (float)cx;
                    if (tmp > 300.0f && hd_fabsf(tmp - value) < 100.0f) {
                        float weight = hd_expf(-((dx * dx + dy * dy) / (2.0f * 3.0f * 3.0f)));
                        sum1 += tmp * weight;
                        sum2 += weight;
                    }
                }
            }
        }
        out = (sum1 / sum2) * adj;
    }
    const int i = y * W + x;
    filtered[i] = out;
    // metriciseDepth (depth_metric_raw.frag:29-42, depth_metric_filtered.frag:29-41)
    const uint32_t hi = (uint32_t)(maxD / depthFactor), lo = (uint32_t)(0.3f / depthFactor);
    const float fhi = maxD / depthFactor, flo = 0.3f / depthFactor;
    uint32_t v = raw[i];
    metric[i] = (v > hi || v < lo) ? 0.0f : (float)v * depthFactor;
    metric_f[i] = (out > fhi || out < flo) ? 0.0f : out * depthFactor;
}

// ---------------------------------------------------------------------------------------------
// P3: vertex / PCA normal / radius (depth_vertex_normal_radius.frag:23-68, geometry.glsl:63-244); getNormalPCA: pca_normal.h
__global__ __launch_bounds__(256) void k_vertex_normal_radius(Cam cam, const float *__restrict__ depth_metric,
                                                              const float *__restrict__ depth_metric_f,
                                                              float4 *__restrict__ vertex_raw,
                                                              float4 *__restrict__ vertex_filtered,
                                                              float4 *__restrict__ normal,
                                                              float4 *__restrict__ normal_pca,
                                                              float *__restrict__ radius, float radius_mult,
                                                              int use_pca)
{
    constexpr int R = 3, TW = TB + 2 * R;
    __shared__ float tile[TW * TW];
    const int W = cam.W, H = cam.H;
    const int bx = blockIdx.x * TB, by = blockIdx.y * TB;
    for (int i = threadIdx.y * TB + threadIdx.x; i < TW * TW; i += TB * TB) {
        int tx = i % TW, ty = i / TW;
        int gx = clampi(bx + tx - R, 0, W - 1), gy = clampi(by + ty - R, 0, H - 1);
        tile[i] = depth_metric_f[gy * W + gx];
    }
    __syncthreads();
    const int px = bx + threadIdx.x, py = by + threadIdx.y;
    if (px >= W || py >= H) return;
    const int i = py * W + px;
    const float cx = cam.cx, cy = cam.cy, camz = cam.camz, camw = cam.camw;
    const float zr = depth_metric[i], zf = tile[(threadIdx.y + R) * TW + threadIdx.x + R];
    f3 vr = mk3(((float)px - cx) * zr * camz, ((float)py - cy) * zr * camw, zr);
    f3 vf = mk3(((float)px - cx) * zf * camz, ((float)py - cy) * zf * camw, zf);
    f3 n = mk3(0.0f, 0.0f, 0.0f);
    if (use_pca) {
        n = pca_normal_tile(tile, TW, bx - R, by - R, W, H, hd_uv_fragment(px, W), hd_uv_fragment(py, H), vf.z, cx, cy, camz, camw);
    } else {
        // central differences (geometry.glsl:36-51) gated by checkNeighbours on the RAW depth (utils.glsl:23-41)
        bool ok = depth_metric[py * W + clampi(px - 1, 0, W - 1)] != 0.0f &&
                  depth_metric[clampi(py - 1, 0, H - 1) * W + px] != 0.0f &&
                  depth_metric[py * W + clampi(px + 1, 0, W - 1)] != 0.0f &&
                  depth_metric[clampi(py + 1, 0, H - 1) * W + px] != 0.0f;
        if (ok) {
            const int lx = threadIdx.x + R, ly = threadIdx.y + R;
            float z;
            z = tile[ly * TW + lx + 1]; f3 vxf = mk3(((float)(px + 1) - cx) * z * camz, ((float)py - cy) * z * camw, z);
            z = tile[ly * TW + lx - 1]; f3 vxb = mk3(((float)(px - 1) - cx) * z * camz, ((float)py - cy) * z * camw, z);
            z = tile[(ly + 1) * TW + lx]; f3 vyf = mk3(((float)px - cx) * z * camz, ((float)(py + 1) - cy) * z * camw, z);
            z = tile[(ly - 1) * TW + lx]; f3 vyb = mk3(((float)px - cx) * z * camz, ((float)(py - 1) - cy) * z * camw, z);
            f3 del_x = sub3(scale3(add3(vxb, vf), 0.5f), scale3(add3(vxf, vf), 0.5f));
            f3 del_y = sub3(scale3(add3(vyb, vf), 0.5f), scale3(add3(vyf, vf), 0.5f));
            n = normalize3(cross3(del_x, del_y));
        }
    }
    float radius_init = radius_mult * get_radius(vf.z, n.z, camz, camw);
    // side output for the fusion: the un-invalidated normal + radius.  data.vert RECOMPUTES both for a new point (data.vert:83-96);
    // where its inputs are the fragment shader's the result is this one, elsewhere k_associate recomputes (pca_normal.h)
    normal_pca[i] = make_float4(n.x, n.y, n.z, radius_init);
    if (len3(n) < 0.3f || vr.z < 0.3f || vf.z < 0.3f) {
        vr = mk3(0, 0, 0); vf = mk3(0, 0, 0); n = mk3(0, 0, 0); radius_init = 0.0f;
    }
    vertex_raw[i] = make_float4(vr.x, vr.y, vr.z,
                                radial_confidence(hd_px_fragment(px, W), hd_px_fragment(py, H), cx, cy, cam.max_dist, 1.0f));
    vertex_filtered[i] = make_float4(vf.x, vf.y, vf.z, 1.0f);
    normal[i] = make_float4(n.x, n.y, n.z, radius_init);
    radius[i] = radius_init;
}

// ---------------------------------------------------------------------------------------------
// P4 + P5: HRBF gradient / Hessian at the pixel's own vertex over its (2w+1)^2 window
// (depth_curvature_gradient.frag:28-142 + hrbfbase.glsl:37-124,147-195).  The neighbour list of the
// GLSL is never materialised: gradient and Hessian sums are accumulated in one pass in the same
// visiting order (x outer, y inner).  LDS: vertex xyz + normal xyzw for tile + halo.
// per-centre quantities are formed once per texel while staging: 10 n, T^2, T^4, 60 / T^4 and -20 / T^2 were recomputed
// (two of them divisions) for each of the <= 49 pixels a texel is a neighbour of
struct alignas(16) NbTexel { float px, py, pz, valid, sx, sy, sz, T2, T2T2, s3, m20, pad; };

// the nine running sums of a pixel: gradient (x, y | z) and the six Hessian entries the curvature consumes, paired the way
// the packed arithmetic of curv_neighbour_fast produces them
struct CurvSums { v2f grxy; float grz; v2f g04; float g1; v2f g25; float g8; };

// One neighbour, literally: getWeightH (hrbfbase.glsl:37-69) and, inside the support, getWeightT (:72-124, only the 18
// entries the Hessian consumes) — r = sqrt(d2 / T2) is formed once for both.
__device__ __forceinline__ void curv_neighbour_literal(const NbTexel &me, const NbTexel &nb, CurvSums &a)
{
    const float sx = nb.sx, sy = nb.sy, sz = nb.sz;
    const float vx = me.px - nb.px, vy = me.py - nb.py, vz = me.pz - nb.pz;
    const float d2 = (vx * vx + vy * vy) + vz * vz;
    const float T2 = nb.T2;
    float h0, h1, h2, h4, h5, h8;
    bool third = false;
    float t0 = 0, t1_ = 0, t2_ = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0, t8 = 0, t13 = 0, t14 = 0, t16 = 0, t17 = 0, t26 = 0;
    if (d2 > T2) { h0 = h1 = h2 = h4 = h5 = h8 = 0.0f; }
    else if (d2 == 0.0f) { h0 = h4 = h8 = nb.m20; h1 = h2 = h5 = 0.0f; }
    else {
        const float r = hd_sqrtf(d2 / T2);
        const float s = 1.0f - r;
        {
            float s2 = s * s;
            float t1 = 20.0f * s2 / (nb.T2T2 * r);
            float t2 = -r * s * T2;
            h0 = t1 * (3.0f * (vx * vx) + t2);
            h1 = t1 * 3.0f * vx * vy;
            h2 = t1 * 3.0f * vx * vz;
            h4 = t1 * (3.0f * (vy * vy) + t2);
            h5 = t1 * 3.0f * vy * vz;
            h8 = t1 * (3.0f * (vz * vz) + t2);
        }
        third = true;
        float s2 = r - 2.0f + 1.0f / r;
        float s3 = nb.s3;
        float s4 = 1.0f / (r * r);
        float prx = vx / (T2 * r), pry = vy / (T2 * r), prz = vz / (T2 * r);
        float qx = prx - s4 * prx, qy = pry - s4 * pry, qz = prz - s4 * prz;
        float tss = T2 * s * s;
        t0 = s3 * (tss * prx + 2.0f * vx * s2 + vx * vx * qx);
        t1_ = s3 * vy * (qx * vx + s2);
        t2_ = s3 * vz * (qx * vx + s2);
        t3 = s3 * (tss * pry + vx * vx * qy);
        t4 = s3 * vx * (qy * vy + s2);
        t5 = s3 * vx * vz * qy;
        t6 = s3 * (tss * prz + vx * vx * qz);
        t7 = s3 * vx * vy * qz;
        t8 = s3 * vx * (qz * vz + s2);
        t13 = s3 * (tss * pry + 2.0f * vy * s2 + vy * vy * qy);
        t14 = s3 * vz * (qy * vy + s2);
        t16 = s3 * (tss * prz + vy * vy * qz);
        t17 = s3 * vy * (qz * vz + s2);
        t26 = s3 * (tss * prz + 2.0f * vz * s2 + vz * vz * qz);
    }
    a.grxy.x -= (sx * h0 + sy * h1) + sz * h2;
    a.grxy.y -= (sx * h1 + sy * h4) + sz * h5;
    a.grz -= (sx * h2 + sy * h5) + sz * h8;
    if (third) {
        a.g04.x -= (sx * t0 + sy * t1_) + sz * t2_;
        a.g1 -= (sx * t3 + sy * t4) + sz * t5;
        a.g25.x -= (sx * t6 + sy * t7) + sz * t8;
        a.g04.y -= (sx * t4 + sy * t13) + sz * t14;      // hw[12] = t[4]
        a.g25.y -= (sx * t7 + sy * t16) + sz * t17;      // hw[15] = t[7]
        a.g8 -= (sx * t8 + sy * t17) + sz * t26;         // hw[24] = t[8], hw[25] = t[17]
    }
    // outside the support the third derivatives are zero: "g -= 0" in the oracle, a no-op in IEEE
}

// ---- the same neighbour for TAME tiles (curv_texel_is_tame): identical IEEE operations, fewer instructions ----
// Correctly rounded division without the range scaling and special-case fix-up of the compiler's expansion
// (v_div_scale / v_div_fmas / v_div_fixup are identities when numerator, denominator and quotient are far from the ends of
// the exponent range — guaranteed by the tame ranges below): reciprocal refined once, quotient refined twice, all in FMAs.
__device__ __forceinline__ float rcp_refined(const float b)
{
    const float y = __builtin_amdgcn_rcpf(b);
    return hd_fmaf(hd_fmaf(-b, y, 1.0f), y, y);
}
__device__ __forceinline__ float div_by(const float a, const float b, const float y /* rcp_refined(b) */)
{
    float q = a * y;
    q = hd_fmaf(hd_fmaf(-b, q, a), y, q);
    return hd_fmaf(hd_fmaf(-b, q, a), y, q);
}
__device__ __forceinline__ v2f div_by(const v2f a, const float b, const float y)
{
    const v2f nb = (v2f)(-b), yy = (v2f)(y);
    v2f q = a * yy;
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(nb, q, a), yy, q);
    return __builtin_elementwise_fma(__builtin_elementwise_fma(nb, q, a), yy, q);
}
// correctly rounded sqrt for arguments >= 2^-96 (hrbf_probe_sqrt_rounding checks every float): v_sqrt_f32 + the two
// residual tests of the compiler's expansion, without its rescaling of tiny arguments and its 0 / inf re-check
__device__ __forceinline__ float sqrt_normal(const float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = hd_u2f(hd_f2u(s) - 1u), up = hd_u2f(hd_f2u(s) + 1u);
    const float ed = hd_fmaf(-dn, s, x), eu = hd_fmaf(-up, s, x);
    const float t = ed <= 0.0f ? dn : s;
    return eu > 0.0f ? up : t;
}
__device__ __forceinline__ float keep_scalar(float x)   // stops the selector from re-pairing two scalar results through v_mov
{
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ v2f swap2(const v2f a) { return __builtin_shufflevector(a, a, 1, 0); }

// What curv_neighbour_fast assumes of every texel that can take part (depth > 0.3): coordinates +0 or between 2^-20 and
// 2^10 in magnitude (differences are then zero or >= 2^-43, squared distances zero or >= 2^-86), T^2 between 2^-26 and
// 2^7, |10 n| below 2^10.  NaNs fail every comparison.
__device__ __forceinline__ bool curv_coord_is_tame(const float c)
{
    const float m = hd_fabsf(c);
    return hd_f2u(c) == 0u || (m >= 0x1p-20f && m < 0x1p10f);   // +0 only: differences of such coordinates are never -0
}
__device__ __forceinline__ bool curv_texel_is_tame(const NbTexel &t)
{
    return curv_coord_is_tame(t.px) && curv_coord_is_tame(t.py) && curv_coord_is_tame(t.pz) && t.T2 >= 0x1p-26f &&
           t.T2 <= 0x1p7f && hd_fabsf(t.sx) < 0x1p10f && hd_fabsf(t.sy) < 0x1p10f && hd_fabsf(t.sz) < 0x1p10f;
}

// Main case (0 < d2 <= T2, r >= 2^-20) on float pairs: x / y entries share a register pair, z entries are scalar.
// Every expression keeps the literal's association; a + b is written b + a in places (the same bits).
__device__ __forceinline__ void curv_neighbour_fast(const v2f V, const float vz, const v2f SQ, const float vz2, const float qd,
                                                    const NbTexel &nb, CurvSums &a)
{
    const v2f S = {nb.sx, nb.sy};
    const float sz = nb.sz, T2 = nb.T2, s3 = nb.s3;
    const float r = sqrt_normal(qd);
    const float s = 1.0f - r;
    const float yr = rcp_refined(r);
    {
        const float den = nb.T2T2 * r;
        const float t1 = div_by(20.0f * (s * s), den, rcp_refined(den));
        const float t2 = -r * s * T2;
        const v2f H04 = t1 * (3.0f * SQ + t2);                   // h0, h4
        const float h8 = t1 * (3.0f * vz2 + t2);
        const v2f AB = (t1 * 3.0f) * V;
        const float h1 = AB.x * V.y;
        const v2f H25 = AB * vz;                                  // h2, h5
        a.grxy -= (S * H04 + swap2(S) * h1) + sz * H25;
        const v2f P = S * H25;
        a.grz -= keep_scalar(P.x + P.y) + sz * h8;
    }
    const float s2 = (r - 2.0f) + div_by(1.0f, r, yr);
    const float rr = r * r;
    const float s4 = div_by(1.0f, rr, rcp_refined(rr));
    const float D = T2 * r, yD = rcp_refined(D);
    const v2f PR = div_by(V, D, yD);
    const float prz = div_by(vz, D, yD);
    const v2f Q = PR - s4 * PR;
    const float qz = prz - s4 * prz;
    const float tss = T2 * s * s;
    const v2f TP = tss * PR;
    const float tpz = tss * prz;
    const v2f A = Q * V + s2;                                     // qx vx + s2, qy vy + s2
    const float az = qz * vz + s2;
    const v2f S3V = s3 * V;
    const float s3vz = s3 * vz;
    const v2f T0_13 = s3 * ((TP + (2.0f * V) * s2) + SQ * Q);     // t0, t13
    const float t26 = s3 * ((tpz + (2.0f * vz) * s2) + vz2 * qz);
    const v2f T6_16 = s3 * (tpz + SQ * qz);                       // t6, t16
    const float t3 = s3 * (TP.y + SQ.x * Q.y);
    const v2f T1_4 = swap2(S3V) * A;                              // t1 = (s3 vy)(qx vx + s2), t4 = (s3 vx)(qy vy + s2)
    const v2f T2_14 = s3vz * A;                                   // t2, t14
    const v2f T8_17 = S3V * az;                                   // t8, t17
    const float t5 = (S3V.x * vz) * Q.y;
    const float t7 = (S3V.x * V.y) * qz;
    a.g04 -= (S * T0_13 + swap2(S) * T1_4) + sz * T2_14;
    a.g1 -= (S.x * t3 + S.y * T1_4.y) + sz * t5;
    a.g25 -= (S * T6_16 + swap2(S) * t7) + sz * T8_17;
    const v2f P8 = S * T8_17;
    a.g8 -= keep_scalar(P8.x + P8.y) + sz * t26;
}

// test probe: div_by against the compiler's division on pseudo-random operands spread over the ranges curv_neighbour_fast
// can meet (numerator 0 or 2^-103 < |a| < 2^24, denominator 2^-74 < b < 2^15); out = {mismatches, cases}
__global__ void k_probe_division(unsigned long long *out, uint32_t rounds)
{
    unsigned long long bad = 0, total = 0;
    uint64_t st = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (uint32_t i = 0; i < rounds; ++i) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t ra = (uint32_t)(st >> 32), rb = (uint32_t)st * 2654435761u ^ (uint32_t)(st >> 17);
        const uint32_t ea = 127u - 102u + (ra >> 23) % 126u, eb = 127u - 73u + (rb >> 23) % 88u;
        float x = hd_u2f((ra & 0x807fffffu) | (ea << 23));
        const float y = hd_u2f((rb & 0x007fffffu) | (eb << 23));
        if ((ra & 0x7f000000u) == 0u) x = 0.0f;                       // exact zeros now and then
        const float ref = x / y, got = div_by(x, y, rcp_refined(y));
        const v2f g2 = div_by((v2f){x, -x}, y, rcp_refined(y));
        // (a -0 numerator comes out +0: tame coordinates exclude -0, so differences are never -0)
        bad += hd_f2u(ref) != hd_f2u(got) || hd_f2u(g2.x) != hd_f2u(ref) || (x != 0.0f && hd_f2u(g2.y) != hd_f2u(-x / y));
        ++total;
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicAdd(&out[1], total);
}

int pre_probe_division(hipStream_t s, unsigned long long *d_out2)
{
    if (hipMemsetAsync(d_out2, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_probe_division, dim3(4096), dim3(256), 0, s, d_out2, 4096u);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// The window is the shader's float-stepped walk, literally (hd_window_axis): x outer, y inner; the tile rows of the y walk are
// the same for every x and are packed once (8 bits each, <= 8 of them)
template <bool TAME>
__device__ __forceinline__ int curv_window(const NbTexel *__restrict__ tile, const NbTexel &me, const hd_window wx, const hd_window wy,
                                           int W, int H, int bx, int by, CurvSums &a)
{
    constexpr int RMAX = 3, TW = TB + 2 * RMAX;
    int n = 0;
    unsigned long long rows = 0ull;
    int ny = 0;
    for (float fj = wy.lo; fj <= wy.hi && ny < 8; fj += wy.step, ++ny)
        rows |= (unsigned long long)(uint32_t)(hd_window_texel(fj, H) - by + RMAX) << (8 * ny);
    for (float fi = wx.lo; fi <= wx.hi; fi += wx.step) {
        const int lx = hd_window_texel(fi, W) - bx + RMAX;
        for (int k = 0; k < ny; ++k) {
            const int ly = (int)((rows >> (8 * k)) & 255ull);
            const NbTexel nb = tile[ly * TW + lx];
            const float vz = me.pz - nb.pz;
            if (!(hd_fabsf(nb.pz - me.pz) < 0.10f && nb.valid > 0.0f)) continue;
            n++;
            if (!TAME) { curv_neighbour_literal(me, nb, a); continue; }
            const v2f V = (v2f){me.px, me.py} - (v2f){nb.px, nb.py};
            const v2f SQ = V * V;
            const float vz2 = vz * vz;
            const float d2 = keep_scalar(SQ.x + SQ.y) + vz2;
            // outside the support: the literal subtracts (s * 0) sums — zeros, since 10 n is finite — from sums that are
            // never -0 (they start at +0 and x - x is +0): nothing to do
            if (d2 > nb.T2) continue;
            if (d2 == 0.0f) {   // the pixel itself: h0 = h4 = h8 = -20 / T^2, the products with the zero entries vanish
                a.grxy -= (v2f){nb.sx, nb.sy} * nb.m20;
                a.grz -= nb.sz * nb.m20;
                continue;
            }
            const float qd = div_by(d2, nb.T2, rcp_refined(nb.T2));
            if (__builtin_amdgcn_ballot_w64(qd < 0x1p-40f) != 0ull) curv_neighbour_literal(me, nb, a);   // r < 2^-20: not tame
            else curv_neighbour_fast(V, vz, SQ, vz2, qd, nb, a);
        }
    }
    return n;
}

template <bool LEVEL0 /* also write level 0 of the registration pyramids for this pixel (frame path) */>
__global__ __launch_bounds__(256) void k_curvature(Cam cam, const float4 *__restrict__ vertex_filtered,
                                                   const float4 *__restrict__ normal_in,
                                                   float4 *__restrict__ curv1, float4 *__restrict__ curv2,
                                                   float *__restrict__ gradmag, float4 *__restrict__ normal_out,
                                                   float win, Level0Args l0)
{
    constexpr int RMAX = 3, TW = TB + 2 * RMAX;
    __shared__ NbTexel tile[TW * TW];
    __shared__ uint32_t s_untame;
    const int W = cam.W, H = cam.H;
    const int bx = blockIdx.x * TB, by = blockIdx.y * TB;
    if (threadIdx.x == 0 && threadIdx.y == 0) s_untame = 0u;
    __syncthreads();
    for (int i = threadIdx.y * TB + threadIdx.x; i < TW * TW; i += TB * TB) {
        int tx = i % TW, ty = i / TW;
        int gx = bx + tx - RMAX, gy = by + ty - RMAX;
        NbTexel t;
        t.px = t.py = t.pz = t.valid = t.sx = t.sy = t.sz = t.T2 = t.T2T2 = t.s3 = t.m20 = t.pad = 0.0f;
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
            float4 v = vertex_filtered[gy * W + gx], n = normal_in[gy * W + gx];
            t.px = v.x; t.py = v.y; t.pz = v.z;
            t.sx = 10.0f * n.x; t.sy = 10.0f * n.y; t.sz = 10.0f * n.z;
            t.T2 = n.w * n.w; t.T2T2 = t.T2 * t.T2; t.s3 = 60.0f / t.T2T2; t.m20 = -20.0f / t.T2;
            t.valid = (v.z > 0.3f && len3(mk3(n.x, n.y, n.z)) > 0.8f) ? 1.0f : 0.0f;
            // a texel takes part as a neighbour (valid) or as the centre (depth > 0.3, |n| > 0.5)
            if (v.z > 0.3f && len3(mk3(n.x, n.y, n.z)) > 0.5f && !curv_texel_is_tame(t)) s_untame = 1u;
        }
        tile[i] = t;
    }
    __syncthreads();
    const int px = bx + threadIdx.x, py = by + threadIdx.y;
    if (px >= W || py >= H) return;
    const int i = py * W + px;
    const NbTexel me = tile[(threadIdx.y + RMAX) * TW + threadIdx.x + RMAX];
    const float4 me_n = normal_in[i];   // the centre's own normal (the tile keeps 10 n)
    float4 pcmax = make_float4(0, 0, 0, 1000.0f), pcmin = make_float4(0, 0, 0, 1000.0f), nopt = make_float4(0, 0, 0, 0);
    float gmag = 0.0f;
    if (me.pz > 0.3f && len3(mk3(me_n.x, me_n.y, me_n.z)) > 0.5f) {
        float k1 = 1000.0f, k2 = 1000.0f;
        f3 pmax = mk3(0, 0, 0), pmin = mk3(0, 0, 0);
        const hd_window wx = hd_window_axis(px, W, win), wy = hd_window_axis(py, H, win);
        CurvSums a;
        a.grxy = (v2f)(0.0f); a.grz = 0.0f; a.g04 = (v2f)(0.0f); a.g1 = 0.0f; a.g25 = (v2f)(0.0f); a.g8 = 0.0f;
        const int n = s_untame ? curv_window<false>(tile, me, wx, wy, W, H, bx, by, a) : curv_window<true>(tile, me, wx, wy, W, H, bx, by, a);
        const float grx = a.grxy.x, gry = a.grxy.y, grz = a.grz;
        const float g0 = a.g04.x, g1 = a.g1, g2 = a.g25.x, g4 = a.g04.y, g5 = a.g25.y, g8 = a.g8;
        if (n > 15) {
            float4 vn = normal_in[i];
            gmag = hd_fabsf((grx * vn.x + gry * vn.y) + grz * vn.z);
            f3 gn = normalize3(mk3(grx, gry, grz));
            nopt = make_float4(gn.x, gn.y, gn.z, vn.w);
            const float ga = grx, gb = gry, gc = grz;
            float g2c = gc * gc * gc;
            float h_x = -ga / gc, h_y = -gb / gc;
            float h_xx = (((2.0f * ga * gc * g2 - ga * ga * g8) - gc * gc * g0)) / g2c;
            float h_xy = (((ga * gc * g5 + gb * gc * g2) - ga * gb * g8) - gc * gc * g1) / g2c;
            float h_yy = (((2.0f * gb * gc * g5 - gb * gb * g8) - gc * gc * g4)) / g2c;
            f3 r_u = mk3(1.0f, 0.0f, h_x), r_v = mk3(0.0f, 1.0f, h_y);
            float E = 1.0f + h_x * h_x, F = h_x * h_y, G = 1.0f + h_y * h_y;
            float length = hd_sqrtf((h_x * h_x + h_y * h_y) + 1.0f);
            float L = h_xx / length, M = h_xy / length, N = h_yy / length;
            float den = E * G - F * F;
            float curvature_g = (L * N - M * M) / den;
            float curvature_m = ((E * N + G * L) - 2.0f * F * M) / (2.0f * den);
            if (!hd_isnanf(curvature_g) && !hd_isnanf(curvature_m)) {
                float delta = curvature_m * curvature_m - curvature_g;
                if (delta < 0.0f) delta = 0.0f;
                float sd = hd_sqrtf(delta);
                k1 = curvature_m + sd;
                k2 = curvature_m - sd;
                float lmax = -(M - k1 * F) / (N - k1 * G);
                float lmin = -(M - k2 * F) / (N - k2 * G);
                pmax = normalize3(add3(r_u, scale3(r_v, lmax)));
                pmin = normalize3(add3(r_u, scale3(r_v, lmin)));
            }
        }
        pcmax = make_float4(pmax.x, pmax.y, pmax.z, k1);
        pcmin = make_float4(pmin.x, pmin.y, pmin.z, k2);
    }
    curv1[i] = pcmax; curv2[i] = pcmin; gradmag[i] = gmag;
    normal_out[i] = nopt;   // updateNormalRad: NORMAL <- NORMAL_OPT (HRBFFusion.cpp:1301-1310)
    if (LEVEL0)   // what k_odo_level0 would read back for this pixel is what was just stored
        odo_level0_pixel(i, W * H, l0.L, l0.src, l0.dp->should_fill_in, l0.f2f, l0.curv_thr, vertex_filtered[i], nopt, pcmax, pcmin,
                         l0.pack ? &l0.dp->pose : nullptr);
}

// VertexConfidence (depth_confidence_evaluation.frag:37-50); weighting lives in device memory so
// the frame needs no host round trip between registration and fusion.
__global__ void k_confidence(Cam cam, const float *__restrict__ gradmag, float *__restrict__ confidence,
                             const float *__restrict__ weighting, int use_conf_eval, float epsilon)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cam.W * cam.H) return;
    int py = i / cam.W, px = i - py * cam.W;
    float conf = radial_confidence(hd_px_fragment(px, cam.W), hd_px_fragment(py, cam.H), cam.cx, cam.cy, cam.max_dist, *weighting);
    if (use_conf_eval > 0) conf = conf * hd_expf(-epsilon / hd_sqrtf(gradmag[i]));
    confidence[i] = conf;
}

// ---------------------------------------------------------------------------------------------
static inline dim3 grid2d(const Cam &c) { return dim3((c.W + TB - 1) / TB, (c.H + TB - 1) / TB); }

void launch_filter_metric(hipStream_t s, const Cam &cam, const uint16_t *raw, float *filtered, float *metric,
                          float *metric_f, float depthFactor, float maxD, int bilateral, const uint8_t *ride_src,
                          uint8_t *ride_dst, size_t ride_bytes)
{
    dim3 b(TB, TB), g = grid2d(cam);
    RideCopy ride = {ride_src, ride_dst, ride_src ? ride_bytes : 0};
    if (ride.bytes) g.y += (unsigned)((ride.bytes / 16 + (size_t)g.x * TB * TB - 1) / ((size_t)g.x * TB * TB));   // one 16-byte word per lane
    if (bilateral)
        hipLaunchKernelGGL(k_filter_metric<true>, g, b, 0, s, cam, raw, filtered, metric, metric_f, depthFactor, maxD, ride);
    else
        hipLaunchKernelGGL(k_filter_metric<false>, g, b, 0, s, cam, raw, filtered, metric, metric_f, depthFactor, maxD, ride);
}
void launch_vertex_normal_radius(hipStream_t s, const Cam &cam, const float *dm, const float *dmf, float4 *vr,
                                 float4 *vf, float4 *n, float4 *npca, float *radius, float radius_mult, int use_pca)
{
    hipLaunchKernelGGL(k_vertex_normal_radius, grid2d(cam), dim3(TB, TB), 0, s, cam, dm, dmf, vr, vf, n, npca, radius,
                       radius_mult, use_pca);
}
void launch_curvature(hipStream_t s, const Cam &cam, const float4 *vf, const float4 *normal_in, float4 *c1, float4 *c2,
                      float *gradmag, float4 *normal_out, float win)
{
    hipLaunchKernelGGL(k_curvature<false>, grid2d(cam), dim3(TB, TB), 0, s, cam, vf, normal_in, c1, c2, gradmag, normal_out, win,
                       Level0Args{});
}
void launch_curvature_level0(hipStream_t s, const Cam &cam, const float4 *vf, const float4 *normal_in, float4 *c1, float4 *c2,
                             float *gradmag, float4 *normal_out, float win, const Level0Args &l0)
{
    hipLaunchKernelGGL(k_curvature<true>, grid2d(cam), dim3(TB, TB), 0, s, cam, vf, normal_in, c1, c2, gradmag, normal_out, win, l0);
}
void launch_confidence(hipStream_t s, const Cam &cam, const float *gradmag, float *conf, const float *weighting,
                       int use_conf_eval, float eps)
{
    int P = cam.W * cam.H;
    hipLaunchKernelGGL(k_confidence, dim3((P + 255) / 256), dim3(256), 0, s, cam, gradmag, conf, weighting, use_conf_eval, eps);
}

// k_odo.hip — frame-to-model RGB-D odometry for gfx950, fully device resident.
//
// Replaces RGBDOdometry (Core/src/Utils/RGBDOdometry.cpp:183-247,660-1249), the CUDA kernels of
// Core/src/Cuda/reduce.cu:253-1359 (icpStep / rgbStep / computeRgbResidual / so3Step) and
// Core/src/Cuda/cudafuncs.cu:57-1028 (map pyramids, copies, Sobel, ...), and the host-side Eigen
// Gauss-Newton loop (LDLT, rodrigues, SE3 update; Core/src/Utils/OdometryProvider.h:35-93).
//
// MI355X design:
//  * the reference synchronises with the host ~70 times per frame (kernel + cudaDeviceSynchronize +
//    29-float download per step).  Here the whole registration is a fixed launch sequence on one
//    stream; the 6x6 / 3x3 solves and the SE3 update run in a one-workgroup kernel and every
//    iteration's state (OdoState) stays in HBM/L2.  No host round trip.
//  * reductions: one pixel per lane, 256-thread workgroups; every addend goes through the exact
//    limb accumulator (hrbf_detmath.h) -> wave64 __shfl_down tree on int64 limbs -> LDS -> one
//    partial row per workgroup -> summed by the solve kernel.  Integer sums make the result
//    independent of wave/block shape and of the number of GPUs (RCCL all-reduce of int64).
//  * warp-32 shuffle emulation through shared memory of the reference (reduce.cu:58-80) has no
//    counterpart: gfx950 shuffles natively across 64 lanes.
#include "common.h"
#include "kernels.h"

#define RB 256               // reduction workgroup = 4 wave64
#define NLIMB(n) ((n) * 3)

struct OdoState {
    float Rprev[9], tprev[3], Rcurr[9], tcurr[3], Rprev_inv[9];
    double Rt[16];                    // resultRt, row-major
    double resultR[9], lastResultR[9];
    float R_lr[9];
    float so3_lastError, so3_lastCount;
    int so3_done;
    float basis[9], kinv[9], krlr[9]; // SO3 step operands
    float krk[9], kt[3];              // RGB residual operands
    float lastRGBError;
    int gn_break;
    float res_icp[2];
    float last_icp_error, last_icp_count;
    float sigmaVal;
    unsigned int ticket;              // last-workgroup election of the fused reduce+solve kernels
    // persistent SO3 kernel: per-iteration chunk tickets and completed-chunk counters, exit counter (publisher election)
    // word[it] = {chunks of iteration `it` done (low 32) | tickets of iteration it+1 drawn (high 32)}: one 64-bit atomic
    // arrives at iteration it AND draws the chunk of iteration it+1; so3_ticket0 = tickets of iteration 0
    unsigned long long so3_word[12];
    unsigned int so3_ticket0, so3_exit;
    int bar_timeout;                  // this frame: a bounded poll of the SO3 kernel gave up (never expected)
    int bar_timeouts_total;           // sticky count of such frames (hrbf_get_status)
};

size_t odo_state_bytes() { return sizeof(OdoState); }
void odo_release(OdoBuffers &ob)
{
    for (int k = 0; k < 2; ++k)
        if (ob.gn_graph_exec[k]) { hipGraphExecDestroy((hipGraphExec_t)ob.gn_graph_exec[k]); ob.gn_graph_exec[k] = nullptr; }
}
#define SO3_ITERS 10
size_t odo_slot_bytes() { return sizeof(long long) * (32 * 87 * 2 + 64 * 2 + SO3_ITERS * 32 * 33); }   // one SO3 slot set per iteration

#define PLN(base, k, rows, cols, y, x) ((base)[((size_t)(k) * (rows) + (y)) * (cols) + (x)])

// ------------------------------------------------------------------------------------------
// exact workgroup reduction of N floats per lane.
//   lane: float -> five signed 25-bit limbs (hd_limbs25, ~20 branch-free VALU ops)  [exact]
//   wave: transposing butterfly over the V = 5N limb registers: every stage pairs lanes (l, l ^ mask), one
//         lane of the pair keeps the lower half of the register file and the other the upper half, so the
//         work halves per stage (~2.3 V instructions in all instead of 6 V for V independent add trees).
//         xor 32 / xor 16 are v_permlane{32,16}_swap (two registers exchanged per instruction), the four
//         in-row stages are DPP adds (row_ror:8 = xor 8, row_half_mirror = xor 7, quad_perm = xor 2, xor 1).
//         64 lanes x 2^25 fits an int32: no carries anywhere.  Afterwards lane l holds ceil(V/64) totals.
//   block: the wave totals are parked in LDS; after the barrier thread t < N folds the waves for value t
//          and converts to the 40-bit limb form
//   grid: one 64-bit atomicAdd per limb into slot (blockIdx % ODO_SLOTS) — integer adds commute, so the
//         result does not depend on arrival order.  The consumer sums the ODO_SLOTS rows and re-zeroes them.
#define ODO_SLOTS 32
#define RES_SLOTS 64

template <int CTRL>
__device__ __forceinline__ int32_t dpp_i32(int32_t v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

// stage S of the butterfly on a[0..V): selector bit 5-S of the lane id
template <int VM, int V, int S>
__device__ __forceinline__ void wave_transpose_sum(int32_t (&a)[VM], const int lane)
{
    if constexpr (S < 6) {
        constexpr int h = (V + 1) / 2;
#pragma unroll
        for (int i = 0; i < h; ++i) {
            const int32_t x = a[i], y = (i + h < V) ? a[i + h] : 0;
            if constexpr (S == 0) {
                const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)y, false, false);
                a[i] = (int32_t)(r[0] + r[1]);
            } else if constexpr (S == 1) {
                const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)y, false, false);
                a[i] = (int32_t)(r[0] + r[1]);
            } else {
                const bool sel = ((lane >> (5 - S)) & 1) != 0;
                const int32_t keep = sel ? y : x, send = sel ? x : y;
                constexpr int ctrl = S == 2 ? 0x128 : S == 3 ? 0x141 : S == 4 ? 0x4e : 0xb1;
                a[i] = keep + dpp_i32<ctrl>(send);
            }
        }
        wave_transpose_sum<VM, h, S + 1>(a, lane);
    }
}

// ---- the same wave totals through doubles, for waves whose values are all below 2^6 in magnitude ----
// q = rint(p * 2^40) is then an integer below 2^46 that a double holds exactly, and so is every partial sum of 64 of them
// (< 2^52): the butterfly adds doubles — the same exact integers as the limb form, 3 instructions to prepare a value
// instead of 7 — and the lane that ends up with a total cuts it into the 25-bit limbs the workgroup stage expects.
__device__ __forceinline__ double f64_from_halves(uint32_t lo, uint32_t hi)
{
    return __hiloint2double((int)hi, (int)lo);
}
template <int VM, int V, int S>
__device__ __forceinline__ void wave_transpose_sum_f64(double (&a)[VM], const int lane)
{
    if constexpr (S < 6) {
        constexpr int h = (V + 1) / 2;
#pragma unroll
        for (int i = 0; i < h; ++i) {
            const double x = a[i], y = (i + h < V) ? a[i + h] : 0.0;
            const uint32_t xl = (uint32_t)__double2loint(x), xh = (uint32_t)__double2hiint(x);
            const uint32_t yl = (uint32_t)__double2loint(y), yh = (uint32_t)__double2hiint(y);
            if constexpr (S == 0) {
                const auto rl = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
                const auto rh = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
                a[i] = f64_from_halves(rl[0], rh[0]) + f64_from_halves(rl[1], rh[1]);
            } else if constexpr (S == 1) {
                const auto rl = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
                const auto rh = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
                a[i] = f64_from_halves(rl[0], rh[0]) + f64_from_halves(rl[1], rh[1]);
            } else {
                const bool sel = ((lane >> (5 - S)) & 1) != 0;
                const double keep = sel ? y : x, send = sel ? x : y;
                constexpr int ctrl = S == 2 ? 0x128 : S == 3 ? 0x141 : S == 4 ? 0x4e : 0xb1;
                const uint32_t sl = (uint32_t)__builtin_amdgcn_update_dpp(0, __double2loint(send), ctrl, 0xf, 0xf, false);
                const uint32_t sh = (uint32_t)__builtin_amdgcn_update_dpp(0, __double2hiint(send), ctrl, 0xf, 0xf, false);
                a[i] = keep + f64_from_halves(sl, sh);
            }
        }
        wave_transpose_sum_f64<VM, h, S + 1>(a, lane);
    }
}
template <int N>
__device__ __forceinline__ void wave_reduce_f64(const float *vals, bool valid, int lane, int32_t *__restrict__ sums)
{
    int vs[7];
    vs[0] = N;
#pragma unroll
    for (int k = 0; k < 6; ++k) vs[k + 1] = (vs[k] + 1) / 2;
    double a[N];
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = hd_rint((double)(valid ? vals[i] : 0.0f) * 0x1p40);
    wave_transpose_sum_f64<N, N, 0>(a, lane);
    static_assert(N <= 64, "one total per lane");
    {
        int idx = 0;
        bool ok = 0 < vs[6];
#pragma unroll
        for (int S = 5; S >= 0; --S) {
            idx += ((lane >> (5 - S)) & 1) * vs[S + 1];
            ok = ok && idx < vs[S];
        }
        if (ok) {   // Q = d0 + d1 2^24 + d2 2^49 with limbs of the sign of Q; d3 = d4 = 0
            const double q = a[0];
            const double d2 = __builtin_trunc(q * 0x1p-49);
            const double r = hd_fma(-d2, 0x1p49, q);
            const double d1 = __builtin_trunc(r * 0x1p-24);
            const double d0 = hd_fma(-d1, 0x1p24, r);
            sums[5 * idx] = (int32_t)d0; sums[5 * idx + 1] = (int32_t)d1; sums[5 * idx + 2] = (int32_t)d2;
            sums[5 * idx + 3] = 0; sums[5 * idx + 4] = 0;
        }
    }
}

// Values up to 2^34: q = rint(p * 2^40) (< 2^74, exact in a double: p has 24 significant bits) is cut into
// q = lo + hi * 2^30 with |lo| < 2^30 and |hi| < 2^44, two doubles whose 64-lane sums (< 2^36, < 2^50) are exact again —
// 6 instructions to prepare a value instead of 10 for three limbs, and 58 registers through the butterfly instead of 87.
// The lane holding a total cuts it into limbs: lo -> limbs 0, 1; hi * 2^30 = h0 * 2^6 * 2^24 + hp * 2^49 -> limbs 1 (added), 2, 3.
template <int N>
__device__ __forceinline__ void wave_reduce_f64x2(const float *vals, bool valid, int lane, int32_t *__restrict__ sums)
{
    constexpr int V = 2 * N;
    static_assert(V <= 64, "one total per lane");
    int vs[7];
    vs[0] = V;
#pragma unroll
    for (int k = 0; k < 6; ++k) vs[k + 1] = (vs[k] + 1) / 2;
    double a[V];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double q = hd_rint((double)(valid ? vals[i] : 0.0f) * 0x1p40);
        const double hi = __builtin_trunc(q * 0x1p-30);
        a[2 * i] = hd_fma(-hi, 0x1p30, q);
        a[2 * i + 1] = hi;
    }
    wave_transpose_sum_f64<V, V, 0>(a, lane);
    int idx = 0;
    bool ok = 0 < vs[6];
#pragma unroll
    for (int S = 5; S >= 0; --S) {
        idx += ((lane >> (5 - S)) & 1) * vs[S + 1];
        ok = ok && idx < vs[S];
    }
    const int v = idx >> 1;
    const double t = a[0];
    if (ok && (idx & 1) == 0) {   // the lo total
        const double d1 = __builtin_trunc(t * 0x1p-24);
        sums[5 * v] = (int32_t)hd_fma(-d1, 0x1p24, t);
        sums[5 * v + 1] = (int32_t)d1;
    }
    // LDS operations of one wave execute in program order: the add below lands on the value just stored
    if (ok && (idx & 1) == 1) {   // the hi total
        const double hp = __builtin_trunc(t * 0x1p-19);
        const double h0 = hd_fma(-hp, 0x1p19, t);
        const double d3 = __builtin_trunc(hp * 0x1p-25);
        atomicAdd(&sums[5 * v + 1], (int32_t)(h0 * 64.0));
        sums[5 * v + 2] = (int32_t)hd_fma(-d3, 0x1p25, hp);
        sums[5 * v + 3] = (int32_t)d3;
        sums[5 * v + 4] = 0;
    }
}

// one wave: NL low limbs of N values per lane -> wave totals parked in sums[5 * value + limb]
template <int N, int NL>
__device__ __forceinline__ void wave_reduce_limbs(const float *vals, bool valid, int lane, int32_t *__restrict__ sums)
{
    constexpr int V = NL * N;
    int vs[7];
    vs[0] = V;
#pragma unroll
    for (int k = 0; k < 6; ++k) vs[k + 1] = (vs[k] + 1) / 2;
    int32_t a[V];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float p = valid ? vals[i] : 0.0f;
        if constexpr (NL == 5) {
            const hd_limbs25 l = hd_limbs25_from_f32(p);
            a[5 * i] = l.d0; a[5 * i + 1] = l.d1; a[5 * i + 2] = l.d2; a[5 * i + 3] = l.d3; a[5 * i + 4] = l.d4;
        } else if constexpr (NL == 3) {
            hd_limbs25_low3(p, &a[3 * i], &a[3 * i + 1], &a[3 * i + 2]);
        } else {
            hd_limbs25_low2(p, &a[2 * i], &a[2 * i + 1]);
        }
    }
    wave_transpose_sum<V, V, 0>(a, lane);
    // register k of lane l now holds the wave total of limb  k + sum_S bit_{5-S}(l) * h_S  (if in range)
#pragma unroll
    for (int k = 0; k < (V + 63) / 64; ++k) {
        int idx = k;
        bool ok = k < vs[6];
#pragma unroll
        for (int S = 5; S >= 0; --S) {
            idx += ((lane >> (5 - S)) & 1) * vs[S + 1];
            ok = ok && idx < vs[S];
        }
        if (ok) sums[5 * (idx / NL) + idx % NL] = a[k];
    }
    if constexpr (NL < 5)   // the limbs that were not computed are zero
        for (int i = lane; i < N * (5 - NL); i += 64) sums[5 * (i / (5 - NL)) + NL + i % (5 - NL)] = 0;
}

// ZEROED: the caller guarantees that lanes with !valid hold exact zeros in vals (no select per value needed)
template <int N, bool ZEROED = false>
__device__ __forceinline__ void block_reduce_exact(const float *vals, bool valid, long long *__restrict__ slots)
{
    constexpr int V = 5 * N;
    __shared__ int32_t s_sum[RB / 64][V];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (__ballot(valid) != 0ull) {
        // wave-uniform choice of how many limbs can be non-zero: |p| < 2^9 -> 2, |p| < 2^34 -> 3, else all 5
        // (non-finite values have the largest magnitudes bits and take the general path, which zeroes them)
        uint32_t m = 0u;
#pragma unroll
        for (int i = 0; i < N; ++i) { const uint32_t b = hd_f2u(vals[i]) & 0x7fffffffu; m = b > m ? b : m; }
        if (!ZEROED && !valid) m = 0u;
        const bool use = ZEROED ? true : valid;
        if (N <= 64 && __ballot(m >= ((127u + 6u) << 23)) == 0ull) wave_reduce_f64<N>(vals, use, lane, s_sum[wid]);
        else if (__ballot(m >= ((127u + 9u) << 23)) == 0ull) wave_reduce_limbs<N, 2>(vals, use, lane, s_sum[wid]);
        else if (2 * N <= 64 && __ballot(m >= ((127u + 34u) << 23)) == 0ull) wave_reduce_f64x2<N>(vals, use, lane, s_sum[wid]);
        else if (__ballot(m >= ((127u + 34u) << 23)) == 0ull) wave_reduce_limbs<N, 3>(vals, use, lane, s_sum[wid]);
        else wave_reduce_limbs<N, 5>(vals, use, lane, s_sum[wid]);
    } else {
        for (int i = lane; i < V; i += 64) s_sum[wid][i] = 0;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        long long s[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            s[j] = 0;
#pragma unroll
            for (int w = 0; w < RB / 64; ++w) s[j] += s_sum[w][threadIdx.x * 5 + j];
        }
        if ((s[0] | s[1] | s[2] | s[3] | s[4]) != 0) {
            const hd_limbs L = hd_limbs25_to_limbs(s[0], s[1], s[2], s[3], s[4]);
            long long *row = slots + (size_t)(blockIdx.x % ODO_SLOTS) * NLIMB(N) + threadIdx.x * 3;
            if (L.l0) atomicAdd((unsigned long long *)&row[0], (unsigned long long)L.l0);
            if (L.l1) atomicAdd((unsigned long long *)&row[1], (unsigned long long)L.l1);
            if (L.l2) atomicAdd((unsigned long long *)&row[2], (unsigned long long)L.l2);
        }
    }
}

// ------------------------------------------------------------------------------------------ O1: map building
// level 0 of every pyramid in one pass (copyMaps, copyCurvatureMap, copyicpWeightMap,
// verticesToDepth, imageBGRToIntensity; cudafuncs.cu:344-470,874-911): odo_level0_pixel in kernels.h.  In the frame
// path the pixel function runs at the tail of k_curvature (k_pre.hip), which has the live values in registers and ALU
// work to hide the traffic behind; this kernel serves the stage API and the frames where that is not possible.
__global__ void k_odo_level0(int P, OdoLevel L, OdoSources src, const DevPose *__restrict__ dp, int f2f, float curv_thr)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    odo_level0_pixel(i, P, L, src, dp->should_fill_in, f2f, curv_thr, src.vertex_filtered[i], src.normal[i], src.curv1[i],
                     src.curv2[i]);
}

// resizeMapKernel / resizeCMapKernel (cudafuncs.cu:526-674): validity plane = 0 (maps) or 3 (curvature)
__device__ __forceinline__ void resize_planar(const float *__restrict__ in, int irows, int icols, float *__restrict__ out,
                                              int orows, int ocols, int x, int y, int valid_plane, bool normalize)
{
    const float qn = hd_nanf();
    const int xs = x * 2, ys = y * 2;
    float a = PLN(in, valid_plane, irows, icols, ys, xs), b = PLN(in, valid_plane, irows, icols, ys, xs + 1),
          c = PLN(in, valid_plane, irows, icols, ys + 1, xs), d = PLN(in, valid_plane, irows, icols, ys + 1, xs + 1);
    if (hd_isnanf(a) || hd_isnanf(b) || hd_isnanf(c) || hd_isnanf(d)) {
        for (int k = 0; k < 4; ++k) PLN(out, k, orows, ocols, y, x) = qn;
        return;
    }
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = (((PLN(in, k, irows, icols, ys, xs) + PLN(in, k, irows, icols, ys, xs + 1)) +
                 PLN(in, k, irows, icols, ys + 1, xs)) + PLN(in, k, irows, icols, ys + 1, xs + 1)) / 4.0f;
    if (normalize) {
        float inv = 1.0f / hd_sqrtf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
        r[0] *= inv; r[1] *= inv; r[2] *= inv;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) PLN(out, k, orows, ocols, y, x) = r[k];
}

// pyrDownKernelGaussF / pyrDownKernelIntensityGauss (cudafuncs.cu:493-524,818-848) incl. border quirk:
// the 5x5 binomial window is anchored at its (clamped) far corner, so at the right/bottom border the taps
// shift instead of being cut.  Rows and columns are visited in increasing source order (tap index 4 -> 0),
// the weights {1,4,6,4,1} x {1,4,6,4,1} are compile-time constants.
template <typename T, typename ValidF>
__device__ __forceinline__ void pyrdown_taps(const T *__restrict__ src, int srows, int scols, int x, int y,
                                             float &sum, int &count, ValidF valid)
{
    const int tx = 2 * x + 3 < scols - 1 ? 2 * x + 3 : scols - 1;
    const int ty = 2 * y + 3 < srows - 1 ? 2 * y + 3 : srows - 1;
    const int lx = 2 * x - 2 > 0 ? 2 * x - 2 : 0, ly = 2 * y - 2 > 0 ? 2 * y - 2 : 0;
    sum = 0.0f; count = 0;
    // All 25 taps are requested before the first is used: a tap the border cuts off reads the clamped texel and is not used.  (With
    // the loads under `if (cy < ly) continue` a lane waited for every tap where its branch ended — 25 dependent round trips, which
    // made the four pyramid tasks the slowest of the 13 this kernel runs side by side.)
    T tap[25];
#pragma unroll
    for (int kr = 4; kr >= 0; --kr) {
        const int cy = ty - 1 - kr;
#pragma unroll
        for (int kc = 4; kc >= 0; --kc) {
            const int cx = tx - 1 - kc;
            tap[kr * 5 + kc] = src[(cy < ly ? ly : cy) * scols + (cx < lx ? lx : cx)];
        }
    }
    {   // pinned: hipcc otherwise sinks every load into the block that uses it, and the chain is back
        uint32_t raw[25];
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            if constexpr (sizeof(T) == 4) raw[t] = __builtin_bit_cast(uint32_t, tap[t]); else raw[t] = (uint32_t)tap[t];
        }
        asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]),
                          "+v"(raw[8]), "+v"(raw[9]), "+v"(raw[10]), "+v"(raw[11]), "+v"(raw[12]));
        asm volatile("" : "+v"(raw[13]), "+v"(raw[14]), "+v"(raw[15]), "+v"(raw[16]), "+v"(raw[17]), "+v"(raw[18]), "+v"(raw[19]),
                          "+v"(raw[20]), "+v"(raw[21]), "+v"(raw[22]), "+v"(raw[23]), "+v"(raw[24]));
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            if constexpr (sizeof(T) == 4) tap[t] = __builtin_bit_cast(T, raw[t]); else tap[t] = (T)raw[t];
        }
    }
#pragma unroll
    for (int kr = 4; kr >= 0; --kr) {
        const int cy = ty - 1 - kr;
#pragma unroll
        for (int kc = 4; kc >= 0; --kc) {
            const int cx = tx - 1 - kc;
            constexpr int w[5] = {1, 4, 6, 4, 1};
            const T s = tap[kr * 5 + kc];
            if (cy >= ly && cx >= lx && valid(s)) { sum += (float)s * (float)(w[kr] * w[kc]); count += w[kr] * w[kc]; }
        }
    }
}
__device__ __forceinline__ float pyrdown_f(const float *__restrict__ src, int srows, int scols, int x, int y)
{
    float sum; int count;
    pyrdown_taps(src, srows, scols, x, y, sum, count, [](float s) { return !hd_isnanf(s); });
    return sum / (float)count;
}
__device__ __forceinline__ uint8_t pyrdown_u8(const uint8_t *__restrict__ src, int srows, int scols, int x, int y)
{
    float sum; int count;
    pyrdown_taps(src, srows, scols, x, y, sum, count, [](uint8_t s) { return s > 0; });
    return count > 0 ? (uint8_t)(int)(sum / (float)count) : (uint8_t)0;
}

// one pyramid level; blockIdx.y selects the map, so the 13 independent resamplings run side by side
#define ODO_DOWN_TASKS 13
__global__ void k_odo_downsample(OdoLevel I, OdoLevel O, DevPose *dp, float frame_wmul)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (dp && i == 0 && blockIdx.y == 0) dp->frame_wmul = frame_wmul;   // rides along: read by the captured k_gn_solve / k_odo_end
    if (i >= O.rows * O.cols) return;
    int y = i / O.cols, x = i - y * O.cols;
    switch (blockIdx.y) {
    case 0: resize_planar(I.vmap_g, I.rows, I.cols, O.vmap_g, O.rows, O.cols, x, y, 0, false); break;
    case 1: resize_planar(I.nmap_g, I.rows, I.cols, O.nmap_g, O.rows, O.cols, x, y, 0, true); break;
    case 2: resize_planar(I.ck1_g, I.rows, I.cols, O.ck1_g, O.rows, O.cols, x, y, 3, false); break;
    case 3: resize_planar(I.ck2_g, I.rows, I.cols, O.ck2_g, O.rows, O.cols, x, y, 3, false); break;
    case 4: resize_planar(I.vmap_c, I.rows, I.cols, O.vmap_c, O.rows, O.cols, x, y, 0, false); break;
    case 5: resize_planar(I.nmap_c, I.rows, I.cols, O.nmap_c, O.rows, O.cols, x, y, 0, true); break;
    case 6: resize_planar(I.ck1_c, I.rows, I.cols, O.ck1_c, O.rows, O.cols, x, y, 3, false); break;
    case 7: resize_planar(I.ck2_c, I.rows, I.cols, O.ck2_c, O.rows, O.cols, x, y, 3, false); break;
    case 8: {   // resizeicpWeightMapKernel cudafuncs.cu:694-726
        const float qn = hd_nanf();
        float a = I.icpw[(2 * y) * I.cols + 2 * x], b = I.icpw[(2 * y) * I.cols + 2 * x + 1],
              c = I.icpw[(2 * y + 1) * I.cols + 2 * x], d = I.icpw[(2 * y + 1) * I.cols + 2 * x + 1];
        O.icpw[i] = (hd_isnanf(a) || hd_isnanf(b) || hd_isnanf(c) || hd_isnanf(d)) ? qn : (((a + b) + c) + d) / 4.0f;
        break;
    }
    case 9: O.last_depth[i] = pyrdown_f(I.last_depth, I.rows, I.cols, x, y); break;
    case 10: O.next_depth[i] = pyrdown_f(I.next_depth, I.rows, I.cols, x, y); break;
    case 11: O.last_image[i] = pyrdown_u8(I.last_image, I.rows, I.cols, x, y); break;
    default: O.next_image[i] = pyrdown_u8(I.next_image, I.rows, I.cols, x, y); break;
    }
}

// tranformMapsKernel / tranformCurvMapsKernel (cudafuncs.cu:213-322), in place, one level
__device__ __forceinline__ void transform_planar(float *m, int rows, int cols, int i, const Rigid &T, bool add_t)
{
    const size_t PP = (size_t)rows * cols;
    float vx = m[i];
    if (hd_isnanf(vx)) return;
    f3 o = rot_mul(T, mk3(vx, m[PP + i], m[2 * PP + i]));
    if (add_t) o = mk3(o.x + T.t[0], o.y + T.t[1], o.z + T.t[2]);
    m[i] = o.x; m[PP + i] = o.y; m[2 * PP + i] = o.z;
}
__device__ __forceinline__ void transform_pixel(const OdoLevel &L, int i, const Rigid &T)
{
    transform_planar(L.vmap_g, L.rows, L.cols, i, T, true);
    transform_planar(L.nmap_g, L.rows, L.cols, i, T, false);
    transform_planar(L.ck1_g, L.rows, L.cols, i, T, false);
    transform_planar(L.ck2_g, L.rows, L.cols, i, T, false);
}

// applyKernel (Sobel, cudafuncs.cu:927-954, running kernelIndex quirk) + projectPointsKernel (:995-1013)
__device__ __forceinline__ void sobel_cloud_pixel(const OdoLevel &L, int i, float fx, float fy, float cx, float cy)
{
    const int rows = L.rows, cols = L.cols;
    int y = i / cols, x = i - y * cols;
    const float gx[9] = {1, 0, -1, 2, 0, -2, 1, 0, -1};
    const float gy[9] = {1, 2, 1, 0, 0, 0, -1, -2, -1};
    float dxv = 0.0f, dyv = 0.0f;
    int ki = 8;
    for (int j = (y - 1 > 0 ? y - 1 : 0); j <= (y + 1 < rows - 1 ? y + 1 : rows - 1); ++j)
        for (int ii = (x - 1 > 0 ? x - 1 : 0); ii <= (x + 1 < cols - 1 ? x + 1 : cols - 1); ++ii) {
            float s = (float)L.next_image[j * cols + ii];
            dxv += s * gx[ki];
            dyv += s * gy[ki];
            --ki;
        }
    L.dIdx[i] = (int16_t)dxv;
    L.dIdy[i] = (int16_t)dyv;
    float invFx = 1.0f / fx, invFy = 1.0f / fy;
    float z = L.last_depth[i];
    const float px_ = ((float)x - cx) * z * invFx, py_ = ((float)y - cy) * z * invFy;
    L.cloud[i * 3] = px_;
    L.cloud[i * 3 + 1] = py_;
    L.cloud[i * 3 + 2] = z;
    if (L.cloud4) {   // in-frame consumers read one texel / one word instead of three floats / two shorts
        L.cloud4[i] = make_float4(px_, py_, z, 0.0f);
        L.dIxy[i] = (int32_t)(((uint32_t)(uint16_t)(int16_t)dxv) | ((uint32_t)(uint16_t)(int16_t)dyv << 16));
    }
}

__global__ void k_first_rgb(int P, const uint8_t *__restrict__ rgb, uint8_t *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) out[i] = intensity_u8(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]);
}
__global__ void k_pyrdown_u8(const uint8_t *__restrict__ src, int srows, int scols, uint8_t *__restrict__ dst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int dcols = scols / 2, drows = srows / 2;
    if (i >= dcols * drows) return;
    int y = i / dcols, x = i - y * dcols;
    dst[i] = pyrdown_u8(src, srows, scols, x, y);
}

// ------------------------------------------------------------------------------------------ small dense algebra
// diagonal-pivoted LDL^T, same arithmetic as the oracle's ldlt (oracle/orc_odo.c).  Everything is indexed
// by compile-time constants (pivot row/column swaps are select chains), so A, perm and y live in registers —
// the one-lane solve kernels are pure latency and runtime-indexed arrays would sit in scratch memory.
// Symmetric swap k <-> piv of the pivoted factorisation: rows, then columns, then the permutation record.  The solve runs
// in ONE lane, so the pivot index is wave-uniform: it goes to a scalar register and the swap is a scalar branch around
// 4 N register moves.  Selecting over every candidate row instead (no branch; N - k - 1 candidates x 4 N selects of 64
// bits) was 700 of the 6x6 solve's 2 200 instructions.  The empty asm keeps the compiler from turning the branch back
// into those selects.  REQUIRES wave-uniform data: every caller runs the solve in a single active lane.
__host__ __device__ __forceinline__ int ldlt_uniform(int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}
template <typename T, int N, int IDX>
__host__ __device__ __forceinline__ void ldlt_swap(T (&A)[N * N], uint32_t &perm, int k, int piv)
{
    if constexpr (IDX < N * N) {
        constexpr int K = IDX / N, I = IDX % N;
        if constexpr (I > K) {
            if (k == K && piv == I) {
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" ::: "memory");
#endif
#pragma unroll
                for (int j = 0; j < N; ++j) { const T t = A[K * N + j]; A[K * N + j] = A[I * N + j]; A[I * N + j] = t; }
#pragma unroll
                for (int j = 0; j < N; ++j) { const T t = A[j * N + K]; A[j * N + K] = A[j * N + I]; A[j * N + I] = t; }
                // the permutation record is one word, 4 bits per entry: an int[N] sat in scratch memory (the asm's memory clobber
                // pins arrays there) — 32 B / lane of round trips on the solve's serial chain
                const uint32_t x = ((perm >> (4 * K)) ^ (perm >> (4 * I))) & 15u;
                perm ^= (x << (4 * K)) | (x << (4 * I));
                return;
            }
        }
        ldlt_swap<T, N, IDX + 1>(A, perm, k, piv);
    }
}

template <typename T, int N>
__host__ __device__ __forceinline__ void ldlt_solve(T (&A)[N * N], const T (&b)[N], T (&x)[N])
{
    static_assert(N <= 8, "4 bits per entry");
    uint32_t perm = 0x76543210u;   // perm[i] = (perm >> 4 i) & 15
    T y[N], rdiag[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        int piv = k;
        T best = A[k * N + k] < 0 ? -A[k * N + k] : A[k * N + k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            T v = A[i * N + i] < 0 ? -A[i * N + i] : A[i * N + i];
            if (v > best) { best = v; piv = i; }
        }
        ldlt_swap<T, N, 0>(A, perm, k, ldlt_uniform(piv));   // rows and columns k <-> piv, perm[k] <-> perm[piv]
        const T d = A[k * N + k];
        rdiag[k] = 0;
        if (d != 0) {
            // ONE division per pivot (oracle/orc_odo.c DEF_LDLT does the same): 6 instead of 21 fp64 divisions on the lane's chain
            const T rd = 1 / d;
            rdiag[k] = rd;
#pragma unroll
            for (int i = k + 1; i < N; ++i) A[i * N + k] = A[i * N + k] * rd;
#pragma unroll
            for (int i = k + 1; i < N; ++i)
#pragma unroll
                for (int j = k + 1; j <= i; ++j) {
                    A[i * N + j] = A[i * N + j] - A[i * N + k] * d * A[j * N + k];
                    A[j * N + i] = A[i * N + j];
                }
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {               // y = P b
        T v = b[0];
#pragma unroll
        for (int j = 1; j < N; ++j) v = ((int)((perm >> (4 * i)) & 15u) == j) ? b[j] : v;
        y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) y[i] = y[i] - A[i * N + j] * y[j];
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = (A[i * N + i] == 0) ? 0 : y[i] * rdiag[i];
#pragma unroll
    for (int i = N - 1; i >= 0; --i)
#pragma unroll
        for (int j = i + 1; j < N; ++j) y[i] = y[i] - A[j * N + i] * y[j];
#pragma unroll
    for (int i = 0; i < N; ++i) {               // x = P^T y
        T v = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) v = ((int)((perm >> (4 * j)) & 15u) == i) ? y[j] : v;
        x[i] = v;
    }
}

__host__ __device__ inline void inv3d(const double *m, double *o)
{
    double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    double c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    double det = (a * c00 + b * c01) + c * c02;
    double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (c * h - b * i) * id; o[2] = (b * f - c * e) * id;
    o[3] = c01 * id; o[4] = (a * i - c * g) * id; o[5] = (c * d - a * f) * id;
    o[6] = c02 * id; o[7] = (b * g - a * h) * id; o[8] = (a * e - b * d) * id;
}
template <typename T>
__host__ __device__ inline void mul3(const T *a, const T *b, T *o)
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
        o[r * 3 + c] = (a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c]) + a[r * 3 + 2] * b[6 + c];
}
__host__ __device__ inline void rodrigues(const double *src, double *R)
{
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    double rx = src[0], ry = src[1], rz = src[2];
    double theta = hd_sqrt((rx * rx + ry * ry) + rz * rz);
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
__host__ __device__ inline void inv3f_cof(const float *m, float *o)
{
    float a = m[0], b = m[1], cc = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    float det = (a * c00 + b * c01) + cc * c02, id = 1.0f / det;
    o[0] = c00 * id; o[1] = (cc * h - b * i) * id; o[2] = (b * f - cc * e) * id;
    o[3] = c01 * id; o[4] = (a * i - cc * g) * id; o[5] = (cc * d - a * f) * id;
    o[6] = c02 * id; o[7] = (b * g - a * h) * id; o[8] = (a * e - b * d) * id;
}

// The workgroup that takes the last ticket of a launch runs the launch's epilogue (slot fold + solve).
// Everything the epilogue consumes from other workgroups arrives through the agent-scope slot atomics, which
// block_reduce_exact issues from wave 0 only (threads < N <= 64).  So wave 0 alone executes the agent-scope
// release fence (a workgroup-scope fence is a no-op for global memory on a single CU and was observed to let
// the ticket overtake the atomics; a fence in every wave costs an L2 write-back each and was 4x slower), the
// barrier collects the workgroup, and thread 0 — a lane of that same wave — takes the ticket.
// The elected workgroup acquires once before it reads.  No spinning: no co-residency requirement.
__device__ __forceinline__ bool elect_last_workgroup(unsigned int *ticket)
{
    __shared__ int s_last;
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1);
        if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return s_last != 0;
}

// s_tot[t] = totals[t] = sum over the slot rows of `part` (and, if given, of `part2` as columns width..2 width-1),
// rows re-zeroed for the next launch.  NGRP * 256 threads: the rows are split into NGRP groups; every thread
// issues all its loads before the first add (one memory latency).
template <bool IN_LAUNCH /* rows written by other workgroups of this very launch */, int NGRP>
__device__ __forceinline__ void fold_slots(long long *__restrict__ part, long long *__restrict__ part2, int width,
                                           long long *__restrict__ s_tot, long long *__restrict__ totals)
{
    __shared__ long long s_grp[NGRP][256];
    const int col = threadIdx.x & 255, grp = threadIdx.x >> 8;
    const int ncols = part2 ? 2 * width : width;
    if (col < ncols && grp < NGRP) {
        long long *base = col < width ? part + col : part2 + (col - width);
        long long v[ODO_SLOTS / NGRP];
#pragma unroll
        for (int b = 0; b < ODO_SLOTS / NGRP; ++b) {
            long long *q = base + (size_t)(grp + b * NGRP) * width;
            v[b] = IN_LAUNCH ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
        }
        long long t = 0;
#pragma unroll
        for (int b = 0; b < ODO_SLOTS / NGRP; ++b) { t += v[b]; base[(size_t)(grp + b * NGRP) * width] = 0; }
        s_grp[grp][col] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < ncols) {
        long long t = 0;
#pragma unroll
        for (int g = 0; g < NGRP; ++g) t += s_grp[g][threadIdx.x];
        s_tot[threadIdx.x] = t; totals[threadIdx.x] = t;
    }
}

__device__ __forceinline__ double limbs_to_double(const long long *t, int i)
{
    return hd_acc_to_double(hd_limbs_combine(t[i * 3], t[i * 3 + 1], t[i * 3 + 2]));
}

// ------------------------------------------------------------------------------------------ step operands
template <class S>
__device__ inline void so3_set_operands(S *st, float fx, float fy, float cx, float cy)
{
    double K[9] = {0}, Kinv[9], KR[9], Hm[9];
    K[0] = fx / 4; K[4] = fy / 4; K[2] = cx / 4; K[5] = cy / 4; K[8] = 1;
    inv3d(K, Kinv);
    mul3<double>(K, st->resultR, KR);
    mul3<double>(KR, Kinv, Hm);
    for (int k = 0; k < 9; ++k) { st->basis[k] = (float)Hm[k]; st->kinv[k] = (float)Kinv[k]; st->krlr[k] = (float)KR[k]; }
}

template <class S>
__device__ inline void gn_set_operands(S *st, float fxl, float fyl, float cxl, float cyl)
{
    double K[9] = {0}, Kinv[9];
    K[0] = fxl; K[4] = fyl; K[2] = cxl; K[5] = cyl; K[8] = 1;
    inv3d(K, Kinv);
    const double *Rt = st->Rt;
    double L[9], Li[9], ti[3];
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) L[r * 3 + k] = Rt[r * 4 + k];
    inv3d(L, Li);
    for (int r = 0; r < 3; ++r) ti[r] = -((Li[r * 3] * Rt[3] + Li[r * 3 + 1] * Rt[7]) + Li[r * 3 + 2] * Rt[11]);
    double KR[9], KRK[9];
    mul3<double>(K, Li, KR); mul3<double>(KR, Kinv, KRK);
    for (int k = 0; k < 9; ++k) st->krk[k] = (float)KRK[k];
    for (int r = 0; r < 3; ++r)
        st->kt[r] = (float)((K[r * 3] * ti[0] + K[r * 3 + 1] * ti[1]) + K[r * 3 + 2] * ti[2]);
}

// first Gauss-Newton operands: resultRt = [R_so3 | 0] (RGBDOdometry.cpp:700-712)
__device__ inline void gn_begin_state(OdoState *st, const OdoConfig &cfg, int level)
{
    for (int k = 0; k < 16; ++k) st->Rt[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (cfg.so3) for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) st->Rt[r * 4 + k] = st->resultR[r * 3 + k];
    const int div = 1 << level;
    gn_set_operands(st, cfg.fx / div, cfg.fy / div, cfg.cx / div, cfg.cy / div);
    st->lastRGBError = 3.402823466e+38f;
}

// begin of registration: latch previous pose, reset SO3 / GN state.  gn_level >= 0: no SO3 stage follows, so
// the Gauss-Newton operands are prepared here as well.
__device__ inline void odo_begin_state(OdoState *st, const DevPose *__restrict__ dp, const OdoConfig &cfg, int gn_level)
{
    for (int k = 0; k < 9; ++k) { st->Rprev[k] = dp->pose.r[k]; st->Rcurr[k] = dp->pose.r[k]; }
    for (int k = 0; k < 3; ++k) { st->tprev[k] = dp->pose.t[k]; st->tcurr[k] = dp->pose.t[k]; }
    inv3f_cof(st->Rprev, st->Rprev_inv);
    for (int k = 0; k < 9; ++k) { st->resultR[k] = st->lastResultR[k] = (k % 4 == 0) ? 1.0 : 0.0; st->R_lr[k] = (k % 4 == 0) ? 1.0f : 0.0f; }
    st->so3_lastError = 3.402823466e+38f / 2.0f;
    st->so3_lastCount = 3.402823466e+38f / 2.0f;
    st->so3_done = cfg.so3 ? 0 : 1;
    st->gn_break = 0;
    st->res_icp[0] = st->res_icp[1] = 0.0f;
    st->ticket = 0u;
    for (int k = 0; k < 12; ++k) st->so3_word[k] = 0ull;
    st->so3_ticket0 = 0u; st->so3_exit = 0u;
    st->bar_timeouts_total += st->bar_timeout; st->bar_timeout = 0;
    if (cfg.so3) so3_set_operands(st, cfg.fx, cfg.fy, cfg.cx, cfg.cy);
    if (gn_level >= 0) gn_begin_state(st, cfg, gn_level);
}

// pose-independent part of RGBResidual::getProducts' pixel test (reduce.cu:981-1010): away from the right / bottom
// border, 4x4 neighbourhood of the live intensity image all non-zero, gradient magnitude above the level's
// threshold, live depth present
__device__ __forceinline__ bool rgb_residual_static_test(const OdoLevel &L, float minScale, int k)
{
    const int rows = L.rows, cols = L.cols;
    const int i = k / cols, j0 = k - i * cols;
    if (!(j0 < cols - 5 && i < rows - 1)) return false;
    bool valid = true;
    for (int u = (i - 2 > 0 ? i - 2 : 0); u < (i + 2 < rows ? i + 2 : rows); ++u)
        for (int v = (j0 - 2 > 0 ? j0 - 2 : 0); v < (j0 + 2 < cols ? j0 + 2 : cols); ++v)
            valid = valid && (L.next_image[u * cols + v] > 0);
    if (!valid) return false;
    int valx = L.dIdx[k], valy = L.dIdy[k];
    float mTwo = (float)((valx * valx) + (valy * valy));
    if (!(mTwo >= minScale)) return false;
    return !hd_isnanf(L.next_depth[k]);
}

// packed ICP operands of one pixel, after the model maps were moved to the global frame (k_odo_prepare)
__device__ __forceinline__ void pack_icp_texels(const OdoLevel &L, int i)
{
    const size_t PP = (size_t)L.rows * L.cols;
    {
        const float vx = L.vmap_c[i], nx = L.nmap_c[i], k1 = L.ck1_c[3 * PP + i], k2 = L.ck2_c[3 * PP + i];
        const bool ok = !(hd_isnanf(vx) || hd_isnanf(nx) || hd_isnanf(k1) || hd_isnanf(k2));
        L.icp_cur[2 * i] = make_float4(vx, L.vmap_c[PP + i], L.vmap_c[2 * PP + i], ok ? 1.0f : 0.0f);
        L.icp_cur[2 * i + 1] = make_float4(nx, L.nmap_c[PP + i], L.nmap_c[2 * PP + i], 0.0f);
    }
    {
        const float vx = L.vmap_g[i], nx = L.nmap_g[i], k1 = L.ck1_g[3 * PP + i], k2 = L.ck2_g[3 * PP + i];
        const bool ok = !(hd_isnanf(vx) || hd_isnanf(nx) || hd_isnanf(k1) || hd_isnanf(k2));
        L.icp_model[2 * i] = make_float4(vx, L.vmap_g[PP + i], L.vmap_g[2 * PP + i], L.icpw[i]);
        L.icp_model[2 * i + 1] = make_float4(nx, L.nmap_g[PP + i], L.nmap_g[2 * PP + i], ok ? 1.0f : 0.0f);
    }
}

// last pass over the pyramids, all levels in one launch (blockIdx.y = level): model maps into the global frame
// (in place), Sobel + back-projected cloud of the live frame; workgroup (0,0) also resets the registration state
struct OdoLevels { OdoLevel lv[HRBF_NUM_PYRS]; };
// one pixel of one level: nothing here depends on another pixel's result
__device__ __forceinline__ void prepare_pixel(const OdoLevel &L, int level, int i, const DevPose *__restrict__ dp,
                                              const OdoConfig &cfg, int do_rgb, int level0_packed)
{
    if (!(level == 0 && level0_packed)) {
        transform_pixel(L, i, dp->pose);
        pack_icp_texels(L, i);
    }
    if (do_rgb) {
        const int div = 1 << level;
        sobel_cloud_pixel(L, i, cfg.fx / div, cfg.fy / div, cfg.cx / div, cfg.cy / div);
        // dIdx / dIdy of this pixel were just written by this very thread; the mask is read by other kernels only
        const float minGrad = level == 0 ? 5.0f : (level == 1 ? 3.0f : 1.0f);   // RGBDOdometry.cpp:78-80
        const float minScale = (float)(((double)minGrad * (double)minGrad) / (0.125 * 0.125));
        L.rgb_mask[i] = rgb_residual_static_test(L, minScale, i) ? 1 : 0;
    }
}
// state_only: the pixel work of all levels rides along in the SO3 kernel (k_so3_persistent, filler workgroups); only the
// state reset and the slot sets remain here
__global__ void k_odo_prepare(OdoLevels all, OdoState *st, const DevPose *__restrict__ dp, OdoConfig cfg, int do_rgb,
                              int gn_level, long long *__restrict__ so3_sets, int level0_packed /* by odo_level0_pixel */,
                              int state_only)
{
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) odo_begin_state(st, dp, cfg, gn_level);
    if (blockIdx.y == 0) {   // the per-iteration SO3 slot sets start from zero every frame
        const int t = blockIdx.x * blockDim.x + threadIdx.x;
        if (t < SO3_ITERS * ODO_SLOTS * 33) so3_sets[t] = 0;
    }
    if (state_only) return;
    const int level = blockIdx.y;
    const OdoLevel &L = all.lv[level];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.rows * L.cols) return;
    prepare_pixel(L, level, i, dp, cfg, do_rgb, level0_packed);
}

// one SO3 step on folded totals (RGBDOdometry.cpp:551-640); S is OdoState or the LDS copy of the persistent kernel
template <class S>
__device__ inline void so3_step(S *st, const long long *s_tot, const OdoConfig &cfg)
{
    if (st->so3_done) return;
    double s[11];
    for (int i = 0; i < 11; ++i) s[i] = limbs_to_double(s_tot, i);
    float jtj[9], jtr[3];
    int shift = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 4; ++j) {
            float value = (float)s[shift++];
            if (j == 3) jtr[i] = value; else jtj[j * 3 + i] = jtj[i * 3 + j] = value;
        }
    float res0 = (float)s[9], res1 = (float)s[10];
    float so3err = hd_sqrtf(res0) / res1, so3cnt = res1;
    if (so3err < st->so3_lastError && st->so3_lastCount == so3cnt) st->so3_done = 1;
    else if (so3err > st->so3_lastError + 0.001f) {
        for (int k = 0; k < 9; ++k) st->resultR[k] = st->lastResultR[k];
        st->so3_done = 1;
    } else {
        st->so3_lastError = so3err; st->so3_lastCount = so3cnt;
        for (int k = 0; k < 9; ++k) st->lastResultR[k] = st->resultR[k];
        float delta[3];
        ldlt_solve<float, 3>(jtj, jtr, delta);
        double dd[3] = {delta[0], delta[1], delta[2]}, rotU[9];
        rodrigues(dd, rotU);
        float rotUf[9], tmp[9];
        for (int k = 0; k < 9; ++k) rotUf[k] = (float)rotU[k];
        mul3<float>(rotUf, st->R_lr, tmp);
        for (int k = 0; k < 9; ++k) { st->R_lr[k] = tmp[k]; st->resultR[k] = tmp[k]; }
        so3_set_operands(st, cfg.fx, cfg.fy, cfg.cx, cfg.cy);
    }
}

// SO3 step (RGBDOdometry.cpp:551-640): fold the slot rows, solve the 3x3 system, update the homography
// operands.  Runs in the elected last workgroup of k_so3_reduce (or alone, from k_so3_solve).
// gn_level >= 0 marks the last SO3 iteration: the Gauss-Newton operands are prepared right away.
__device__ __forceinline__ void so3_solve_block(OdoState *st, long long *__restrict__ part, long long *__restrict__ totals,
                                                int do_reduce, const OdoConfig &cfg, int gn_level)
{
    __shared__ long long s_tot[33];
    if (do_reduce == 2) fold_slots<true, 1>(part, nullptr, 33, s_tot, totals);
    else if (do_reduce) fold_slots<false, 1>(part, nullptr, 33, s_tot, totals);
    else if (threadIdx.x < 33) s_tot[threadIdx.x] = totals[threadIdx.x];
    __syncthreads();
    if (threadIdx.x != 0) return;
    so3_step(st, s_tot, cfg);
    if (gn_level >= 0) gn_begin_state(st, cfg, gn_level);
}

// stand-alone SO3 solve on totals that were summed elsewhere (row-sharded multi-GPU: all-reduce of the limb sums)
__global__ __launch_bounds__(RB) void k_so3_solve(OdoState *st, long long *__restrict__ part,
                                                  long long *__restrict__ totals, int do_reduce, OdoConfig cfg,
                                                  int gn_level)
{
    so3_solve_block(st, part, totals, do_reduce, cfg, gn_level);
}

// ------------------------------------------------------------------------------------------ O2: SO3 pre-alignment
// one pixel of the SO3 photometric system (so3Step, reduce.cu:1102-1215): 3x4 upper products + residual + count
template <class S>
__device__ __forceinline__ bool so3_pixel(const OdoLevel &L, const S *st, int i, float (&row4)[11])
{
    const int rows = L.rows, cols = L.cols;
    bool valid = false;
    if (i < rows * cols) {
        const int y = i / cols, x = i - y * cols;
        const uint8_t *lastImage = L.last_next_image, *nextImage = L.next_image;
        f3 un = mk3((float)x, (float)y, 1.0f);
        f3 wp = m33_mul(st->basis, un);
        float fu = wp.x / wp.z, fv = wp.y / wp.z;
        if (fu > -1.0e9f && fu < 1.0e9f && fv > -1.0e9f && fv < 1.0e9f) {
            int wx = (int)hd_rintf(fu), wy = (int)hd_rintf(fv);
            if (wx >= 1 && wx < cols - 1 && wy >= 1 && wy < rows - 1 && x >= 1 && x < cols - 1 && y >= 1 &&
                y < rows - 1) {
                valid = true;
                float actu = (float)nextImage[wy * cols + wx];
                float back = (float)nextImage[wy * cols + wx - 1], fore = (float)nextImage[wy * cols + wx + 1];
                float gnx = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
                back = (float)nextImage[(wy - 1) * cols + wx]; fore = (float)nextImage[(wy + 1) * cols + wx];
                float gny = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
                actu = (float)lastImage[y * cols + x];
                back = (float)lastImage[y * cols + x - 1]; fore = (float)lastImage[y * cols + x + 1];
                float glx = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
                back = (float)lastImage[(y - 1) * cols + x]; fore = (float)lastImage[(y + 1) * cols + x];
                float gly = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
                float gx = (gnx + glx) / 2.0f, gy = (gny + gly) / 2.0f;
                f3 point = m33_mul(st->kinv, un);
                float z2 = point.z * point.z;
                const float *k = st->krlr;
                float a = k[0], b = k[1], cc = k[2], d = k[3], e = k[4], f = k[5], g = k[6], h = k[7], i_ = k[8];
                float fxp = (float)x, fyp = (float)y;
                f3 lp = mk3((((point.z * (d * gy + a * gx)) - (gy * g * fyp)) - (gx * g * fxp)) / z2,
                            (((point.z * (e * gy + b * gx)) - (gy * h * fyp)) - (gx * h * fxp)) / z2,
                            (((point.z * (f * gy + cc * gx)) - (gy * i_ * fyp)) - (gx * i_ * fxp)) / z2);
                f3 jr = cross3(lp, point);
                float r[4] = {jr.x, jr.y, jr.z, -((float)nextImage[wy * cols + wx] - (float)lastImage[y * cols + x])};
                int q = 0;
#pragma unroll
                for (int ii = 0; ii < 3; ++ii)
#pragma unroll
                    for (int jj = ii; jj < 4; ++jj) row4[q++] = r[ii] * r[jj];
                row4[9] = r[3] * r[3];
                row4[10] = 1.0f;
            }
        }
    }
    return valid;
}

__global__ __launch_bounds__(RB) void k_so3_reduce(OdoLevel L, OdoState *st, long long *__restrict__ part,
                                                   long long *__restrict__ totals, OdoConfig cfg, int fused_solve,
                                                   int gn_level, int p0, int p1 /* pixel range of this rank */)
{
    const int i = p0 + blockIdx.x * RB + threadIdx.x;
    float row4[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) row4[k] = 0.0f;
    const bool valid = !st->so3_done && i < p1 && so3_pixel(L, st, i, row4);
    block_reduce_exact<11>(row4, valid, part);
    if (fused_solve && elect_last_workgroup(&st->ticket)) so3_solve_block(st, part, totals, 2, cfg, gn_level);
}

// LDS copy of the SO3 part of the state: every workgroup of the persistent kernel advances its own copy with the
// same deterministic arithmetic, so nothing but the slot sums has to be exchanged
struct So3Local {
    double resultR[9], lastResultR[9];
    float R_lr[9];
    float so3_lastError, so3_lastCount;
    int so3_done;
    float basis[9], kinv[9], krlr[9];
};

// All SO3 iterations in one launch.  The level-2 image is cut into `nchunk` chunks of RB pixels; per iteration the
// running workgroups draw chunks from a ticket, add their exact products into slot set `it` and count the chunk as
// done; a workgroup then waits until all chunks of the iteration are done (claiming further chunks while any is
// unclaimed), folds the set and takes the step on its LDS copy of the state.  Every chunk a waiter depends on was
// claimed by a workgroup that is running and finishes it without waiting for anybody, so the kernel makes progress
// with ANY number of resident workgroups — other contexts, RCCL kernels or a grid larger than the device cannot
// deadlock it.  A workgroup that starts late replays the iterations from the retained slot sets (one set per
// iteration) with the same deterministic arithmetic.  The ticket of iteration it+1 is drawn together with the
// arrival of iteration it, so the usual (co-resident) case pays no extra round trip.  The loop ends at convergence
// (typically 4 iterations).  The LAST workgroup to leave publishes the final state: by then every other workgroup
// has read the initial one.
// Filler: the SO3 iterations are 75 workgroups chasing each other through memory round trips for 36 us while the rest of
// the chip idles, and k_odo_prepare's per-pixel work (all levels; needed only by the Gauss-Newton loop that follows) is
// independent of them.  Workgroups n_so3 .. gridDim.x - 1 do that work and leave; they take no ticket, count in no
// election and are dispatched after the SO3 workgroups (dispatch follows blockIdx).
struct So3Filler { OdoLevels all; const DevPose *dp; int do_rgb, level0_packed; unsigned int first[HRBF_NUM_PYRS + 1]; };
__global__ __launch_bounds__(RB) void k_so3_persistent(OdoLevel L, OdoState *st, long long *__restrict__ part,
                                                       OdoConfig cfg, int gn_level, unsigned int nchunk, unsigned int n_so3,
                                                       So3Filler fill)
{
    if (blockIdx.x >= n_so3) {
        const unsigned int f = blockIdx.x - n_so3;
#pragma unroll
        for (int lv = 0; lv < HRBF_NUM_PYRS; ++lv)
            if (f >= fill.first[lv] && f < fill.first[lv + 1]) {
                const OdoLevel &FL = fill.all.lv[lv];
                const int i = (int)((f - fill.first[lv]) * RB + threadIdx.x);
                if (i < FL.rows * FL.cols) prepare_pixel(FL, lv, i, fill.dp, cfg, fill.do_rgb, fill.level0_packed);
            }
        return;
    }
    __shared__ So3Local S;
    __shared__ long long s_tot[33];
    __shared__ unsigned int s_chunk, s_next;
    __shared__ int s_more;
    if (threadIdx.x == 0) {
        s_chunk = __hip_atomic_fetch_add(&st->so3_ticket0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = 0; k < 9; ++k) {
            S.resultR[k] = st->resultR[k]; S.lastResultR[k] = st->lastResultR[k]; S.R_lr[k] = st->R_lr[k];
            S.basis[k] = st->basis[k]; S.kinv[k] = st->kinv[k]; S.krlr[k] = st->krlr[k];
        }
        S.so3_lastError = st->so3_lastError; S.so3_lastCount = st->so3_lastCount; S.so3_done = st->so3_done;
    }
    __syncthreads();
    unsigned int chunk = s_chunk;
    for (int it = 0; it < SO3_ITERS; ++it) {
        if (S.so3_done) break;                      // identical in every workgroup
        long long *set = part + (size_t)it * ODO_SLOTS * 33;
        bool drawn = false;
        for (;;) {
            if (chunk < nchunk) {
                float row4[11];
#pragma unroll
                for (int k = 0; k < 11; ++k) row4[k] = 0.0f;
                const bool valid = so3_pixel(L, &S, (int)(chunk * RB + threadIdx.x), row4);
                block_reduce_exact<11>(row4, valid, set);
            }
            __syncthreads();
            if (threadIdx.x == 0) {   // wave 0 issued this workgroup's slot atomics: the release fence orders them first
                // one 64-bit atomic: +1 chunk done (if one was processed) and, the first time round, +1 ticket of it+1
                const unsigned long long inc = (chunk < nchunk ? 1ull : 0ull) | (drawn ? 0ull : (1ull << 32));
                if (inc) {
                    if (chunk < nchunk) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    const unsigned long long old = __hip_atomic_fetch_add(&st->so3_word[it], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!drawn) s_next = (unsigned int)(old >> 32);
                }
                int more = 0, spins = 0;
                for (;;) {
                    if ((unsigned int)__hip_atomic_load(&st->so3_word[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nchunk) break;
                    // chunks of this iteration nobody has claimed yet (only when fewer workgroups run than there are chunks)
                    const unsigned int drawn_it = it == 0 ? __hip_atomic_load(&st->so3_ticket0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                          : (unsigned int)(__hip_atomic_load(&st->so3_word[it - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
                    if (drawn_it < nchunk) {
                        const unsigned int c = it == 0 ? __hip_atomic_fetch_add(&st->so3_ticket0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                       : (unsigned int)(__hip_atomic_fetch_add(&st->so3_word[it - 1], 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
                        if (c < nchunk) { s_chunk = c; more = 1; break; }
                    }
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) { st->bar_timeout = 1; break; }   // cannot happen by construction; loud if it does
                }
                s_more = more;
            }
            drawn = true;
            __syncthreads();
            if (!s_more) break;
            chunk = s_chunk;
            __syncthreads();   // s_chunk / s_more are rewritten by the next round
        }
        if (threadIdx.x < 33) {
            long long v[ODO_SLOTS];
#pragma unroll
            for (int b = 0; b < ODO_SLOTS; ++b)
                v[b] = __hip_atomic_load(&set[b * 33 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long t = 0;
#pragma unroll
            for (int b = 0; b < ODO_SLOTS; ++b) t += v[b];
            s_tot[threadIdx.x] = t;
        }
        chunk = s_next;
        __syncthreads();
        if (threadIdx.x == 0) so3_step(&S, s_tot, cfg);
        __syncthreads();
    }
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(&st->so3_exit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_so3 - 1u) {
        for (int k = 0; k < 9; ++k) {
            st->resultR[k] = S.resultR[k]; st->lastResultR[k] = S.lastResultR[k]; st->R_lr[k] = S.R_lr[k];
            st->basis[k] = S.basis[k]; st->kinv[k] = S.kinv[k]; st->krlr[k] = S.krlr[k];
        }
        st->so3_lastError = S.so3_lastError; st->so3_lastCount = S.so3_lastCount; st->so3_done = S.so3_done;
        if (gn_level >= 0) gn_begin_state(st, cfg, gn_level);
    }
}

// ------------------------------------------------------------------------------------------ O3 + O4 fused launch
// The Gauss-Newton kernels are chains of memory round trips (VALU busy 11-12 %): what hipcc does with `load; if (field) return;
// use the rest` is to fetch the tested field alone and the rest behind the branch — one more dependent round trip per early exit
// (the ICP pixel took five: c0.w, c0.xyz, m1.w, the other nine floats, m0.w; two are needed).  hold2 / hold4 pin whole texels
// that were requested together in registers before the first test: same values, same arithmetic, fewer round trips.
__device__ __forceinline__ void hold2(float4 &a, float4 &b)
{
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
}
struct IcpArgs {
    const float *vmap_c, *nmap_c, *ck1_c, *ck2_c, *vmap_g, *nmap_g, *ck1_g, *ck2_g, *icpw;
    const float4 *cur_tex, *model_tex;   // packed operands (pack_icp_texels); null in the icpStep seam
    int rows, cols;
    float fx, fy, cx, cy, distThres, angleThres;
    int use_search, radius, use_weight;
    float4 *sparse;       // sparse (ADMM) ICP side image of this level, or null: {lambda.xyz, corres} {z.xyz, -}
    int sparse_first;     // first iteration of the level: the multiplier starts from zero (RGBDOdometry.cpp:964-977)
};
// per-pixel in/out of the sparse variant: multiplier in; shrunk residual and chosen model pixel out
struct SparseIo { f3 lambda, z; int bx, by; };

// The same association for the common case (no windowed search) on the packed operands written once per frame by
// pack_icp_texels: current pixel {v.xyz, valid} {n.xyz, -}, model pixel {v.xyz, icp weight} {n.xyz, valid} — two
// 16-byte loads + two 16-byte gathers instead of eight 4-byte plane loads + nine 4-byte plane gathers per pixel
// and iteration.  `valid` folds the four NaN tests of the planar version; every arithmetic step is the same.
// the 27 upper-triangle products of the weighted row (JtJ | Jtr), the weighted squared residual and the count
// (reduce.cu:374-395 / :764-786): what both Gauss-Newton terms hand to the exact reduction
__device__ __forceinline__ void products29(const float (&row)[7], float weight, float *out)
{
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 7; ++j) out[k++] = weight * row[i] * row[j];
    out[27] = weight * row[6] * row[6];
    out[28] = 1.0f;
}

__device__ __forceinline__ bool icp_pixel_packed_rows(const IcpArgs &A, const float *Rcurr, f3 tcurr, const float *Rpi, f3 tprev,
                                                      int x, int y, float (&row)[7], float &weight);
__device__ __forceinline__ bool icp_pixel_packed_rows_pre(const IcpArgs &A, const float *Rcurr, f3 tcurr, const float *Rpi, f3 tprev,
                                                          const float4 c0, const float4 c1, float (&row)[7], float &weight);

__device__ __forceinline__ bool icp_pixel_packed(const IcpArgs &A, const float *Rcurr, f3 tcurr, const float *Rpi, f3 tprev,
                                                 int x, int y, float *out)
{
    float row[7], weight;
    if (!icp_pixel_packed_rows(A, Rcurr, tcurr, Rpi, tprev, x, y, row, weight)) return false;
    products29(row, weight, out);
    return true;
}

// ... returning the row and its weight instead of the 29 products: the Gauss-Newton kernel forms the products after the
// control flow has merged (8 values to carry through the early exits instead of 29)
__device__ __forceinline__ bool icp_pixel_packed_rows(const IcpArgs &A, const float *Rcurr, f3 tcurr, const float *Rpi, f3 tprev,
                                                      int x, int y, float (&row)[7], float &weight)
{
    const int cols = A.cols;
    float4 c0 = A.cur_tex[2 * (y * cols + x)], c1 = A.cur_tex[2 * (y * cols + x) + 1];
    hold2(c0, c1);
    return icp_pixel_packed_rows_pre(A, Rcurr, tcurr, Rpi, tprev, c0, c1, row, weight);
}
// ... with the pixel's two texels already in registers (the Gauss-Newton kernel requests them beside the registration state)
__device__ __forceinline__ bool icp_pixel_packed_rows_pre(const IcpArgs &A, const float *Rcurr, f3 tcurr, const float *Rpi, f3 tprev,
                                                          const float4 c0, const float4 c1, float (&row)[7], float &weight)
{
    const int rows = A.rows, cols = A.cols;
    if (c0.w == 0.0f) return false;
    const f3 vcur = mk3(c0.x, c0.y, c0.z), ncur = mk3(c1.x, c1.y, c1.z);
    f3 vg_ = add3(m33_mul(Rcurr, vcur), tcurr);
    f3 vcp = m33_mul(Rpi, sub3(vg_, tprev));
    float fu = vcp.x * A.fx / vcp.z + A.cx, fv = vcp.y * A.fy / vcp.z + A.cy;
    if (hd_isnanf(fu) || hd_isnanf(fv)) return false;
    if (!(fu > -1.0e9f && fu < 1.0e9f && fv > -1.0e9f && fv < 1.0e9f)) return false;
    int ux = (int)hd_rintf(fu), uy = (int)hd_rintf(fv);
    if (ux < 0 || uy < 0 || ux >= cols || uy >= rows || vcp.z < 0.0f) return false;
    f3 ncur_g = m33_mul(Rcurr, ncur);
    float4 m0 = A.model_tex[2 * (uy * cols + ux)], m1 = A.model_tex[2 * (uy * cols + ux) + 1];
    hold2(m0, m1);
    if (m1.w == 0.0f) return false;
    const f3 bv = mk3(m0.x, m0.y, m0.z), bn = mk3(m1.x, m1.y, m1.z);
    float dist = len3(sub3(bv, vg_)), sine = len3(cross3(ncur_g, bn));
    if (sine > A.angleThres || dist > A.distThres) return false;
    f3 s_cp = m33_mul(Rpi, sub3(vg_, tprev));
    f3 d_cp = m33_mul(Rpi, sub3(bv, tprev));
    f3 n_cp = m33_mul(Rpi, bn);
    weight = 1.0f;
    if (A.use_weight) { float w = m0.w; weight = hd_isnanf(w) ? 0.0f : w; }
    f3 cr = cross3(s_cp, n_cp);
    row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z; row[3] = cr.x; row[4] = cr.y; row[5] = cr.z;
    row[6] = dot3(n_cp, sub3(s_cp, d_cp));
    return true;
}

template <bool SPARSE = false>
__device__ __forceinline__ bool icp_pixel(const IcpArgs &A, const float *Rcurr, f3 tcurr, const float *Rpi, f3 tprev,
                                          int x, int y, float *out, SparseIo *io = nullptr)
{
    if (!SPARSE && A.cur_tex && !A.use_search) return icp_pixel_packed(A, Rcurr, tcurr, Rpi, tprev, x, y, out);
    const int rows = A.rows, cols = A.cols;
    f3 vcur = mk3(PLN(A.vmap_c, 0, rows, cols, y, x), PLN(A.vmap_c, 1, rows, cols, y, x), PLN(A.vmap_c, 2, rows, cols, y, x));
    f3 ncur = mk3(PLN(A.nmap_c, 0, rows, cols, y, x), PLN(A.nmap_c, 1, rows, cols, y, x), PLN(A.nmap_c, 2, rows, cols, y, x));
    float ck1 = PLN(A.ck1_c, 3, rows, cols, y, x), ck2 = PLN(A.ck2_c, 3, rows, cols, y, x);
    if (hd_isnanf(vcur.x) || hd_isnanf(ncur.x) || hd_isnanf(ck1) || hd_isnanf(ck2)) return false;
    f3 vg_ = add3(m33_mul(Rcurr, vcur), tcurr);
    f3 vcp = m33_mul(Rpi, sub3(vg_, tprev));
    float fu = vcp.x * A.fx / vcp.z + A.cx, fv = vcp.y * A.fy / vcp.z + A.cy;
    if (hd_isnanf(fu) || hd_isnanf(fv)) return false;
    if (!(fu > -1.0e9f && fu < 1.0e9f && fv > -1.0e9f && fv < 1.0e9f)) return false;
    int ux = (int)hd_rintf(fu), uy = (int)hd_rintf(fv);
    if (ux < 0 || uy < 0 || ux >= cols || uy >= rows || vcp.z < 0.0f) return false;
    f3 ncur_g = m33_mul(Rcurr, ncur);
    const int R = A.use_search ? A.radius : 0;
    bool found = false;
    int bx = -1, by = -1;
    f3 bv = mk3(0, 0, 0), bn = mk3(0, 0, 0);
    float p_smallest = 1e8f, DpR = -1e8f;
    if (A.use_search)
        for (int cy_ = uy - R; cy_ < uy + R + 1; ++cy_)
            for (int cx_ = ux - R; cx_ < ux + R + 1; ++cx_) {
                if (cx_ < 0 || cy_ < 0 || cx_ >= cols || cy_ >= rows) continue;
                f3 vp = mk3(PLN(A.vmap_g, 0, rows, cols, cy_, cx_), PLN(A.vmap_g, 1, rows, cols, cy_, cx_), PLN(A.vmap_g, 2, rows, cols, cy_, cx_));
                f3 np = mk3(PLN(A.nmap_g, 0, rows, cols, cy_, cx_), PLN(A.nmap_g, 1, rows, cols, cy_, cx_), PLN(A.nmap_g, 2, rows, cols, cy_, cx_));
                float c1 = PLN(A.ck1_g, 3, rows, cols, cy_, cx_), c2 = PLN(A.ck2_g, 3, rows, cols, cy_, cx_);
                if (hd_isnanf(vp.x) || hd_isnanf(np.x) || hd_isnanf(c1) || hd_isnanf(c2)) continue;
                float dist = len3(sub3(vp, vg_)), sine = len3(cross3(ncur_g, np));
                if (sine > A.angleThres || dist > A.distThres) continue;
                if (dist > DpR) DpR = dist;
            }
    for (int cy_ = uy - R; cy_ < uy + R + 1; ++cy_)
        for (int cx_ = ux - R; cx_ < ux + R + 1; ++cx_) {
            if (cx_ < 0 || cy_ < 0 || cx_ >= cols || cy_ >= rows) continue;
            f3 vp = mk3(PLN(A.vmap_g, 0, rows, cols, cy_, cx_), PLN(A.vmap_g, 1, rows, cols, cy_, cx_), PLN(A.vmap_g, 2, rows, cols, cy_, cx_));
            f3 np = mk3(PLN(A.nmap_g, 0, rows, cols, cy_, cx_), PLN(A.nmap_g, 1, rows, cols, cy_, cx_), PLN(A.nmap_g, 2, rows, cols, cy_, cx_));
            float c1 = PLN(A.ck1_g, 3, rows, cols, cy_, cx_), c2 = PLN(A.ck2_g, 3, rows, cols, cy_, cx_);
            if (hd_isnanf(vp.x) || hd_isnanf(np.x) || hd_isnanf(c1) || hd_isnanf(c2)) continue;
            float dist = len3(sub3(vp, vg_)), sine = len3(cross3(ncur_g, np));
            if (sine > A.angleThres || dist > A.distThres) continue;
            float p = 1.0f;
            if (A.use_search) {
                float a1 = hd_fabsf(c1), a2 = hd_fabsf(c2);
                float ckmax = a1 > a2 ? a1 : a2;
                float D_p = dist / DpR;
                float D_n = 1.0f - dot3(np, ncur_g);
                float D_c = 1.0f - hd_expf(-hd_fabsf(c1 - ck1) / ckmax) * hd_expf(-hd_fabsf(c2 - ck2) / ckmax);
                p = (0.333f * D_p + 0.333f * D_n) + 0.333f * D_c;
            }
            if (p < p_smallest) { bx = cx_; by = cy_; bv = vp; bn = np; p_smallest = p; }
            found = true;
        }
    if (!found) return false;
    f3 s_cp = m33_mul(Rpi, sub3(vg_, tprev));
    f3 d_cp = m33_mul(Rpi, sub3(bv, tprev));
    f3 n_cp = m33_mul(Rpi, bn);
    if (SPARSE) {   // reduce.cu:479-492: the target moves by the shrunk residual minus the scaled multiplier
        io->bx = bx; io->by = by;
        const f3 lm = mk3(io->lambda.x / HD_SPARSE_MU, io->lambda.y / HD_SPARSE_MU, io->lambda.z / HD_SPARSE_MU);
        const f3 h = add3(sub3(s_cp, d_cp), lm);
        const float beta = hd_sparse_shrink_factor(len3(h));
        io->z = mk3(beta * h.x, beta * h.y, beta * h.z);
        d_cp = sub3(add3(d_cp, io->z), lm);
    }
    float weight = 1.0f;
    if (A.use_weight) { float w = bx < 0 ? 0.0f : A.icpw[by * cols + bx]; weight = hd_isnanf(w) ? 0.0f : w; }   // bx < 0: see oracle/orc_odo.c
    float row[7];
    f3 cr = cross3(s_cp, n_cp);
    row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z; row[3] = cr.x; row[4] = cr.y; row[5] = cr.z;
    row[6] = dot3(n_cp, sub3(s_cp, d_cp));
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 7; ++j) out[k++] = weight * row[i] * row[j];
    out[27] = weight * row[6] * row[6];
    out[28] = 1.0f;
    return true;
}

// Sparse (ADMM) ICP, use_sparse_icp: updateLambdaMapKernel (cudafuncs.cu:1030-1080) of the previous iteration is
// folded into the head of this one — it only touches the pixel's own multiplier, at the pose this iteration runs at —
// then the association with the shrink step; {lambda, corres} {z} go back to the side image for the next iteration.
// The update tests corresp.x > 0, so matches in model column 0 never move their multiplier (kept).
__device__ __forceinline__ bool icp_pixel_sparse(const IcpArgs &A, const float *Rcurr, f3 tcurr, const float *Rpi, f3 tprev,
                                                 int x, int y, float *out)
{
    const int rows = A.rows, cols = A.cols, k = y * cols + x;
    SparseIo io;
    io.lambda = mk3(0, 0, 0); io.z = mk3(0, 0, 0); io.bx = -1; io.by = -1;
    if (!A.sparse_first) {
        const float4 s0 = A.sparse[2 * k], s1 = A.sparse[2 * k + 1];
        io.lambda = mk3(s0.x, s0.y, s0.z);
        const int cp = __float_as_int(s0.w);
        const int ux = cp == -1 ? -1 : (cp & 0xffff), uy = cp >> 16;
        if (ux > 0) {
            const f3 vcur = mk3(PLN(A.vmap_c, 0, rows, cols, y, x), PLN(A.vmap_c, 1, rows, cols, y, x), PLN(A.vmap_c, 2, rows, cols, y, x));
            const f3 vlp = m33_mul(Rpi, sub3(add3(m33_mul(Rcurr, vcur), tcurr), tprev));
            const f3 vp = m33_mul(Rpi, sub3(mk3(PLN(A.vmap_g, 0, rows, cols, uy, ux), PLN(A.vmap_g, 1, rows, cols, uy, ux),
                                                 PLN(A.vmap_g, 2, rows, cols, uy, ux)), tprev));
            const f3 d = sub3(sub3(vlp, vp), mk3(s1.x, s1.y, s1.z));
            io.lambda = add3(io.lambda, mk3(HD_SPARSE_MU * d.x, HD_SPARSE_MU * d.y, HD_SPARSE_MU * d.z));
        }
    }
    const bool found = icp_pixel<true>(A, Rcurr, tcurr, Rpi, tprev, x, y, out, &io);
    const int cp = io.bx < 0 ? -1 : (int)((uint32_t)io.bx | ((uint32_t)io.by << 16));
    A.sparse[2 * k] = make_float4(io.lambda.x, io.lambda.y, io.lambda.z, __int_as_float(cp));
    A.sparse[2 * k + 1] = make_float4(io.z.x, io.z.y, io.z.z, 0.0f);
    return found;
}

// RGBResidual::getProducts (reduce.cu:981-1060).  The correspondence (u0, v0, x, y, valid) and the intensity
// difference are returned in registers; the multi-launch path parks them in corres / corres_diff.
struct RgbCorr { int16_t c0, c1, c2, c3, c4; float diff; };
template <class S>
__device__ __forceinline__ RgbCorr rgb_residual_pixel(const OdoLevel &L, const S *st, float minScale, int k,
                                                      long long &cnt, long long &sig)
{
    const int rows = L.rows, cols = L.cols;
    const int i = k / cols, j0 = k - i * cols;
    (void)rows;
    RgbCorr r; r.c0 = r.c1 = r.c2 = r.c3 = r.c4 = 0; r.diff = 0.0f;
    // the window / gradient / depth part of the test does not depend on the pose: k_odo_prepare evaluates it once per
    // frame (rgb_mask), the iterations read one byte instead of 16 + 2 + 1 values; the seam kernels pass no mask
    // the pixel's own three values are requested together, and so are the two of the model pixel further down (five dependent
    // round trips otherwise: mask, depth, model depth, model intensity, own intensity)
    float d1 = L.next_depth[k];
    int own_i = (int)L.next_image[k];
    int mask_b = (int)*(L.rgb_mask ? L.rgb_mask + k : L.next_image + k);   // (an unconditional load: no wait where a branch would end)
    float krk[9], kt[3];   // the warp's operands travel beside the pixel's values, not behind them
#pragma unroll
    for (int q = 0; q < 9; ++q) krk[q] = st->krk[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) kt[q] = st->kt[q];
    asm volatile("" : "+s"(krk[0]), "+s"(krk[1]), "+s"(krk[2]), "+s"(krk[3]), "+s"(krk[4]), "+s"(krk[5]), "+s"(krk[6]), "+s"(krk[7]),
                      "+s"(krk[8]), "+s"(kt[0]), "+s"(kt[1]), "+s"(kt[2]));
    asm volatile("" : "+v"(d1), "+v"(own_i), "+v"(mask_b));
    const bool pre = L.rgb_mask ? mask_b != 0 : rgb_residual_static_test(L, minScale, k);   // (no mask: the seam kernels)
    if (!pre) return r;
    const int y = i, x = j0;
    float td1 = d1 * ((krk[6] * (float)x + krk[7] * (float)y) + krk[8]) + kt[2];
    float fu = (d1 * ((krk[0] * (float)x + krk[1] * (float)y) + krk[2]) + kt[0]) / td1;
    float fv = (d1 * ((krk[3] * (float)x + krk[4] * (float)y) + krk[5]) + kt[1]) / td1;
    if (fu > -1.0e9f && fu < 1.0e9f && fv > -1.0e9f && fv < 1.0e9f) {
        int u0 = (int)hd_rintf(fu), v0 = (int)hd_rintf(fv);
        if (u0 >= 0 && v0 >= 0 && u0 < cols && v0 < rows) {
            float d0 = L.last_depth[v0 * cols + u0];
            int model_i = (int)L.last_image[v0 * cols + u0];
            asm volatile("" : "+v"(d0), "+v"(model_i));
            if (d0 > 0.0f && hd_fabsf(td1 - d0) <= 0.07f && model_i != 0) {
                float diff = (float)own_i - (float)model_i;
                r.c0 = (int16_t)u0; r.c1 = (int16_t)v0; r.c2 = (int16_t)x; r.c3 = (int16_t)y; r.c4 = 1;
                r.diff = diff;
                cnt += 1;
                sig += (long long)(diff * diff);
            }
        }
    }
    return r;
}

// RGBReduction::getProducts (reduce.cu:717-808) for one correspondence: operands already fetched
__device__ __forceinline__ void rgb_row_core(float diff, float cpx, float cpy, float cpz, int gx_raw, int gy_raw,
                                             float sigma, float fx, float fy, int use_grad, float (&row)[7], float &rw)
{
    float w = sigma + hd_fabsf(diff);
    w = w > 1.19209290e-07f ? 1.0f / w : 1.0f;
    if (sigma == -1.0f) w = 1.0f;
    row[6] = -w * diff;
    float invz = 1.0f / cpz;
    float dIx = w * 0.125f * (float)gx_raw;
    float dIy = w * 0.125f * (float)gy_raw;
    float v0 = dIx * fx * invz, v1 = dIy * fy * invz;
    float v2 = -(v0 * cpx + v1 * cpy) * invz;
    row[0] = v0; row[1] = v1; row[2] = v2;
    row[3] = -cpz * v1 + cpy * v2;
    row[4] = cpz * v0 - cpx * v2;
    row[5] = -cpy * v0 + cpx * v1;
    rw = 1.0f;
    if (use_grad) {
        float gm = hd_sqrtf(dIx * dIx + dIy * dIy);
        rw = hd_expf(-0.5f * (10.0f / gm) * (10.0f / gm));
    }
}
__device__ __forceinline__ void rgb_products_core(float diff, float cpx, float cpy, float cpz, int gx_raw, int gy_raw,
                                                  float sigma, float fx, float fy, int use_grad, float (&out)[29])
{
    float row[7], rw;
    rgb_row_core(diff, cpx, cpy, cpz, gx_raw, gy_raw, sigma, fx, fy, use_grad, row, rw);
    products29(row, rw, out);
}
// the same on the seam layout: DataTerm as 6 x int16 + diff plane, cloud as 3 floats, gradients as two short planes
__device__ __forceinline__ bool rgb_products_pixel(const OdoLevel &L, const RgbCorr &co, float sigma, float fx, float fy,
                                                   int use_grad, float (&out)[29])
{
    if (!co.c4) return false;
    const int cols = L.cols;
    const float *cpp = &L.cloud[((size_t)co.c1 * cols + co.c0) * 3];
    rgb_products_core(co.diff, cpp[0], cpp[1], cpp[2], (int)L.dIdx[co.c3 * cols + co.c2], (int)L.dIdy[co.c3 * cols + co.c2],
                      sigma, fx, fy, use_grad, out);
    return true;
}

// blocks [0, nb) : ICP products ; blocks [nb, 2nb) : RGB residual.  Both read the same OdoState.
template <bool SPARSE>
__global__ __launch_bounds__(RB) void k_gn_icp_residual(OdoLevel L, IcpArgs A, const OdoState *__restrict__ st, int nb,
                                                       int do_icp, int do_rgb, float minScale,
                                                       long long *__restrict__ icp_part, long long *__restrict__ res_part,
                                                       int16_t *__restrict__ corres, float *__restrict__ corres_diff,
                                                       int p0, int p1 /* pixel range of this rank */)
{
    if ((int)blockIdx.x < nb) {
        float out[29];
#pragma unroll
        for (int k = 0; k < 29; ++k) out[k] = 0.0f;
        bool valid = false;
        const int i = p0 + blockIdx.x * RB + threadIdx.x;
        // the usual case (packed operands, no windowed search): the pixel's two texels and the registration state are requested side
        // by side, before the loop's stop flag is looked at — hipcc had them in a chain (flag, arguments, texels, state: four
        // dependent round trips in front of the model gather)
        const bool usual = !SPARSE && A.cur_tex && !A.use_search;
        // (unconditional loads — the head of the registration state stands in where there is no texel to fetch: a load under a
        //  branch is waited for where the branch ends, in front of the state)
        const float4 *tex = (usual && do_icp && i < p1) ? A.cur_tex + 2 * (size_t)i : reinterpret_cast<const float4 *>(st);
        float4 c0 = tex[0], c1 = tex[1];
        float Rc[9], Rp[9], tc[3], tp[3];
        int brk = st->gn_break;
#pragma unroll
        for (int k = 0; k < 9; ++k) { Rc[k] = st->Rcurr[k]; Rp[k] = st->Rprev_inv[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { tc[k] = st->tcurr[k]; tp[k] = st->tprev[k]; }
        asm volatile("" : "+s"(brk), "+s"(Rc[0]), "+s"(Rc[1]), "+s"(Rc[2]), "+s"(Rc[3]), "+s"(Rc[4]), "+s"(Rc[5]), "+s"(Rc[6]), "+s"(Rc[7]),
                          "+s"(Rc[8]), "+s"(tc[0]), "+s"(tc[1]), "+s"(tc[2]));
        asm volatile("" : "+s"(Rp[0]), "+s"(Rp[1]), "+s"(Rp[2]), "+s"(Rp[3]), "+s"(Rp[4]), "+s"(Rp[5]), "+s"(Rp[6]), "+s"(Rp[7]), "+s"(Rp[8]),
                          "+s"(tp[0]), "+s"(tp[1]), "+s"(tp[2]));
        hold2(c0, c1);
        if (do_icp && !brk && i < p1) {
            const int y = i / A.cols, x = i - y * A.cols;
            if (SPARSE)
                valid = icp_pixel_sparse(A, Rc, mk3(tc[0], tc[1], tc[2]), Rp, mk3(tp[0], tp[1], tp[2]), x, y, out);
            else if (usual) {   // products formed after the early exits
                float row[7], weight;
                valid = icp_pixel_packed_rows_pre(A, Rc, mk3(tc[0], tc[1], tc[2]), Rp, mk3(tp[0], tp[1], tp[2]), c0, c1, row, weight);
                if (!valid) {
                    weight = 0.0f;
#pragma unroll
                    for (int k = 0; k < 7; ++k) row[k] = 0.0f;
                }
                products29(row, weight, out);          // a lane without a match contributes 0 * 0 * 0
                out[28] = valid ? 1.0f : 0.0f;
            } else
                valid = icp_pixel(A, Rc, mk3(tc[0], tc[1], tc[2]), Rp, mk3(tp[0], tp[1], tp[2]), x, y, out);
        }
        // icp_part rows are indexed by blockIdx.x in [0, nb); every path leaves exact zeros in the lanes it rejects
        block_reduce_exact<29, true>(out, valid, icp_part);
    } else {
        __shared__ long long s_c[RB / 64], s_s[RB / 64];
        const int b = blockIdx.x - nb;
        const int k = p0 + b * RB + threadIdx.x;
        long long cnt = 0, sig = 0;
        if (do_rgb && !st->gn_break && k < p1) {
            const RgbCorr r = rgb_residual_pixel(L, st, minScale, k, cnt, sig);
            // compact in-frame record (the pixel's own x, y are implied by k): {u0 | v0 << 16 or -1, diff}
            reinterpret_cast<int2 *>(corres)[k] =
                make_int2(r.c4 ? (int)((uint32_t)(uint16_t)r.c0 | ((uint32_t)(uint16_t)r.c1 << 16)) : -1, __float_as_int(r.diff));
        }
        for (int d = 32; d > 0; d >>= 1) { cnt += __shfl_down(cnt, d); sig += __shfl_down(sig, d); }
        if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = cnt; s_s[threadIdx.x >> 6] = sig; }
        __syncthreads();
        if (threadIdx.x == 0) {
            long long c = 0, s = 0;
            for (int w = 0; w < RB / 64; ++w) { c += s_c[w]; s += s_s[w]; }
            if (c) {
                atomicAdd((unsigned long long *)&res_part[(b % RES_SLOTS) * 2], (unsigned long long)c);
                atomicAdd((unsigned long long *)&res_part[(b % RES_SLOTS) * 2 + 1], (unsigned long long)s);
            }
        }
    }
}

__device__ inline void frame_weighting_state(DevPose *dp, float weight_multiplier);   // defined with the pose kernels

// end of registration: 0.3 m guard (RGBDOdometry.cpp:1232-1236), publish the pose
__device__ inline void odo_end_state(OdoState *st, DevPose *dp, const OdoConfig &cfg)
{
    const int rgb = cfg.rgb_only || cfg.icp_weight < 100.0f;
    if (rgb) {
        f3 d = mk3(st->tcurr[0] - st->tprev[0], st->tcurr[1] - st->tprev[1], st->tcurr[2] - st->tprev[2]);
        if (len3(d) > 0.3f) {
            for (int k = 0; k < 9; ++k) st->Rcurr[k] = st->Rprev[k];
            for (int k = 0; k < 3; ++k) st->tcurr[k] = st->tprev[k];
        }
    }
    if (st->bar_timeout) st->tcurr[0] = hd_nanf();   // a grid barrier gave up: make the failure impossible to miss
    for (int k = 0; k < 9; ++k) dp->pose.r[k] = st->Rcurr[k];
    for (int k = 0; k < 3; ++k) dp->pose.t[k] = st->tcurr[k];
    dp->tinv = rigid_inverse(dp->pose);
    dp->last_icp_error = st->last_icp_error;
    dp->last_icp_count = st->last_icp_count;
}

// solve + SE3 update (RGBDOdometry.cpp:1162-1204, OdometryProvider.h:73-93) from the folded limb totals in
// s_tot (LDS, icp 0..86 | rgb 87..173); also prepares the operands of the next iteration (possibly on the next
// pyramid level).  res_c / res_s: the folded RGB residual (count, sigma).  Whole workgroup; S is OdoState or the
// LDS copy of the persistent kernel.  dp != null marks the last iteration of the registration: the pose is
// published (only meaningful when S is the global state).
template <class S>
__device__ __forceinline__ void gn_step_from_totals(S *st, const long long *s_tot, long long res_c, long long res_s,
                                                    const OdoConfig &cfg, int next_level, int level_changes)
{
    __shared__ double s_val[58];
    __shared__ double s_A[36], s_b[6];
    const int tid = threadIdx.x;
    const int rgbOnly = cfg.rgb_only;
    const int icp = !rgbOnly && cfg.icp_weight > 0.0f;
    const int rgb = rgbOnly || cfg.icp_weight < 100.0f;
    if (tid < 58) s_val[tid] = limbs_to_double(s_tot + (tid < 29 ? 0 : 87), tid < 29 ? tid : tid - 29);
    __syncthreads();
    if (tid < 42) {   // entries of the combined normal equations (RGBDOdometry.cpp:1162-1178)
        const int r = tid < 36 ? tid / 6 : tid - 36, c = tid < 36 ? tid % 6 : 6;
        const int i = r < c ? r : c, j = r < c ? c : r;
        const int shift = i * 7 - (i * (i - 1)) / 2 + (j - i);   // row-major upper triangle incl. the rhs column
        const float vi = icp ? (float)s_val[shift] : 0.0f, vr = rgb ? (float)s_val[29 + shift] : 0.0f;
        const double w = cfg.icp_weight;
        double v;
        if (icp && rgb) v = (double)vr + (tid < 36 ? w * w : w) * (double)vi;
        else v = icp ? (double)vi : (double)vr;
        if (tid < 36) s_A[tid] = v; else s_b[tid - 36] = v;
    }
    __syncthreads();
    if (tid == 0) {
        // the reference sums diff^2 in a 32-bit int (reduce.cu:985-1046, RGBDOdometry.cpp:994-1018): the sum wraps beyond 2^31
        const long long c = res_c, sg = (long long)(int)(unsigned int)(unsigned long long)res_s;
        if (rgb) {
            float rgbError = (float)(hd_sqrt((double)sg) / (double)(c == 0 ? 1 : c));
            if (rgbOnly && rgbError > st->lastRGBError) st->gn_break = 1;
            if (!st->gn_break) st->lastRGBError = rgbError;
        }
        if (!st->gn_break) {
            if (icp) { st->res_icp[0] = (float)s_val[27]; st->res_icp[1] = (float)s_val[28]; }
            st->last_icp_error = hd_sqrtf(st->res_icp[0]) / st->res_icp[1];
            st->last_icp_count = st->res_icp[1];
            double lastA[36], lastb[6], result[6];
#pragma unroll
            for (int k = 0; k < 36; ++k) lastA[k] = s_A[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) lastb[k] = s_b[k];
            ldlt_solve<double, 6>(lastA, lastb, result);
            double rv[3] = {result[3], result[4], result[5]}, Ru[9], U[16], N[16];
            rodrigues(rv, Ru);
            for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) U[r * 4 + k] = Ru[r * 3 + k]; U[r * 4 + 3] = result[r]; }
            U[12] = U[13] = U[14] = 0; U[15] = 1;
            double *Rt = st->Rt;
            for (int r = 0; r < 4; ++r) for (int k = 0; k < 4; ++k)
                N[r * 4 + k] = ((U[r * 4] * Rt[k] + U[r * 4 + 1] * Rt[4 + k]) + U[r * 4 + 2] * Rt[8 + k]) + U[r * 4 + 3] * Rt[12 + k];
            for (int k = 0; k < 16; ++k) Rt[k] = N[k];
            float oR[9], ot[3];
            for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) oR[r * 3 + k] = (float)Rt[r * 4 + k]; ot[r] = (float)Rt[r * 4 + 3]; }
            float iR[9], it_[3];
            for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) iR[r * 3 + k] = oR[k * 3 + r];
            for (int r = 0; r < 3; ++r) it_[r] = -((iR[r * 3] * ot[0] + iR[r * 3 + 1] * ot[1]) + iR[r * 3 + 2] * ot[2]);
            float Rc[9];
            mul3<float>(st->Rprev, iR, Rc);
            for (int k = 0; k < 9; ++k) st->Rcurr[k] = Rc[k];
            f3 rt = m33_mul(st->Rprev, mk3(it_[0], it_[1], it_[2]));
            st->tcurr[0] = rt.x + st->tprev[0]; st->tcurr[1] = rt.y + st->tprev[1]; st->tcurr[2] = rt.z + st->tprev[2];
        }
        if (level_changes) { st->gn_break = 0; st->lastRGBError = 3.402823466e+38f; }
        if (next_level >= 0) {
            const int div = 1 << next_level;
            gn_set_operands(st, cfg.fx / div, cfg.fy / div, cfg.cx / div, cfg.cy / div);
        }
    }
    __syncthreads();
}

// multi-launch path: fold the slot rows (or take totals that were summed elsewhere), then the step
__device__ __forceinline__ void gn_solve_block(OdoState *st, long long *__restrict__ icp_part,
                                               long long *__restrict__ rgb_part, long long *__restrict__ res_part,
                                               long long *__restrict__ totals, int do_reduce, long long res_c,
                                               long long res_s, const OdoConfig &cfg, int next_level, int level_changes,
                                               DevPose *dp)
{
    // icp_part and rgb_part are adjacent (OdoBuffers): 32 slot rows x (87 + 87) limbs
    __shared__ long long s_tot[176];
    const int tid = threadIdx.x;
    if (do_reduce) {
        // 256 threads: one column of the 174 per thread, its 32 slot rows loaded before the first add.  (1024 threads split the
        // rows four ways but capped the kernel at 128 VGPRs: the single-lane fp64 solve then spilled 80 B / lane to scratch, on
        // the serial chain that dominates this kernel.)
        fold_slots<false, 1>(icp_part, rgb_part, 87, s_tot, totals);
        for (int t = tid; t < RES_SLOTS * 2; t += blockDim.x) res_part[t] = 0;
    } else {
        for (int t = tid; t < 174; t += blockDim.x) s_tot[t] = totals[t];
    }
    __syncthreads();
    gn_step_from_totals(st, s_tot, res_c, res_s, cfg, next_level, level_changes);
    if (tid == 0 && dp) odo_end_state(st, dp, cfg);
}

// every workgroup folds the residual slots into (count, sigma) and derives sigmaVal (RGBDOdometry.cpp:1017-1030)
template <bool IN_LAUNCH>
__device__ __forceinline__ void fold_residual(const long long *__restrict__ res_part, int gn_break, float lastRGBError,
                                              int rgb_only, float *s_sigma, int *s_break, long long *s_res)
{
    __shared__ long long s_c[RB / 64], s_s[RB / 64];
    long long cnt = 0, sig = 0;
    for (int b = threadIdx.x; b < RES_SLOTS; b += RB) {
        if (IN_LAUNCH) {
            cnt += __hip_atomic_load(&res_part[b * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sig += __hip_atomic_load(&res_part[b * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else { cnt += res_part[b * 2]; sig += res_part[b * 2 + 1]; }
    }
    for (int d = 32; d > 0; d >>= 1) { cnt += __shfl_down(cnt, d); sig += __shfl_down(sig, d); }
    if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = cnt; s_s[threadIdx.x >> 6] = sig; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long c = 0, s = 0;
        for (int w = 0; w < RB / 64; ++w) { c += s_c[w]; s += s_s[w]; }
        s = (long long)(int)(unsigned int)(unsigned long long)s;   // the reference's `int sigma`: wraps beyond 2^31 (reduce.cu:985-1046)
        float sigmaVal = hd_sqrtf((((float)s / (float)c) == 0.0f) ? 1.0f : (float)c);
        int brk = gn_break;
        if (rgb_only) {   // only this mode consults the error here (an fp64 sqrt + division on every workgroup's critical path)
            const float rgbError = (float)(hd_sqrt((double)s) / (double)(c == 0 ? 1 : c));
            if (rgbError > lastRGBError) brk = 1;
            sigmaVal = -1.0f;
        }
        *s_sigma = sigmaVal; *s_break = brk;
        s_res[0] = c; s_res[1] = s;
    }
    __syncthreads();
}

// RGBReduction::getProducts (reduce.cu:717-808) over the correspondences k_gn_icp_residual parked in corres
__global__ __launch_bounds__(RB) void k_gn_rgb_step(OdoLevel L, const OdoState *__restrict__ st, int nb, float fx,
                                                    float fy, int rgb_only, int use_grad,
                                                    const long long *__restrict__ res_part,
                                                    const int16_t *__restrict__ corres,
                                                    const float *__restrict__ corres_diff, long long *__restrict__ rgb_part,
                                                    long long *__restrict__ totals, int p0, int p1)
{
    __shared__ float s_sigma;
    __shared__ int s_break;
    __shared__ long long s_res[2];
    // The pixel's record and gradients, then the model point it names, are requested BEFORE the residual slots are folded (they
    // do not depend on sigma): the fold's own round trip and its two barriers then run under the gather instead of in front of the
    // record (residual slots -> record -> point were three dependent round trips).  A record left by an earlier iteration (the
    // loop has stopped: `brk`) names a pixel of this level; it is read and not used.
    const int k = p0 + blockIdx.x * RB + threadIdx.x;
    int brk0 = st->gn_break;
    float last_err = st->lastRGBError;
    asm volatile("" : "+s"(brk0), "+s"(last_err));   // (thread 0 read them behind the fold's first barrier)
    int2 rec = make_int2(-1, 0);
    int g = 0;
    if (k < p1) {
        rec = reinterpret_cast<const int2 *>(corres)[k];
        g = L.dIxy[k];
        asm volatile("" : "+v"(rec.x), "+v"(rec.y), "+v"(g));
    }
    // an unconditional load (texel 0 stands in where there is no point to fetch): a load under a branch is waited for where the
    // branch ends, in front of the fold
    const uint32_t u0 = (uint32_t)rec.x & 0xffffu, v0 = (uint32_t)rec.x >> 16;
    const bool has_point = rec.x != -1 && u0 < (uint32_t)L.cols && v0 < (uint32_t)L.rows;
    const float4 cp = L.cloud4[has_point ? (size_t)v0 * L.cols + u0 : (size_t)0];
    fold_residual<false>(res_part, brk0, last_err, rgb_only, &s_sigma, &s_break, s_res);
    if (blockIdx.x == 0 && threadIdx.x == 0) { totals[174] = s_res[0]; totals[175] = s_res[1]; }
    const float sigma = s_sigma;
    const int brk = s_break;
    float out[29], row[7] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, rw = 0.0f;
    bool valid = false;
    if (!brk && k < p1 && rec.x != -1) {
        rgb_row_core(__int_as_float(rec.y), cp.x, cp.y, cp.z, (int)(int16_t)(g & 0xffff), (int)(int16_t)((uint32_t)g >> 16),
                     sigma, fx, fy, use_grad, row, rw);
        valid = true;
    }
    products29(row, rw, out);   // formed after the control flow has merged; a lane without a correspondence contributes zeros
    out[28] = valid ? 1.0f : 0.0f;
    block_reduce_exact<29, true>(out, valid, rgb_part);
}

// ---- row-sharded path: slot rows -> totals on every rank, all-reduce in between (launch_odometry)
__global__ __launch_bounds__(256) void k_fold_rows(long long *__restrict__ part, int width, long long *__restrict__ totals,
                                                   long long *__restrict__ also_zero, int n_zero)
{
    for (int col = threadIdx.x; col < width; col += blockDim.x) {
        long long t = 0;
        for (int b = 0; b < ODO_SLOTS; ++b) { t += part[(size_t)b * width + col]; part[(size_t)b * width + col] = 0; }
        totals[col] = t;
    }
    for (int k = threadIdx.x; k < n_zero; k += blockDim.x) also_zero[k] = 0;
}
__global__ void k_fold_residual_slots(long long *__restrict__ res_part, long long *__restrict__ totals2)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long c = 0, s = 0;
    for (int b = 0; b < RES_SLOTS; ++b) { c += res_part[b * 2]; s += res_part[b * 2 + 1]; res_part[b * 2] = 0; res_part[b * 2 + 1] = 0; }
    totals2[0] = c; totals2[1] = s;
}
__global__ void k_residual_to_slot0(const long long *__restrict__ totals2, long long *__restrict__ res_part)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    res_part[0] = totals2[0]; res_part[1] = totals2[1];   // the other slots are zero: k_gn_rgb_step folds the global sums
}

// stand-alone solve on totals that were summed elsewhere (row-sharded multi-GPU: all-reduce of the limb sums)
__global__ __launch_bounds__(256) void k_gn_solve(OdoState *st, long long *__restrict__ icp_part,
                                                   long long *__restrict__ rgb_part, long long *__restrict__ res_part,
                                                   long long *__restrict__ totals, int do_reduce, OdoConfig cfg,
                                                   int next_level, int level_changes, DevPose *dp, int weighting)
{
    gn_solve_block(st, icp_part, rgb_part, res_part, totals, do_reduce, totals[174], totals[175], cfg, next_level,
                   level_changes, dp);
    // last iteration of the registration: the frame's velocity weighting rides along (no separate launch); the multiplier is
    // the device word k_odo_downsample wrote this frame, NOT a launch argument: the captured graph stays valid whatever the caller passes
    if (threadIdx.x == 0 && dp && weighting) frame_weighting_state(dp, dp->frame_wmul);
}

// registration without any Gauss-Newton iteration configured: only the guard + publish
__global__ void k_odo_end(OdoState *st, DevPose *dp, OdoConfig cfg, int weighting)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    odo_end_state(st, dp, cfg);
    if (weighting) frame_weighting_state(dp, dp->frame_wmul);
}

// ------------------------------------------------------------------------------------------ pose bookkeeping
__global__ void k_pose_set(DevPose *dp, Rigid p, int also_prev)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    dp->pose = p;
    dp->tinv = rigid_inverse(p);
    if (also_prev) dp->prev = p;
}
__global__ void k_pose_commit_prev(DevPose *dp)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) dp->prev = dp->pose;
}

// HRBFFusion::rodrigues2 (HRBFFusion.cpp:2004-2050) without the JacobiSVD re-orthonormalisation
__device__ inline f3 rodrigues2(const float *R)
{
    double rx = (double)R[7] - (double)R[5], ry = (double)R[2] - (double)R[6], rz = (double)R[3] - (double)R[1];
    double s = hd_sqrt(((rx * rx + ry * ry) + rz * rz) * 0.25);
    double cth = ((double)((R[0] + R[4]) + R[8]) - 1.0) * 0.5;
    cth = cth > 1.0 ? 1.0 : (cth < -1.0 ? -1.0 : cth);
    double theta = hd_acos(cth);
    if (s < 1e-5) {
        if (cth > 0) rx = ry = rz = 0;
        else {
            double t;
            t = ((double)R[0] + 1.0) * 0.5; rx = hd_sqrt(t > 0.0 ? t : 0.0);
            t = ((double)R[4] + 1.0) * 0.5; ry = hd_sqrt(t > 0.0 ? t : 0.0) * (R[1] < 0 ? -1.0 : 1.0);
            t = ((double)R[8] + 1.0) * 0.5; rz = hd_sqrt(t > 0.0 ? t : 0.0) * (R[2] < 0 ? -1.0 : 1.0);
            double arx = rx < 0 ? -rx : rx, ary = ry < 0 ? -ry : ry, arz = rz < 0 ? -rz : rz;
            if (arx < ary && arx < arz && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= hd_sqrt((rx * rx + ry * ry) + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = 1.0 / (2.0 * s);
        vth *= theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    return mk3((float)rx, (float)ry, (float)rz);
}

// velocity weighting (HRBFFusion.cpp:1112-1123): diff = currPose^-1 * lastPose
__device__ inline void frame_weighting_state(DevPose *dp, float weight_multiplier);
__global__ void k_frame_weighting(DevPose *dp, float weight_multiplier)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    frame_weighting_state(dp, weight_multiplier);
}
// velocity weighting of the frame (HRBFFusion.cpp:1112-1123)
__device__ inline void frame_weighting_state(DevPose *dp, float weight_multiplier)
{
    const Rigid inv = dp->tinv, last = dp->prev;
    float Rm[9];
    mul3<float>(inv.r, last.r, Rm);
    f3 lt = m33_mul(inv.r, mk3(last.t[0], last.t[1], last.t[2]));
    f3 dt = mk3(lt.x + inv.t[0], lt.y + inv.t[1], lt.z + inv.t[2]);
    float a = len3(dt), b = len3(rodrigues2(Rm));
    float weighting = a > b ? a : b;
    const float largest = 0.01f, minWeight = 0.5f;
    if (weighting > largest) weighting = largest;
    float wv = 1.0f - (weighting / largest);
    dp->weighting = (wv > minWeight ? wv : minWeight) * weight_multiplier;
}

void launch_pose_set(hipStream_t s, DevPose *dp, const float p[16], int also_prev)
{
    Rigid r;
    for (int i = 0; i < 3; ++i) { for (int k = 0; k < 3; ++k) r.r[i * 3 + k] = p[k * 4 + i]; r.t[i] = p[12 + i]; }
    hipLaunchKernelGGL(k_pose_set, dim3(1), dim3(1), 0, s, dp, r, also_prev);
}
void launch_frame_epilogue(hipStream_t s, DevPose *dp, float weight_multiplier, int tracked)
{
    (void)tracked;
    hipLaunchKernelGGL(k_frame_weighting, dim3(1), dim3(1), 0, s, dp, weight_multiplier);
}
void launch_pose_commit_prev(hipStream_t s, DevPose *dp)
{
    hipLaunchKernelGGL(k_pose_commit_prev, dim3(1), dim3(1), 0, s, dp);
}

void launch_odo_first_rgb(hipStream_t s, const OdoBuffers &ob, const uint8_t *rgb)
{
    const OdoLevel &L0 = ob.lv[0];
    int P = L0.rows * L0.cols;
    hipLaunchKernelGGL(k_first_rgb, dim3((P + 255) / 256), dim3(256), 0, s, P, rgb, L0.last_next_image);
    for (int i = 0; i + 1 < HRBF_NUM_PYRS; ++i) {
        int n = ob.lv[i + 1].rows * ob.lv[i + 1].cols;
        hipLaunchKernelGGL(k_pyrdown_u8, dim3((n + 255) / 256), dim3(256), 0, s, ob.lv[i].last_next_image, ob.lv[i].rows,
                           ob.lv[i].cols, ob.lv[i + 1].last_next_image);
    }
}

static IcpArgs make_icp_args(const OdoLevel &L, const OdoConfig &cfg, int level)
{
    IcpArgs A;
    A.vmap_c = L.vmap_c; A.nmap_c = L.nmap_c; A.ck1_c = L.ck1_c; A.ck2_c = L.ck2_c;
    A.vmap_g = L.vmap_g; A.nmap_g = L.nmap_g; A.ck1_g = L.ck1_g; A.ck2_g = L.ck2_g; A.icpw = L.icpw;
    A.cur_tex = L.icp_cur; A.model_tex = L.icp_model;
    A.rows = L.rows; A.cols = L.cols;
    const int div = 1 << level;
    A.fx = cfg.fx / div; A.fy = cfg.fy / div; A.cx = cfg.cx / div; A.cy = cfg.cy / div;
    A.distThres = 0.1f; A.angleThres = 0.3420201433f;   // sin(20 deg), RGBDOdometry.h:65-66
    A.use_search = cfg.use_search; A.radius = cfg.search_radius; A.use_weight = cfg.use_weighted;
    A.sparse = cfg.use_sparse ? L.sparse : nullptr; A.sparse_first = 1;
    return A;
}

void launch_odometry(hipStream_t s, OdoBuffers &ob, const OdoSources &src, const OdoConfig &cfg, DevPose *dp,
                     const OdoComm *oc, float weight_multiplier, int level0_done)
{
    // sharded = the slot rows of this process do not hold the whole image: fold -> all-reduce -> stand-alone solve
    const bool sharded = oc != nullptr && (oc->allreduce_i64 != nullptr || oc->virtual_world > 1);
    const int vworld = sharded ? (oc->virtual_world > 1 ? oc->virtual_world : oc->world) : 1;
    const int vfirst = sharded && oc->virtual_world <= 1 ? oc->rank : 0;            // ranks this process plays
    const int vlast = sharded && oc->virtual_world <= 1 ? oc->rank + 1 : vworld;
    auto strip = [&](const OdoLevel &L, int r, int &p0, int &p1) {                  // rows [r, r+1) * rows / world
        p0 = (int)((long long)L.rows * r / vworld) * L.cols;
        p1 = (int)((long long)L.rows * (r + 1) / vworld) * L.cols;
    };
    auto allreduce = [&](long long *buf, size_t n) {
        if (sharded && oc->ar_count) { oc->ar_count[0] += 1; oc->ar_count[1] += sizeof(long long) * n; }
        if (sharded && oc->allreduce_i64 && oc->allreduce_i64(oc->ar_ctx, buf, n, s) != 0 && oc->ar_failed) *oc->ar_failed = 1;
    };
    const int rgb = cfg.rgb_only || cfg.icp_weight < 100.0f;
    const int icp = !cfg.rgb_only && cfg.icp_weight > 0.0f;
    const int P = ob.lv[0].rows * ob.lv[0].cols;
    // O1: pyramids
    if (!level0_done)
        hipLaunchKernelGGL(k_odo_level0, dim3((P + 255) / 256), dim3(256), 0, s, P, ob.lv[0], src, dp, cfg.frame_to_frame_rgb,
                           cfg.curv_thr);
    for (int i = 1; i < HRBF_NUM_PYRS; ++i) {
        int n = ob.lv[i].rows * ob.lv[i].cols;
        hipLaunchKernelGGL(k_odo_downsample, dim3((n + 255) / 256, ODO_DOWN_TASKS), dim3(256), 0, s, ob.lv[i - 1], ob.lv[i],
                           i == 1 ? dp : (DevPose *)nullptr, weight_multiplier);
    }
    // the slot rows (icp | rgb | res | so3) are zeroed once at allocation; every fold re-zeroes what it read
    int iterations[3] = {cfg.fast_odom ? 3 : 10, cfg.pyramid ? 5 : 0, cfg.pyramid ? 4 : 0};
    int first_level = -1, last_level = -1;
    for (int i = HRBF_NUM_PYRS - 1; i >= 0; --i) if (iterations[i] > 0) { first_level = i; break; }
    for (int i = 0; i < HRBF_NUM_PYRS; ++i) if (iterations[i] > 0) { last_level = i; break; }
    const int gn_level = first_level < 0 ? 0 : first_level;
    {
        OdoLevels all;
        for (int i = 0; i < HRBF_NUM_PYRS; ++i) all.lv[i] = ob.lv[i];
        // single-GPU path with the SO3 stage: the pixel work goes into the SO3 kernel as filler (see there)
        const bool filler = cfg.so3 && !sharded;
        if (filler)
            hipLaunchKernelGGL(k_odo_prepare, dim3((SO3_ITERS * ODO_SLOTS * 33 + 255) / 256, 1), dim3(256), 0, s, all, ob.state, dp,
                               cfg, rgb, -1, ob.so3_part, 0, 1);
        else
            hipLaunchKernelGGL(k_odo_prepare, dim3((P + 255) / 256, HRBF_NUM_PYRS), dim3(256), 0, s, all, ob.state, dp, cfg, rgb,
                               cfg.so3 ? -1 : gn_level, ob.so3_part, level0_done == 2 ? 1 : 0, 0);
    }
    // O2: SO3 pre-alignment on level 2: one persistent launch for all iterations (75 chunks at VGA).  A plain launch on
    // purpose: hipLaunchCooperativeKernel serialises against the whole device and cost 30 frames/s in the benchmark,
    // and the kernel does not need co-residency (see k_so3_persistent).  The row-sharded multi-GPU path keeps one
    // launch per iteration (the all-reduce sits between the reduction and the step).
    if (cfg.so3) {
        const OdoLevel &L = ob.lv[2];
        const int nb = (L.rows * L.cols + RB - 1) / RB;
        static long long capacity = -1;   // co-resident workgroups of the persistent kernel, probed once
        if (capacity < 0) {
            int dev = 0, per_cu = 0;
            hipDeviceProp_t prop;
            hipGetDevice(&dev);
            hipGetDeviceProperties(&prop, dev);
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_so3_persistent, RB, 0);
            capacity = per_cu > 0 ? (long long)per_cu * prop.multiProcessorCount : 0;
        }
        if (sharded) {
            for (int it = 0; it < SO3_ITERS; ++it) {
                for (int r = vfirst; r < vlast; ++r) {
                    int p0, p1; strip(L, r, p0, p1);
                    if (p1 > p0)
                        hipLaunchKernelGGL(k_so3_reduce, dim3((p1 - p0 + RB - 1) / RB), dim3(RB), 0, s, L, ob.state, ob.so3_part,
                                           ob.totals + 176, cfg, 0, -1, p0, p1);
                }
                hipLaunchKernelGGL(k_fold_rows, dim3(1), dim3(256), 0, s, ob.so3_part, 33, ob.totals + 176, (long long *)nullptr, 0);
                allreduce(ob.totals + 176, 33);
                hipLaunchKernelGGL(k_so3_solve, dim3(1), dim3(RB), 0, s, ob.state, ob.so3_part, ob.totals + 176, 0, cfg,
                                   it == SO3_ITERS - 1 ? gn_level : -1);
            }
        } else {   // any grid size is safe (ticketed chunks); one workgroup per chunk while the device can hold them all
            const long long grid = capacity > 0 && nb > capacity ? capacity : nb;
            So3Filler fill;
            unsigned int nfill = 0;
            for (int i = 0; i < HRBF_NUM_PYRS; ++i) fill.all.lv[i] = ob.lv[i];
            // filler ranges in workgroup units, level by level
            for (int lv = 0; lv < HRBF_NUM_PYRS; ++lv) {
                fill.first[lv] = nfill;
                nfill += (unsigned int)((ob.lv[lv].rows * ob.lv[lv].cols + RB - 1) / RB);
            }
            fill.first[HRBF_NUM_PYRS] = nfill;
            fill.dp = dp; fill.do_rgb = rgb; fill.level0_packed = level0_done == 2 ? 1 : 0;
            hipLaunchKernelGGL(k_so3_persistent, dim3((unsigned)grid + nfill), dim3(RB), 0, s, L, ob.state, ob.so3_part, cfg,
                               gn_level, (unsigned int)nb, (unsigned int)grid, fill);
        }
    }
    // O3-O6: coarse-to-fine Gauss-Newton, three launches per iteration.  On the single-GPU path the whole loop (57
    // launches whose arguments only depend on the configuration and on which of the two image-pointer parities is
    // current) is captured once into a hipGraph and replayed: one submission instead of 57.
    static const bool use_graph = getenv("HRBF_NO_GN_GRAPH") == nullptr;
    const int par = ob.swap_parity & 1;
    const int weighting = weight_multiplier >= 0.0f ? 1 : 0;   // < 0: a caller that does its own frame weighting (operator seams)
    bool replayed = false, capturing = false;
    if (use_graph && !sharded) {
        if (ob.gn_graph_exec[par] && memcmp(&ob.gn_graph_cfg[par], &cfg, sizeof(cfg)) == 0 &&
            ob.gn_graph_weighting[par] == weighting && ob.gn_graph_dp[par] == (void *)dp) {
            replayed = hipGraphLaunch((hipGraphExec_t)ob.gn_graph_exec[par], s) == hipSuccess;
        } else {
            if (ob.gn_graph_exec[par]) { hipGraphExecDestroy((hipGraphExec_t)ob.gn_graph_exec[par]); ob.gn_graph_exec[par] = nullptr; }
            capturing = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
        }
    }
    auto enqueue_gn = [&]() {
    const float minGrad[3] = {5, 3, 1};
    for (int i = HRBF_NUM_PYRS - 1; i >= 0; --i) {
        const OdoLevel &L = ob.lv[i];
        const int nb = (L.rows * L.cols + RB - 1) / RB;
        const int div = 1 << i;
        IcpArgs A = make_icp_args(L, cfg, i);
        const float minScale = (float)(((double)minGrad[i] * (double)minGrad[i]) / (0.125 * 0.125));
        const auto k_icp_res = A.sparse ? k_gn_icp_residual<true> : k_gn_icp_residual<false>;
        for (int j = 0; j < iterations[i]; ++j) {
            A.sparse_first = (j == 0);
            // operands for the next iteration: same level, or the next non-empty finer level
            const bool last_of_level = (j == iterations[i] - 1);
            int next_level = i;
            if (last_of_level) {
                next_level = -1;
                for (int k = i - 1; k >= 0; --k) if (iterations[k] > 0) { next_level = k; break; }
            }
            const bool last_of_all = last_of_level && i == last_level;
            if (sharded) {
                for (int r = vfirst; r < vlast; ++r) {
                    int p0, p1; strip(L, r, p0, p1);
                    const int nbr = (p1 - p0 + RB - 1) / RB;
                    if (nbr > 0)
                        hipLaunchKernelGGL(k_icp_res, dim3(2 * nbr), dim3(RB), 0, s, L, A, ob.state, nbr, icp, rgb, minScale,
                                           ob.icp_part, ob.res_part, ob.corres, ob.corres_diff, p0, p1);
                }
                // the robust weight needs the residual sums of the WHOLE image before any RGB product is formed
                hipLaunchKernelGGL(k_fold_residual_slots, dim3(1), dim3(1), 0, s, ob.res_part, ob.totals + 174);
                allreduce(ob.totals + 174, 2);
                hipLaunchKernelGGL(k_residual_to_slot0, dim3(1), dim3(1), 0, s, ob.totals + 174, ob.res_part);
                for (int r = vfirst; r < vlast; ++r) {
                    int p0, p1; strip(L, r, p0, p1);
                    const int nbr = (p1 - p0 + RB - 1) / RB;
                    if (nbr > 0)
                        hipLaunchKernelGGL(k_gn_rgb_step, dim3(nbr), dim3(RB), 0, s, L, ob.state, nbr, cfg.fx / div, cfg.fy / div,
                                           cfg.rgb_only, cfg.rgb_use_grad, ob.res_part, ob.corres, ob.corres_diff, ob.rgb_part,
                                           ob.totals, p0, p1);
                }
                // icp | rgb rows -> totals[0..173] (adjacent parts: one 174-wide view is NOT contiguous per row, fold each)
                hipLaunchKernelGGL(k_fold_rows, dim3(1), dim3(256), 0, s, ob.icp_part, 87, ob.totals, ob.res_part, RES_SLOTS * 2);
                hipLaunchKernelGGL(k_fold_rows, dim3(1), dim3(256), 0, s, ob.rgb_part, 87, ob.totals + 87, (long long *)nullptr, 0);
                allreduce(ob.totals, 174);
                hipLaunchKernelGGL(k_gn_solve, dim3(1), dim3(256), 0, s, ob.state, ob.icp_part, ob.rgb_part, ob.res_part,
                                   ob.totals, 0, cfg, next_level, last_of_level ? 1 : 0,
                                   last_of_all ? dp : (DevPose *)nullptr, weighting);
                continue;
            }
            hipLaunchKernelGGL(k_icp_res, dim3(2 * nb), dim3(RB), 0, s, L, A, ob.state, nb, icp, rgb, minScale,
                               ob.icp_part, ob.res_part, ob.corres, ob.corres_diff, 0, L.rows * L.cols);
            // fusing the solve into the last workgroup of k_gn_rgb_step was measured slower (the fold then reads the
            // slot rows from memory instead of L2 and pays a ticket round trip): 16.1 vs 6.3 + 8.8 us on level 2
            hipLaunchKernelGGL(k_gn_rgb_step, dim3(nb), dim3(RB), 0, s, L, ob.state, nb, cfg.fx / div, cfg.fy / div,
                               cfg.rgb_only, cfg.rgb_use_grad, ob.res_part, ob.corres, ob.corres_diff, ob.rgb_part,
                               ob.totals, 0, L.rows * L.cols);
            hipLaunchKernelGGL(k_gn_solve, dim3(1), dim3(256), 0, s, ob.state, ob.icp_part, ob.rgb_part, ob.res_part,
                               ob.totals, 1, cfg, next_level, last_of_level ? 1 : 0,
                               last_of_all ? dp : (DevPose *)nullptr, weighting);
        }
    }
    };
    if (capturing) {
        enqueue_gn();
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        const bool ok = hipStreamEndCapture(s, &g) == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess;
        if (g) hipGraphDestroy(g);
        if (ok) {
            ob.gn_graph_exec[par] = (void *)ge; ob.gn_graph_cfg[par] = cfg; ob.gn_graph_weighting[par] = weighting;
            ob.gn_graph_captures++;
            ob.gn_graph_dp[par] = (void *)dp;
            replayed = hipGraphLaunch(ge, s) == hipSuccess;
        } else (void)hipGetLastError();
    }
    if (!replayed) enqueue_gn();   // sharded path, graphs switched off, or capture / replay refused: plain launches
    if (last_level < 0) hipLaunchKernelGGL(k_odo_end, dim3(1), dim3(1), 0, s, ob.state, dp, cfg, weighting);
    if (cfg.so3)   // swap NextImage <-> lastNextImage (RGBDOdometry.cpp:1239-1245): pointer swap, no copy
    {
        for (int i = 0; i < HRBF_NUM_PYRS; ++i) { uint8_t *t = ob.lv[i].last_next_image; ob.lv[i].last_next_image = ob.lv[i].next_image; ob.lv[i].next_image = t; }
        ob.swap_parity ^= 1;
    }
}

// ------------------------------------------------------------------------------------------ icpStep seam
__global__ __launch_bounds__(RB) void k_icp_only(IcpArgs A, Rigid cur, Rigid prevInv_and_tprev, long long *__restrict__ part)
{
    float out[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) out[k] = 0.0f;
    bool valid = false;
    const int i = blockIdx.x * RB + threadIdx.x;
    if (i < A.rows * A.cols) {
        const int y = i / A.cols, x = i - y * A.cols;
        valid = icp_pixel(A, cur.r, mk3(cur.t[0], cur.t[1], cur.t[2]), prevInv_and_tprev.r,
                          mk3(prevInv_and_tprev.t[0], prevInv_and_tprev.t[1], prevInv_and_tprev.t[2]), x, y, out);
    }
    block_reduce_exact<29>(out, valid, part);
}

// the sparse (ADMM) form of the seam: lambdaMap in, z_thrinkMap and corresICP out (reduce.cu:455-492); all three are
// device images, lambda / z as 3 interleaved floats per pixel, corres as 2 int32 per pixel ((-1,-1) = no match)
__global__ __launch_bounds__(RB) void k_icp_only_sparse(IcpArgs A, Rigid cur, Rigid prevInv_and_tprev,
                                                        long long *__restrict__ part, const float *__restrict__ lambda3,
                                                        float *__restrict__ z3, int32_t *__restrict__ corres2)
{
    float out[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) out[k] = 0.0f;
    bool valid = false;
    const int i = blockIdx.x * RB + threadIdx.x;
    if (i < A.rows * A.cols) {
        const int y = i / A.cols, x = i - y * A.cols;
        SparseIo io;
        io.lambda = mk3(lambda3[3 * i], lambda3[3 * i + 1], lambda3[3 * i + 2]);
        io.z = mk3(0, 0, 0); io.bx = -1; io.by = -1;
        valid = icp_pixel<true>(A, cur.r, mk3(cur.t[0], cur.t[1], cur.t[2]), prevInv_and_tprev.r,
                                mk3(prevInv_and_tprev.t[0], prevInv_and_tprev.t[1], prevInv_and_tprev.t[2]), x, y, out, &io);
        z3[3 * i] = io.z.x; z3[3 * i + 1] = io.z.y; z3[3 * i + 2] = io.z.z;
        corres2[2 * i] = io.bx; corres2[2 * i + 1] = io.by;
    }
    block_reduce_exact<29>(out, valid, part);
}

// updateLambdaMapKernel (cudafuncs.cu:1030-1080) as its own launch, on caller-owned device images
__global__ void k_update_lambda(int rows, int cols, Rigid cur, Rigid prevInv_and_tprev, const float *__restrict__ vmap_c,
                                const float *__restrict__ vmap_g, const int32_t *__restrict__ corres2,
                                const float *__restrict__ z3, float *__restrict__ lambda3)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int ux = corres2[2 * i], uy = corres2[2 * i + 1];
    if (ux <= 0) return;   // the reference tests corresp.x > 0
    const int y = i / cols, x = i - y * cols;
    const f3 tcurr = mk3(cur.t[0], cur.t[1], cur.t[2]), tprev = mk3(prevInv_and_tprev.t[0], prevInv_and_tprev.t[1], prevInv_and_tprev.t[2]);
    const f3 vcur = mk3(PLN(vmap_c, 0, rows, cols, y, x), PLN(vmap_c, 1, rows, cols, y, x), PLN(vmap_c, 2, rows, cols, y, x));
    const f3 vlp = m33_mul(prevInv_and_tprev.r, sub3(add3(m33_mul(cur.r, vcur), tcurr), tprev));
    const f3 vp = m33_mul(prevInv_and_tprev.r, sub3(mk3(PLN(vmap_g, 0, rows, cols, uy, ux), PLN(vmap_g, 1, rows, cols, uy, ux),
                                                         PLN(vmap_g, 2, rows, cols, uy, ux)), tprev));
    const f3 d = sub3(sub3(vlp, vp), mk3(z3[3 * i], z3[3 * i + 1], z3[3 * i + 2]));
    lambda3[3 * i] = lambda3[3 * i] + HD_SPARSE_MU * d.x;
    lambda3[3 * i + 1] = lambda3[3 * i + 1] + HD_SPARSE_MU * d.y;
    lambda3[3 * i + 2] = lambda3[3 * i + 2] + HD_SPARSE_MU * d.z;
}

int run_update_lambda_map(hipStream_t s, const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                          const float Rprev_inv[9], const float tprev[3], const float *vmap_g_prev, const int32_t *corres,
                          const float *z_map, float *lambda_map, int rows, int cols)
{
    Rigid cur, prv;
    for (int k = 0; k < 9; ++k) { cur.r[k] = Rcurr[k]; prv.r[k] = Rprev_inv[k]; }
    for (int k = 0; k < 3; ++k) { cur.t[k] = tcurr[k]; prv.t[k] = tprev[k]; }
    const int n = rows * cols;
    hipLaunchKernelGGL(k_update_lambda, dim3((n + 255) / 256), dim3(256), 0, s, rows, cols, cur, prv, vmap_curr, vmap_g_prev,
                       corres, z_map, lambda_map);
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) { hrbf_set_error("update_lambda_map: %s", hipGetErrorString(e)); return HRBF_ERR_DEVICE; }
    return HRBF_OK;
}

int run_icp_step(hipStream_t s, const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                 const float *nmap_curr, const float *ck1_curr, const float *ck2_curr, const float Rprev_inv[9],
                 const float tprev[3], float fx, float fy, float cx, float cy, const float *vmap_g_prev,
                 const float *nmap_g_prev, const float *ck1_g_prev, const float *ck2_g_prev, const float *icpw, int rows,
                 int cols, float dist_thresh, float angle_thresh, int use_weight, double A_out[36], double b_out[6],
                 double residual_out[2], const float *lambda_map, float *z_map_out, int32_t *corres_out)
{
    IcpArgs A;
    A.vmap_c = vmap_curr; A.nmap_c = nmap_curr; A.ck1_c = ck1_curr; A.ck2_c = ck2_curr;
    A.vmap_g = vmap_g_prev; A.nmap_g = nmap_g_prev; A.ck1_g = ck1_g_prev; A.ck2_g = ck2_g_prev; A.icpw = icpw;
    A.cur_tex = nullptr; A.model_tex = nullptr;
    A.rows = rows; A.cols = cols; A.fx = fx; A.fy = fy; A.cx = cx; A.cy = cy;
    A.distThres = dist_thresh; A.angleThres = angle_thresh; A.use_search = 0; A.radius = 0; A.use_weight = use_weight;
    A.sparse = nullptr; A.sparse_first = 1;
    Rigid cur, prv;
    for (int k = 0; k < 9; ++k) { cur.r[k] = Rcurr[k]; prv.r[k] = Rprev_inv[k]; }
    for (int k = 0; k < 3; ++k) { cur.t[k] = tcurr[k]; prv.t[k] = tprev[k]; }
    const int nb = (rows * cols + RB - 1) / RB;
    long long *part = nullptr;
    const size_t bytes = sizeof(long long) * 87 * ODO_SLOTS;
    HIP_CHECK(hipMalloc(&part, bytes));
    hipError_t e = hipMemsetAsync(part, 0, bytes, s);
    if (lambda_map)
        hipLaunchKernelGGL(k_icp_only_sparse, dim3(nb), dim3(RB), 0, s, A, cur, prv, part, lambda_map, z_map_out, corres_out);
    else
        hipLaunchKernelGGL(k_icp_only, dim3(nb), dim3(RB), 0, s, A, cur, prv, part);
    long long *h = (long long *)malloc(bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(h, part, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    hipFree(part);
    if (e != hipSuccess) { free(h); hrbf_set_error("icp_step: %s", hipGetErrorString(e)); return HRBF_ERR_DEVICE; }
    long long tot[87];
    for (int t = 0; t < 87; ++t) tot[t] = 0;
    for (int b = 0; b < ODO_SLOTS; ++b) for (int t = 0; t < 87; ++t) tot[t] += h[(size_t)b * 87 + t];
    free(h);
    double sums[29];
    for (int i = 0; i < 29; ++i) sums[i] = hd_acc_to_double(hd_limbs_combine(tot[i * 3], tot[i * 3 + 1], tot[i * 3 + 2]));
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            double v = sums[shift++];
            if (j == 6) b_out[i] = v; else A_out[j * 6 + i] = A_out[i * 6 + j] = v;
        }
    residual_out[0] = sums[27]; residual_out[1] = sums[28];
    return HRBF_OK;
}

// ------------------------------------------------------------------------------------------ standalone seams
// so3Step / computeRgbResidual / rgbStep (cudafuncs.cuh:118-162) on caller-provided DEVICE images, like run_icp_step.
struct So3Operands { float basis[9], kinv[9], krlr[9]; };
struct RgbOperands { float krk[9], kt[3]; };

__global__ __launch_bounds__(RB) void k_so3_only(OdoLevel L, So3Operands op, long long *__restrict__ part)
{
    float row4[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) row4[k] = 0.0f;
    const bool valid = so3_pixel(L, &op, blockIdx.x * RB + threadIdx.x, row4);
    block_reduce_exact<11>(row4, valid, part);
}

__global__ __launch_bounds__(RB) void k_rgb_residual_only(OdoLevel L, RgbOperands op, float minScale,
                                                          int16_t *__restrict__ corres, float *__restrict__ corres_diff,
                                                          unsigned long long *__restrict__ count_sigma)
{
    __shared__ long long s_c[RB / 64], s_s[RB / 64];
    const int k = blockIdx.x * RB + threadIdx.x;
    long long cnt = 0, sig = 0;
    if (k < L.rows * L.cols) {
        const RgbCorr r = rgb_residual_pixel(L, &op, minScale, k, cnt, sig);
        int16_t *co = &corres[(size_t)k * 6];
        co[0] = r.c0; co[1] = r.c1; co[2] = r.c2; co[3] = r.c3; co[4] = r.c4; co[5] = 0;
        corres_diff[k] = r.diff;
    }
    for (int d = 32; d > 0; d >>= 1) { cnt += __shfl_down(cnt, d); sig += __shfl_down(sig, d); }
    if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = cnt; s_s[threadIdx.x >> 6] = sig; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long c = 0, s = 0;
        for (int w = 0; w < RB / 64; ++w) { c += s_c[w]; s += s_s[w]; }
        if (c) { atomicAdd(&count_sigma[0], (unsigned long long)c); atomicAdd(&count_sigma[1], (unsigned long long)s); }
    }
}

__global__ __launch_bounds__(RB) void k_rgb_only(OdoLevel L, const int16_t *__restrict__ corres,
                                                 const float *__restrict__ corres_diff, float sigma, float fx, float fy,
                                                 int use_grad, long long *__restrict__ part)
{
    float out[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) out[k] = 0.0f;
    bool valid = false;
    const int k = blockIdx.x * RB + threadIdx.x;
    if (k < L.rows * L.cols) {
        const int16_t *co = &corres[(size_t)k * 6];
        RgbCorr r; r.c0 = co[0]; r.c1 = co[1]; r.c2 = co[2]; r.c3 = co[3]; r.c4 = co[4]; r.diff = corres_diff[k];
        valid = rgb_products_pixel(L, r, sigma, fx, fy, use_grad, out);
    }
    block_reduce_exact<29>(out, valid, part);
}

// slot rows -> exact sums on the host
static int fetch_sums(hipStream_t s, long long *d_part, int nvals, double *sums, const char *what)
{
    const int width = nvals * 3;
    const size_t bytes = sizeof(long long) * (size_t)width * ODO_SLOTS;
    long long *h = (long long *)malloc(bytes);
    hipError_t e = hipMemcpyAsync(h, d_part, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { free(h); hrbf_set_error("%s: %s", what, hipGetErrorString(e)); return HRBF_ERR_DEVICE; }
    for (int i = 0; i < nvals; ++i) {
        long long t[3] = {0, 0, 0};
        for (int b = 0; b < ODO_SLOTS; ++b) for (int l = 0; l < 3; ++l) t[l] += h[(size_t)b * width + i * 3 + l];
        sums[i] = hd_acc_to_double(hd_limbs_combine(t[0], t[1], t[2]));
    }
    free(h);
    return HRBF_OK;
}

int run_so3_step(hipStream_t s, const uint8_t *last_image, const uint8_t *next_image, int rows, int cols,
                 const float basis[9], const float kinv[9], const float krlr[9], double A_out[9], double b_out[3],
                 double residual_out[2])
{
    OdoLevel L; memset(&L, 0, sizeof(L));
    L.rows = rows; L.cols = cols; L.last_next_image = (uint8_t *)last_image; L.next_image = (uint8_t *)next_image;
    So3Operands op;
    for (int k = 0; k < 9; ++k) { op.basis[k] = basis[k]; op.kinv[k] = kinv[k]; op.krlr[k] = krlr[k]; }
    long long *part = nullptr;
    const size_t bytes = sizeof(long long) * 33 * ODO_SLOTS;
    HIP_CHECK(hipMalloc(&part, bytes));
    hipMemsetAsync(part, 0, bytes, s);
    hipLaunchKernelGGL(k_so3_only, dim3((rows * cols + RB - 1) / RB), dim3(RB), 0, s, L, op, part);
    double sums[11];
    const int r = fetch_sums(s, part, 11, sums, "so3_step");
    hipFree(part);
    if (r) return r;
    int shift = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 4; ++j) {
            const double v = sums[shift++];
            if (j == 3) b_out[i] = v; else A_out[j * 3 + i] = A_out[i * 3 + j] = v;
        }
    residual_out[0] = sums[9]; residual_out[1] = sums[10];
    return HRBF_OK;
}

int run_rgb_residual(hipStream_t s, float min_scale, const int16_t *dIdx, const int16_t *dIdy, const float *last_depth,
                     const float *next_depth, const uint8_t *last_image, const uint8_t *next_image, int rows, int cols,
                     const float kt[3], const float krkinv[9], int16_t *corres_out, float *diff_out, long long *count,
                     long long *sigma)
{
    OdoLevel L; memset(&L, 0, sizeof(L));
    L.rows = rows; L.cols = cols; L.dIdx = (int16_t *)dIdx; L.dIdy = (int16_t *)dIdy;
    L.last_depth = (float *)last_depth; L.next_depth = (float *)next_depth;
    L.last_image = (uint8_t *)last_image; L.next_image = (uint8_t *)next_image;
    RgbOperands op;
    for (int k = 0; k < 9; ++k) op.krk[k] = krkinv[k];
    for (int k = 0; k < 3; ++k) op.kt[k] = kt[k];
    unsigned long long *cs = nullptr;
    HIP_CHECK(hipMalloc(&cs, 16));
    hipMemsetAsync(cs, 0, 16, s);
    hipLaunchKernelGGL(k_rgb_residual_only, dim3((rows * cols + RB - 1) / RB), dim3(RB), 0, s, L, op, min_scale, corres_out,
                       diff_out, cs);
    unsigned long long h[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(h, cs, 16, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    hipFree(cs);
    if (e != hipSuccess) { hrbf_set_error("rgb_residual: %s", hipGetErrorString(e)); return HRBF_ERR_DEVICE; }
    *count = (long long)h[0]; *sigma = (long long)h[1];
    return HRBF_OK;
}

int run_rgb_step(hipStream_t s, const int16_t *corres, const float *corres_diff, float sigma, const float *cloud, float fx,
                 float fy, const int16_t *dIdx, const int16_t *dIdy, int use_grad_weight, int rows, int cols,
                 double A_out[36], double b_out[6], double residual_out[2])
{
    OdoLevel L; memset(&L, 0, sizeof(L));
    L.rows = rows; L.cols = cols; L.dIdx = (int16_t *)dIdx; L.dIdy = (int16_t *)dIdy; L.cloud = (float *)cloud;
    long long *part = nullptr;
    const size_t bytes = sizeof(long long) * 87 * ODO_SLOTS;
    HIP_CHECK(hipMalloc(&part, bytes));
    hipMemsetAsync(part, 0, bytes, s);
    hipLaunchKernelGGL(k_rgb_only, dim3((rows * cols + RB - 1) / RB), dim3(RB), 0, s, L, corres, corres_diff, sigma, fx, fy,
                       use_grad_weight, part);
    double sums[29];
    const int r = fetch_sums(s, part, 29, sums, "rgb_step");
    hipFree(part);
    if (r) return r;
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const double v = sums[shift++];
            if (j == 6) b_out[i] = v; else A_out[j * 6 + i] = A_out[i * 6 + j] = v;
        }
    residual_out[0] = sums[27]; residual_out[1] = sums[28];
    return HRBF_OK;
}

// sticky count of frames whose SO3 kernel ran into its poll bound (hrbf_get_status); synchronises the stream
int odo_read_timeouts(hipStream_t s, OdoState *st, int clear)
{
    int v[2] = {0, 0};
    if (hipMemcpyAsync(v, &st->bar_timeout, sizeof(v), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    if (clear) { const int z[2] = {0, 0}; hipMemcpyAsync(&st->bar_timeout, z, sizeof(z), hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }
    return v[0] + v[1];
}

// ------------------------------------------------------------------------------------------ measurement probe
// The per-iteration pixel work of ONE pyramid level done by ONE workgroup (the "single-workgroup Gauss-Newton loop"
// question): the ICP pixel + exact reduction over all pixels, then the RGB products + exact reduction
// over the correspondences the last frame left behind.  Same pixel functions, same limbs; scratch slot rows.  Returns
// the kernel time in ms of `iters` such passes (one launch), for comparison with the three launches of an iteration.
__global__ __launch_bounds__(RB) void k_probe_single_wg_iteration(OdoLevel L, IcpArgs A, const OdoState *__restrict__ st, float fx,
                                                                  float fy, int use_grad, const int16_t *__restrict__ corres,
                                                                  long long *__restrict__ scratch, int iters)
{
    const int n = L.rows * L.cols;
    for (int it = 0; it < iters; ++it) {
        for (int base = 0; base < n; base += RB) {
            float out[29];
#pragma unroll
            for (int k = 0; k < 29; ++k) out[k] = 0.0f;
            bool valid = false;
            const int i = base + threadIdx.x;
            if (i < n) {
                const int y = i / A.cols, x = i - y * A.cols;
                valid = icp_pixel(A, st->Rcurr, mk3(st->tcurr[0], st->tcurr[1], st->tcurr[2]), st->Rprev_inv,
                                  mk3(st->tprev[0], st->tprev[1], st->tprev[2]), x, y, out);
            }
            block_reduce_exact<29>(out, valid, scratch);
            __syncthreads();
        }
        for (int base = 0; base < n; base += RB) {
            float out[29];
#pragma unroll
            for (int k = 0; k < 29; ++k) out[k] = 0.0f;
            bool valid = false;
            const int k = base + threadIdx.x;
            if (k < n) {
                const int2 rec = reinterpret_cast<const int2 *>(corres)[k];
                if (rec.x != -1) {
                    // the records were left by the last level-0 iteration: fold their coordinates into this level (timing only)
                    const int u0 = (rec.x & 0xffff) % L.cols, v0 = (int)((uint32_t)rec.x >> 16) % L.rows;
                    {
                        const float4 cp = L.cloud4[(size_t)v0 * L.cols + u0];
                        const int g = L.dIxy[k];
                        rgb_products_core(__int_as_float(rec.y), cp.x, cp.y, cp.z, (int)(int16_t)(g & 0xffff), (int)(int16_t)((uint32_t)g >> 16),
                                          10.0f, fx, fy, use_grad, out);
                        valid = true;
                    }
                }
            }
            block_reduce_exact<29>(out, valid, scratch + ODO_SLOTS * 87);
            __syncthreads();
        }
    }
}

int odo_probe_single_wg(hipStream_t s, OdoBuffers &ob, const OdoConfig &cfg, int level, int iters, float *ms_out)
{
    if (level < 0 || level >= HRBF_NUM_PYRS || iters < 1 || !ms_out) return HRBF_ERR_INVALID;
    const OdoLevel &L = ob.lv[level];
    IcpArgs A = make_icp_args(L, cfg, level);
    long long *scratch = nullptr;
    HIP_CHECK(hipMalloc(&scratch, sizeof(long long) * 87 * ODO_SLOTS * 2));
    HIP_CHECK(hipMemsetAsync(scratch, 0, sizeof(long long) * 87 * ODO_SLOTS * 2, s));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int div = 1 << level;
    // the correspondence records of the LAST frame's final level-0 iteration are laid out for level 0; any level's pixel
    // count indexes a valid prefix of them, which is all a timing probe needs (indices are range-checked above)
    hipLaunchKernelGGL(k_probe_single_wg_iteration, dim3(1), dim3(RB), 0, s, L, A, ob.state, cfg.fx / div, cfg.fy / div, cfg.rgb_use_grad,
                       ob.corres, scratch, 1);   // warm-up (code + L2)
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(k_probe_single_wg_iteration, dim3(1), dim3(RB), 0, s, L, A, ob.state, cfg.fx / div, cfg.fy / div, cfg.rgb_use_grad,
                       ob.corres, scratch, iters);
    hipEventRecord(e1, s);
    hipError_t e = hipStreamSynchronize(s);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(scratch);
    if (e != hipSuccess) { hrbf_set_error("probe: %s", hipGetErrorString(e)); return HRBF_ERR_DEVICE; }
    *ms_out = ms;
    return HRBF_OK;
}

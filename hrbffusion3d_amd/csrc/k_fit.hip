// k_fit.hip — OPTIONAL extension, not part of processFrame: a true Hermite-RBF fit per pixel, the "batched-HRBF small-GEMM on
// MFMA" BASELINE config 5 names.  NO counterpart in the reference: hrbfbase.glsl:132,153,173 use the closed-form coefficients
// 10 * n_i where a Hermite-RBF interpolant solves a (4k x 4k) symmetric positive definite system over its k centres.  The default
// path (k_curvature) is untouched; this operator (hrbf_fit_curvature) is held to a float64 numpy statement of the same algorithm
// (oracle/hrbf_fit_ref.py) within a tolerance and to analytic plane / sphere / cylinder answers (tests/test_hrbf_fit.py).
//
// One WAVE per pixel (64-thread workgroups, no cross-wave hand-over), everything out of LDS:
//   gather   the valid pixels of the (2w+1)^2 window (w <= 2: k <= 25 centres), common support rho = support * max |p_q - p_c|,
//            dimensionless coordinates u = (p - p_c) / rho
//   assemble the 4k x 4k system of Wendland-C4 blocks [[psi, -grad psi^T], [grad psi, -H psi]](u_i - u_j) + ridge, padded with an
//            identity to 112 = 7 x 7 blocks of 16 x 16 (3 x 3 for the 3 x 3 window), lower triangle only (28 blocks, rows padded to
//            17 floats: bank-conflict-free for the MFMA operand reads; 31 KB: five waves per CU); the right-hand side rides as the
//            LAST ROW of the matrix, so the forward substitution L y = b is done by the factorisation itself
//   factor   right-looking blocked Cholesky.  A lane owns one ROW (16 registers): lanes 0..15 the diagonal block's rows, lanes
//            16..63 the rows of three panel blocks; column j of all of them is the same recurrence s = a_ij - sum_{t<j} l_it l_jt,
//            l_jt broadcast from lane j by v_readlane_b32 — the panel's triangular solve costs nothing beside the factorisation.
//            Trailing updates C(ib, jb) -= X(ib) X(jb)^T are 16 x 16 x 16 products on v_mfma_f32_16x16x4_f32 (exact f32:
//            bit-for-bit a k-ordered fmaf chain), accumulators loaded from / stored to LDS in the C/D register layout.
//            (First version: 32 x 32 blocks on v_mfma_f32_32x32x2_f32, 43 KB and three waves per CU: 26.7 ms per 640 x 480 frame —
//            small blocks move the flops from the lanes' recurrences into the matrix core and fit more systems on a CU.)
//   read out the gradient (3) and Hessian (6) of the interpolant at the pixel are LINEAR functionals e_m of the coefficients; their
//            rows ride in the matrix below the right-hand side, so e_m^T A^-1 b = (L^-1 e_m) . (L^-1 b) is computed by the
//            factorisation's own trailing updates and stands in L[RB + 1 + m][RB] afterwards: no back substitution (it was 22 %
//            of the kernel), then the shape operator, principal curvatures + directions on one lane
#include "common.h"
#include "kernels.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

#define FIT_B 16                        // block edge (v_mfma_f32_16x16x4_f32)
#define FIT_LD 17                       // floats per block row in LDS (odd: the MFMA operand reads of 16 lanes hit 16 banks)
#define FIT_BLK (FIT_B * FIT_LD)        // floats per block
#define FIT_MAXK 25
#define FIT_MIN_CENTRES 8
#define FIT_SENTINEL 1000.0f

__device__ __forceinline__ int fit_blk(int I, int J) { return (I * (I + 1) / 2 + J) * FIT_BLK; }
__device__ __forceinline__ float &fit_at(float *A, int r, int c)   // element (r, c), r >= c
{
    return A[fit_blk(r >> 4, c >> 4) + (r & 15) * FIT_LD + (c & 15)];
}
__device__ __forceinline__ float fit_readlane(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// Wendland C4: phi(r) = (1 - r)^6 (35 r^2 + 18 r + 3) / 3;  F = phi'/r, G = F'/r, K = G'/r
__device__ __forceinline__ void fit_wendland(float r, float &phi, float &F, float &G, float &K)
{
    if (!(r < 1.0f)) { phi = F = G = K = 0.0f; return; }
    const float t = 1.0f - r, t2 = t * t, t3 = t2 * t, t4 = t2 * t2;
    phi = t4 * t2 * (35.0f * r * r + 18.0f * r + 3.0f) * (1.0f / 3.0f);
    F = -(56.0f / 3.0f) * t4 * t * (5.0f * r + 1.0f);
    G = 560.0f * t4;
    K = r > 0.0f ? -2240.0f * t3 / r : 0.0f;
}

// Columns 0..15 of up to four block rows at once: a lane owns one ROW (16 registers); lanes 0..15 hold the rows of the diagonal
// block, lanes 16..63 the rows of three panel blocks.  FACTOR: the diagonal block is being factored in the same sweep; otherwise
// lanes 0..15 hold its finished rows and only broadcast.  Column j of every row is the same recurrence
// s = a_ij - sum_{t<j} l_it l_jt, l_jt broadcast from lane j (v_readlane_b32): the panel's triangular solve costs nothing beside
// the factorisation.  Four columns advance together — four INDEPENDENT chains over t < j0 (a wave has at most one partner on its
// SIMD to hide a dependent chain behind) — and are then finished one by one.  Each element takes its terms in ascending t.
template <bool FACTOR>
__device__ __forceinline__ void fit_finish(float (&row)[FIT_B], int lane, int j, float s)
{
    // v_rsq_f32 / v_rcp_f32 (1 ulp) instead of the IEEE sqrt + division (~25 instructions per column): this operator is held to a
    // tolerance against a float64 statement, not to bits
    float inv;
    if (FACTOR) inv = __builtin_amdgcn_rsqf(fit_readlane(s, j));
    else inv = __builtin_amdgcn_rcpf(fit_readlane(row[j], j));
    const float val = s * inv;
    if (FACTOR) { if (lane >= j) row[j] = val; }       // lane j: pivot / sqrt(pivot) = l_jj
    else if (lane >= FIT_B) row[j] = val;
}
template <bool FACTOR>
__device__ __forceinline__ void fit_columns(float (&row)[FIT_B], int lane)
{
#pragma unroll
    for (int j0 = 0; j0 < FIT_B; j0 += 4) {
        float s[4] = {row[j0], row[j0 + 1], row[j0 + 2], row[j0 + 3]};
#pragma unroll
        for (int t = 0; t < j0; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] = fmaf(-row[t], fit_readlane(row[t], j0 + q), s[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int t = j0; t < j0 + q; ++t) s[q] = fmaf(-row[t], fit_readlane(row[t], j0 + q), s[q]);
            fit_finish<FACTOR>(row, lane, j0 + q, s[q]);
        }
    }
}

__device__ __forceinline__ void fit_load_row(const float *blk, int lane, bool lower_only, float (&row)[FIT_B])
{
    const int r = lane & 15;
#pragma unroll
    for (int c = 0; c < FIT_B; ++c) row[c] = (!lower_only || c <= r) ? blk[r * FIT_LD + c] : 0.0f;
}
__device__ __forceinline__ void fit_store_row(float *blk, int lane, const float (&row)[FIT_B])
{
    const int r = lane & 15;
#pragma unroll
    for (int c = 0; c < FIT_B; ++c) blk[r * FIT_LD + c] = row[c];
}

// C(ib, jb .. jb + N - 1) -= X(ib, kb) * X(jb .., kb)^T on the matrix core: N x 4 v_mfma_f32_16x16x4_f32 (exact f32: a k-ordered
// fmaf chain).  N blocks of one block row at a time: X(ib) is read once, the N dependent accumulator chains (40 cycles per link,
// 32 per issue) interleave, and the LDS round trips in front of and behind the MFMAs are paid once per N blocks (measured: 440
// cycles per block update in pairs).  Requesting block jb + 1's operands before block jb's MFMAs — ONE chain, software-pipelined
// over a block row — measured slower (580).
template <int N>
__device__ __forceinline__ void fit_update_n(float *A, int ib, int jb, int kb, int lane)
{
    const float *Xi = A + fit_blk(ib, kb);
    const int col = lane & 15, quad = lane >> 4;
    floatx4 acc[N];
    float a[4], b[N][4];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float *C = A + fit_blk(ib, jb + n), *Xj = A + fit_blk(jb + n, kb);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[n][r] = C[(4 * quad + r) * FIT_LD + col];      // C/D: row = 4 * (lane >> 4) + reg, col = lane & 15
#pragma unroll
        for (int s = 0; s < 4; ++s) b[n][s] = Xj[col * FIT_LD + 4 * s + quad];         // B[k = lane >> 4][j = lane & 15] = X(jb)[j][k]
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = -Xi[col * FIT_LD + 4 * s + quad];               // A[i = lane & 15][k = lane >> 4]
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[n][s], acc[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < N; ++n) {
        float *C = A + fit_blk(ib, jb + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(4 * quad + r) * FIT_LD + col] = acc[n][r];
    }
}
__device__ __forceinline__ void fit_update_row(float *A, int ib, int kb, int lane)
{
    int jb = kb + 1;
    for (; jb + 3 <= ib; jb += 4) fit_update_n<4>(A, ib, jb, kb, lane);
    const int left = ib - jb + 1;
    if (left == 3) fit_update_n<3>(A, ib, jb, kb, lane);
    else if (left == 2) fit_update_n<2>(A, ib, jb, kb, lane);
    else if (left == 1) fit_update_n<1>(A, ib, jb, kb, lane);
}

// Panel blocks beyond the two that ride in the factor sweep: X(ib) = P(ib) * W with W = L(kb, kb)^-T, which the sweep produced as
// the "panel solve" of an identity block (X L^T = I) and left in the diagonal block's place (with the functional rows there is no
// back substitution, so L(kb, kb) itself is never read again).  In place, N blocks at a time, W read once.
template <int N>
__device__ __forceinline__ void fit_trsm_n(float *A, int ib, int kb, int lane)
{
    const float *Wb = A + fit_blk(kb, kb);
    const int col = lane & 15, quad = lane >> 4;
    floatx4 acc[N];
    float a[N][4], b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) b[s] = Wb[(4 * s + quad) * FIT_LD + col];              // B[k][j] = W[k][j]
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float *P = A + fit_blk(ib + n, kb);
#pragma unroll
        for (int s = 0; s < 4; ++s) a[n][s] = P[col * FIT_LD + 4 * s + quad];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[n][r] = 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[n][s], b[s], acc[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < N; ++n) {
        float *P = A + fit_blk(ib + n, kb);
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(4 * quad + r) * FIT_LD + col] = acc[n][r];
    }
}

#ifdef FIT_TIMING   // measurement build: cycles per phase and wave, plain stores (tools/probes/fit_phases.py sums them)
#define FIT_TIMING_WAVES (640 * 480)
__device__ unsigned int g_fit_phase_w[FIT_TIMING_WAVES * 8];
extern "C" int hrbf_probe_fit_phases(unsigned int *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fit_phase_w), sizeof(unsigned int) * FIT_TIMING_WAVES * 8) != hipSuccess) return -1;
    if (reset) { void *p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fit_phase_w)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned int) * FIT_TIMING_WAVES * 8) != hipSuccess) return -1; }
    return 0;
}
#define FIT_T(i) do { const unsigned long long t_now = __builtin_readcyclecounter(); if (lane == 0 && pi < FIT_TIMING_WAVES) g_fit_phase_w[pi * 8 + (i)] += (unsigned int)(t_now - t_last); t_last = t_now; } while (0)
#else
#define FIT_T(i)
#endif

// NBR block rows of 16: 7 for the 5 x 5 window (4 * 25 unknowns + the right-hand-side row <= 112), 3 for the 3 x 3 window (<= 48)
template <int NBR>
__global__ __launch_bounds__(64) void k_hrbf_fit(Cam cam, const float4 *__restrict__ vertex, const float4 *__restrict__ normal,
                                                 int w, float support, float ridge, float jump, float4 *__restrict__ out_c1,
                                                 float4 *__restrict__ out_c2, float4 *__restrict__ out_n)
{
    // rows: [0, 4k) the unknowns, identity padding up to RB, row RB the right-hand side b, rows RB + 1 .. RB + 9 the nine linear
    // functionals that read the interpolant's gradient (3) and Hessian (6) at the pixel, two rows of padding.  After the
    // factorisation row RB of L is y = L^-1 b and row RB + 1 + m is z_m = L^-1 e_m, so the wanted values q_m = e_m^T A^-1 b =
    // z_m . y appear — with a minus sign, divided by the pivot of column RB — at L[RB + 1 + m][RB]: the factorisation's own
    // trailing updates compute them and NO back substitution is needed (it was 22 % of the kernel: dependent steps on 16 lanes).
    constexpr int NBLK = NBR * (NBR + 1) / 2, NPAD = NBR * FIT_B, RB = NPAD - 12;
    __shared__ __attribute__((aligned(16))) float A[NBLK * FIT_BLK];
    __shared__ float s_u[FIT_MAXK][3], s_n[FIT_MAXK][3];
    const int lane = threadIdx.x, pi = blockIdx.x;
    const int W = cam.W, H = cam.H;
    const int px = pi % W, py = pi / W;
    const int side = 2 * w + 1, nwin = side * side, tc = nwin / 2;
#ifdef FIT_TIMING
    unsigned long long t_last = __builtin_readcyclecounter();
#endif

    // ---- gather
    float4 v = make_float4(0, 0, 0, 0), nn = make_float4(0, 0, 0, 0);
    bool ok = false;
    if (lane < nwin) {
        const int qx = px + lane % side - w, qy = py + lane / side - w;
        if (qx >= 0 && qy >= 0 && qx < W && qy < H) {
            v = vertex[qy * W + qx]; nn = normal[qy * W + qx];
            const float nl = sqrtf((nn.x * nn.x + nn.y * nn.y) + nn.z * nn.z);
            ok = v.z > 0.0f && isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(nl) && nl > 0.5f;
        }
    }
    {   // the matrix is zeroed while the window's loads are in flight
        static_assert((NBLK * FIT_BLK) % 4 == 0, "block storage is a multiple of 16 bytes");
        float4 *A4 = reinterpret_cast<float4 *>(A);
        for (int i = lane; i < NBLK * FIT_BLK / 4; i += 64) A4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const float pcx = fit_readlane(v.x, tc), pcy = fit_readlane(v.y, tc), pcz = fit_readlane(v.z, tc);
    const bool okc = (__ballot(ok) >> tc) & 1ull;
    const float dx = v.x - pcx, dy = v.y - pcy, dz = v.z - pcz;
    const float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
    ok = ok && dist <= jump * (float)w * pcz * cam.camz;
    const unsigned long long mask = __ballot(ok);
    const int k = __popcll(mask);
    float maxd = ok ? dist : 0.0f;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) maxd = fmaxf(maxd, __shfl_xor(maxd, d));
    const float rho = support * maxd;
    if (!okc || k < FIT_MIN_CENTRES || !(rho > 0.0f)) {
        if (lane == 0) {
            out_c1[pi] = make_float4(0, 0, 0, FIT_SENTINEL); out_c2[pi] = make_float4(0, 0, 0, FIT_SENTINEL);
            out_n[pi] = make_float4(0, 0, 0, 0);
        }
        return;
    }
    const int slot = __popcll(mask & ((1ull << lane) - 1ull));
    if (ok) {
        s_u[slot][0] = dx / rho; s_u[slot][1] = dy / rho; s_u[slot][2] = dz / rho;
        s_n[slot][0] = nn.x; s_n[slot][1] = nn.y; s_n[slot][2] = nn.z;
    }
    __syncthreads();
    FIT_T(0);

    // ---- assemble (lower triangle), identity padding, the right-hand side as the last row
    const int n4 = 4 * k;
    for (int r = n4 + lane; r < RB; r += 64) fit_at(A, r, r) = 1.0f;
    if (lane < 12) fit_at(A, RB + lane, RB + lane) = lane < 10 ? 1.0e30f : 1.0f;   // pivots of the extra rows: large, so that they stay positive
    for (int c = lane; c < n4; c += 64) fit_at(A, RB, c) = (c & 3) ? s_n[c >> 2][(c & 3) - 1] : 0.0f;
    if (lane < k) {   // the nine functionals at u = 0: centre `lane` contributes its four columns to each
        const int j = lane;
        const float d[3] = {-s_u[j][0], -s_u[j][1], -s_u[j][2]};
        const float r = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        float phi, F, G, K;
        fit_wendland(r, phi, F, G, K);
#pragma unroll
        for (int a = 0; a < 3; ++a) {          // gradient component a
            fit_at(A, RB + 1 + a, 4 * j) = F * d[a];
#pragma unroll
            for (int b = 0; b < 3; ++b) fit_at(A, RB + 1 + a, 4 * j + 1 + b) = -((a == b ? F : 0.0f) + G * d[a] * d[b]);
        }
        int m = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = a; b < 3; ++b) {      // Hessian entry (a, b): xx xy xz yy yz zz
                const float dl = a == b ? 1.0f : 0.0f;
                fit_at(A, RB + 4 + m, 4 * j) = F * dl + G * d[a] * d[b];
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    fit_at(A, RB + 4 + m, 4 * j + 1 + c) = -(G * (d[c] * dl + (a == c ? d[b] : 0.0f) + (b == c ? d[a] : 0.0f)) + K * d[a] * d[b] * d[c]);
                ++m;
            }
    }
    const int npairs = k * (k + 1) / 2;
    for (int p = lane; p < npairs; p += 64) {
        int i = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
        while (i * (i + 1) / 2 > p) --i;
        while ((i + 1) * (i + 2) / 2 <= p) ++i;
        const int j = p - i * (i + 1) / 2;
        if (i == j) {
            fit_at(A, 4 * i, 4 * i) = 1.0f + ridge;
#pragma unroll
            for (int a = 1; a < 4; ++a) fit_at(A, 4 * i + a, 4 * i + a) = 56.0f / 3.0f + ridge;
            continue;
        }
        const float d[3] = {s_u[i][0] - s_u[j][0], s_u[i][1] - s_u[j][1], s_u[i][2] - s_u[j][2]};
        const float r = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        float phi, F, G, K;
        fit_wendland(r, phi, F, G, K);
        fit_at(A, 4 * i, 4 * j) = phi;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            fit_at(A, 4 * i, 4 * j + 1 + a) = -F * d[a];
            fit_at(A, 4 * i + 1 + a, 4 * j) = F * d[a];
#pragma unroll
            for (int b = 0; b < 3; ++b) fit_at(A, 4 * i + 1 + a, 4 * j + 1 + b) = -((a == b ? F : 0.0f) + G * d[a] * d[b]);
        }
    }
    __syncthreads();
    FIT_T(1);

    // ---- blocked Cholesky, right-looking over block columns
    const int grp = lane >> 4;
    for (int kb = 0; kb < NBR; ++kb) {
        float row[FIT_B];
        const int nb = NBR - 1 - kb;                     // panel blocks below the diagonal block
        if (nb <= 3) {   // the diagonal block is factored while all panel blocks below it are solved in the same sweep
            const int ib = kb + grp;
            if (grp == 0) fit_load_row(A + fit_blk(kb, kb), lane, true, row);
            else if (ib < NBR) fit_load_row(A + fit_blk(ib, kb), lane, false, row);
            else {
#pragma unroll
                for (int c = 0; c < FIT_B; ++c) row[c] = 0.0f;
            }
            fit_columns<true>(row, lane);
            if (ib < NBR) fit_store_row(A + fit_blk(ib, kb), lane, row);
            __syncthreads();
        } else {         // more panels than lanes: lanes 16..31 solve an IDENTITY block (-> W = L^-T), lanes 32..63 two panels,
                         // the other nb - 2 panels are multiplied by W on the matrix core (one sweep per block column, not two)
            if (grp == 0) fit_load_row(A + fit_blk(kb, kb), lane, true, row);
            else if (grp == 1) {
#pragma unroll
                for (int c = 0; c < FIT_B; ++c) row[c] = c == (lane & 15) ? 1.0f : 0.0f;
            } else fit_load_row(A + fit_blk(kb + grp - 1, kb), lane, false, row);
            fit_columns<true>(row, lane);
            __syncthreads();                              // every lane has read its rows: the diagonal block's place is free
            if (grp == 1) fit_store_row(A + fit_blk(kb, kb), lane, row);
            else if (grp > 1) fit_store_row(A + fit_blk(kb + grp - 1, kb), lane, row);
            __syncthreads();
            int ib = kb + 3;
            for (; ib + 3 < NBR; ib += 4) fit_trsm_n<4>(A, ib, kb, lane);
            const int left = NBR - ib;
            if (left == 3) fit_trsm_n<3>(A, ib, kb, lane);
            else if (left == 2) fit_trsm_n<2>(A, ib, kb, lane);
            else if (left == 1) fit_trsm_n<1>(A, ib, kb, lane);
            __syncthreads();
        }
        FIT_T(2);
        for (int ib = kb + 1; ib < NBR; ++ib) fit_update_row(A, ib, kb, lane);
        __syncthreads();
        FIT_T(3);
    }

    // ---- read the nine values out of the factor: q_m = -L[RB + 1 + m][RB] * L[RB][RB]
    FIT_T(4);
    float g[3], h[6];
    {
        const float *D = A + fit_blk(NBR - 1, NBR - 1);
        constexpr int lc = RB - FIT_B * (NBR - 1);            // local column of the b row in the last diagonal block
        const float piv = D[lc * FIT_LD + lc];
#pragma unroll
        for (int a = 0; a < 3; ++a) g[a] = -D[(lc + 1 + a) * FIT_LD + lc] * piv;
#pragma unroll
        for (int a = 0; a < 6; ++a) h[a] = -D[(lc + 4 + a) * FIT_LD + lc] * piv;
    }
    FIT_T(5);
    if (lane != 0) return;
    const float gn = sqrtf((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]);
    if (!(gn > 0.0f) || !isfinite(gn)) {
        out_c1[pi] = make_float4(0, 0, 0, FIT_SENTINEL); out_c2[pi] = make_float4(0, 0, 0, FIT_SENTINEL); out_n[pi] = make_float4(0, 0, 0, 0);
        return;
    }
    const f3 n = mk3(g[0] / gn, g[1] / gn, g[2] / gn);
    const f3 ax = fabsf(n.x) < 0.9f ? mk3(1, 0, 0) : mk3(0, 1, 0);
    const f3 t1 = normalize3(cross3(n, ax)), t2 = cross3(n, t1);
    const float sc = 1.0f / (rho * gn);       // H_x = H_u / rho; shape operator = tangential H_x / |grad|
    auto Hv = [&](f3 x) { return mk3((h[0] * x.x + h[1] * x.y) + h[2] * x.z, (h[1] * x.x + h[3] * x.y) + h[4] * x.z, (h[2] * x.x + h[4] * x.y) + h[5] * x.z); };
    const f3 H1 = Hv(t1), H2 = Hv(t2);
    const float m00 = dot3(t1, H1) * sc, m01 = dot3(t1, H2) * sc, m11 = dot3(t2, H2) * sc;
    const float mean = 0.5f * (m00 + m11), diff = 0.5f * (m00 - m11), rad = sqrtf(diff * diff + m01 * m01);
    const float kmax = mean + rad, kmin = mean - rad;
    float vx, vy;       // eigenvector of kmax in the (t1, t2) basis
    if (fabsf(m01) > 1.0e-12f * (fabsf(m00) + fabsf(m11) + 1.0e-30f)) { vx = m01; vy = kmax - m00; }
    else if (diff >= 0.0f) { vx = 1.0f; vy = 0.0f; }
    else { vx = 0.0f; vy = 1.0f; }
    const float vl = sqrtf(vx * vx + vy * vy);
    vx /= vl; vy /= vl;
    const f3 dmax = add3(scale3(t1, vx), scale3(t2, vy)), dmin = add3(scale3(t1, -vy), scale3(t2, vx));
    out_c1[pi] = make_float4(dmax.x, dmax.y, dmax.z, kmax);
    out_c2[pi] = make_float4(dmin.x, dmin.y, dmin.z, kmin);
    out_n[pi] = make_float4(n.x, n.y, n.z, gn);
}

void launch_hrbf_fit(hipStream_t s, const Cam &cam, const float4 *vertex, const float4 *normal, int w, float support, float ridge,
                     float jump, float4 *out_c1, float4 *out_c2, float4 *out_n)
{
    if (w >= 2)
        hipLaunchKernelGGL(k_hrbf_fit<7>, dim3(cam.W * cam.H), dim3(64), 0, s, cam, vertex, normal, w, support, ridge, jump, out_c1,
                           out_c2, out_n);
    else
        hipLaunchKernelGGL(k_hrbf_fit<3>, dim3(cam.W * cam.H), dim3(64), 0, s, cam, vertex, normal, w, support, ridge, jump, out_c1,
                           out_c2, out_n);
}

// k_fit.hip — OPTIONAL extension, not part of processFrame: a true Hermite-RBF fit per pixel, the "batched-HRBF small-GEMM on
// MFMA" BASELINE config 5 names.  NO counterpart in the reference: hrbfbase.glsl:132,153,173 use the closed-form coefficients
// 10 * n_i where a Hermite-RBF interpolant solves a (4k x 4k) symmetric positive definite system over its k centres.  The default
// path (k_curvature) is untouched; this operator (hrbf_fit_curvature) is held to a float64 numpy statement of the same algorithm
// (oracle/hrbf_fit_ref.py) within a tolerance and to analytic plane / sphere / cylinder answers (tests/test_hrbf_fit.py).
//
// One WAVE per pixel (64-thread workgroups, no cross-wave hand-over); THE MATRIX LIVES IN REGISTERS:
//   layout   a 16 x 16 block M is four registers per lane: lane l (row rr = l & 15, quarter g = l >> 4) holds M[rr][4g .. 4g + 3].
//            With that ONE layout for every block, v_mfma_f32_16x16x4_f32 with srcA = P's registers and srcB = Q's registers
//            (step s = register s: the k index runs over the quarters, a permutation of the columns that A and B share)
//            accumulates Q P^T and leaves it IN THE SAME LAYOUT (the C/D layout of the instruction is the transpose of its A / B
//            layout) — products chain from registers to registers with no shuffle and no LDS:
//              trailing update  C(i, j) -= X(i) X(j)^T :  srcA = -X(j), srcB = X(i), srcC = C(i, j)
//              panel solve      X(i) = C(i, k) L^-T    :  srcA = L(k, k)^-1, srcB = C(i, k)
//            the 28 lower-triangle blocks of the 112 x 112 system are 112 registers; three waves per SIMD (the LDS version held
//            31 KB per system: five waves per CU, 7.23 ms per 640 x 480 frame).
//   gather   the valid pixels of the (2w+1)^2 window (w <= 2: k <= 25 centres), common support rho = support * max |p_q - p_c|,
//            dimensionless coordinates u = (p - p_c) / rho
//   assemble every lane computes its own entries: one centre pair (row centre, column centre 4J + g) and one row component per
//            block — Wendland-C4 blocks [[psi, -grad psi^T], [grad psi, -H psi]](u_i - u_j) + ridge, identity padding up to
//            112 = 7 x 7 blocks (3 x 3 for the 3 x 3 window); the right-hand side and the nine read-out functionals ride as rows
//   factor   right-looking blocked Cholesky.  Only the DIAGONAL block leaves the registers: through 1.3 KB of LDS into one row per
//            lane (lanes 0..15), an identity block beside it (lanes 16..31); the column sweep (l_jt broadcast by v_readlane_b32)
//            factors the block and solves the identity into W = L^-T, which goes back through LDS into the block layout.
//   read out the gradient (3) and Hessian (6) of the interpolant at the pixel are LINEAR functionals e_m of the coefficients; their
//            rows ride in the matrix below the right-hand side, so e_m^T A^-1 b = (L^-1 e_m) . (L^-1 b) is computed by the
//            factorisation's own trailing updates and stands in L[RB + 1 + m][RB] afterwards: no forward, no back substitution,
//            then the shape operator, principal curvatures + directions on one lane
#include "common.h"
#include "kernels.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));
#ifdef FIT_NO_MFMA   // measurement only (wrong results): what the kernel costs without its matrix-core instructions
#define FIT_MFMA(a, b, c) (c)
#else
#define FIT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#endif

#define FIT_B 16                        // block edge (v_mfma_f32_16x16x4_f32)
#define FIT_LD 20                       // floats per row of the diagonal block's LDS tile (16-byte aligned rows, conflict-free transposed reads)
#define FIT_MAXK 25
#define FIT_MIN_CENTRES 8
#define FIT_SENTINEL 1000.0f
#ifndef FIT_WAVES
#define FIT_WAVES 4                     // waves per SIMD the register budget is cut for (128 registers)
#endif

__device__ __forceinline__ constexpr int fit_blk(int I, int J) { return I * (I + 1) / 2 + J; }
__device__ __forceinline__ float fit_readlane(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// Wendland C4: phi(r) = (1 - r)^6 (35 r^2 + 18 r + 3) / 3;  F = phi'/r, G = F'/r (K = G'/r where it is needed); zero beyond r = 1
__device__ __forceinline__ void fit_wendland(float r, float &phi, float &F, float &G, float &t)
{
    t = fmaxf(1.0f - r, 0.0f);
    const float t2 = t * t, t4 = t2 * t2;
    phi = (t4 * t2) * fmaf(fmaf(35.0f / 3.0f, r, 6.0f), r, 1.0f);
    F = (t4 * t) * fmaf(-280.0f / 3.0f, r, -56.0f / 3.0f);
    G = 560.0f * t4;
}

// Columns 0..NCOL-1 of the diagonal block and of the identity block beside it: a lane owns one ROW (16 registers, as eight
// pairs); lanes 0..15 hold the rows of the diagonal block, lanes 16..31 the rows of an identity (lanes 32..63 repeat lanes 0..31).
// Column j of every row is the same recurrence s = a_ij - sum_{t<j} l_it l_jt, so the identity rows come out as W = L^-T beside
// the factor.  Four columns advance together: the factor's finished rows stand in LDS (Tr[j][t] = l_jt, written 16 bytes at a
// time) and l_{j, t..t+3} arrives as ONE broadcast 16-byte read; the sum runs over PAIRS of t on v_pk_fma_f32 — (row[t], row[t+1])
// and (l_jt, l_jt+1) are natural register pairs, the two partial sums are added at the end.  (The first version took every l_jt
// through v_readlane_b32 + an SGPR: 844 broadcasts and as many hazard nops per system.)  Inside the group of four the six l_jt
// come by v_readlane_b32.  v_rsq_f32 (1 ulp) instead of the IEEE sqrt + division: this operator is held to a tolerance against a
// float64 statement, not to bits.
typedef float floatx2 __attribute__((ext_vector_type(2)));
template <int NCOL>
__device__ __forceinline__ void fit_columns(floatx2 (&row)[FIT_B / 2], int lane, float *Tr)
{
#pragma unroll
    for (int j0 = 0; j0 < NCOL; j0 += 4) {
        floatx2 acc[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
        for (int t = 0; t < j0; t += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (j0 + q >= NCOL) break;
                const floatx4 l = *reinterpret_cast<const floatx4 *>(&Tr[(j0 + q) * FIT_B + t]);
                acc[q] = __builtin_elementwise_fma(row[t / 2], floatx2{l[0], l[1]}, acc[q]);
                acc[q] = __builtin_elementwise_fma(row[t / 2 + 1], floatx2{l[2], l[3]}, acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (j0 + q >= NCOL) break;
            const int j = j0 + q;
            float s = row[j / 2][j & 1] - (acc[q][0] + acc[q][1]);
#pragma unroll
            for (int t = j0; t < j; ++t) s = fmaf(-row[t / 2][t & 1], fit_readlane(row[t / 2][t & 1], j), s);
            row[j / 2][j & 1] = s * __builtin_amdgcn_rsqf(fit_readlane(s, j));     // lane j: pivot / sqrt(pivot) = l_jj; the rows above
            // the diagonal (lanes < j) take a value nobody reads: their later columns are above the diagonal too
        }
        if (j0 + 4 < NCOL) {
            if (lane < FIT_B) *reinterpret_cast<floatx4 *>(&Tr[lane * FIT_B + j0]) = floatx4{row[j0 / 2][0], row[j0 / 2][1], row[j0 / 2 + 1][0], row[j0 / 2 + 1][1]};
            __syncthreads();
        }
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
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FIT_WAVES, FIT_WAVES)))
void k_hrbf_fit(Cam cam, const float4 *__restrict__ vertex, const float4 *__restrict__ normal, int w, float support, float ridge,
                float jump, float4 *__restrict__ out_c1, float4 *__restrict__ out_c2, float4 *__restrict__ out_n)
{
    // rows: [0, 4k) the unknowns, identity padding up to RB, row RB the right-hand side b, rows RB + 1 .. RB + 9 the nine linear
    // functionals that read the interpolant's gradient (3) and Hessian (6) at the pixel, two rows of padding.  After the
    // factorisation row RB of L is y = L^-1 b and row RB + 1 + m is z_m = L^-1 e_m, so the wanted values q_m = e_m^T A^-1 b =
    // z_m . y appear — with a minus sign, divided by the pivot of column RB — at L[RB + 1 + m][RB]: the factorisation's own
    // trailing updates compute them and NO substitution is needed.
    constexpr int NBLK = NBR * (NBR + 1) / 2, NPAD = NBR * FIT_B, RB = NPAD - 12;
    constexpr int LC = RB - FIT_B * (NBR - 1);            // local row / column of the b row in the last block row (= 4)
    __shared__ __attribute__((aligned(16))) float s_u[4 * NBR][4];      // centre: u.xyz; slots >= k hold zeros
    __shared__ __attribute__((aligned(16))) float s_n[4 * NBR][4];      // its normal
    __shared__ __attribute__((aligned(16))) float S[2 * FIT_B * FIT_LD];    // rows 0..15: the diagonal block on its way to / from the row sweep; rows 16..31: an identity
    __shared__ __attribute__((aligned(16))) float T[FIT_B * FIT_B];         // the factor's finished rows, T[j][t] = l_jt
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
        *reinterpret_cast<float4 *>(s_u[slot]) = make_float4(dx / rho, dy / rho, dz / rho, 0.0f);
        *reinterpret_cast<float4 *>(s_n[slot]) = make_float4(nn.x, nn.y, nn.z, 0.0f);
    }
    if (lane < FIT_B) {
#pragma unroll
        for (int c = 0; c < FIT_B; c += 4)
            *reinterpret_cast<floatx4 *>(&S[(FIT_B + lane) * FIT_LD + c]) = floatx4{c == lane ? 1.0f : 0.0f, c + 1 == lane ? 1.0f : 0.0f, c + 2 == lane ? 1.0f : 0.0f, c + 3 == lane ? 1.0f : 0.0f};
    }
    if (lane >= k && lane < 4 * NBR) {   // padding: centres far from everything (and from each other) with no normal — their
        // sub-blocks with every other centre vanish (compact support), their own is diag(1, 56/3, 56/3, 56/3): an inert tail
        *reinterpret_cast<float4 *>(s_u[lane]) = make_float4(1000.0f * (float)(lane + 1), 0.0f, 0.0f, 0.0f);
        *reinterpret_cast<float4 *>(s_n[lane]) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    __syncthreads();
    FIT_T(0);

    // ---- assemble, LAZILY: block (I, J) is computed when the factorisation reaches block column J (below).  Lane (rr, g) computes
    // M[rr][4g .. 4g + 3].  Its row is component p = rr & 3 of centre 4I + (rr >> 2); its four columns are the components of centre
    // 4J + g: with d = u_row - u_col, c0 = [p == 0] and w = onehot(p - 1), dp = w . d, the row of the sub-block is
    //     [ c0 phi + F dp ,  -( (G dp + c0 F) d + w F ) ]
    // — no selects.  In the last block row the rows rr >= LC are the right-hand side (rr = LC), the gradient functionals (the same
    // sub-block rows 1..3 with u_row = 0), the six Hessian functionals and two rows of padding (c0 = 0, w = 0: zeros).
    const int rr = lane & 15, g = lane >> 4, p = rr & 3;
    auto assemble = [&](const int I, const float4 ub4, const float4 nb, const bool diagonal) -> floatx4 {
        const bool tail = I == NBR - 1 && rr >= LC;                   // a row below the unknowns
        const int tr = rr - LC;                                       // tail row: 0 b, 1..3 gradient, 4..9 Hessian, 10..11 padding
        const int pp = tail ? (tr < 4 ? tr : -1) : p;                 // sub-block row this lane evaluates (-1: none)
        const float c0 = pp == 0 && !tail ? 1.0f : 0.0f;
        const float w0 = pp == 1 ? 1.0f : 0.0f, w1 = pp == 2 ? 1.0f : 0.0f, w2 = pp == 3 ? 1.0f : 0.0f;
        const float4 ua4 = *reinterpret_cast<const float4 *>(s_u[4 * I + (rr >> 2)]);
        const f3 ua = tail ? mk3(0.0f, 0.0f, 0.0f) : mk3(ua4.x, ua4.y, ua4.z);
        const float d0 = ua.x - ub4.x, d1 = ua.y - ub4.y, d2 = ua.z - ub4.z;
        const float r = __builtin_amdgcn_sqrtf(fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
        float phi, F, G, t;
        fit_wendland(r, phi, F, G, t);
        const float dp = fmaf(w2, d2, fmaf(w1, d1, w0 * d0));
        const float c0F = c0 * F, nA = fmaf(-G, dp, -c0F);
        floatx4 e;
        e[0] = fmaf(F, dp, c0 * phi);
        e[1] = fmaf(nA, d0, -(w0 * F));
        e[2] = fmaf(nA, d1, -(w1 * F));
        e[3] = fmaf(nA, d2, -(w2 * F));
        if (I == NBR - 1) {
            if (tail && tr == 0) { e[0] = 0.0f; e[1] = nb.x; e[2] = nb.y; e[3] = nb.z; }      // b: the normals
            if (tail && tr >= 4 && tr < 10) {
                const int hm = tr - 4;                                // xx xy xz yy yz zz
                const int ha = hm < 3 ? 0 : hm < 5 ? 1 : 2, hb = hm < 3 ? hm : hm < 5 ? hm - 2 : 2;
                const float K = r > 0.0f ? -2240.0f * (t * t * t) * __builtin_amdgcn_rcpf(r) : 0.0f;
                const float da = ha == 0 ? d0 : ha == 1 ? d1 : d2, db = hb == 0 ? d0 : hb == 1 ? d1 : d2;
                const float dl = ha == hb ? 1.0f : 0.0f, Kab = K * da * db;
                e[0] = fmaf(G * da, db, F * dl);
                e[1] = -fmaf(Kab, d0, G * (fmaf(d0, dl, (ha == 0 ? db : 0.0f)) + (hb == 0 ? da : 0.0f)));
                e[2] = -fmaf(Kab, d1, G * (fmaf(d1, dl, (ha == 1 ? db : 0.0f)) + (hb == 1 ? da : 0.0f)));
                e[3] = -fmaf(Kab, d2, G * (fmaf(d2, dl, (ha == 2 ? db : 0.0f)) + (hb == 2 ? da : 0.0f)));
            }
        }
        if (diagonal) {   // the diagonal entry of this row is column p of quarter rr >> 2; extra rows: large pivots, so that they stay positive
            const float dg = g != (rr >> 2) ? 0.0f : tail ? (tr < 10 ? 1.0e30f : 1.0f) : ridge;
            e[0] += p == 0 ? dg : 0.0f; e[1] += p == 1 ? dg : 0.0f; e[2] += p == 2 ? dg : 0.0f; e[3] += p == 3 ? dg : 0.0f;
        }
        return e;
    };

    // ---- blocked Cholesky, LEFT-looking over block columns: column kb is assembled when its turn comes, takes the products of
    // the finished panels X(i, j) X(kb, j)^T (j < kb), is factored, and becomes the panels X(i, kb).  Live at any time: the panels
    // of the rows still to come and one block column — at most 16 blocks = 64 registers (the right-looking order keeps all 28
    // blocks = 112 registers alive from the first update to the last).
    float g9[9];
    floatx4 X[NBLK];              // X[fit_blk(i, j)], i > j
#pragma unroll
    for (int kb = 0; kb < NBR; ++kb) {
        floatx4 Cc[NBR];
        {
            const float4 ub4 = *reinterpret_cast<const float4 *>(s_u[4 * kb + g]);
            const float4 nb = *reinterpret_cast<const float4 *>(s_n[4 * kb + g]);
#pragma unroll
            for (int i = kb; i < NBR; ++i) Cc[i] = assemble(i, ub4, nb, i == kb);
        }
        FIT_T(1);
#pragma unroll
        for (int j = 0; j < kb; ++j) {
            const floatx4 nx = -X[fit_blk(kb, j)];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = kb; i < NBR; ++i)
                    Cc[i] = FIT_MFMA(nx[s], i == kb ? -nx[s] : X[fit_blk(i, j)][s], Cc[i]);
        }
        FIT_T(3);
        // the diagonal block -> one row per lane
        *reinterpret_cast<floatx4 *>(&S[rr * FIT_LD + 4 * g]) = Cc[kb];
        __syncthreads();
        floatx2 row[FIT_B / 2];
#pragma unroll
        for (int c = 0; c < FIT_B; c += 4) {
            const floatx4 t = *reinterpret_cast<const floatx4 *>(&S[(lane & 31) * FIT_LD + c]);
            row[c / 2] = floatx2{t[0], t[1]}; row[c / 2 + 1] = floatx2{t[2], t[3]};
        }
        __syncthreads();
        if (kb == NBR - 1) {      // the last block: columns 0 .. LC are all that is read
            fit_columns<LC + 1>(row, lane, T);
            const float piv = fit_readlane(row[LC / 2][LC & 1], LC);
#pragma unroll
            for (int m = 0; m < 9; ++m) g9[m] = -fit_readlane(row[LC / 2][LC & 1], LC + 1 + m) * piv;
            break;
        }
        fit_columns<FIT_B>(row, lane, T);
        if (g == 1) {
#pragma unroll
            for (int c = 0; c < FIT_B; c += 4) *reinterpret_cast<floatx4 *>(&S[rr * FIT_LD + c]) = floatx4{row[c / 2][0], row[c / 2][1], row[c / 2 + 1][0], row[c / 2 + 1][1]};
        }
        __syncthreads();
        floatx4 li;               // L(kb, kb)^-1 in the block layout: Linv[rr][4g + r] = W[4g + r][rr]
#pragma unroll
        for (int r = 0; r < 4; ++r) li[r] = S[(4 * g + r) * FIT_LD + rr];
        __syncthreads();
        FIT_T(2);
        // panels: X(i, kb) = C(i, kb) L^-T
#pragma unroll
        for (int i = kb + 1; i < NBR; ++i) X[fit_blk(i, kb)] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = kb + 1; i < NBR; ++i) X[fit_blk(i, kb)] = FIT_MFMA(li[s], Cc[i][s], X[fit_blk(i, kb)]);
        FIT_T(3);
    }

    // ---- the nine values: gradient g9[0..2], Hessian g9[3..8] (xx xy xz yy yz zz)
    FIT_T(4);
    if (lane != 0) return;
    const float *gq = g9, *h = g9 + 3;
    const float g2 = (gq[0] * gq[0] + gq[1] * gq[1]) + gq[2] * gq[2];
    const float ign = __builtin_amdgcn_rsqf(g2), gn = g2 * ign;
    if (!(g2 > 0.0f) || !isfinite(gn)) {
        out_c1[pi] = make_float4(0, 0, 0, FIT_SENTINEL); out_c2[pi] = make_float4(0, 0, 0, FIT_SENTINEL); out_n[pi] = make_float4(0, 0, 0, 0);
        return;
    }
    const f3 n = mk3(gq[0] * ign, gq[1] * ign, gq[2] * ign);
    const f3 ax = fabsf(n.x) < 0.9f ? mk3(1, 0, 0) : mk3(0, 1, 0);
    const f3 c1 = cross3(n, ax);
    const f3 t1 = scale3(c1, __builtin_amdgcn_rsqf(dot3(c1, c1))), t2 = cross3(n, t1);
    const float sc = __builtin_amdgcn_rcpf(rho * gn);       // H_x = H_u / rho; shape operator = tangential H_x / |grad|
    auto Hv = [&](f3 x) { return mk3((h[0] * x.x + h[1] * x.y) + h[2] * x.z, (h[1] * x.x + h[3] * x.y) + h[4] * x.z, (h[2] * x.x + h[4] * x.y) + h[5] * x.z); };
    const f3 H1 = Hv(t1), H2 = Hv(t2);
    const float m00 = dot3(t1, H1) * sc, m01 = dot3(t1, H2) * sc, m11 = dot3(t2, H2) * sc;
    const float mean = 0.5f * (m00 + m11), diff = 0.5f * (m00 - m11), rad = __builtin_amdgcn_sqrtf(diff * diff + m01 * m01);
    const float kmax = mean + rad, kmin = mean - rad;
    float vx, vy;       // eigenvector of kmax in the (t1, t2) basis
    if (fabsf(m01) > 1.0e-12f * (fabsf(m00) + fabsf(m11) + 1.0e-30f)) { vx = m01; vy = kmax - m00; }
    else if (diff >= 0.0f) { vx = 1.0f; vy = 0.0f; }
    else { vx = 0.0f; vy = 1.0f; }
    const float ivl = __builtin_amdgcn_rsqf(vx * vx + vy * vy);
    vx *= ivl; vy *= ivl;
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

// Measurement probe (not part of the path): what rate does v_mfma_f32_16x16x4_f32 really sustain on this part, in
// CYCLES (s_memtime, the shader clock) so that the answer does not depend on the clock the board happens to run at, and what that
// clock is under the load (s_memtime against the constant 100 MHz s_memrealtime).
//   ACC independent accumulators per wave (ACC = 1: one dependent chain), W waves per SIMD, long runs
//   (ACC = 8 is not run: hipcc gives the eighth accumulator a misaligned source tuple a[2:5] and repairs it with accvgpr moves and an
//   s_nop 7 per iteration — 40 cycles per MFMA that are the compiler's, not the matrix core's)
//   (>= 100 ms per configuration: the short runs of mfma_valu_overlap.hip finish before the clock has settled).
// The guide (MI355X_MICROARCH.md:41,389,434) states 32 cycles per SIMD issue, 40 cycles dependent latency, 155 TF measured.
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_rate.hip -o /tmp/mfr && /tmp/mfr
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int ACC, int VALU>
__global__ __launch_bounds__(256) void k_rate(unsigned long long *out, int iters)
{
    floatx4 acc[8] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i);
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {   // 8 MFMAs per iteration over ACC accumulators
            acc[m % ACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % ACC], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < VALU; ++q) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double *)&v[(q & 3) * 2]) : "v"(*(const double *)&v[0]), "v"(*(const double *)&v[2]));
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = c1 - c0; out[2 * w + 1] = r1 - r0;
    }
    if (s == 12345.678f) out[0] = (unsigned long long)s;
}

template <int ACC, int VALU>
static void run(int waves_per_simd, int iters)
{
    const int blocks = 256 * waves_per_simd, waves = blocks * 4;      // 256 CUs x 4 SIMDs, one wave per SIMD per workgroup
    unsigned long long *d; hipMalloc(&d, sizeof(unsigned long long) * 2 * waves);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<ACC, VALU>), dim3(blocks), dim3(256), 0, 0, d, iters / 4);     // warm-up: let the clock settle under this load
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_rate<ACC, VALU>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * waves);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), mhz(waves);
    for (int w = 0; w < waves; ++w) { cyc[w] = (double)h[2 * w]; mhz[w] = (double)h[2 * w] / ((double)h[2 * w + 1] / 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
    const double per_wave = cyc[waves / 2] / ((double)iters * 8.0);            // cycles a wave spends per MFMA it issues
    const double per_simd = per_wave / waves_per_simd;                           // cycles of SIMD time per MFMA
    const double clock = mhz[waves / 2];
    const double tf = (double)waves * iters * 8.0 * 2048.0 / (ms * 1e-3) / 1e12; // 16 x 16 x 4 MACs = 2048 flop per instruction
    printf("ACC=%d VALU/MFMA=%d waves/SIMD=%d : %8.2f ms  %6.1f cycles per MFMA per wave = %5.1f per SIMD; clock %4.0f MHz (s_memtime / s_memrealtime)  %6.1f TFLOP/s = %.2f of 157.3\n",
           ACC, VALU, waves_per_simd, ms, per_wave, per_simd, clock, tf, tf / 157.3);
    hipFree(d);
}

int main()
{
    const int it = 400000;      // x 8 MFMAs x 32 cycles ~ 100 M cycles ~ 45-60 ms at one wave per SIMD
    run<1, 0>(1, it / 2);       // one dependent chain: the latency
    run<2, 0>(1, it);
    run<4, 0>(1, it);           // the guide's shape, one wave
    run<4, 0>(2, it / 2);
    run<4, 0>(4, it / 4);       // the guide's shape: >= 4 accumulators, 4 waves per SIMD
    run<1, 0>(4, it / 4);       // four waves of ONE dependent chain each: what hides the 40-cycle latency
    run<4, 0>(8, it / 8);
    run<4, 2>(4, it / 4);       // beside vector work (v_pk_fma_f32, 2 per MFMA): do the times add?
    run<4, 8>(4, it / 8);
    return 0;
}

// Measurement probe (not part of the path): does v_mfma_f32_16x16x4_f32 execute beside independent VALU work on gfx950?
// Each wave runs ITER x (M independent MFMAs + V independent v_fma_f32); W waves per SIMD.  If the two co-execute, the time per
// iteration stays at max(32 M, 4 V) cycles; if the f32 matrix instruction occupies the vector ALU, it is their sum.
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef short shortx8 __attribute__((ext_vector_type(8)));

template <int M, int V, int KIND>
__global__ __launch_bounds__(256) void k_probe(float *out, int iters)
{
    floatx4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i);
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    shortx8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (short)(0x3f80 + threadIdx.x); hb[i] = (short)(0x3f80 - i); }
    for (int it = 0; it < iters; ++it) {
        // program order: one MFMA, then its share of the VALU work — a wave issues in order, so the VALU instructions must not
        // queue behind a second MFMA that waits for the pipe
#pragma unroll
        for (int m = 0; m < (M ? M : 1); ++m) {
            if (M) {
                if (KIND == 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
                else acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[m & 3], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < V / (M ? M : 1); ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(a), "v"(b));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    if (s == 12345.678f) out[0] = s;
}

template <int M, int V, int KIND>
static void run(const char *name, int waves_per_simd)
{
    float *d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * waves_per_simd;      // 256 CUs, 4 waves (one per SIMD) per workgroup
    hipLaunchKernelGGL((k_probe<M, V, KIND>), dim3(blocks), dim3(256), 0, 0, d, 100);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_probe<M, V, KIND>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s M=%d V=%2d waves/SIMD=%d : %7.3f ms  -> %6.1f ns/iter/wave-slot (at 2.4 GHz: %6.1f cycles per iteration per SIMD)\n", name, M, V, waves_per_simd,
           ms, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
    hipFree(d);
}

int main()
{
    run<4, 0, 0>("f32 16x16x4: MFMA only", 1);
    run<0, 32, 0>("VALU only", 1);
    run<4, 32, 0>("f32 16x16x4 + VALU", 1);
    run<4, 32, 0>("f32 16x16x4 + VALU", 2);
    run<4, 0, 0>("f32 16x16x4: MFMA only", 2);
    run<0, 32, 0>("VALU only", 2);
    run<4, 16, 0>("f32 16x16x4 + VALU", 2);
    run<4, 0, 1>("bf16 16x16x32: MFMA only", 1);
    run<4, 32, 1>("bf16 16x16x32 + VALU", 1);
    run<4, 32, 1>("bf16 16x16x32 + VALU", 2);
    return 0;
}

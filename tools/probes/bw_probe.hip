// Measurement probe (not part of the path): what a plain streaming kernel reaches on this GPU at the sizes of the fuse pass.
//   read  : one float4 plane of N surfels (16 B / surfel), summed            -> the floor of k_project / pass A's position stream
//   copy5 : five float4 planes read and written out of place (160 B / surfel) -> the ceiling of pass B (in-place compaction)
// hipcc --offload-arch=gfx950 -O3 tools/probes/bw_probe.hip -o /tmp/bw_probe && /tmp/bw_probe 4343735
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ p, size_t n, float *out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.w; }
    if (acc == 12345.678f) *out = acc;
}
template <int IPT>
__global__ __launch_bounds__(512) void k_copy5(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n, size_t plane)
{
    for (size_t base = (size_t)blockIdx.x * blockDim.x * IPT; base < n; base += (size_t)gridDim.x * blockDim.x * IPT) {
        float4 v[IPT][5];
#pragma unroll
        for (int k = 0; k < IPT; ++k) { size_t i = base + k * blockDim.x + threadIdx.x; if (i < n) for (int q = 0; q < 5; ++q) v[k][q] = a[q * plane + i]; }
#pragma unroll
        for (int k = 0; k < IPT; ++k) { size_t i = base + k * blockDim.x + threadIdx.x; if (i < n) for (int q = 0; q < 5; ++q) b[q * plane + i] = v[k][q]; }
    }
}
// copy5 with the workgroup shape as a parameter: all loads of a trip first (IPT x 5 float4 per thread), then the stores
template <int THREADS, int IPT>
__global__ __launch_bounds__(THREADS) void k_copy5s(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n, size_t plane)
{
    for (size_t base = (size_t)blockIdx.x * THREADS * IPT; base < n; base += (size_t)gridDim.x * THREADS * IPT) {
        float4 v[IPT][5];
#pragma unroll
        for (int k = 0; k < IPT; ++k) { size_t i = base + k * THREADS + threadIdx.x; if (i < n) for (int q = 0; q < 5; ++q) v[k][q] = a[q * plane + i]; }
#pragma unroll
        for (int k = 0; k < IPT; ++k) { size_t i = base + k * THREADS + threadIdx.x; if (i < n) for (int q = 0; q < 5; ++q) b[q * plane + i] = v[k][q]; }
    }
}
// V1/V2: one contiguous stream of 5 n float4 (what hipMemcpy sees), U float4 per thread per trip
typedef float f4v __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_flat(const float4 *__restrict__ a4, float4 *__restrict__ b4, size_t n)
{
    const f4v *a = reinterpret_cast<const f4v *>(a4); f4v *b = reinterpret_cast<f4v *>(b4);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride * U) {
        f4v v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) if (i + k * stride < n) v[k] = NT ? __builtin_nontemporal_load(&a[i + k * stride]) : a[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) if (i + k * stride < n) { if (NT) __builtin_nontemporal_store(v[k], &b[i + k * stride]); else b[i + k * stride] = v[k]; }
    }
}
// V3: a workgroup takes a tile of T surfels and moves it plane by plane (one stream in, one out at a time)
template <int T>
__global__ __launch_bounds__(512) void k_copy_tile_planes(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n, size_t plane)
{
    const size_t tiles = (n + T - 1) / T;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        for (int q = 0; q < 5; ++q) {
            float4 v[T / 512];
#pragma unroll
            for (int k = 0; k < T / 512; ++k) { size_t i = t * T + k * 512 + threadIdx.x; if (i < n) v[k] = a[q * plane + i]; }
#pragma unroll
            for (int k = 0; k < T / 512; ++k) { size_t i = t * T + k * 512 + threadIdx.x; if (i < n) b[q * plane + i] = v[k]; }
        }
    }
}
// V4 (round 3): the in-place left shift of pass B — record i + S moves to i inside ONE buffer — record-major (every lane moves its
// record through the five planes, what k_fuse_stream does) against plane-major (the grid finishes plane 0, then plane 1, ...: two
// streams at a time).  Bandwidth only: no hazard protocol (S is far larger than what the grid holds in flight).
template <int IPT>
__global__ __launch_bounds__(512) void k_shift_record_major(float4 *__restrict__ a, size_t n, size_t plane, size_t S)
{
    for (size_t base = (size_t)blockIdx.x * 512 * IPT; base + S < n; base += (size_t)gridDim.x * 512 * IPT) {
        float4 v[IPT][5];
#pragma unroll
        for (int k = 0; k < IPT; ++k) { size_t i = base + k * 512 + threadIdx.x; if (i + S < n) for (int q = 0; q < 5; ++q) v[k][q] = a[q * plane + i + S]; }
#pragma unroll
        for (int k = 0; k < IPT; ++k) { size_t i = base + k * 512 + threadIdx.x; if (i + S < n) for (int q = 0; q < 5; ++q) a[q * plane + i] = v[k][q]; }
    }
}
// the same record-major tile, but the 5 x IPT loads of a lane issued plane by plane (IPT consecutive KB of one plane per wave, then
// the next plane) instead of record by record; SPLIT: loads and stores of one plane back to back (plane-major inside the tile)
template <int IPT, bool SPLIT>
__global__ __launch_bounds__(512) void k_shift_tile_plane_order(float4 *__restrict__ a, size_t n, size_t plane, size_t S)
{
    for (size_t base = (size_t)blockIdx.x * 512 * IPT; base + S < n; base += (size_t)gridDim.x * 512 * IPT) {
        float4 v[5][IPT];
        if (!SPLIT) {
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int k = 0; k < IPT; ++k) { size_t i = base + k * 512 + threadIdx.x; if (i + S < n) v[q][k] = a[q * plane + i + S]; }
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int k = 0; k < IPT; ++k) { size_t i = base + k * 512 + threadIdx.x; if (i + S < n) a[q * plane + i] = v[q][k]; }
        } else {
#pragma unroll
            for (int q = 0; q < 5; ++q) {
#pragma unroll
                for (int k = 0; k < IPT; ++k) { size_t i = base + k * 512 + threadIdx.x; if (i + S < n) v[q][k] = a[q * plane + i + S]; }
#pragma unroll
                for (int k = 0; k < IPT; ++k) { size_t i = base + k * 512 + threadIdx.x; if (i + S < n) a[q * plane + i] = v[q][k]; }
            }
        }
    }
}
template <int IPT>
__global__ __launch_bounds__(512) void k_shift_plane_major(float4 *__restrict__ a, size_t n, size_t plane, size_t S)
{
    for (int q = 0; q < 5; ++q)
        for (size_t base = (size_t)blockIdx.x * 512 * IPT; base + S < n; base += (size_t)gridDim.x * 512 * IPT) {
            float4 v[IPT];
#pragma unroll
            for (int k = 0; k < IPT; ++k) { size_t i = base + k * 512 + threadIdx.x; if (i + S < n) v[k] = a[q * plane + i + S]; }
#pragma unroll
            for (int k = 0; k < IPT; ++k) { size_t i = base + k * 512 + threadIdx.x; if (i + S < n) a[q * plane + i] = v[k]; }
        }
}
int main(int argc, char **argv)
{
    size_t n = argc > 1 ? atoll(argv[1]) : 4343735;
    float4 *a, *b; float *o;
    hipMalloc(&a, n * 80); hipMalloc(&b, n * 80); hipMalloc(&o, 4);
    hipMemset(a, 1, n * 80); hipMemset(b, 0, n * 80);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, double bytes, auto launch) {
        for (int w = 0; w < 3; ++w) launch();
        float best = 1e9, sum = 0; const int R = 20;
        for (int r = 0; r < R; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms; }
        printf("%-28s avg %7.1f us  best %7.1f us  -> %.2f TB/s (best %.2f)\n", name, 1e3 * sum / R, 1e3 * best, bytes / (sum / R * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
    };
    for (int blocks : {256, 512, 1024, 2048, 4096, 16384})
        timeit(("read16 blocks=" + std::to_string(blocks)).c_str(), n * 16.0, [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, o); });
    for (int blocks : {256, 512, 1024, 2048})
        timeit(("copy5 ipt1 blocks=" + std::to_string(blocks)).c_str(), n * 160.0, [&] { hipLaunchKernelGGL(k_copy5<1>, dim3(blocks), dim3(512), 0, 0, a, b, n, n); });
    for (int blocks : {256, 512, 1024})
        timeit(("copy5 ipt2 blocks=" + std::to_string(blocks)).c_str(), n * 160.0, [&] { hipLaunchKernelGGL(k_copy5<2>, dim3(blocks), dim3(512), 0, 0, a, b, n, n); });
    for (int blocks : {256, 512})
        timeit(("copy5 ipt4 blocks=" + std::to_string(blocks)).c_str(), n * 160.0, [&] { hipLaunchKernelGGL(k_copy5<4>, dim3(blocks), dim3(512), 0, 0, a, b, n, n); });
    for (int blocks : {256, 512, 1024, 2048, 4096}) {
        timeit(("copy5 T256 ipt1 blocks=" + std::to_string(blocks)).c_str(), n * 160.0, [&] { hipLaunchKernelGGL((k_copy5s<256, 1>), dim3(blocks), dim3(256), 0, 0, a, b, n, n); });
        timeit(("copy5 T256 ipt2 blocks=" + std::to_string(blocks)).c_str(), n * 160.0, [&] { hipLaunchKernelGGL((k_copy5s<256, 2>), dim3(blocks), dim3(256), 0, 0, a, b, n, n); });
        timeit(("copy5 T128 ipt1 blocks=" + std::to_string(blocks)).c_str(), n * 160.0, [&] { hipLaunchKernelGGL((k_copy5s<128, 1>), dim3(blocks), dim3(128), 0, 0, a, b, n, n); });
        timeit(("copy5 T1024 ipt1 blocks=" + std::to_string(blocks)).c_str(), n * 160.0, [&] { hipLaunchKernelGGL((k_copy5s<1024, 1>), dim3(blocks), dim3(1024), 0, 0, a, b, n, n); });
    }
    // does the distance between the planes matter (channel / bank interleave)?  plane stride = n + pad elements of 16 B
    {
        float4 *a2, *b2; const size_t big = n + (1u << 20);
        hipMalloc(&a2, big * 80); hipMalloc(&b2, big * 80); hipMemset(a2, 1, big * 80);
        for (size_t pad : {(size_t)0, (size_t)1, (size_t)4, (size_t)16, (size_t)64, (size_t)100, (size_t)256, (size_t)1000, (size_t)4096, (size_t)65536 + 4, (size_t)(1u << 20)})
            timeit(("copy5 T256 ipt1 b=512 pad=" + std::to_string(pad)).c_str(), n * 160.0, [&] { hipLaunchKernelGGL((k_copy5s<256, 1>), dim3(512), dim3(256), 0, 0, a2, b2, n, n + pad); });
        hipFree(a2); hipFree(b2);
    }
    for (int blocks : {512, 1024, 2048, 4096, 8192}) {
        timeit(("flat U1 blocks=" + std::to_string(blocks)).c_str(), n * 160.0, [&] { hipLaunchKernelGGL((k_copy_flat<1, false>), dim3(blocks), dim3(256), 0, 0, a, b, n * 5); });
        timeit(("flat U4 blocks=" + std::to_string(blocks)).c_str(), n * 160.0, [&] { hipLaunchKernelGGL((k_copy_flat<4, false>), dim3(blocks), dim3(256), 0, 0, a, b, n * 5); });
    }
    timeit("flat U4 NT blocks=2048", n * 160.0, [&] { hipLaunchKernelGGL((k_copy_flat<4, true>), dim3(2048), dim3(256), 0, 0, a, b, n * 5); });
    timeit("flat U8 blocks=1024", n * 160.0, [&] { hipLaunchKernelGGL((k_copy_flat<8, false>), dim3(1024), dim3(256), 0, 0, a, b, n * 5); });
    for (int blocks : {256, 512, 1024})
        timeit(("tile2048 planes blocks=" + std::to_string(blocks)).c_str(), n * 160.0, [&] { hipLaunchKernelGGL(k_copy_tile_planes<2048>, dim3(blocks), dim3(512), 0, 0, a, b, n, n); });
    {
        const size_t S = 200000;
        const double bytes = (double)(n - S) * 160.0;
        for (int blocks : {256, 512, 1024}) {
            timeit(("shift record-major ipt1 b=" + std::to_string(blocks)).c_str(), bytes, [&] { hipLaunchKernelGGL(k_shift_record_major<1>, dim3(blocks), dim3(512), 0, 0, a, n, n, S); });
            timeit(("shift record-major ipt4 b=" + std::to_string(blocks)).c_str(), bytes, [&] { hipLaunchKernelGGL(k_shift_record_major<4>, dim3(blocks), dim3(512), 0, 0, a, n, n, S); });
            timeit(("shift tile plane-order ipt4 b=" + std::to_string(blocks)).c_str(), bytes, [&] { hipLaunchKernelGGL((k_shift_tile_plane_order<4, false>), dim3(blocks), dim3(512), 0, 0, a, n, n, S); });
            timeit(("shift tile plane-split ipt4 b=" + std::to_string(blocks)).c_str(), bytes, [&] { hipLaunchKernelGGL((k_shift_tile_plane_order<4, true>), dim3(blocks), dim3(512), 0, 0, a, n, n, S); });
            timeit(("shift tile plane-order ipt2 b=" + std::to_string(blocks)).c_str(), bytes, [&] { hipLaunchKernelGGL((k_shift_tile_plane_order<2, false>), dim3(blocks), dim3(512), 0, 0, a, n, n, S); });
            timeit(("shift record-major ipt2 b=" + std::to_string(blocks)).c_str(), bytes, [&] { hipLaunchKernelGGL(k_shift_record_major<2>, dim3(blocks), dim3(512), 0, 0, a, n, n, S); });
            timeit(("shift plane-major  ipt1 b=" + std::to_string(blocks)).c_str(), bytes, [&] { hipLaunchKernelGGL(k_shift_plane_major<1>, dim3(blocks), dim3(512), 0, 0, a, n, n, S); });
            timeit(("shift plane-major  ipt4 b=" + std::to_string(blocks)).c_str(), bytes, [&] { hipLaunchKernelGGL(k_shift_plane_major<4>, dim3(blocks), dim3(512), 0, 0, a, n, n, S); });
            timeit(("shift plane-major  ipt8 b=" + std::to_string(blocks)).c_str(), bytes, [&] { hipLaunchKernelGGL(k_shift_plane_major<8>, dim3(blocks), dim3(512), 0, 0, a, n, n, S); });
        }
    }
    timeit("hipMemcpyDtoD 5 planes", n * 160.0, [&] { hipMemcpyAsync(b, a, n * 80, hipMemcpyDeviceToDevice, 0); });
    return 0;
}

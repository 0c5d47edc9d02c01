// Micro-benchmark: ceiling of the k_fused store pattern.  N units x 5 planes x 512 x 512 float32, written as float4 per lane by
// 256-thread workgroups in three tilings: (a) linear fill, (b) 256 x 16 tiles (wave <-> rows w, w+4, ...: the current k_fused),
// (c) 256 x 16 tiles with 4 contiguous rows per wave, (d) 512 x 16 tiles (full rows: 16 KB contiguous per plane and wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int CROP = 512, PLANES = 5;

__global__ __launch_bounds__(256) void k_linear(float* out, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n4; i += stride) reinterpret_cast<float4*>(out)[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// MODE 0: rows w + 4r (interleaved), 256-wide tile;  MODE 1: rows 4w + r (contiguous per wave), 256-wide;  MODE 2: 512-wide tile,
// rows 4w + r, two float4 per lane and row
template <int MODE>
__global__ __launch_bounds__(256) void k_tiles(float* out) {
    const int u = blockIdx.z, by = blockIdx.y, bx = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t plane = (size_t)CROP * CROP;
    float* o = out + (size_t)u * PLANES * plane;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = by * 16 + (MODE == 0 ? wv + 4 * r : 4 * wv + r);
        if (MODE == 2) {
#pragma unroll
            for (int c = 0; c < PLANES; ++c) {
                float* row = o + c * plane + (size_t)y * CROP;
                *reinterpret_cast<float4*>(row + 4 * lane) = make_float4(1.f, 2.f, 3.f, (float)c);
                *reinterpret_cast<float4*>(row + 256 + 4 * lane) = make_float4(1.f, 2.f, 3.f, (float)c);
            }
        } else {
            const size_t off = (size_t)y * CROP + bx * 256 + 4 * lane;
#pragma unroll
            for (int c = 0; c < PLANES; ++c) *reinterpret_cast<float4*>(o + c * plane + off) = make_float4(1.f, 2.f, 3.f, (float)c);
        }
    }
}

int main() {
    const int N = 168;
    const size_t bytes = (size_t)N * PLANES * CROP * CROP * 4;
    float* out;
    CHECK(hipMalloc(&out, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        float best = 1e9f, sum = 0.f;
        for (int i = 0; i < 10; ++i) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best; sum += ms;
        }
        printf("%-46s avg %.1f us  best %.1f us  %.2f TB/s (avg)\n", name, sum * 100.f, best * 1e3f, bytes / (sum / 10 * 1e-3) / 1e12);
    };
    timeit("linear fill, 4096 workgroups", [&] { hipLaunchKernelGGL(k_linear, dim3(4096), dim3(256), 0, 0, out, bytes / 16); });
    timeit("linear fill, 1024 workgroups", [&] { hipLaunchKernelGGL(k_linear, dim3(1024), dim3(256), 0, 0, out, bytes / 16); });
    timeit("256x16 tiles, wave rows interleaved (k_fused)", [&] { hipLaunchKernelGGL(k_tiles<0>, dim3(2, 32, N), dim3(256), 0, 0, out); });
    timeit("256x16 tiles, 4 contiguous rows per wave", [&] { hipLaunchKernelGGL(k_tiles<1>, dim3(2, 32, N), dim3(256), 0, 0, out); });
    timeit("512x16 tiles, 4 contiguous rows per wave", [&] { hipLaunchKernelGGL(k_tiles<2>, dim3(1, 32, N), dim3(256), 0, 0, out); });
    return 0;
}

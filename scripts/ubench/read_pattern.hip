// Micro-benchmark: ceiling of a read-only streaming reduction (what the BatchNorm statistics kernels do): sum of a buffer with
// 16-byte loads, U loads in flight per thread, grid-stride over the whole buffer, for buffer sizes 0.075 / 0.3 / 1.2 GB.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_sum(const uint4* __restrict__ in, size_t n16, float* out) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * 256;
    float s = 0.f;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) s += __uint_as_float(v[u].x) + __uint_as_float(v[u].y) + __uint_as_float(v[u].z) + __uint_as_float(v[u].w);
    }
    for (; i < n16; i += stride) { const uint4 v = in[i]; s += __uint_as_float(v.x) + __uint_as_float(v.w); }
    if (s == 12345.678f) out[0] = s;          // keep the loads
}

template <int U>
int run(const uint4* buf, size_t bytes, int blocks, float* out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t n16 = bytes / 16;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_sum<U>, dim3(blocks), dim3(256), 0, 0, buf, n16, out);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_sum<U>, dim3(blocks), dim3(256), 0, 0, buf, n16, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %.3f GB, U = %d, %5d blocks: %7.1f us  %.2f TB/s\n", bytes / 1e9, U, blocks, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
    return 0;
}

int main() {
    const size_t big = (size_t)1208 * 1000 * 1000;
    uint4* buf; float* out;
    CHECK(hipMalloc(&buf, big)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemset(buf, 0, big));
    for (size_t bytes : {(size_t)75 * 1000 * 1000, (size_t)302 * 1000 * 1000, big})
        for (int blocks : {1024, 2048, 4096, 16384}) {
            if (run<4>(buf, bytes, blocks, out)) return 1;
            if (run<8>(buf, bytes, blocks, out)) return 1;
        }
    return 0;
}

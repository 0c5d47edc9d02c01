// Stride-2 pixel sub-sampling of NCHW planes, y[p][i][j] = x[p][2i][2j], and its gradient (dy scattered to the even
// positions, zeros elsewhere).  A 1x1 convolution with stride 2 (the down-sampling shortcut of a ResNet stage) is this
// followed by a stride-1 1x1 convolution; the library instead transposes the whole activation NCHW -> CNHW and back around
// its GEMM and zero-fills the input gradient in a separate pass (~5 ms per step for the two shortcuts at N = 144 x 512^2).
// Both kernels here are single streaming passes: 16-byte stores, every byte of dx written exactly once.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

// bf16: one lane = 8 outputs (16-byte store) from 16 inputs (two 16-byte loads); float: 4 outputs from 8 inputs
template <typename T> struct Sub;
template <> struct Sub<float> {
    static constexpr int V = 4;
    static __device__ __forceinline__ void gather(const float* row, float* y) {
        const float4 a = aadg_load_stream(row), b = aadg_load_stream(row + 4);
        *reinterpret_cast<float4*>(y) = make_float4(a.x, a.z, b.x, b.z);
    }
    static __device__ __forceinline__ void scatter(const float* g, float* even_row, float* odd_row) {
        const float4 v = aadg_load_stream(g);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(even_row) = make_float4(v.x, 0.f, v.y, 0.f);
        *reinterpret_cast<float4*>(even_row + 4) = make_float4(v.z, 0.f, v.w, 0.f);
        if (odd_row != nullptr) { *reinterpret_cast<float4*>(odd_row) = z; *reinterpret_cast<float4*>(odd_row + 4) = z; }
    }
};
template <> struct Sub<__hip_bfloat16> {
    static constexpr int V = 8;
    static __device__ __forceinline__ void gather(const __hip_bfloat16* row, __hip_bfloat16* y) {
        const uint4 a = aadg_load_stream(row), b = aadg_load_stream(row + 8);
        // the even elements are the low halves of each 32-bit pair
        *reinterpret_cast<uint4*>(y) = make_uint4((a.x & 0xFFFFu) | (a.y << 16), (a.z & 0xFFFFu) | (a.w << 16),
                                                  (b.x & 0xFFFFu) | (b.y << 16), (b.z & 0xFFFFu) | (b.w << 16));
    }
    static __device__ __forceinline__ void scatter(const __hip_bfloat16* g, __hip_bfloat16* even_row, __hip_bfloat16* odd_row) {
        const uint4 v = aadg_load_stream(g);
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(even_row) = make_uint4(v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16);
        *reinterpret_cast<uint4*>(even_row + 8) = make_uint4(v.z & 0xFFFFu, v.z >> 16, v.w & 0xFFFFu, v.w >> 16);
        if (odd_row != nullptr) { *reinterpret_cast<uint4*>(odd_row) = z; *reinterpret_cast<uint4*>(odd_row + 8) = z; }
    }
};

template <typename T>
__global__ __launch_bounds__(256) void k_subsample2(const T* __restrict__ x, T* __restrict__ y, int H, int W, int Ho, int Wo,
                                                    long long items) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= items) return;
    constexpr int V = Sub<T>::V;
    const int per_row = Wo / V;
    const int cg = (int)(q % per_row);
    const long long t = q / per_row;
    const int i = (int)(t % Ho);
    const long long plane = t / Ho;
    Sub<T>::gather(x + ((size_t)plane * H + 2 * i) * W + 2 * V * cg, y + ((size_t)plane * Ho + i) * Wo + V * cg);
}

// one lane = V gradients of output row i -> 2V elements of input row 2i and the 2V zeros of row 2i + 1 below it
template <typename T>
__global__ __launch_bounds__(256) void k_subsample2_bwd(const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int Ho, int Wo,
                                                        long long items) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= items) return;
    constexpr int V = Sub<T>::V;
    const int per_row = Wo / V;
    const int cg = (int)(q % per_row);
    const long long t = q / per_row;
    const int i = (int)(t % Ho);
    const long long plane = t / Ho;
    T* even = dx + ((size_t)plane * H + 2 * i) * W + 2 * V * cg;
    Sub<T>::scatter(dy + ((size_t)plane * Ho + i) * Wo + V * cg, even, 2 * i + 1 < H ? even + W : nullptr);
}

inline bool sub_ok(long long planes, int H, int W, int dtype) {
    const int V = dtype == 0 ? 4 : 8;
    return planes > 0 && H >= 2 && (H % 2) == 0 && W >= 2 * V && (W % (2 * V)) == 0;
}

}  // namespace

extern "C" int aadg_subsample2x2_supported(int H, int W, int dtype) { return (dtype == 0 || dtype == 1) && sub_ok(1, H, W, dtype) ? 1 : 0; }

extern "C" int aadg_subsample2x2(const void* x, void* y, int planes, int H, int W, int dtype, void* stream) {
    if (x == nullptr || y == nullptr || ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0) || (dtype != 0 && dtype != 1)) return AADG_E_BADARG;
    if (!sub_ok(planes, H, W, dtype)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int Ho = H / 2, Wo = W / 2;
    const long long items = (long long)planes * Ho * (Wo / (dtype == 0 ? 4 : 8));
    if (items > 0x7FFFFFFFLL * 256) return AADG_E_UNSUPPORTED;
    const dim3 grid((unsigned)((items + 255) / 256));
    if (dtype == 0) hipLaunchKernelGGL(k_subsample2<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, H, W, Ho, Wo, items);
    else hipLaunchKernelGGL(k_subsample2<__hip_bfloat16>, grid, dim3(256), 0, st, (const __hip_bfloat16*)x, (__hip_bfloat16*)y, H, W, Ho, Wo, items);
    AADG_LAUNCH_CHECK();
    return 0;
}

extern "C" int aadg_subsample2x2_backward(const void* dy, void* dx, int planes, int H, int W, int dtype, void* stream) {
    if (dy == nullptr || dx == nullptr || ((((uintptr_t)dy | (uintptr_t)dx) & 15u) != 0) || (dtype != 0 && dtype != 1)) return AADG_E_BADARG;
    if (!sub_ok(planes, H, W, dtype)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int Ho = H / 2, Wo = W / 2;
    const long long items = (long long)planes * Ho * (Wo / (dtype == 0 ? 4 : 8));
    if (items > 0x7FFFFFFFLL * 256) return AADG_E_UNSUPPORTED;
    const dim3 grid((unsigned)((items + 255) / 256));
    if (dtype == 0) hipLaunchKernelGGL(k_subsample2_bwd<float>, grid, dim3(256), 0, st, (const float*)dy, (float*)dx, H, W, Ho, Wo, items);
    else hipLaunchKernelGGL(k_subsample2_bwd<__hip_bfloat16>, grid, dim3(256), 0, st, (const __hip_bfloat16*)dy, (__hip_bfloat16*)dx, H, W, Ho, Wo, items);
    AADG_LAUNCH_CHECK();
    return 0;
}

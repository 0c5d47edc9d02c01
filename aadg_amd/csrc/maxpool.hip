// 3x3 / stride 2 / padding 1 max-pooling over NCHW planes (the ResNet stem pool, [N, 64, h/2, w/2] -> h/4) forward and
// backward.  The forward stores the arg-max as the position inside the 3x3 window (one byte per output instead of the
// library's int64 index: 8 bytes); the backward is then a pure gather -- every lane owns 8 consecutive input columns of one
// row, looks at the <= 2 x 5 outputs whose windows cover them and writes one 16-byte vector: no atomics, no LDS, and the
// pooled input (4x the output's size) is neither saved for nor read by the backward.
//
// Arithmetic = ATen max_pool2d: padding never wins, ties go to the first element in row-major window order.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

template <typename T> struct Px;
template <> struct Px<float> {
    static __device__ __forceinline__ void load8(const float* p, float* v) {
        const float4 a = aadg_load_stream(p), b = aadg_load_stream(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void load4(const float* p, float* v) {
        const float4 a = aadg_load_stream(p);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    }
    static __device__ __forceinline__ float load1(const float* p) { return *p; }
    static __device__ __forceinline__ void store4(float* p, const float* v) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    static __device__ __forceinline__ void store8(float* p, const float* v) { store4(p, v); store4(p + 4, v + 4); }
};
template <> struct Px<__hip_bfloat16> {
    static __device__ __forceinline__ void load8(const __hip_bfloat16* p, float* v) {
        const uint4 t = aadg_load_stream(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
    static __device__ __forceinline__ void load4(const __hip_bfloat16* p, float* v) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
    }
    static __device__ __forceinline__ float load1(const __hip_bfloat16* p) {
        return __uint_as_float((uint32_t)(*reinterpret_cast<const uint16_t*>(p)) << 16);
    }
    static __device__ __forceinline__ void store4(__hip_bfloat16* p, const float* v) {
        uint2 t;
        t.x = aadg_f2bf_pk(v[0], v[1]);
        t.y = aadg_f2bf_pk(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = t;
    }
    static __device__ __forceinline__ void store8(__hip_bfloat16* p, const float* v) {
        uint4 t;
        t.x = aadg_f2bf_pk(v[0], v[1]);
        t.y = aadg_f2bf_pk(v[2], v[3]);
        t.z = aadg_f2bf_pk(v[4], v[5]);
        t.w = aadg_f2bf_pk(v[6], v[7]);
        *reinterpret_cast<uint4*>(p) = t;
    }
};

// forward: one lane = 4 consecutive outputs of one row (8 input columns + the one to their left, 3 input rows)
template <typename T, bool IDX>
__global__ __launch_bounds__(256) void k_maxpool_fwd(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int H,
                                                     int W, int Ho, int Wo, long long quads) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= quads) return;
    const int w4 = Wo / 4;
    const int cg = (int)(q % w4);
    const long long t = q / w4;
    const int i = (int)(t % Ho);
    const long long plane = t / Ho;
    const T* px = x + (size_t)plane * H * W;
    float out[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    uint32_t at[4] = {4u, 4u, 4u, 4u};                       // the centre is always inside the image
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int r = 2 * i - 1 + a;
        if (r < 0 || r >= H) continue;
        const T* row = px + (size_t)r * W + 8 * cg;
        float v[9];
        Px<T>::load8(row, v + 1);
        v[0] = cg > 0 ? Px<T>::load1(row - 1) : -INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float e = v[2 * k + b];
                if (e > out[k]) { out[k] = e; at[k] = 3 * a + b; }
            }
    }
    const size_t o = (size_t)plane * Ho * Wo + (size_t)i * Wo + 4 * cg;
    Px<T>::store4(y + o, out);
    if (IDX) *reinterpret_cast<uint32_t*>(idx + o) = at[0] | (at[1] << 8) | (at[2] << 16) | (at[3] << 24);
}

// backward: one lane = 8 consecutive input columns c0 .. c0+7 of input row r.  Output (i, j) covers input rows 2i-1 .. 2i+1
// and columns 2j-1 .. 2j+1, so row r is covered by output row r >> 1 (window row 1 or 2) and, for odd r, by (r + 1) >> 1
// (window row 0); likewise columns: the 8 columns see outputs j0 .. j0+4 with j0 = c0 / 2.
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool_bwd(const uint8_t* __restrict__ idx, const T* __restrict__ dy, T* __restrict__ dx,
                                                     int H, int W, int Ho, int Wo, long long octs) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= octs) return;
    const int w8 = W / 8;
    const int cg = (int)(q % w8);
    const long long t = q / w8;
    const int r = (int)(t % H);
    const long long plane = t / H;
    const int j0 = 4 * cg;
    const bool has5 = j0 + 4 < Wo;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int i = (r + s) >> 1;
        const int a = r - 2 * i + 1;                         // window row of input row r in output row i: s = 0 -> 1 or 2, s = 1 -> 0
        if ((s == 1 && !(r & 1)) || i >= Ho) continue;
        const size_t o = (size_t)plane * Ho * Wo + (size_t)i * Wo + j0;
        const uint32_t w = *reinterpret_cast<const uint32_t*>(idx + o);
        float g[5];
        {
            float g4[4];
            Px<T>::load4(dy + o, g4);
            g[0] = g4[0]; g[1] = g4[1]; g[2] = g4[2]; g[3] = g4[3];
        }
        uint32_t p[5] = {w & 255u, (w >> 8) & 255u, (w >> 16) & 255u, w >> 24, 255u};
        g[4] = 0.0f;
        if (has5) { p[4] = idx[o + 4]; g[4] = Px<T>::load1(dy + o + 4); }
        const uint32_t base = 3u * (uint32_t)a;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = k >> 1;                             // output whose window has this column at position 1 (even k) / 2 (odd k)
            if (p[j] == base + 1u + (uint32_t)(k & 1)) acc[k] += g[j];
            if ((k & 1) && p[j + 1] == base) acc[k] += g[j + 1];   // odd columns are also the left edge of the next window
        }
    }
    Px<T>::store8(dx + (size_t)plane * H * W + (size_t)r * W + 8 * cg, acc);
}

inline bool mp_ok(long long planes, int H, int W) { return planes > 0 && H >= 2 && W >= 8 && (W % 8) == 0; }

template <typename T>
int mp_forward(const T* x, T* y, uint8_t* idx, int planes, int H, int W, hipStream_t st) {
    const int Ho = (H - 1) / 2 + 1, Wo = W / 2;
    const long long quads = (long long)planes * Ho * (Wo / 4);
    const dim3 grid((unsigned)((quads + 255) / 256));
    if (idx != nullptr) hipLaunchKernelGGL((k_maxpool_fwd<T, true>), grid, dim3(256), 0, st, x, y, idx, H, W, Ho, Wo, quads);
    else hipLaunchKernelGGL((k_maxpool_fwd<T, false>), grid, dim3(256), 0, st, x, y, idx, H, W, Ho, Wo, quads);
    AADG_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int mp_backward(const uint8_t* idx, const T* dy, T* dx, int planes, int H, int W, hipStream_t st) {
    const int Ho = (H - 1) / 2 + 1, Wo = W / 2;
    const long long octs = (long long)planes * H * (W / 8);
    hipLaunchKernelGGL((k_maxpool_bwd<T>), dim3((unsigned)((octs + 255) / 256)), dim3(256), 0, st, idx, dy, dx, H, W, Ho, Wo, octs);
    AADG_LAUNCH_CHECK();
    return 0;
}

inline bool mp_fits(int planes, int H, int W) {             // one-dimensional grids of 256-lane workgroups
    return (long long)planes * H * (W / 8) <= 0x7FFFFFFFLL * 256;
}

}  // namespace

extern "C" int aadg_maxpool3x3s2_supported(int H, int W) { return mp_ok(1, H, W) ? 1 : 0; }

extern "C" size_t aadg_maxpool3x3s2_index_bytes(int planes, int H, int W) {
    if (!mp_ok(planes, H, W)) return 0;
    return (size_t)planes * (size_t)((H - 1) / 2 + 1) * (size_t)(W / 2);
}

extern "C" int aadg_maxpool3x3s2_forward(const void* x, void* y, void* index, int planes, int H, int W, int dtype, void* stream) {
    if (x == nullptr || y == nullptr || ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0) || (((uintptr_t)index) & 3u) != 0) return AADG_E_BADARG;
    if (!mp_ok(planes, H, W) || !mp_fits(planes, H, W)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) return mp_forward<float>((const float*)x, (float*)y, (uint8_t*)index, planes, H, W, st);
    if (dtype == 1) return mp_forward<__hip_bfloat16>((const __hip_bfloat16*)x, (__hip_bfloat16*)y, (uint8_t*)index, planes, H, W, st);
    return AADG_E_BADARG;
}

extern "C" int aadg_maxpool3x3s2_backward(const void* index, const void* dy, void* dx, int planes, int H, int W, int dtype, void* stream) {
    if (index == nullptr || dy == nullptr || dx == nullptr || (((uintptr_t)dx) & 15u) != 0 || ((((uintptr_t)index) & 3u) != 0) ||
        (((uintptr_t)dy) & 15u) != 0)
        return AADG_E_BADARG;
    if (!mp_ok(planes, H, W) || !mp_fits(planes, H, W)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) return mp_backward<float>((const uint8_t*)index, (const float*)dy, (float*)dx, planes, H, W, st);
    if (dtype == 1)
        return mp_backward<__hip_bfloat16>((const uint8_t*)index, (const __hip_bfloat16*)dy, (__hip_bfloat16*)dx, planes, H, W, st);
    return AADG_E_BADARG;
}

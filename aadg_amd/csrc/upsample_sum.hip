// Fused "resize to the stride-4 grid and add" of the all-MLP segmentation head (aadg_amd/models/segformer.py; the reference head
// resizes every stage's projection with mmseg's resize(..., mode='bilinear', align_corners=False) and concatenates,
// models/mmseg/models/decode_heads/segformer_head.py:66-80):
//
//     out[n, c, y, x] = full[n, c, y, x] + sum_i bilinear(low_i[n, c])(y, x)          i < n_low <= 3, align_corners = False
//
// ATen runs each resize as its own kernel at ~40-80 GB/s on these shapes ([48, 768, 128, 128]: 15 ms each) and each addition
// as another pass; here the output is written once (16-byte stores), `full` is read once and the small low-resolution planes
// come out of L2.  Backward: d full = d out; d low_i = the transposed interpolation, gathered (no atomics): one thread per
// low-resolution pixel walks the <= 2F x 2F outputs it feeds with the forward's exact weights.
//
// Arithmetic = ATen upsample_bilinear2d(align_corners=False): src = max(0, (dst + 0.5) * in / out - 0.5); i0 = int(src);
// i1 = i0 + (i0 < in - 1); l1 = src - i0; float accumulation.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

constexpr int US_MAX_LOW = 3;

struct LowPlanes {
    const void* p[US_MAX_LOW];
    int h[US_MAX_LOW], w[US_MAX_LOW];
    float sy[US_MAX_LOW], sx[US_MAX_LOW];      // in / out
    int n;
};

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<__hip_bfloat16>(const __hip_bfloat16* p) {
    return __uint_as_float((uint32_t)(*reinterpret_cast<const uint16_t*>(p)) << 16);
}

__device__ __forceinline__ void tap(float scale, int dst, int in_size, int* i0, int* i1, float* l1) {
    float src = ((float)dst + 0.5f) * scale - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    const int a = (int)src;
    *i0 = a < in_size - 1 ? a : in_size - 1;
    *i1 = *i0 + (*i0 < in_size - 1 ? 1 : 0);
    *l1 = src - (float)*i0;
}

// grid (ceil(H * ceil(W / 8) / 256), planes): thread <-> 8 consecutive output pixels of one row; consecutive threads walk the
// plane row by row, so every lane is busy whatever the width
template <typename T>
__global__ __launch_bounds__(256) void k_upsample_sum(const T* __restrict__ full, LowPlanes lows, T* __restrict__ out, int H, int W) {
    const size_t plane = blockIdx.y;
    const int w8 = (W + 7) >> 3;
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int y = v / w8, x0 = (v - y * w8) * 8;
    if (y >= H) return;
    float acc[8];
    const size_t o = plane * (size_t)H * W + (size_t)y * W + x0;
    const bool whole = x0 + 8 <= W && (W & 7) == 0;
    if (full != nullptr) {
        if (whole) {
            if (sizeof(T) == 2) {
                const uint4 t = *reinterpret_cast<const uint4*>(full + o);
                const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[2 * i] = __uint_as_float(u[i] << 16); acc[2 * i + 1] = __uint_as_float(u[i] & 0xFFFF0000u); }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = ldf(full + o + i);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = x0 + i < W ? ldf(full + o + i) : 0.0f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
    }
    for (int k = 0; k < lows.n; ++k) {
        const int h = lows.h[k], w = lows.w[k];
        const T* lp = reinterpret_cast<const T*>(lows.p[k]) + plane * (size_t)h * w;
        int y0, y1;
        float ly1;
        tap(lows.sy[k], y, h, &y0, &y1, &ly1);
        const float ly0 = 1.0f - ly1;
        const T* r0 = lp + (size_t)y0 * w;
        const T* r1 = lp + (size_t)y1 * w;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int xa, xb;
            float lx1;
            tap(lows.sx[k], min(x0 + i, W - 1), w, &xa, &xb, &lx1);
            const float lx0 = 1.0f - lx1;
            acc[i] += ly0 * (lx0 * ldf(r0 + xa) + lx1 * ldf(r0 + xb)) + ly1 * (lx0 * ldf(r1 + xa) + lx1 * ldf(r1 + xb));
        }
    }
    if (whole && sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(out + o) = make_uint4(aadg_f2bf_pk(acc[0], acc[1]), aadg_f2bf_pk(acc[2], acc[3]),
                                                        aadg_f2bf_pk(acc[4], acc[5]), aadg_f2bf_pk(acc[6], acc[7]));
    } else if (whole) {
        float* po = reinterpret_cast<float*>(out + o);
        *reinterpret_cast<float4*>(po) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(po + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    } else {
        for (int i = 0; i < 8 && x0 + i < W; ++i) {
            if (sizeof(T) == 2) *reinterpret_cast<uint16_t*>(out + o + i) = (uint16_t)aadg_f2bf_bits(acc[i]);
            else *reinterpret_cast<float*>(out + o + i) = acc[i];
        }
    }
}

// first output index whose first tap is >= i (taps are non-decreasing in the output index)
__device__ __forceinline__ int first_dst(float scale, int i, int in_size, int OUT) {
    if (i <= 0) return 0;
    int X = (int)(((float)i + 0.5f) / scale - 0.5f) - 2;
    X = X < 0 ? 0 : (X > OUT ? OUT : X);
    int a, b;
    float l;
    while (X < OUT) {
        tap(scale, X, in_size, &a, &b, &l);
        if (a >= i) break;
        ++X;
    }
    return X;
}

// d low[n, c, i, j] = sum over the outputs (Y, X) that read (i, j): grid (ceil(h * w / 256), planes)
template <typename T>
__global__ __launch_bounds__(256) void k_upsample_sum_bwd(const T* __restrict__ dout, T* __restrict__ dlow, int h, int w, int H, int W,
                                                          float sy, float sx) {
    const size_t plane = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= h * w) return;
    const int i = idx / w, j = idx - i * w;
    // outputs whose taps include row i: first tap i - 1 (second tap i) or first tap i
    const int Ya = first_dst(sy, i - 1, h, H), Yb = first_dst(sy, i + 1, h, H);
    const int Xa = first_dst(sx, j - 1, w, W), Xb = first_dst(sx, j + 1, w, W);
    const T* pd = dout + plane * (size_t)H * W;
    float acc = 0.0f;
    for (int Y = Ya; Y < Yb; ++Y) {
        int y0, y1;
        float ly1;
        tap(sy, Y, h, &y0, &y1, &ly1);
        const float wy = (y0 == i ? 1.0f - ly1 : 0.0f) + (y1 == i ? ly1 : 0.0f);
        if (wy == 0.0f) continue;
        float row = 0.0f;
        for (int X = Xa; X < Xb; ++X) {
            int xa, xb;
            float lx1;
            tap(sx, X, w, &xa, &xb, &lx1);
            const float wx = (xa == j ? 1.0f - lx1 : 0.0f) + (xb == j ? lx1 : 0.0f);
            row = fmaf(wx, ldf(pd + (size_t)Y * W + X), row);
        }
        acc = fmaf(wy, row, acc);
    }
    if (sizeof(T) == 2) *reinterpret_cast<uint16_t*>(dlow + plane * (size_t)h * w + idx) = (uint16_t)aadg_f2bf_bits(acc);
    else *reinterpret_cast<float*>(dlow + plane * (size_t)h * w + idx) = acc;
}

}  // namespace

/* out [planes, H, W] = full [planes, H, W] (or 0 when NULL) + sum_i bilinear(low_i [planes, h_i, w_i]), align_corners = False.
 * dtype 0 = float32, 1 = bfloat16 (all tensors); n_low <= 3. */
extern "C" int aadg_upsample_sum(const void* full, const void* const* lows, const int* low_h, const int* low_w, int n_low, void* out,
                                 int planes, int H, int W, int dtype, void* stream) {
    if (out == nullptr || planes <= 0 || H <= 0 || W <= 0 || n_low < 0 || n_low > US_MAX_LOW || (dtype != 0 && dtype != 1)) return AADG_E_BADARG;
    if (n_low > 0 && (lows == nullptr || low_h == nullptr || low_w == nullptr)) return AADG_E_BADARG;
    if ((((uintptr_t)out | (uintptr_t)full) & 15u) != 0) return AADG_E_BADARG;
    LowPlanes L;
    L.n = n_low;
    for (int k = 0; k < US_MAX_LOW; ++k) { L.p[k] = nullptr; L.h[k] = L.w[k] = 1; L.sy[k] = L.sx[k] = 1.0f; }
    for (int k = 0; k < n_low; ++k) {
        if (lows[k] == nullptr || low_h[k] <= 0 || low_w[k] <= 0) return AADG_E_BADARG;
        L.p[k] = lows[k]; L.h[k] = low_h[k]; L.w[k] = low_w[k];
        L.sy[k] = (float)low_h[k] / (float)H; L.sx[k] = (float)low_w[k] / (float)W;
    }
    hipStream_t st = (hipStream_t)stream;
    const int w8 = (W + 7) / 8;
    const size_t esz = dtype == 0 ? 4 : 2;
    for (int p0 = 0; p0 < planes; p0 += 65535) {
        const int np = planes - p0 < 65535 ? planes - p0 : 65535;
        LowPlanes Lp = L;
        for (int k = 0; k < n_low; ++k) Lp.p[k] = (const char*)L.p[k] + (size_t)p0 * L.h[k] * L.w[k] * esz;
        const char* f = full ? (const char*)full + (size_t)p0 * H * W * esz : nullptr;
        char* o = (char*)out + (size_t)p0 * H * W * esz;
        const dim3 g((H * w8 + 255) / 256, np);
        if (dtype == 0) hipLaunchKernelGGL(k_upsample_sum<float>, g, dim3(256), 0, st, (const float*)f, Lp, (float*)o, H, W);
        else hipLaunchKernelGGL(k_upsample_sum<__hip_bfloat16>, g, dim3(256), 0, st, (const __hip_bfloat16*)f, Lp, (__hip_bfloat16*)o, H, W);
        AADG_LAUNCH_CHECK();
    }
    return 0;
}

/* dlow [planes, h, w] = gradient of aadg_upsample_sum w.r.t. one low-resolution input, from dout [planes, H, W]. */
extern "C" int aadg_upsample_sum_backward(const void* dout, void* dlow, int planes, int h, int w, int H, int W, int dtype, void* stream) {
    if (dout == nullptr || dlow == nullptr || planes <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || (dtype != 0 && dtype != 1)) return AADG_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const size_t esz = dtype == 0 ? 4 : 2;
    for (int p0 = 0; p0 < planes; p0 += 65535) {
        const int np = planes - p0 < 65535 ? planes - p0 : 65535;
        const char* d = (const char*)dout + (size_t)p0 * H * W * esz;
        char* o = (char*)dlow + (size_t)p0 * h * w * esz;
        const dim3 g((h * w + 255) / 256, np);
        if (dtype == 0) hipLaunchKernelGGL(k_upsample_sum_bwd<float>, g, dim3(256), 0, st, (const float*)d, (float*)o, h, w, H, W, sy, sx);
        else hipLaunchKernelGGL(k_upsample_sum_bwd<__hip_bfloat16>, g, dim3(256), 0, st, (const __hip_bfloat16*)d, (__hip_bfloat16*)o, h, w, H, W, sy, sx);
        AADG_LAUNCH_CHECK();
    }
    return 0;
}

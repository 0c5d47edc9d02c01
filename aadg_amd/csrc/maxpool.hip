// 3x3 / stride 2 / padding 1 max-pooling over NCHW planes (the ResNet stem pool, [N, 64, h/2, w/2] -> h/4) forward and
// backward.  The backward recomputes the arg-max from the saved input instead of reading an int64 index tensor
// (8 bytes per output in the library version): one workgroup stages a 19 x 131 input patch in LDS, scatters the
// 9 x 65 output gradients of its tile (+ one halo row / column of outputs) into an LDS gradient patch and writes its
// own 16 x 128 input region with 16-byte stores.
//
// Arithmetic = ATen max_pool2d: padding never wins, ties go to the first element in row-major window order.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

template <typename T> struct Px;
template <> struct Px<float> {
    static __device__ __forceinline__ void load8(const float* p, float* v) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ float load1(const float* p) { return *p; }
    static __device__ __forceinline__ void store4(float* p, const float* v) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    static __device__ __forceinline__ void store8(float* p, const float* v) { store4(p, v); store4(p + 4, v + 4); }
};
template <> struct Px<__hip_bfloat16> {
    static __device__ __forceinline__ void load8(const __hip_bfloat16* p, float* v) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
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
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool_fwd(const T* __restrict__ x, T* __restrict__ y, int H, int W, int Ho, int Wo,
                                                     long long quads) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= quads) return;
    const int w4 = Wo / 4;
    const int cg = (int)(q % w4);
    const long long t = q / w4;
    const int i = (int)(t % Ho);
    const long long plane = t / Ho;
    const T* px = x + (size_t)plane * H * W;
    float out[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int r = 2 * i - 1 + a;
        if (r < 0 || r >= H) continue;
        const T* row = px + (size_t)r * W + 8 * cg;
        float v[8];
        Px<T>::load8(row, v);
        const float left = cg > 0 ? Px<T>::load1(row - 1) : -INFINITY;
        out[0] = fmaxf(out[0], fmaxf(left, fmaxf(v[0], v[1])));
#pragma unroll
        for (int k = 1; k < 4; ++k) out[k] = fmaxf(out[k], fmaxf(v[2 * k - 1], fmaxf(v[2 * k], v[2 * k + 1])));
    }
    Px<T>::store4(y + (size_t)plane * Ho * Wo + (size_t)i * Wo + 4 * cg, out);
}

constexpr int BT_OH = 8, BT_OW = 64;                    // outputs whose input region this workgroup owns
constexpr int BT_IH = 2 * BT_OH, BT_IW = 2 * BT_OW;     // 16 x 128 input elements written
constexpr int BT_PH = BT_IH + 3, BT_PW = BT_IW + 3;     // staged patch: 19 x 131
constexpr int BT_LD = BT_PW + 1;                        // LDS row pitch (floats)

// grid (ceil(W / 128), ceil(H / 16), planes)
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool_bwd(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int H,
                                                     int W, int Ho, int Wo) {
    __shared__ float xL[BT_PH * BT_LD];
    __shared__ float dL[BT_PH * BT_LD];
    const size_t plane = blockIdx.z;
    const T* px = x + plane * (size_t)H * W;
    const int row0 = blockIdx.y * BT_IH - 1, col0 = blockIdx.x * BT_IW - 1;     // image coordinates of patch (0, 0)
    const int tid = threadIdx.x;
    // this lane's output gradients (up to 3 of the 9 x 65 outputs of the tile + halo): issued first, so that their
    // latency overlaps the staging of the input patch instead of following the barrier
    const int oi0 = blockIdx.y * BT_OH, oj0 = blockIdx.x * BT_OW;
    const T* pdy = dy + plane * (size_t)Ho * Wo;
    constexpr int NOUT = (BT_OH + 1) * (BT_OW + 1), PER = (NOUT + 255) / 256;
    float gq[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int o = tid + 256 * q;
        const int oi = o / (BT_OW + 1), oj = o - oi * (BT_OW + 1);
        const int i = oi0 + oi, j = oj0 + oj;
        gq[q] = (o < NOUT && i < Ho && j < Wo) ? Px<T>::load1(pdy + (size_t)i * Wo + j) : 0.0f;
    }
    for (int i = tid; i < BT_PH * BT_LD; i += 256) dL[i] = 0.0f;
    // aligned middle of every patch row: 16 vectors of 8 elements; then the 3 edge columns
    for (int i = tid; i < BT_PH * 16; i += 256) {
        const int rr = i >> 4, v8 = i & 15;
        const int r = row0 + rr, c = col0 + 1 + 8 * v8;
        float v[8];
        if (r >= 0 && r < H && c < W) Px<T>::load8(px + (size_t)r * W + c, v);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = -INFINITY;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) xL[rr * BT_LD + 1 + 8 * v8 + k] = v[k];
    }
    for (int i = tid; i < BT_PH * 3; i += 256) {
        const int rr = i / 3, e = i - rr * 3;
        const int cc = e == 0 ? 0 : BT_IW + e;            // patch columns 0, 129, 130
        const int r = row0 + rr, c = col0 + cc;
        xL[rr * BT_LD + cc] = (r >= 0 && r < H && c >= 0 && c < W) ? Px<T>::load1(px + (size_t)r * W + c) : -INFINITY;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int o = tid + 256 * q;
        if (o >= NOUT) continue;
        const int oi = o / (BT_OW + 1), oj = o - oi * (BT_OW + 1);
        const int i = oi0 + oi, j = oj0 + oj;
        if (i >= Ho || j >= Wo) continue;
        const float* w = xL + (2 * oi) * BT_LD + 2 * oj;     // window rows 2i-1..2i+1 -> patch rows 2oi..2oi+2
        float best = -INFINITY;
        int at = BT_LD + 1;                                  // the centre is always inside the image
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float v = w[a * BT_LD + b];
                if (v > best) { best = v; at = a * BT_LD + b; }
            }
        __hip_atomic_fetch_add(&dL[(2 * oi) * BT_LD + 2 * oj + at], gq[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    {
        const int rr = tid >> 4, v8 = tid & 15;              // 16 rows x 16 vectors: the owned region
        const int r = blockIdx.y * BT_IH + rr, c = blockIdx.x * BT_IW + 8 * v8;
        if (r < H && c < W) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = dL[(rr + 1) * BT_LD + 1 + 8 * v8 + k];
            Px<T>::store8(dx + plane * (size_t)H * W + (size_t)r * W + c, v);
        }
    }
}

inline bool mp_ok(int planes, int H, int W) { return planes > 0 && planes <= 65535 * 16 && H >= 2 && W >= 8 && (W % 8) == 0; }

template <typename T>
int mp_forward(const T* x, T* y, int planes, int H, int W, hipStream_t st) {
    const int Ho = (H - 1) / 2 + 1, Wo = W / 2;
    const long long quads = (long long)planes * Ho * (Wo / 4);
    hipLaunchKernelGGL((k_maxpool_fwd<T>), dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, x, y, H, W, Ho, Wo, quads);
    AADG_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int mp_backward(const T* x, const T* dy, T* dx, int planes, int H, int W, hipStream_t st) {
    const int Ho = (H - 1) / 2 + 1, Wo = W / 2;
    for (int p0 = 0; p0 < planes; p0 += 65535) {          // gridDim.z limit
        const int np = planes - p0 < 65535 ? planes - p0 : 65535;
        hipLaunchKernelGGL((k_maxpool_bwd<T>), dim3((W + BT_IW - 1) / BT_IW, (H + BT_IH - 1) / BT_IH, np), dim3(256), 0, st,
                           x + (size_t)p0 * H * W, dy + (size_t)p0 * Ho * Wo, dx + (size_t)p0 * H * W, H, W, Ho, Wo);
        AADG_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

extern "C" int aadg_maxpool3x3s2_supported(int H, int W) { return mp_ok(1, H, W) ? 1 : 0; }

extern "C" int aadg_maxpool3x3s2_forward(const void* x, void* y, int planes, int H, int W, int dtype, void* stream) {
    if (x == nullptr || y == nullptr || ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0)) return AADG_E_BADARG;
    if (!mp_ok(planes, H, W)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) return mp_forward<float>((const float*)x, (float*)y, planes, H, W, st);
    if (dtype == 1) return mp_forward<__hip_bfloat16>((const __hip_bfloat16*)x, (__hip_bfloat16*)y, planes, H, W, st);
    return AADG_E_BADARG;
}

extern "C" int aadg_maxpool3x3s2_backward(const void* x, const void* dy, void* dx, int planes, int H, int W, int dtype, void* stream) {
    if (x == nullptr || dy == nullptr || dx == nullptr || ((((uintptr_t)x | (uintptr_t)dx) & 15u) != 0)) return AADG_E_BADARG;
    if (!mp_ok(planes, H, W)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) return mp_backward<float>((const float*)x, (const float*)dy, (float*)dx, planes, H, W, st);
    if (dtype == 1)
        return mp_backward<__hip_bfloat16>((const __hip_bfloat16*)x, (const __hip_bfloat16*)dy, (__hip_bfloat16*)dx, planes, H, W, st);
    return AADG_E_BADARG;
}

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

// The same with one workgroup per plane: the low-resolution planes (64^2 + 32^2 + 16^2 values for a 128^2 output) and the column
// taps of every level sit in LDS, so a pixel costs 4 LDS reads per level instead of 4 scalar loads through L1 and the tap
// arithmetic of its column; the kernel above ran at 0.8 TB/s on [48 x 768, 128, 128] (3.05 ms).
constexpr int USP_MAX_LOW_ELEMS = 64 * 64 + 32 * 32 + 16 * 16 + 64;
constexpr int USP_MAX_W = 256;
template <typename T>
__global__ __launch_bounds__(256) void k_upsample_sum_plane(const T* __restrict__ full, LowPlanes lows, T* __restrict__ out, int H, int W) {
    __shared__ float lowv[USP_MAX_LOW_ELEMS];
    __shared__ int xa_s[US_MAX_LOW][USP_MAX_W];
    __shared__ float xl_s[US_MAX_LOW][USP_MAX_W];
    const size_t plane = blockIdx.x;
    const int tid = threadIdx.x;
    int loff[US_MAX_LOW + 1];
    loff[0] = 0;
#pragma unroll
    for (int k = 0; k < US_MAX_LOW; ++k) loff[k + 1] = loff[k] + (k < lows.n ? lows.h[k] * lows.w[k] : 0);
    for (int k = 0; k < lows.n; ++k) {
        const int hw = lows.h[k] * lows.w[k];
        const T* lp = reinterpret_cast<const T*>(lows.p[k]) + plane * (size_t)hw;
        for (int v = tid; v < hw; v += 256) lowv[loff[k] + v] = ldf(lp + v);
        for (int X = tid; X < W; X += 256) {
            int a, b;
            float l;
            tap(lows.sx[k], X, lows.w[k], &a, &b, &l);
            xa_s[k][X] = a | ((b - a) << 16);
            xl_s[k][X] = l;
        }
    }
    __syncthreads();
    const int w8 = W >> 3, chunks = H * w8;
    const T* pf = full != nullptr ? full + plane * (size_t)H * W : nullptr;
    T* po = out + plane * (size_t)H * W;
    for (int v = tid; v < chunks; v += 256) {
        const int y = v / w8, x0 = (v - y * w8) * 8;
        float acc[8];
        if (pf != nullptr) {
            if (sizeof(T) == 2) {
                const uint4 t = *reinterpret_cast<const uint4*>(pf + (size_t)8 * v);
                const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[2 * i] = __uint_as_float(u[i] << 16); acc[2 * i + 1] = __uint_as_float(u[i] & 0xFFFF0000u); }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = ldf(pf + (size_t)8 * v + i);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
        }
        for (int k = 0; k < lows.n; ++k) {
            int y0, y1;
            float ly1;
            tap(lows.sy[k], y, lows.h[k], &y0, &y1, &ly1);
            const float ly0 = 1.0f - ly1;
            const float* r0 = lowv + loff[k] + y0 * lows.w[k];
            const float* r1 = lowv + loff[k] + y1 * lows.w[k];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pk = xa_s[k][x0 + i];
                const int xa = pk & 0xFFFF, xb = xa + (pk >> 16);
                const float lx1 = xl_s[k][x0 + i], lx0 = 1.0f - lx1;
                acc[i] += ly0 * (lx0 * r0[xa] + lx1 * r0[xb]) + ly1 * (lx0 * r1[xa] + lx1 * r1[xb]);
            }
        }
        if (sizeof(T) == 2) {
            *reinterpret_cast<uint4*>(po + (size_t)8 * v) = make_uint4(aadg_f2bf_pk(acc[0], acc[1]), aadg_f2bf_pk(acc[2], acc[3]),
                                                                        aadg_f2bf_pk(acc[4], acc[5]), aadg_f2bf_pk(acc[6], acc[7]));
        } else {
            float* pof = reinterpret_cast<float*>(po + (size_t)8 * v);
            *reinterpret_cast<float4*>(pof) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(pof + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
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

// ---- all low-resolution gradients in ONE pass over d out -------------------------------------------------------------------
// One workgroup per plane: the H x W gradient plane is read once (16-byte loads) into LDS as float32 bits of bfloat16 / float32, then
// for every low-resolution level the transposed interpolation is applied SEPARABLY out of LDS: tmp[Y][j] = sum_X wx(j, X) d[Y][X]
// (<= 2F + 1 terms), d low[i][j] = sum_Y wy(i, Y) tmp[Y][j].  The per-level kernel above reads the plane once per level with one
// thread per low-resolution pixel and (2F)^2 scalar loads each: 3.3 ms per level on [48 x 768, 128, 128] (0.37 TB/s).
// Tap tables (first tap + weight of every output column / row, first output of every low column / row) are built per level in LDS.
struct LowOut {
    void* p[US_MAX_LOW];
    int h[US_MAX_LOW], w[US_MAX_LOW];
    int n;
};
constexpr int USB_MAX_HW = 128 * 128;      // plane elements that fit LDS as float
constexpr int USB_MAX_TMP = 128 * 64;      // H * w of the largest level
constexpr int USB_NT = 512;                // threads per plane (two planes per CU: 16 waves to hide the LDS latencies)
constexpr int USB_KT = 17;                 // outputs that read one low-resolution column / row: <= 2 F + 1, F <= 8

// x pass of k_upsample_sum_bwd_all: tmp[Y][j] = sum over the <= KT outputs X that read column j of their weight times d[Y][X].
// Thread <-> column j (fixed) and every rpp-th row: the weights sit in registers, a tap is one LDS read + one fma.
template <typename T, int KT>
__device__ __forceinline__ void xpass(const T* d, float* tmp, const int* xa, const float* xl, const int* first, int H, int W, int w) {
    const int tid = threadIdx.x;
    const int rpp = USB_NT / w, jj = tid % w, y0 = tid / w;
    if (y0 >= rpp) return;
    int X0 = first[jj];
    const int X1 = first[jj + 2];
    // the outputs clamped at the left border have first tap 0 and second-tap weight 0: they sit in column 1's range without
    // contributing -- skip them, or that range exceeds 2 F + 1
    while (X0 < X1 && !(xa[X0] == jj || (xa[X0] + 1 == jj && xl[X0] != 0.0f))) ++X0;
    const int nx = X1 - X0;
    float wt[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        wt[t] = 0.0f;
        if (t < nx) {
            const int a = xa[X0 + t];
            const float l = xl[X0 + t];
            wt[t] = a == jj ? 1.0f - l : (a + 1 == jj ? l : 0.0f);
        }
    }
    const int xlast = W - 1 - X0;
#pragma unroll 2
    for (int Y = y0; Y < H; Y += rpp) {
        const T* row = d + Y * W + X0;
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < KT; ++t) acc = fmaf(wt[t], ldf(row + min(t, xlast)), acc);
        tmp[Y * w + jj] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(USB_NT) void k_upsample_sum_bwd_all(const T* __restrict__ dout, LowOut lows, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* tmp = sm;                             // [H][w]
    int* xa = reinterpret_cast<int*>(tmp + USB_MAX_TMP);     // [max(H, W)] first tap of an output index (x pass, then reused for y)
    float* xl = reinterpret_cast<float*>(xa + 256);          // its second-tap weight
    int* first = reinterpret_cast<int*>(xl + 256);           // [w + 2]: first[j] = first output whose first tap is >= j - 1
    T* d = reinterpret_cast<T*>(first + 264);                // [H][W], kept in the tensor's own type (bfloat16: 32 KB for 128 x 128)
    const size_t plane = blockIdx.x;
    const int tid = threadIdx.x, HW = H * W;
    const T* pd = dout + plane * (size_t)HW;
    constexpr int V = 16 / sizeof(T);
    if ((HW % V) == 0) {
        const int nv = HW / V;
        for (int v0 = tid; v0 < nv; v0 += 8 * USB_NT) {           // 8 loads in flight per thread
            uint4 r[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] = v0 + USB_NT * q < nv ? reinterpret_cast<const uint4*>(pd)[v0 + USB_NT * q] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) if (v0 + USB_NT * q < nv) reinterpret_cast<uint4*>(d)[v0 + USB_NT * q] = r[q];
        }
    } else {
        for (int v = tid; v < HW; v += USB_NT) d[v] = pd[v];
    }
    for (int k = 0; k < lows.n; ++k) {
        const int h = lows.h[k], w = lows.w[k];
        const float sy = (float)h / (float)H, sx = (float)w / (float)W;
        __syncthreads();                         // d is loaded / the previous level is done with the tables and tmp
        for (int X = tid; X < W; X += USB_NT) { int a, b; float l; tap(sx, X, w, &a, &b, &l); xa[X] = a; xl[X] = b > a ? l : 0.0f; }
        for (int j = tid; j <= w + 1; j += USB_NT) first[j] = first_dst(sx, j - 1, w, W);       // first[j] = first X with tap >= j - 1
        __syncthreads();
        // x pass: tmp[Y][j] = sum over X in [first[j], first[j + 2]) of the weight X gives to column j.  Thread <-> column j (fixed)
        // and every rpp-th row: its <= USB_KT weights sit in registers, a tap is one LDS read + one fma
        {
            const int kt = 2 * ((W + w - 1) / w) + 1;        // uniform bound on the outputs that feed one column
            if (kt <= 5) xpass<T, 5>(d, tmp, xa, xl, first, H, W, w);
            else if (kt <= 9) xpass<T, 9>(d, tmp, xa, xl, first, H, W, w);
            else xpass<T, USB_KT>(d, tmp, xa, xl, first, H, W, w);
        }
        __syncthreads();
        for (int Y = tid; Y < H; Y += USB_NT) { int a, b; float l; tap(sy, Y, h, &a, &b, &l); xa[Y] = a; xl[Y] = b > a ? l : 0.0f; }
        for (int i = tid; i <= h + 1; i += USB_NT) first[i] = first_dst(sy, i - 1, h, H);
        __syncthreads();
        T* po = reinterpret_cast<T*>(lows.p[k]) + plane * (size_t)h * w;
        for (int v = tid; v < h * w; v += USB_NT) {
            const int i = v / w, j = v - i * w;
            float acc = 0.0f;
            for (int Y = first[i]; Y < first[i + 2]; ++Y) {
                const int a = xa[Y];
                const float l = xl[Y];
                const float wy = a == i ? 1.0f - l : (a + 1 == i ? l : 0.0f);
                acc = fmaf(wy, tmp[Y * w + j], acc);
            }
            if (sizeof(T) == 2) *reinterpret_cast<uint16_t*>(po + v) = (uint16_t)aadg_f2bf_bits(acc);
            else *reinterpret_cast<float*>(po + v) = acc;
        }
    }
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
    int low_elems = 0;
    for (int k = 0; k < n_low; ++k) low_elems += low_h[k] * low_w[k];
    if (n_low > 0 && (W & 7) == 0 && W <= USP_MAX_W && low_elems <= USP_MAX_LOW_ELEMS && (long long)H * W <= 256 * 256) {
        // one workgroup per plane, low-resolution planes and column taps in LDS
        if (dtype == 0) hipLaunchKernelGGL(k_upsample_sum_plane<float>, dim3(planes), dim3(256), 0, st, (const float*)full, L, (float*)out, H, W);
        else hipLaunchKernelGGL(k_upsample_sum_plane<__hip_bfloat16>, dim3(planes), dim3(256), 0, st, (const __hip_bfloat16*)full, L,
                                (__hip_bfloat16*)out, H, W);
        AADG_LAUNCH_CHECK();
        return 0;
    }
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

extern "C" int aadg_upsample_sum_backward_all_supported(int H, int W, const int* low_h, const int* low_w, int n_low) {
    if (H <= 0 || W <= 0 || H > 256 || W > 256 || H * W > USB_MAX_HW || n_low < 1 || n_low > US_MAX_LOW || low_h == nullptr || low_w == nullptr)
        return 0;
    for (int k = 0; k < n_low; ++k)
        if (low_h[k] <= 0 || low_w[k] <= 0 || low_h[k] > H || low_w[k] > W || low_h[k] + 2 > 256 || low_w[k] + 2 > 256 || H * low_w[k] > USB_MAX_TMP ||
            2 * ((W + low_w[k] - 1) / low_w[k]) + 1 > USB_KT)
            return 0;
    return 1;
}

/* all dlow_i [planes, h_i, w_i] of aadg_upsample_sum from dout [planes, H, W] in one pass over dout (one workgroup per plane,
 * plane resident in LDS: H * W <= 128 * 128; else use aadg_upsample_sum_backward per level) */
extern "C" int aadg_upsample_sum_backward_all(const void* dout, void* const* dlows, const int* low_h, const int* low_w, int n_low, int planes,
                                              int H, int W, int dtype, void* stream) {
    if (dout == nullptr || dlows == nullptr || planes <= 0 || (dtype != 0 && dtype != 1)) return AADG_E_BADARG;
    if (!aadg_upsample_sum_backward_all_supported(H, W, low_h, low_w, n_low)) return AADG_E_UNSUPPORTED;
    if (((uintptr_t)dout & 15u) != 0) return AADG_E_BADARG;
    LowOut L;
    L.n = n_low;
    for (int k = 0; k < US_MAX_LOW; ++k) { L.p[k] = nullptr; L.h[k] = L.w[k] = 1; }
    for (int k = 0; k < n_low; ++k) {
        if (dlows[k] == nullptr) return AADG_E_BADARG;
        L.p[k] = dlows[k]; L.h[k] = low_h[k]; L.w[k] = low_w[k];
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t tables = (size_t)USB_MAX_TMP * 4 + 256 * 4 * 2 + 264 * 4;
    const size_t lds = tables + (size_t)H * W * (dtype == 0 ? 4 : 2);
    static bool attr_set = false;
    if (!attr_set) {
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_upsample_sum_bwd_all<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(tables + (size_t)USB_MAX_HW * 4)));
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_upsample_sum_bwd_all<__hip_bfloat16>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(tables + (size_t)USB_MAX_HW * 2)));
        attr_set = true;
    }
    if (dtype == 0) hipLaunchKernelGGL(k_upsample_sum_bwd_all<float>, dim3(planes), dim3(USB_NT), lds, st, (const float*)dout, L, H, W);
    else hipLaunchKernelGGL(k_upsample_sum_bwd_all<__hip_bfloat16>, dim3(planes), dim3(USB_NT), lds, st, (const __hip_bfloat16*)dout, L, H, W);
    AADG_LAUNCH_CHECK();
    return 0;
}

// Bilinear up-sampling (align_corners = True), NCHW, float32 or bfloat16 -- the two x4 up-samplings of the
// DeepLabV3+ decoder ([N,256,h/16,w/16] -> h/4 and the K-channel logits h/4 -> h).  They sit between the hot
// path's producer (augmentation) and consumer (BCE/Dice kernel); the ATen kernel they replace runs at ~40 GB/s on
// this shape (7-15 ms per call at N=144, 512x512), this one streams the output with one 16-byte store per lane.
//
// Arithmetic = ATen upsample_bilinear2d (float accumulation): s = (in-1)/(out-1); src = s*dst; i0 = int(src);
// i1 = i0 + (i0 < in-1); l1 = src - i0; l0 = 1 - l1; v = l0y*(l0x*a + l1x*b) + l1y*(l0x*c + l1x*d).
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    using type = float4;
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void ld4(const float* p, float* v) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static constexpr int V16 = 4;                       // elements per 16-byte load
    static __device__ __forceinline__ void ld16(const float* p, float* v) { ld4(p, v); }
    static __device__ __forceinline__ void st4(float* p, float a, float b, float c, float d, bool stream = false) {
        aadg_store_out(p, make_float4(a, b, c, d), stream);
    }
    static __device__ __forceinline__ void st8(float* p, const float* v, bool stream = false) {
        st4(p, v[0], v[1], v[2], v[3], stream); st4(p + 4, v[4], v[5], v[6], v[7], stream);
    }
    static __device__ __forceinline__ void st1(float* p, float a) { *p = a; }
};
template <> struct Vec4<__hip_bfloat16> {
    static __device__ __forceinline__ float ld(const __hip_bfloat16* p) { return __bfloat162float(*p); }
    static __device__ __forceinline__ void ld4(const __hip_bfloat16* p, float* v) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
    }
    static constexpr int V16 = 8;
    static __device__ __forceinline__ void ld16(const __hip_bfloat16* p, float* v) {
        const uint4 t = aadg_load_stream(p);
        const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(u[i] << 16); v[2 * i + 1] = __uint_as_float(u[i] & 0xFFFF0000u); }
    }
    static __device__ __forceinline__ void st4(__hip_bfloat16* p, float a, float b, float c, float d, bool stream = false) {
        uint2 v;
        v.x = aadg_f2bf_pk(a, b);
        v.y = aadg_f2bf_pk(c, d);
        *reinterpret_cast<uint2*>(p) = v;
    }
    static __device__ __forceinline__ void st8(__hip_bfloat16* p, const float* v, bool stream = false) {      // one 16-byte store
        aadg_store_out(p, make_uint4(aadg_f2bf_pk(v[0], v[1]), aadg_f2bf_pk(v[2], v[3]), aadg_f2bf_pk(v[4], v[5]), aadg_f2bf_pk(v[6], v[7])), stream);
    }
    static __device__ __forceinline__ void st1(__hip_bfloat16* p, float a) { *p = __float2bfloat16(a); }
};

// grid (ceil(W/4/64) * rows-chunks, planes): a wave writes 256 consecutive output pixels of one row
template <typename T>
__global__ __launch_bounds__(256) void k_upsample(const T* __restrict__ in, T* __restrict__ out, int h, int w, int H, int W,
                                                  float sy, float sx, int rows_per_block, int C, long long out_img_stride,
                                                  int plane0) {
    const size_t plane = (size_t)plane0 + blockIdx.y;
    const T* pin = in + plane * (size_t)h * w;
    // the output may be a channel slice of a wider tensor (written straight into a concatenation buffer)
    T* po = out + (plane / C) * (size_t)out_img_stride + (plane % C) * (size_t)H * W;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int xgroups = (W + 255) / 256;
    const int xg = blockIdx.x % xgroups, yb = (blockIdx.x / xgroups) * rows_per_block;
    const int x0 = xg * 256 + lane * 4;
    if (x0 >= W) return;
    // per-lane x taps for 4 consecutive columns
    int xi0[4], xi1[4];
    float lx1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int X = min(x0 + k, W - 1);
        const float src = sx * (float)X;
        const int i0 = (int)src;
        xi0[k] = i0;
        xi1[k] = i0 + (i0 < w - 1 ? 1 : 0);
        lx1[k] = src - (float)i0;
    }
    const bool vec = (W & 3) == 0;
    for (int r = wv; r < rows_per_block; r += 4) {
        const int Y = yb + r;
        if (Y >= H) break;
        const float srcy = sy * (float)Y;
        const int y0 = (int)srcy;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const float ly1 = srcy - (float)y0, ly0 = 1.0f - ly1;
        const T* r0 = pin + (size_t)y0 * w;
        const T* r1 = pin + (size_t)y1 * w;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lx0 = 1.0f - lx1[k];
            const float a = Vec4<T>::ld(r0 + xi0[k]), b = Vec4<T>::ld(r0 + xi1[k]);
            const float c = Vec4<T>::ld(r1 + xi0[k]), d = Vec4<T>::ld(r1 + xi1[k]);
            v[k] = ly0 * (lx0 * a + lx1[k] * b) + ly1 * (lx0 * c + lx1[k] * d);
        }
        T* dst = po + (size_t)Y * W + x0;
        if (vec) Vec4<T>::st4(dst, v[0], v[1], v[2], v[3]);
        else
            for (int k = 0; k < 4 && x0 + k < W; ++k) Vec4<T>::st1(dst + k, v[k]);
    }
}

// Small input planes (h * w <= UP_LDS_MAX, e.g. the 32 x 32 ASPP map): the plane is staged once in LDS as float, so the
// taps of a lane's outputs are LDS reads instead of 2-byte global gathers.  An item = OPI consecutive outputs of one row
// (16 bytes of T when the width allows); the items of `rows` output rows are dealt round-robin to the 256 lanes, so every
// lane is busy whatever the width.  grid (ceil(H / rows), planes); rows = H (one staging per plane) for outputs <= 128 x 128.
constexpr int UP_LDS_MAX = 4096, UP_ROWS = 32, UP_WHOLE_PLANE = 16384;

template <typename T, int OPI>
__global__ __launch_bounds__(256) void k_upsample_lds(const T* __restrict__ in, T* __restrict__ out, int h, int w, int H, int W,
                                                      float sy, float sx, int C, long long out_img_stride, int plane0, int rows, int stream) {
    __shared__ __attribute__((aligned(16))) float P[UP_LDS_MAX];
    const size_t plane = (size_t)plane0 + blockIdx.y;
    const T* pin = in + plane * (size_t)h * w;
    constexpr int V = Vec4<T>::V16;
    const int hw = h * w;
    if ((hw % V) == 0 && (((uintptr_t)pin) & 15u) == 0) {
        for (int i = threadIdx.x; i < hw / V; i += 256) {
            float v[V];
            Vec4<T>::ld16(pin + (size_t)i * V, v);
#pragma unroll
            for (int e = 0; e < V; e += 4) *reinterpret_cast<float4*>(P + (size_t)i * V + e) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
        }
    } else {
        for (int i = threadIdx.x; i < hw; i += 256) P[i] = Vec4<T>::ld(pin + i);
    }
    __syncthreads();
    // the output may be a channel slice of a wider tensor (written straight into a concatenation buffer)
    T* po = out + (plane / C) * (size_t)out_img_stride + (plane % C) * (size_t)H * W;
    const int Y0 = blockIdx.x * rows, nrows = min(H, Y0 + rows) - Y0;
    const int per_row = (W + OPI - 1) / OPI;
    const bool vec = (W % OPI) == 0;
    for (int item = threadIdx.x; item < nrows * per_row; item += 256) {
        const int yy = item / per_row, x0 = (item - yy * per_row) * OPI;
        const int Y = Y0 + yy;
        const float srcy = sy * (float)Y;
        const int y0 = (int)srcy;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const float ly1 = srcy - (float)y0, ly0 = 1.0f - ly1;
        const float* r0 = P + y0 * w;
        const float* r1 = P + y1 * w;
        float v[OPI];
#pragma unroll
        for (int k = 0; k < OPI; ++k) {
            const int X = min(x0 + k, W - 1);
            const float src = sx * (float)X;
            const int i0 = (int)src;
            const int i1 = i0 + (i0 < w - 1 ? 1 : 0);
            const float lx1 = src - (float)i0, lx0 = 1.0f - lx1;
            v[k] = ly0 * (lx0 * r0[i0] + lx1 * r0[i1]) + ly1 * (lx0 * r1[i0] + lx1 * r1[i1]);
        }
        T* dst = po + (size_t)Y * W + x0;
        if (vec) {
            if constexpr (OPI == 8) Vec4<T>::st8(dst, v, stream != 0); else Vec4<T>::st4(dst, v[0], v[1], v[2], v[3], stream != 0);
        } else {
            for (int k = 0; k < OPI && x0 + k < W; ++k) Vec4<T>::st1(dst + k, v[k]);
        }
    }
}

// ---- backward: dx[i][j] = sum_Y sum_X wy(Y, i) * wx(X, j) * dy[Y][X], separable, gathered (no atomics) ---------------
// The taps of an input row / column (which outputs touch it, with which weight) depend on the geometry only, so a tiny
// kernel tabulates them once per call (forward arithmetic re-derived exactly); the streaming kernel then stages the
// rectangle of dy that feeds an 8 x 32 input tile in LDS, reduces it along y with 16-byte LDS reads and along x.
constexpr int BT_I = 8, BT_J = 32, BT_MAXTAP = 16, BT_CAP = 12288;   // BT_CAP: staged floats per workgroup (48 KiB)
constexpr int TAP_STRIDE = BT_MAXTAP + 2;                            // floats per table row: first, count, weights

__device__ __forceinline__ int first_out_reaching(float s, int i, int OUT) {   // smallest X with (int)(s * X) >= i
    if (i <= 0) return 0;
    int X = s > 0.0f ? (int)((float)i / s) - 1 : OUT;
    if (X < 0) X = 0;
    if (X > OUT) X = OUT;
    while (X < OUT && (int)(s * (float)X) < i) ++X;
    return X;
}

// table[idx] = {first output, tap count, weights[BT_MAXTAP]} for idx < in_size; grid 2 (y axis, x axis)
__global__ __launch_bounds__(256) void k_upsample_bwd_taps(int h, int w, int H, int W, float sy, float sx, float* tab) {
    const bool isx = blockIdx.x == 1;
    const int in_size = isx ? w : h, OUT = isx ? W : H;
    const float s = isx ? sx : sy;
    float* t = tab + (isx ? (size_t)h * TAP_STRIDE : 0);
    for (int idx = threadIdx.x; idx < in_size; idx += 256) {
        const int a = first_out_reaching(s, idx - 1, OUT), b = first_out_reaching(s, idx + 1, OUT);
        int cnt = b - a;
        if (cnt > BT_MAXTAP) cnt = BT_MAXTAP;        // excluded by _supported()
        float* row = t + (size_t)idx * TAP_STRIDE;
        reinterpret_cast<int*>(row)[0] = a;
        reinterpret_cast<int*>(row)[1] = cnt;
        for (int k = 0; k < BT_MAXTAP; ++k) {
            const int X = a + k;
            const float src = s * (float)X;
            const int x0 = (int)src, x1 = x0 + (x0 < in_size - 1 ? 1 : 0);
            const float l1 = src - (float)x0;
            row[2 + k] = k < cnt ? (x0 == idx ? 1.0f - l1 : 0.0f) + (x1 == idx ? l1 : 0.0f) : 0.0f;
        }
    }
}

// grid (ceil(w / BT_J) * ceil(h / BT_I), planes); ldc = LDS pitch of the staged rectangle (multiple of 4)
template <typename T>
__global__ __launch_bounds__(256) void k_upsample_bwd(const T* __restrict__ dy, T* __restrict__ dx, int h, int w, int H, int W,
                                                      const float* __restrict__ tab, int ld_rows, int ldc, int C,
                                                      long long dy_img_stride, int plane0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float taps_y[BT_I * TAP_STRIDE];
    __shared__ float taps_x[BT_J * TAP_STRIDE];
    const size_t plane = (size_t)plane0 + blockIdx.y;
    const int tj = (w + BT_J - 1) / BT_J;
    const int i0 = (blockIdx.x / tj) * BT_I, j0 = (blockIdx.x % tj) * BT_J;
    const int i1 = min(h, i0 + BT_I), j1 = min(w, j0 + BT_J);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* ty = tab + (size_t)i0 * TAP_STRIDE;
    const float* tx = tab + (size_t)h * TAP_STRIDE + (size_t)j0 * TAP_STRIDE;
    for (int t = tid; t < (i1 - i0) * TAP_STRIDE; t += 256) taps_y[t] = ty[t];
    for (int t = tid; t < (j1 - j0) * TAP_STRIDE; t += 256) taps_x[t] = tx[t];
    // rectangle of dy feeding this tile (uniform: scalar loads)
    const int Ylo = reinterpret_cast<const int*>(ty)[0];
    const int* ly = reinterpret_cast<const int*>(ty + (size_t)(i1 - 1 - i0) * TAP_STRIDE);
    const int Yhi = ly[0] + ly[1];
    const int Xlo = reinterpret_cast<const int*>(tx)[0] & ~3;          // 4-aligned start: float4 LDS reads line up
    const int* lx = reinterpret_cast<const int*>(tx + (size_t)(j1 - 1 - j0) * TAP_STRIDE);
    const int Xhi = lx[0] + lx[1];
    const int nr = Yhi - Ylo, nc = Xhi - Xlo;
    float* D = lds;                                   // [nr][ldc]
    float* V = lds + (size_t)ld_rows * ldc;           // [BT_I][ldc]: reduced along y
    // dy may be a channel slice of a wider tensor (the gradient of a concatenation): image stride given by the caller
    const T* pdy = dy + (plane / C) * (size_t)dy_img_stride + (plane % C) * (size_t)H * W;
    for (int rb = wv; rb < nr; rb += 16) {            // 4 rows x 3 column groups in flight per lane
        float v[4][3];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int r = rb + 4 * a, c = lane + 64 * b;
                v[a][b] = (r < nr && c < nc) ? Vec4<T>::ld(pdy + (size_t)(Ylo + r) * W + Xlo + c) : 0.0f;
            }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int r = rb + 4 * a, c = lane + 64 * b;
                if (r < nr && c < ldc) D[r * ldc + c] = v[a][b];      // columns in [nc, ldc) are zero padding
            }
    }
    __syncthreads();
    // along y: thread <-> (input row ii, 4 consecutive staged columns)
    for (int t = tid; t < BT_I * (ldc >> 2); t += 256) {
        const int ii = t / (ldc >> 2), c4 = (t - ii * (ldc >> 2)) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i0 + ii < i1) {
            const float* tp = taps_y + ii * TAP_STRIDE;
            const int yf = reinterpret_cast<const int*>(tp)[0] - Ylo, yn = reinterpret_cast<const int*>(tp)[1];
#pragma unroll
            for (int k = 0; k < BT_MAXTAP; ++k)        // fixed trip count: all LDS reads issue before the first wait
                if (k < yn) {
                    const float wgt = tp[2 + k];
                    const float4 d = *reinterpret_cast<const float4*>(D + (yf + k) * ldc + c4);
                    acc.x = fmaf(wgt, d.x, acc.x); acc.y = fmaf(wgt, d.y, acc.y); acc.z = fmaf(wgt, d.z, acc.z); acc.w = fmaf(wgt, d.w, acc.w);
                }
        }
        *reinterpret_cast<float4*>(V + ii * ldc + c4) = acc;
    }
    __syncthreads();
    {                                                  // along x: one output per thread (256 / BT_J == BT_I)
        const int jj = tid % BT_J, ii = tid / BT_J, i = i0 + ii, j = j0 + jj;
        if (i < i1 && j < j1) {
            const float* tp = taps_x + jj * TAP_STRIDE;
            const int xf = reinterpret_cast<const int*>(tp)[0] - Xlo, xn = reinterpret_cast<const int*>(tp)[1];
            const float* vrow = V + ii * ldc + xf;
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < BT_MAXTAP; ++k)
                if (k < xn) acc = fmaf(tp[2 + k], vrow[k], acc);
            Vec4<T>::st1(dx + plane * (size_t)h * w + (size_t)i * w + j, acc);
        }
    }
}

// Whole-plane variant for small outputs (H * W <= UPB_MAX_OUT, e.g. 128 x 128 -> 32 x 32): one workgroup per plane.  The
// reduction along y reads dy straight from global memory with 16-byte loads -- every tap of an item is an independent load
// (<= BT_MAXTAP in flight per lane), an output row feeds at most two input rows, so the second read hits L1 / L2 -- into
// V[h][W] in LDS (16 KiB at 32 x 128: eight workgroups per CU instead of the two a staged copy of dy allowed); then along x.
constexpr int UPB_MAX_OUT = 16384, UPB_MAX_IN_ROWS = 64;

template <typename T, int CPI>          // CPI = columns per item of the first pass (16 bytes of T, or 4)
__global__ __launch_bounds__(256) void k_upsample_bwd_plane(const T* __restrict__ dy, T* __restrict__ dx, int h, int w, int H, int W,
                                                            const float* __restrict__ tab, int C, long long dy_img_stride) {
    extern __shared__ __attribute__((aligned(16))) float V[];      // [h][W]
    const size_t plane = blockIdx.x;
    // dy may be a channel slice of a wider tensor (the gradient of a concatenation): image stride given by the caller
    const T* pdy = dy + (plane / C) * (size_t)dy_img_stride + (plane % C) * (size_t)H * W;
    const int tid = threadIdx.x;
    const float* ty = tab;
    const float* tx = tab + (size_t)h * TAP_STRIDE;
    const int wc = W / CPI;
    for (int t = tid; t < h * wc; t += 256) {          // along y: (input row i, CPI consecutive output columns)
        const int i = t / wc, c0 = (t - i * wc) * CPI;
        const float* tp = ty + (size_t)i * TAP_STRIDE;
        const int yf = reinterpret_cast<const int*>(tp)[0], yn = reinterpret_cast<const int*>(tp)[1];
        float acc[CPI];
#pragma unroll
        for (int e = 0; e < CPI; ++e) acc[e] = 0.f;
        const T* col = pdy + (size_t)yf * W + c0;
#pragma unroll
        for (int k = 0; k < BT_MAXTAP; ++k)
            if (k < yn) {
                const float wgt = tp[2 + k];
                float d[CPI];
                if (CPI == 4) Vec4<T>::ld4(col + (size_t)k * W, d); else Vec4<T>::ld16(col + (size_t)k * W, d);
#pragma unroll
                for (int e = 0; e < CPI; ++e) acc[e] = fmaf(wgt, d[e], acc[e]);
            }
#pragma unroll
        for (int e = 0; e < CPI; e += 4)
            *reinterpret_cast<float4*>(V + (size_t)i * W + c0 + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
    }
    __syncthreads();
    T* pdx = dx + plane * (size_t)h * w;
    for (int t = tid; t < h * w; t += 256) {           // along x
        const int i = t / w, j = t - i * w;
        const float* tp = tx + (size_t)j * TAP_STRIDE;
        const int xf = reinterpret_cast<const int*>(tp)[0], xn = reinterpret_cast<const int*>(tp)[1];
        const float* vrow = V + (size_t)i * W + xf;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < BT_MAXTAP; ++k)
            if (k < xn) acc = fmaf(tp[2 + k], vrow[k], acc);
        Vec4<T>::st1(pdx + t, acc);
    }
}

// rows / columns of dy a tile may need (upper bound used to size the LDS image)
inline int span_bound(int tile, float s, int OUT) {
    if (s <= 0.0f) return OUT;
    const int n = (int)((float)(tile + 1) / s) + 3;
    return n < OUT ? n : OUT;
}
inline int staged_pitch(int nc) { return ((nc + 3 + 3) & ~3) + 4; }   // + up to 3 alignment columns, multiple of 4, + pad

}  // namespace

/* in [N*C, h, w] contiguous -> out plane (n, c) at n * out_image_stride + c * H * W elements (out_image_stride = C * H * W for
 * a contiguous output; larger when `out` is a channel slice of a concatenation buffer) */
extern "C" int aadg_upsample_bilinear2d_strided(const void* in, void* out, int N, int C, int h, int w, int H, int W,
                                                long long out_image_stride, int dtype, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || out_image_stride < (long long)C * H * W)
        return AADG_E_BADARG;
    if (dtype != 0 && dtype != 1) return AADG_E_BADARG;
    if ((long long)N * C > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    const int planes = N * C;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    const bool lds_path = h * w <= UP_LDS_MAX;
    const int rows_per_block = 16;
    const int xgroups = (W + 255) / 256;
    // an output too large to stay cached until its consumer reads it is written with streaming stores
    const int stream_out = (size_t)planes * H * W * (dtype == 0 ? 4 : 2) > ((size_t)128 << 20) ? 1 : 0;
    for (int p0 = 0; p0 < planes; p0 += 65535) {          // gridDim.y limit
        const int np = planes - p0 < 65535 ? planes - p0 : 65535;
        if (lds_path) {
            const int rows = (long long)H * W <= UP_WHOLE_PLANE ? H : UP_ROWS;
            const dim3 gl((H + rows - 1) / rows, np);
            // 16-byte stores need the whole destination aligned: base pointer, image stride and plane size
            const bool wide = dtype == 1 && (W & 7) == 0 && (((uintptr_t)out) & 15u) == 0 && (out_image_stride & 7) == 0 && (((long long)H * W) & 7) == 0;
            if (dtype == 0)
                hipLaunchKernelGGL((k_upsample_lds<float, 4>), gl, dim3(256), 0, st, reinterpret_cast<const float*>(in),
                                   reinterpret_cast<float*>(out), h, w, H, W, sy, sx, C, out_image_stride, p0, rows, stream_out);
            else if (wide)
                hipLaunchKernelGGL((k_upsample_lds<__hip_bfloat16, 8>), gl, dim3(256), 0, st, reinterpret_cast<const __hip_bfloat16*>(in),
                                   reinterpret_cast<__hip_bfloat16*>(out), h, w, H, W, sy, sx, C, out_image_stride, p0, rows, stream_out);
            else
                hipLaunchKernelGGL((k_upsample_lds<__hip_bfloat16, 4>), gl, dim3(256), 0, st, reinterpret_cast<const __hip_bfloat16*>(in),
                                   reinterpret_cast<__hip_bfloat16*>(out), h, w, H, W, sy, sx, C, out_image_stride, p0, rows, stream_out);
        } else {
            const dim3 g(xgroups * ((H + rows_per_block - 1) / rows_per_block), np);
            if (dtype == 0)
                hipLaunchKernelGGL(k_upsample<float>, g, dim3(256), 0, st, reinterpret_cast<const float*>(in),
                                   reinterpret_cast<float*>(out), h, w, H, W, sy, sx, rows_per_block, C, out_image_stride, p0);
            else
                hipLaunchKernelGGL(k_upsample<__hip_bfloat16>, g, dim3(256), 0, st, reinterpret_cast<const __hip_bfloat16*>(in),
                                   reinterpret_cast<__hip_bfloat16*>(out), h, w, H, W, sy, sx, rows_per_block, C, out_image_stride, p0);
        }
        AADG_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int aadg_upsample_bilinear2d(const void* in, void* out, int planes, int h, int w, int H, int W, int dtype,
                                        void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0) return AADG_E_BADARG;
    return aadg_upsample_bilinear2d_strided(in, out, 1, planes, h, w, H, W, (long long)planes * H * W, dtype, stream);
}

extern "C" int aadg_upsample_bilinear2d_backward_supported(int h, int w, int H, int W) {
    if (h <= 0 || w <= 0 || H <= 0 || W <= 0) return 0;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    const int nr = span_bound(BT_I, sy, H), nc = span_bound(BT_J, sx, W);
    if (span_bound(1, sy, H) > BT_MAXTAP || span_bound(1, sx, W) > BT_MAXTAP) return 0;   // taps per input row / column
    const int ldc = staged_pitch(nc);
    if (ldc > 192) return 0;                                                             // 3 column groups of 64 lanes
    return (size_t)(nr + BT_I) * ldc <= (size_t)BT_CAP ? 1 : 0;
}

extern "C" size_t aadg_upsample_bilinear2d_backward_workspace_bytes(int h, int w) {
    return h > 0 && w > 0 ? (size_t)(h + w) * TAP_STRIDE * sizeof(float) : 0;
}

/* dx [N*C, h, w] (contiguous) = gradient of aadg_upsample_bilinear2d w.r.t. its input, from dy: plane (n, c) of dy starts at
 * n * dy_image_stride + c * H * W elements -- dy may be a channel slice of a wider tensor (the gradient of a concatenation). */
extern "C" int aadg_upsample_bilinear2d_backward_strided(const void* dy, void* dx, int N, int C, int h, int w, int H, int W,
                                                         long long dy_image_stride, int dtype, void* ws, size_t ws_bytes,
                                                         void* stream) {
    if (!dy || !dx || !ws || N <= 0 || C <= 0 || (long long)N * C > 65535LL * 64 || dy_image_stride < (long long)C * H * W)
        return AADG_E_BADARG;
    if (dtype != 0 && dtype != 1) return AADG_E_BADARG;
    if (!aadg_upsample_bilinear2d_backward_supported(h, w, H, W)) return AADG_E_UNSUPPORTED;
    if (ws_bytes < aadg_upsample_bilinear2d_backward_workspace_bytes(h, w)) return AADG_E_WORKSPACE;
    const int planes = N * C;
    const long long img_stride = dy_image_stride;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    float* tab = reinterpret_cast<float*>(ws);
    hipLaunchKernelGGL(k_upsample_bwd_taps, dim3(2), dim3(256), 0, st, h, w, H, W, sy, sx, tab);
    AADG_LAUNCH_CHECK();
    const size_t esz = dtype == 0 ? 4 : 2;
    if (H * W <= UPB_MAX_OUT && (size_t)h * W <= (size_t)UPB_MAX_OUT && (W & 3) == 0 && h <= UPB_MAX_IN_ROWS && (((uintptr_t)dy) & 15u) == 0 && (((size_t)H * W) % 8) == 0 &&
        ((size_t)img_stride * esz) % 16 == 0) {
        const size_t lds_p = (size_t)h * W * sizeof(float);                         // <= 64 KiB
        if (dtype == 0)
            hipLaunchKernelGGL((k_upsample_bwd_plane<float, 4>), dim3(planes), dim3(256), lds_p, st, reinterpret_cast<const float*>(dy),
                               reinterpret_cast<float*>(dx), h, w, H, W, (const float*)tab, C, img_stride);
        else if ((W & 7) == 0)
            hipLaunchKernelGGL((k_upsample_bwd_plane<__hip_bfloat16, 8>), dim3(planes), dim3(256), lds_p, st,
                               reinterpret_cast<const __hip_bfloat16*>(dy), reinterpret_cast<__hip_bfloat16*>(dx), h, w, H, W,
                               (const float*)tab, C, img_stride);
        else
            hipLaunchKernelGGL((k_upsample_bwd_plane<__hip_bfloat16, 4>), dim3(planes), dim3(256), lds_p, st,
                               reinterpret_cast<const __hip_bfloat16*>(dy), reinterpret_cast<__hip_bfloat16*>(dx), h, w, H, W,
                               (const float*)tab, C, img_stride);
        AADG_LAUNCH_CHECK();
        return 0;
    }
    const int nr = span_bound(BT_I, sy, H), nc = span_bound(BT_J, sx, W);
    const int ldc = staged_pitch(nc);
    const size_t lds = (size_t)(nr + BT_I) * ldc * sizeof(float);
    const int tiles = ((w + BT_J - 1) / BT_J) * ((h + BT_I - 1) / BT_I);
    for (int p0 = 0; p0 < planes; p0 += 65535) {
        const int np = planes - p0 < 65535 ? planes - p0 : 65535;
        const dim3 g(tiles, np);
        if (dtype == 0)
            hipLaunchKernelGGL(k_upsample_bwd<float>, g, dim3(256), lds, st, reinterpret_cast<const float*>(dy),
                               reinterpret_cast<float*>(dx), h, w, H, W, (const float*)tab, nr, ldc, C, img_stride, p0);
        else
            hipLaunchKernelGGL(k_upsample_bwd<__hip_bfloat16>, g, dim3(256), lds, st, reinterpret_cast<const __hip_bfloat16*>(dy),
                               reinterpret_cast<__hip_bfloat16*>(dx), h, w, H, W, (const float*)tab, nr, ldc, C, img_stride, p0);
        AADG_LAUNCH_CHECK();
    }
    return 0;
}

/* contiguous dy [planes, H, W] */
extern "C" int aadg_upsample_bilinear2d_backward(const void* dy, void* dx, int planes, int h, int w, int H, int W, int dtype,
                                                 void* ws, size_t ws_bytes, void* stream) {
    if (planes <= 0) return AADG_E_BADARG;
    return aadg_upsample_bilinear2d_backward_strided(dy, dx, 1, planes, h, w, H, W, (long long)planes * H * W, dtype, ws, ws_bytes,
                                                     stream);
}

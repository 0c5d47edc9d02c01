// Depthwise 3x3 convolution (stride 1, padding = dilation, no bias) over NCHW planes: the atrous separable
// convolutions of the DeepLabV3+ head (ASPP rates 12 / 24 / 36 on [N, C_enc, h/16, w/16], the 3x3 fuse convs) that
// sit between the augmentation batch and the loss kernel.  The library paths they replace are a one-thread-per-
// output direct kernel (~0.3 TB/s) and MIOpen's naive fallback (double accumulation, 3-6 ms per call at N = 144).
// Every pass here streams planes through LDS:
//
//   k_dw3x3        y = conv(x, w)            one workgroup per (plane, row tile): rows + halo -> LDS (float), 4 outputs
//                  (also dx = conv(dy, flip w))   per lane per row, 16-byte loads / 8-or-16-byte stores
//   k_dw3x3_wgrad  dw[c] = sum_n dy (*) x    grid (split, C): partial 3x3 sums per workgroup, combined in float64
//
// Weights stay float32 (the master copy): no cast kernel, float32 weight gradients.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

constexpr int DW_MAX_SPLIT = 32;

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int V = 4;     // elements per 16-byte load
    static __device__ __forceinline__ void load16(const float* p, float* v) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void load4(const float* p, float* v) { load16(p, v); }
    static __device__ __forceinline__ void store4(float* p, const float* v) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
__device__ __forceinline__ uint32_t dw_f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
template <> struct Elem<__hip_bfloat16> {
    static constexpr int V = 8;
    static __device__ __forceinline__ void load16(const __hip_bfloat16* p, float* v) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
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
    static __device__ __forceinline__ void store4(__hip_bfloat16* p, const float* v) {
        uint2 t;
        t.x = dw_f2bf(v[0]) | (dw_f2bf(v[1]) << 16);
        t.y = dw_f2bf(v[2]) | (dw_f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(p) = t;
    }
};

struct DwGeom {
    int H, W, d, TH, tiles;     // tile = TH rows x full width; tiles per plane
    int w4, rows_per_pass;      // W / 4 lanes per row; rows covered by one pass of the workgroup
    int lds_rows;               // rows staged (TH + 2d, clipped to H)
};
inline bool dw_geom(int H, int W, int d, int elems_per_vec, DwGeom* g) {
    if (H <= 0 || W <= 0 || d <= 0 || W > 256 || (W % elems_per_vec) != 0) return false;
    g->H = H; g->W = W; g->d = d;
    int th = 2048 / W;
    if (th < 8) th = 8;
    if (th > H) th = H;
    g->TH = th;
    g->tiles = (H + th - 1) / th;
    g->w4 = W / 4;
    g->rows_per_pass = 256 / g->w4;
    const int need = th + 2 * d;
    g->lds_rows = need < H ? need : H;
    return (size_t)g->lds_rows * W * sizeof(float) <= 64 * 1024;
}

// stage rows [R0, R1) of the plane into LDS as float (full rows are contiguous in memory: flat 16-byte loads)
template <typename T>
__device__ __forceinline__ void stage_rows(const T* __restrict__ plane, int W, int R0, int R1, float* L) {
    constexpr int V = Elem<T>::V;
    const int nvec = (R1 - R0) * W / V;
    const T* src = plane + (size_t)R0 * W;
    for (int i = threadIdx.x; i < nvec; i += 256) {
        float v[V];
        Elem<T>::load16(src + (size_t)i * V, v);
#pragma unroll
        for (int k = 0; k < V; k += 4) *reinterpret_cast<float4*>(L + (size_t)i * V + k) = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
    }
}

// grid: planes * tiles.  flip = 1 applies the kernel rotated by 180 degrees (gradient w.r.t. the input).
template <typename T>
__global__ __launch_bounds__(256) void k_dw3x3(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, int C,
                                               DwGeom g, int flip) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    const int plane = blockIdx.x / g.tiles, tile = blockIdx.x - plane * g.tiles;
    const int c = plane % C;
    const int r0 = tile * g.TH, r1 = min(g.H, r0 + g.TH);
    const int R0 = max(0, r0 - g.d), R1 = min(g.H, r1 + g.d);
    const T* px = x + (size_t)plane * g.H * g.W;
    stage_rows<T>(px, g.W, R0, R1, L);
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[c * 9 + (flip ? 8 - i : i)];
    __syncthreads();
    const int cg = threadIdx.x % g.w4, ro = threadIdx.x / g.w4;
    if (ro >= g.rows_per_pass) return;
    const int j0 = cg * 4;
    T* py = y + (size_t)plane * g.H * g.W;
    for (int i = r0 + ro; i < r1; i += g.rows_per_pass) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int ri = i + (a - 1) * g.d;
            if (ri < 0 || ri >= g.H) continue;
            const float* row = L + (size_t)(ri - R0) * g.W;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int cj = j0 + (b - 1) * g.d;
                const float kv = k[a * 3 + b];
                if (cj >= 0 && cj + 3 < g.W) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = fmaf(kv, row[cj + t], acc[t]);
                } else if (cj + 3 >= 0 && cj < g.W) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (cj + t >= 0 && cj + t < g.W) acc[t] = fmaf(kv, row[cj + t], acc[t]);
                }
            }
        }
        Elem<T>::store4(py + (size_t)i * g.W + j0, acc);
    }
}

// grid (split, C): partial[c][split][9] = sum over this workgroup's (plane, tile) items of dy[i][j] * x[i+(a-1)d][j+(b-1)d]
template <typename T>
__global__ __launch_bounds__(256) void k_dw3x3_wgrad(const T* __restrict__ x, const T* __restrict__ dy, int N, int C, DwGeom g,
                                                     float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    __shared__ float red[4][9];
    const int c = blockIdx.y, S = gridDim.x;
    const int cg = threadIdx.x % g.w4, ro = threadIdx.x / g.w4;
    const int j0 = cg * 4;
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
    const int items = N * g.tiles;
    for (int q = blockIdx.x; q < items; q += S) {
        const int n = q / g.tiles, tile = q - n * g.tiles;
        const size_t pl = ((size_t)n * C + c) * g.H * g.W;
        const int r0 = tile * g.TH, r1 = min(g.H, r0 + g.TH);
        const int R0 = max(0, r0 - g.d), R1 = min(g.H, r1 + g.d);
        __syncthreads();                       // previous item's readers are done with L
        stage_rows<T>(x + pl, g.W, R0, R1, L);
        __syncthreads();
        if (ro < g.rows_per_pass) {
            for (int i = r0 + ro; i < r1; i += g.rows_per_pass) {
                float gy[4];
                Elem<T>::load4(dy + pl + (size_t)i * g.W + j0, gy);
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const int ri = i + (a - 1) * g.d;
                    if (ri < 0 || ri >= g.H) continue;
                    const float* row = L + (size_t)(ri - R0) * g.W;
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const int cj = j0 + (b - 1) * g.d;
                        float s = 0.f;
                        if (cj >= 0 && cj + 3 < g.W) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) s = fmaf(gy[t], row[cj + t], s);
                        } else if (cj + 3 >= 0 && cj < g.W) {
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                if (cj + t >= 0 && cj + t < g.W) s = fmaf(gy[t], row[cj + t], s);
                        }
                        acc[a * 3 + b] += s;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = wave_sum(acc[i]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int i = 0; i < 9; ++i) red[threadIdx.x >> 6][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < 9)
        partial[((size_t)c * DW_MAX_SPLIT + blockIdx.x) * 9 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void k_dw3x3_wgrad_final(const float* __restrict__ partial, int split, int C, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * 9) return;
    const int c = i / 9, k = i - c * 9;
    double s = 0.0;
    for (int j = 0; j < split; ++j) s += (double)partial[((size_t)c * DW_MAX_SPLIT + j) * 9 + k];
    dw[i] = (float)s;
}

template <typename T>
int dw_forward(const T* x, const float* w, T* y, int N, int C, int H, int W, int d, int flip, hipStream_t st) {
    DwGeom g;
    if (!dw_geom(H, W, d, Elem<T>::V, &g) || (long long)N * C * g.tiles > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    const size_t lds = (size_t)g.lds_rows * W * sizeof(float);
    hipLaunchKernelGGL((k_dw3x3<T>), dim3((unsigned)(N * C * g.tiles)), dim3(256), lds, st, x, w, y, C, g, flip);
    AADG_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int dw_wgrad(const T* x, const T* dy, float* dw, int N, int C, int H, int W, int d, float* ws, hipStream_t st) {
    DwGeom g;
    if (!dw_geom(H, W, d, Elem<T>::V, &g) || C > 65535) return AADG_E_UNSUPPORTED;
    int split = (4096 + C - 1) / C;
    const int items = N * g.tiles;
    if (split > items) split = items;
    if (split > DW_MAX_SPLIT) split = DW_MAX_SPLIT;
    const size_t lds = (size_t)g.lds_rows * W * sizeof(float);
    hipLaunchKernelGGL((k_dw3x3_wgrad<T>), dim3(split, C), dim3(256), lds, st, x, dy, N, C, g, ws);
    AADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_dw3x3_wgrad_final, dim3((C * 9 + 255) / 256), dim3(256), 0, st, (const float*)ws, split, C, dw);
    AADG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" size_t aadg_dwconv3x3_workspace_bytes(int C) { return C > 0 ? (size_t)C * DW_MAX_SPLIT * 9 * sizeof(float) : 0; }

extern "C" int aadg_dwconv3x3_supported(int H, int W, int dilation, int dtype) {
    DwGeom g;
    return (dtype == 0 || dtype == 1) && dw_geom(H, W, dilation, dtype == 0 ? 4 : 8, &g) ? 1 : 0;
}

extern "C" int aadg_dwconv3x3(const void* x, const float* weight, void* y, int N, int C, int H, int W, int dilation, int flip,
                              int dtype, void* stream) {
    if (x == nullptr || weight == nullptr || y == nullptr || N <= 0 || C <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0) return AADG_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) return dw_forward<float>((const float*)x, weight, (float*)y, N, C, H, W, dilation, flip, st);
    if (dtype == 1)
        return dw_forward<__hip_bfloat16>((const __hip_bfloat16*)x, weight, (__hip_bfloat16*)y, N, C, H, W, dilation, flip, st);
    return AADG_E_BADARG;
}

extern "C" int aadg_dwconv3x3_wgrad(const void* x, const void* dy, float* dweight, int N, int C, int H, int W, int dilation,
                                    int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (x == nullptr || dy == nullptr || dweight == nullptr || ws == nullptr || N <= 0 || C <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)x | (uintptr_t)dy) & 15u) != 0) return AADG_E_BADARG;
    if (ws_bytes < aadg_dwconv3x3_workspace_bytes(C)) return AADG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) return dw_wgrad<float>((const float*)x, (const float*)dy, dweight, N, C, H, W, dilation, (float*)ws, st);
    if (dtype == 1)
        return dw_wgrad<__hip_bfloat16>((const __hip_bfloat16*)x, (const __hip_bfloat16*)dy, dweight, N, C, H, W, dilation,
                                        (float*)ws, st);
    return AADG_E_BADARG;
}

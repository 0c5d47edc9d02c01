// Residual add + LayerNorm over the last dimension, forward and backward, for the pre-norm transformer blocks of the SegFormer
// backbone (BASELINE configs[4]; reference: models/mmseg/models/backbones/mix_transformer.py:108-117 `x = x + drop_path(attn(norm1(x)))`,
// `x = x + drop_path(mlp(norm2(x)))`, nn.LayerNorm :96-104).
//
//   s = x + rscale[sample] * r          (optional: the previous branch's output joins the residual stream here; rscale = the
//                                        stochastic-depth factor mask / keep of the sample, or NULL for 1)
//   y = (s - mean(s)) * rstd(s) * gamma + beta
//
// What it replaces under torch autocast: the float32 residual add, the cast of its result, ATen's float32 layer_norm (autocast runs
// it in float32) and the cast of the normalised tokens back to bfloat16 for the next Linear -- five launches and ~20 B per element --
// by one pass that reads x and r and writes s and y in the activations' own type (12 B per element in bfloat16); statistics and
// the normalisation are evaluated in float32 on the ROUNDED s, which is also what the backward recomputes from.
// HBM-bound: a lane owns 8 consecutive channels (16-byte accesses in bfloat16), LPR = C / 8 lanes share a row (64 / LPR rows per
// wave), the row reductions are wave shuffles.  C % 8 == 0, C <= 512 (MiT: 64 / 128 / 320 / 512).
// Backward: dx = ds = rstd * (g - mean(g) - x^ * mean(g x^)) + ds_extra with g = dy * gamma; dr = rscale * ds; the parameter
// gradients are per-workgroup column sums (registers -> LDS -> workspace) finished in a fixed order by a second kernel.
#include "common.h"

namespace {

constexpr int LN_THREADS = 256;
constexpr int LN_VEC = 8;

template <typename T> struct LnIO;
template <> struct LnIO<float> {
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* v) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    static __device__ __forceinline__ float round(float v) { return v; }
};
template <> struct LnIO<uint16_t> {            // bfloat16 bits
    static __device__ __forceinline__ void load(const uint16_t* p, float* v) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u); }
    }
    static __device__ __forceinline__ void store(uint16_t* p, const float* v) {
        *reinterpret_cast<uint4*>(p) = make_uint4(aadg_f2bf_pk(v[0], v[1]), aadg_f2bf_pk(v[2], v[3]), aadg_f2bf_pk(v[4], v[5]), aadg_f2bf_pk(v[6], v[7]));
    }
    static __device__ __forceinline__ float round(float v) { return __uint_as_float(aadg_f2bf_bits(v) << 16); }
};

template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// grid: ceil(R / rows per workgroup); a wave handles 64 / LPR rows at a time
template <typename T, int LPR>
__global__ __launch_bounds__(LN_THREADS) void k_ln_fwd(const T* __restrict__ x, const T* __restrict__ r, const float* __restrict__ rscale,
                                                       int rows_per_sample, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, T* __restrict__ s_out, T* __restrict__ y, float* __restrict__ mean,
                                                       float* __restrict__ rstd, int R, int C) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane / LPR, l = lane % LPR;
    const int row = (blockIdx.x * (LN_THREADS / 64) + wv) * RPW + sub;
    const int c0 = l * LN_VEC;
    const bool live = row < R && c0 < C;
    float v[LN_VEC];
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) v[i] = 0.f;
    if (live) {
        LnIO<T>::load(x + (size_t)row * C + c0, v);
        if (r != nullptr) {
            float q[LN_VEC];
            LnIO<T>::load(r + (size_t)row * C + c0, q);
            const float sc = rscale != nullptr ? rscale[row / rows_per_sample] : 1.0f;
#pragma unroll
            for (int i = 0; i < LN_VEC; ++i) v[i] = LnIO<T>::round(v[i] + sc * q[i]);
            LnIO<T>::store(s_out + (size_t)row * C + c0, v);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) sum += v[i];
    const float m = row_sum<LPR>(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) { const float d = live ? v[i] - m : 0.f; sq += d * d; }
    const float rs = rsqrtf(row_sum<LPR>(sq) / (float)C + eps);
    if (!live) return;
    float g[LN_VEC], b[LN_VEC], o[LN_VEC];
    LnIO<float>::load(gamma + c0, g);
    LnIO<float>::load(beta + c0, b);
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) o[i] = (v[i] - m) * rs * g[i] + b[i];
    LnIO<T>::store(y + (size_t)row * C + c0, o);
    if (l == 0) { mean[row] = m; rstd[row] = rs; }
}

// grid: nblk workgroups, each walks rows with stride nblk * rows-per-workgroup; column sums of dy * x^ and dy per workgroup -> ws
template <typename T, int LPR>
__global__ __launch_bounds__(LN_THREADS) void k_ln_bwd(const T* __restrict__ s, const T* __restrict__ dy, const T* __restrict__ ds_extra,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ rscale, int rows_per_sample,
                                                       T* __restrict__ dx, T* __restrict__ dr, float* __restrict__ part /*[nblk][2][C]*/,
                                                       int R, int C) {
    constexpr int RPW = 64 / LPR, RPB = RPW * (LN_THREADS / 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane / LPR, l = lane % LPR;
    const int c0 = l * LN_VEC;
    const bool col_live = c0 < C;
    float g[LN_VEC], dg[LN_VEC], db[LN_VEC];
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i) { g[i] = 0.f; dg[i] = 0.f; db[i] = 0.f; }
    if (col_live) LnIO<float>::load(gamma + c0, g);
    for (int row = blockIdx.x * RPB + wv * RPW + sub; row < R; row += gridDim.x * RPB) {     // (the LPR lanes of a row leave together)
        const bool live = row < R && col_live;
        float v[LN_VEC], d[LN_VEC];
#pragma unroll
        for (int i = 0; i < LN_VEC; ++i) { v[i] = 0.f; d[i] = 0.f; }
        float m = 0.f, rs = 0.f;
        if (live) {
            LnIO<T>::load(s + (size_t)row * C + c0, v);
            LnIO<T>::load(dy + (size_t)row * C + c0, d);
            m = mean[row]; rs = rstd[row];
        }
        float xh[LN_VEC], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_VEC; ++i) {
            xh[i] = (v[i] - m) * rs;
            const float gi = d[i] * g[i];
            s1 += gi; s2 += gi * xh[i];
            dg[i] += d[i] * xh[i]; db[i] += d[i];
        }
        const float c1 = row_sum<LPR>(s1) / (float)C, c2 = row_sum<LPR>(s2) / (float)C;
        if (live) {
            float o[LN_VEC];
#pragma unroll
            for (int i = 0; i < LN_VEC; ++i) o[i] = rs * (d[i] * g[i] - c1 - xh[i] * c2);
            if (ds_extra != nullptr) {
                float e[LN_VEC];
                LnIO<T>::load(ds_extra + (size_t)row * C + c0, e);
#pragma unroll
                for (int i = 0; i < LN_VEC; ++i) o[i] += e[i];
            }
            LnIO<T>::store(dx + (size_t)row * C + c0, o);
            if (dr != nullptr) {
                const float sc = rscale != nullptr ? rscale[row / rows_per_sample] : 1.0f;
#pragma unroll
                for (int i = 0; i < LN_VEC; ++i) o[i] = LnIO<T>::round(o[i]) * sc;
                LnIO<T>::store(dr + (size_t)row * C + c0, o);
            }
        }
    }
    // lanes sub = 0 .. RPW-1 of a wave hold the same columns: fold them, then the four waves through LDS
#pragma unroll
    for (int i = 0; i < LN_VEC; ++i)
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) { dg[i] += __shfl_xor(dg[i], o, 64); db[i] += __shfl_xor(db[i], o, 64); }
    __shared__ float red[LN_THREADS / 64][2][512];
    if (sub == 0 && col_live) {
#pragma unroll
        for (int i = 0; i < LN_VEC; ++i) { red[wv][0][c0 + i] = dg[i]; red[wv][1][c0 + i] = db[i]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += LN_THREADS) {
        const int k = i / C, c = i - k * C;
        part[(size_t)blockIdx.x * 2 * C + i] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
    }
}

// column sums over the workgroups' partial records: a workgroup owns 16 columns, 16 groups of 16 lanes each walk every 16th record
// (independent loads, unrolled), LDS tree over the groups -- fixed order, no atomics.  (One thread per column walking all records one
// after the other took 225 us per call at 1024 records: 12 ms per SegFormer step.)
__global__ __launch_bounds__(256) void k_ln_bwd_final(const float* __restrict__ part, int nblk, int C, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta) {
    const int col = blockIdx.x * 16 + (threadIdx.x & 15), grp = threadIdx.x >> 4;
    float a = 0.f;
    if (col < 2 * C) {
#pragma unroll 8
        for (int b = grp; b < nblk; b += 16) a += part[(size_t)b * 2 * C + col];
    }
    __shared__ float red[16][17];
    red[grp][threadIdx.x & 15] = a;
    __syncthreads();
    if (threadIdx.x < 16 && col < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x];
        if (col < C) dgamma[col] = t; else dbeta[col - C] = t;
    }
}

int ln_lpr(int C) { return C <= 64 ? 8 : (C <= 128 ? 16 : 64); }
int ln_nblk(int R, int C) {
    const int rpb = (64 / ln_lpr(C)) * (LN_THREADS / 64);
    const int need = (R + rpb - 1) / rpb;
    return need < 1 ? 1 : (need > 1024 ? 1024 : need);
}

}  // namespace

extern "C" int aadg_layernorm_supported(int R, int C, int dtype) {
    return R > 0 && C > 0 && (C % 8) == 0 && C <= 512 && (dtype == 0 || dtype == 1);
}

extern "C" size_t aadg_layernorm_workspace_bytes(int R, int C) {
    if (R <= 0 || C <= 0) return 0;
    return aadg_align_up((size_t)ln_nblk(R, C) * 2 * C * sizeof(float), 256);
}

#define LN_LAUNCH(KERNEL, T, GRID, ...)                                                                       \
    do {                                                                                                      \
        const int lpr = ln_lpr(C);                                                                            \
        if (lpr == 8) hipLaunchKernelGGL((KERNEL<T, 8>), GRID, dim3(LN_THREADS), 0, st, __VA_ARGS__);         \
        else if (lpr == 16) hipLaunchKernelGGL((KERNEL<T, 16>), GRID, dim3(LN_THREADS), 0, st, __VA_ARGS__);  \
        else hipLaunchKernelGGL((KERNEL<T, 64>), GRID, dim3(LN_THREADS), 0, st, __VA_ARGS__);                 \
    } while (0)

template <typename T> static inline const T* cp(const void* p) { return reinterpret_cast<const T*>(p); }
template <typename T> static inline T* mp(void* p) { return reinterpret_cast<T*>(p); }

extern "C" int aadg_layernorm_forward(const void* x, const void* r, const float* rscale, int rows_per_sample, const float* gamma,
                                      const float* beta, float eps, void* s_out, void* y, float* mean, float* rstd, int R, int C,
                                      int dtype, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd) return AADG_E_BADARG;
    if (!aadg_layernorm_supported(R, C, dtype)) return AADG_E_UNSUPPORTED;
    if (r != nullptr && s_out == nullptr) return AADG_E_BADARG;
    if (rscale != nullptr && (rows_per_sample <= 0 || R % rows_per_sample)) return AADG_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)r | (uintptr_t)s_out | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) return AADG_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rpb = (64 / ln_lpr(C)) * (LN_THREADS / 64);
    const dim3 grid((R + rpb - 1) / rpb);
    if (dtype == 0)
        LN_LAUNCH(k_ln_fwd, float, grid, cp<float>(x), cp<float>(r), rscale, rows_per_sample, gamma, beta, eps, mp<float>(s_out), mp<float>(y), mean, rstd, R, C);
    else
        LN_LAUNCH(k_ln_fwd, uint16_t, grid, cp<uint16_t>(x), cp<uint16_t>(r), rscale, rows_per_sample, gamma, beta, eps, mp<uint16_t>(s_out), mp<uint16_t>(y), mean, rstd, R, C);
    AADG_LAUNCH_CHECK();
    return 0;
}

extern "C" int aadg_layernorm_backward(const void* s, const void* dy, const void* ds_extra, const float* gamma, const float* mean,
                                       const float* rstd, const float* rscale, int rows_per_sample, void* dx, void* dr, float* dgamma,
                                       float* dbeta, void* ws, size_t ws_bytes, int R, int C, int dtype, void* stream) {
    if (!s || !dy || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !ws) return AADG_E_BADARG;
    if (!aadg_layernorm_supported(R, C, dtype)) return AADG_E_UNSUPPORTED;
    if (ws_bytes < aadg_layernorm_workspace_bytes(R, C)) return AADG_E_WORKSPACE;
    if (rscale != nullptr && (rows_per_sample <= 0 || R % rows_per_sample)) return AADG_E_BADARG;
    if (((uintptr_t)s | (uintptr_t)dy | (uintptr_t)ds_extra | (uintptr_t)dx | (uintptr_t)dr | (uintptr_t)gamma) & 15) return AADG_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nblk = ln_nblk(R, C);
    float* part = reinterpret_cast<float*>(ws);
    if (dtype == 0)
        LN_LAUNCH(k_ln_bwd, float, dim3(nblk), cp<float>(s), cp<float>(dy), cp<float>(ds_extra), gamma, mean, rstd, rscale, rows_per_sample, mp<float>(dx), mp<float>(dr), part, R, C);
    else
        LN_LAUNCH(k_ln_bwd, uint16_t, dim3(nblk), cp<uint16_t>(s), cp<uint16_t>(dy), cp<uint16_t>(ds_extra), gamma, mean, rstd, rscale, rows_per_sample, mp<uint16_t>(dx), mp<uint16_t>(dr), part, R, C);
    AADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ln_bwd_final, dim3((2 * C + 15) / 16), dim3(256), 0, st, part, nblk, C, dgamma, dbeta);
    AADG_LAUNCH_CHECK();
    return 0;
}

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
        const float4 t = aadg_load_stream(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void load4(const float* p, float* v) { load16(p, v); }
    static __device__ __forceinline__ void store4(float* p, const float* v, bool stream) {
        aadg_store_out(p, make_float4(v[0], v[1], v[2], v[3]), stream);
    }
};
template <> struct Elem<__hip_bfloat16> {
    static constexpr int V = 8;
    static __device__ __forceinline__ void load16(const __hip_bfloat16* p, float* v) {
        const uint4 t = aadg_load_stream(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
    static __device__ __forceinline__ void load4(const __hip_bfloat16* p, float* v) {
        typedef uint32_t u32x2_l __attribute__((ext_vector_type(2)));
        const u32x2_l tq = __builtin_nontemporal_load(reinterpret_cast<const u32x2_l*>(p));       // dy of the weight gradient: read once
        const uint2 t = make_uint2(tq.x, tq.y);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
    }
    static __device__ __forceinline__ void store4(__hip_bfloat16* p, const float* v, bool stream) {
        uint2 t;
        t.x = aadg_f2bf_pk(v[0], v[1]);
        t.y = aadg_f2bf_pk(v[2], v[3]);
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t q = {t.x, t.y};
        // an output too large to stay cached until its consumer reads it: streaming store
        if (stream) __builtin_nontemporal_store(q, reinterpret_cast<u32x2_t*>(p)); else *reinterpret_cast<uint2*>(p) = t;
    }
};

constexpr int DW_MAX_PP = 16;    // planes per workgroup when a whole plane is one tile

struct DwGeom {
    int H, W, d, TH, tiles;     // tile = TH rows x full width; tiles per plane
    int PP;                     // planes per workgroup (> 1 only when tiles == 1)
    int w4, rows_per_pass;      // W / 4 lanes per row; rows covered by one pass of the workgroup
    int lds_rows;               // rows staged per plane (TH + 2d, clipped to H)
    int stream;                 // forward: write the output with streaming stores (set by the launcher for outputs > 128 MB)
};
inline bool dw_geom(int H, int W, int d, int elems_per_vec, DwGeom* g) {
    if (H <= 0 || W <= 0 || d <= 0 || W > 256 || (W % elems_per_vec) != 0) return false;
    g->H = H; g->W = W; g->d = d; g->stream = 0;
    int th = 8192 / W;                                   // ~8 output quads per lane
    while (th > 8 && (size_t)(th + 2 * d < H ? th + 2 * d : H) * W * sizeof(float) > 48 * 1024) th /= 2;
    if (th > H) th = H;
    g->TH = th;
    g->tiles = (H + th - 1) / th;
    g->PP = 1;
    if (g->tiles == 1) {
        int pp = 8192 / (H * W);
        if (pp > DW_MAX_PP) pp = DW_MAX_PP;
        if (pp < 1) pp = 1;
        g->PP = pp;
    }
    g->w4 = W / 4;
    g->rows_per_pass = 256 / g->w4;
    const int need = th + 2 * d;
    g->lds_rows = need < H ? need : H;
    return (size_t)g->PP * g->lds_rows * W * sizeof(float) <= 64 * 1024;
}

// stage `count` contiguous elements (a multiple of the 16-byte vector) into LDS as float
template <typename T>
__device__ __forceinline__ void stage_flat(const T* __restrict__ src, int count, float* L) {
    constexpr int V = Elem<T>::V;
    const int nvec = count / V;
#pragma unroll 4
    for (int i = threadIdx.x; i < nvec; i += 256) {
        float v[V];
        Elem<T>::load16(src + (size_t)i * V, v);
#pragma unroll
        for (int k = 0; k < V; k += 4) *reinterpret_cast<float4*>(L + (size_t)i * V + k) = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
    }
}

// the 12 inputs of one staged row that the 4 outputs at columns j0..j0+3 see: tap b of output t = in[b][t]
// (zero outside the row).  d == 1 and d % 4 == 0 (the ASPP rates) use 16-byte LDS reads.
__device__ __forceinline__ void dw_row_taps(const float* row, int W, int d, int j0, float in[3][4]) {
    if (d == 1) {
        const float4 m = *reinterpret_cast<const float4*>(row + j0);
        const float l = j0 > 0 ? row[j0 - 1] : 0.0f, r = j0 + 4 < W ? row[j0 + 4] : 0.0f;
        in[0][0] = l;   in[0][1] = m.x; in[0][2] = m.y; in[0][3] = m.z;
        in[1][0] = m.x; in[1][1] = m.y; in[1][2] = m.z; in[1][3] = m.w;
        in[2][0] = m.y; in[2][1] = m.z; in[2][2] = m.w; in[2][3] = r;
    } else if ((d & 3) == 0) {          // a shifted quad is entirely inside or entirely outside the row
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int cj = j0 + (b - 1) * d;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cj >= 0 && cj < W) v = *reinterpret_cast<const float4*>(row + cj);
            in[b][0] = v.x; in[b][1] = v.y; in[b][2] = v.z; in[b][3] = v.w;
        }
    } else {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int cj = j0 + (b - 1) * d;
#pragma unroll
            for (int t = 0; t < 4; ++t) in[b][t] = (cj + t >= 0 && cj + t < W) ? row[cj + t] : 0.0f;
        }
    }
}

// 4 outputs of row i (columns j0..j0+3) of one plane from its staged rows (row R0 of the plane at L)
__device__ __forceinline__ void dw_quad(const float* L, const float* k, const DwGeom& g, int R0, int i, int j0, float* acc) {
    acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int ri = i + (a - 1) * g.d;
        if (ri < 0 || ri >= g.H) continue;
        float in[3][4];
        dw_row_taps(L + (size_t)(ri - R0) * g.W, g.W, g.d, j0, in);
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = fmaf(k[a * 3 + b], in[b][t], acc[t]);
    }
}
// the 3x3 products of 4 output gradients with the staged input: acc[a*3+b] += sum_t gy[t] * x[i+(a-1)d][j0+t+(b-1)d]
__device__ __forceinline__ void dw_quad_wgrad(const float* L, const float* gy, const DwGeom& g, int R0, int i, int j0, float* acc) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int ri = i + (a - 1) * g.d;
        if (ri < 0 || ri >= g.H) continue;
        float in[3][4];
        dw_row_taps(L + (size_t)(ri - R0) * g.W, g.W, g.d, j0, in);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) s = fmaf(gy[t], in[b][t], s);
            acc[a * 3 + b] += s;
        }
    }
}

// grid: ceil(planes / PP) * tiles.  flip = 1 applies the kernel rotated by 180 degrees (gradient w.r.t. the input).
template <typename T>
__global__ __launch_bounds__(256) void k_dw3x3(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, int planes,
                                               int C, DwGeom g, int flip) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    __shared__ float kw[DW_MAX_PP * 9];
    const int grp = blockIdx.x / g.tiles, tile = blockIdx.x - grp * g.tiles;
    const int plane0 = grp * g.PP, np = min(g.PP, planes - plane0);
    const int r0 = tile * g.TH, r1 = min(g.H, r0 + g.TH);
    const int R0 = max(0, r0 - g.d), R1 = min(g.H, r1 + g.d);
    const size_t psz = (size_t)g.H * g.W;
    // tiles == 1: the np planes are whole and adjacent in memory; else one plane, rows [R0, R1)
    stage_flat<T>(x + (size_t)plane0 * psz + (size_t)R0 * g.W, (g.tiles == 1 ? np * g.H : R1 - R0) * g.W, L);
    if (threadIdx.x < np * 9) {
        const int pl = threadIdx.x / 9, i = threadIdx.x - pl * 9;
        kw[threadIdx.x] = w[((plane0 + pl) % C) * 9 + (flip ? 8 - i : i)];
    }
    __syncthreads();
    const int cg = threadIdx.x % g.w4, ro = threadIdx.x / g.w4;
    if (ro >= g.rows_per_pass) return;
    const int j0 = cg * 4;
    const int nrows = r1 - r0, vrows = np * nrows;          // virtual rows over the planes of this workgroup
    const int strip = (vrows + g.rows_per_pass - 1) / g.rows_per_pass;
    if (g.d == 1 && strip > 1 && nrows % strip == 0) {
        // d = 1: a lane walks `strip` CONSECUTIVE rows of one plane with a 3-row window of taps in registers: one new row of
        // LDS reads per output row instead of three (the strided mapping below spent ~70 % of its time on those reads)
        const int v0 = ro * strip;
        if (v0 >= vrows) return;
        const int pl = v0 / nrows, i0 = r0 + (v0 - pl * nrows);
        const float* Lp = L + (size_t)pl * psz;
        const float* k = kw + pl * 9;
        float w0[3][4], w1[3][4], w2[3][4];
        auto fetch = [&](int ri, float in[3][4]) {
            if (ri >= 0 && ri < g.H) dw_row_taps(Lp + (size_t)(ri - R0) * g.W, g.W, 1, j0, in);
            else
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int t = 0; t < 4; ++t) in[b][t] = 0.0f;
        };
        fetch(i0 - 1, w0);
        fetch(i0, w1);
        T* py = y + (size_t)(plane0 + pl) * psz;
        for (int r = 0; r < strip; ++r) {
            fetch(i0 + r + 1, w2);
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[t] = fmaf(k[b], w0[b][t], acc[t]);
                    acc[t] = fmaf(k[3 + b], w1[b][t], acc[t]);
                    acc[t] = fmaf(k[6 + b], w2[b][t], acc[t]);
                }
            Elem<T>::store4(py + (size_t)(i0 + r) * g.W + j0, acc, g.stream != 0);
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int t = 0; t < 4; ++t) { w0[b][t] = w1[b][t]; w1[b][t] = w2[b][t]; }
        }
        return;
    }
    for (int v = ro; v < vrows; v += g.rows_per_pass) {
        const int pl = v / nrows, i = r0 + (v - pl * nrows);
        float acc[4];
        dw_quad(L + (size_t)pl * psz, kw + pl * 9, g, R0, i, j0, acc);
        Elem<T>::store4(y + (size_t)(plane0 + pl) * psz + (size_t)i * g.W + j0, acc, g.stream != 0);
    }
}

// grid (C, split): partial[c][split][9] = sum over this workgroup's (plane group, tile) items of
// dy[i][j] * x[i+(a-1)d][j+(b-1)d];  with tiles == 1 an item is PP images' planes of channel c
template <typename T>
__global__ __launch_bounds__(256) void k_dw3x3_wgrad(const T* __restrict__ x, const T* __restrict__ dy, int N, int C, DwGeom g,
                                                     float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    __shared__ float red[4][9];
    const int c = blockIdx.x, S = gridDim.y;           // grid (C, split): neighbouring workgroups read neighbouring channels of one image (round 6; it was (split, C))
    const int cg = threadIdx.x % g.w4, ro = threadIdx.x / g.w4;
    const int j0 = cg * 4;
    const size_t psz = (size_t)g.H * g.W;
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
    const int groups = (N + g.PP - 1) / g.PP;
    const int items = groups * g.tiles;
    for (int q = blockIdx.y; q < items; q += S) {
        const int grp = q / g.tiles, tile = q - grp * g.tiles;
        const int n0 = grp * g.PP, np = min(g.PP, N - n0);
        const int r0 = tile * g.TH, r1 = min(g.H, r0 + g.TH);
        const int R0 = max(0, r0 - g.d), R1 = min(g.H, r1 + g.d);
        const int srows = R1 - R0;
        __syncthreads();                       // previous item's readers are done with L
        for (int pl = 0; pl < np; ++pl)        // channel c of consecutive images: planes C * psz apart
            stage_flat<T>(x + ((size_t)(n0 + pl) * C + c) * psz + (size_t)R0 * g.W, srows * g.W, L + (size_t)pl * srows * g.W);
        __syncthreads();
        if (ro < g.rows_per_pass) {
            const int nrows = r1 - r0, vrows = np * nrows;
            const int strip = (vrows + g.rows_per_pass - 1) / g.rows_per_pass;
            if (g.d == 1 && strip > 1 && nrows % strip == 0) {      // consecutive rows, 3-row window of input taps (see k_dw3x3)
                const int v0 = ro * strip;
                if (v0 < vrows) {
                    const int pl = v0 / nrows, i0 = r0 + (v0 - pl * nrows);
                    const float* Lp = L + (size_t)pl * srows * g.W;
                    const T* pdy = dy + ((size_t)(n0 + pl) * C + c) * psz;
                    float w0[3][4], w1[3][4], w2[3][4];
                    auto fetch = [&](int ri, float in[3][4]) {
                        if (ri >= 0 && ri < g.H) dw_row_taps(Lp + (size_t)(ri - R0) * g.W, g.W, 1, j0, in);
                        else
#pragma unroll
                            for (int b = 0; b < 3; ++b)
#pragma unroll
                                for (int t = 0; t < 4; ++t) in[b][t] = 0.0f;
                    };
                    fetch(i0 - 1, w0);
                    fetch(i0, w1);
                    for (int r = 0; r < strip; ++r) {
                        fetch(i0 + r + 1, w2);
                        float gy[4];
                        Elem<T>::load4(pdy + (size_t)(i0 + r) * g.W + j0, gy);
#pragma unroll
                        for (int b = 0; b < 3; ++b) {
                            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                s0 = fmaf(gy[t], w0[b][t], s0);
                                s1 = fmaf(gy[t], w1[b][t], s1);
                                s2 = fmaf(gy[t], w2[b][t], s2);
                            }
                            acc[b] += s0; acc[3 + b] += s1; acc[6 + b] += s2;
                        }
#pragma unroll
                        for (int b = 0; b < 3; ++b)
#pragma unroll
                            for (int t = 0; t < 4; ++t) { w0[b][t] = w1[b][t]; w1[b][t] = w2[b][t]; }
                    }
                }
            } else
            for (int v = ro; v < vrows; v += g.rows_per_pass) {
                const int pl = v / nrows, i = r0 + (v - pl * nrows);
                float gy[4];
                Elem<T>::load4(dy + ((size_t)(n0 + pl) * C + c) * psz + (size_t)i * g.W + j0, gy);
                dw_quad_wgrad(L + (size_t)pl * srows * g.W, gy, g, R0, i, j0, acc);
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
        partial[((size_t)c * DW_MAX_SPLIT + blockIdx.y) * 9 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Dilation 1, bfloat16, W in {32, 64, 128}: the weight gradient without LDS and without barriers.  LPR = W / 8 lanes hold one image row
// (8 pixels = one 16-byte load each); such a lane group streams down a strip of DWR_ROWS rows of one plane, keeping a three-row window
// of x in registers as packed bfloat16 pairs -- as loaded, and shifted by one pixel to either side (v_alignbit; the pixel of the
// neighbouring lane through a DPP row shift, zero at the row ends = the padding column) -- and accumulates the nine taps with
// v_dot2c_f32_bf16 (two pixels per instruction, no unpacking).  Every byte of x and dy is read once (streaming loads; the two halo rows of
// a strip twice).  grid (split, C): the lane groups of a workgroup take the (image, strip) tasks of channel c round-robin.
constexpr int DWR_ROWS = 32;
typedef __bf16 dw_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dw_dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dw_bf16x2, a), __builtin_bit_cast(dw_bf16x2, b), c, false);
}

struct DwRow3 { uint32_t l[4], m[4], r[4]; };       // x[j - 1], x[j], x[j + 1] of a lane's 8 pixels, packed pairs

template <int LPR>
__device__ __forceinline__ void dw_row_load(const uint16_t* __restrict__ px, int row, int H, int W, int lane_in_row, DwRow3& o) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row >= 0 && row < H) v = aadg_load_stream(px + (size_t)row * W + lane_in_row * 8);
    o.m[0] = v.x; o.m[1] = v.y; o.m[2] = v.z; o.m[3] = v.w;
    // neighbours' edge words: DPP shifts within a 16-lane row; a lane group narrower than that masks its own ends
    uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x111, 0xF, 0xF, true);     // row_shr:1
    uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x101, 0xF, 0xF, true);     // row_shl:1
    if (LPR < 16) {
        if (lane_in_row == 0) prev = 0u;
        if (lane_in_row == LPR - 1) next = 0u;
    }
    o.l[0] = __builtin_amdgcn_alignbit(v.x, prev, 16); o.l[1] = __builtin_amdgcn_alignbit(v.y, v.x, 16);
    o.l[2] = __builtin_amdgcn_alignbit(v.z, v.y, 16);  o.l[3] = __builtin_amdgcn_alignbit(v.w, v.z, 16);
    o.r[0] = __builtin_amdgcn_alignbit(v.y, v.x, 16);  o.r[1] = __builtin_amdgcn_alignbit(v.z, v.y, 16);
    o.r[2] = __builtin_amdgcn_alignbit(v.w, v.z, 16);  o.r[3] = __builtin_amdgcn_alignbit(next, v.w, 16);
}

template <int LPR>
__global__ __launch_bounds__(256) void k_dw3x3_wgrad_rows(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, int N, int C, int H,
                                                          float* __restrict__ partial) {
    constexpr int W = LPR * 8, GROUPS = 256 / LPR;
    __shared__ float red[4][9];
    const int c = blockIdx.x, S = gridDim.y;           // grid (C, split): neighbouring workgroups read neighbouring channels of one image (round 6; it was (split, C))
    const int grp = threadIdx.x / LPR, lir = threadIdx.x % LPR;
    const int strips = (H + DWR_ROWS - 1) / DWR_ROWS, tasks = N * strips;
    const size_t psz = (size_t)H * W;
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
    for (int t = blockIdx.y * GROUPS + grp; t < tasks; t += S * GROUPS) {
        const int n = t / strips, st = t - n * strips;
        const int r0 = st * DWR_ROWS, r1 = min(H, r0 + DWR_ROWS);
        const uint16_t* px = x + ((size_t)n * C + c) * psz;
        const uint16_t* pg = dy + ((size_t)n * C + c) * psz;
        DwRow3 a, b, cc;                                   // rows r - 1, r, r + 1
        dw_row_load<LPR>(px, r0 - 1, H, W, lir, a);
        dw_row_load<LPR>(px, r0, H, W, lir, b);
        uint4 g = aadg_load_stream(pg + (size_t)r0 * W + lir * 8);
        for (int r = r0; r < r1; ++r) {
            dw_row_load<LPR>(px, r + 1, H, W, lir, cc);
            uint4 gn = make_uint4(0, 0, 0, 0);
            if (r + 1 < r1) gn = aadg_load_stream(pg + (size_t)(r + 1) * W + lir * 8);      // next row's dy in flight during the 36 dots
            const uint32_t gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[0] = dw_dot2(gw[i], a.l[i], acc[0]);  acc[1] = dw_dot2(gw[i], a.m[i], acc[1]);  acc[2] = dw_dot2(gw[i], a.r[i], acc[2]);
                acc[3] = dw_dot2(gw[i], b.l[i], acc[3]);  acc[4] = dw_dot2(gw[i], b.m[i], acc[4]);  acc[5] = dw_dot2(gw[i], b.r[i], acc[5]);
                acc[6] = dw_dot2(gw[i], cc.l[i], acc[6]); acc[7] = dw_dot2(gw[i], cc.m[i], acc[7]); acc[8] = dw_dot2(gw[i], cc.r[i], acc[8]);
            }
            a = b; b = cc; g = gn;
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = wave_sum(acc[i]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int i = 0; i < 9; ++i) red[threadIdx.x >> 6][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < 9)
        partial[((size_t)c * DW_MAX_SPLIT + blockIdx.y) * 9 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// (split, C) grid of the register-window kernel: every lane group gets the same number k of tasks (k = the fewest that keeps split <= DW_MAX_SPLIT)
static inline int dw_even_split(int tasks, int groups) {
    const int per = (tasks + groups - 1) / groups;              // lane-group rounds of one channel
    const int k = (per + DW_MAX_SPLIT - 1) / DW_MAX_SPLIT;
    const int split = (per + k - 1) / k;
    return split < 1 ? 1 : split;
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
    const size_t lds = (size_t)g.PP * g.lds_rows * W * sizeof(float);
    const int planes = N * C, groups = (planes + g.PP - 1) / g.PP;
    g.stream = (size_t)planes * H * W * sizeof(T) > ((size_t)128 << 20) ? 1 : 0;
    hipLaunchKernelGGL((k_dw3x3<T>), dim3((unsigned)(groups * g.tiles)), dim3(256), lds, st, x, w, y, planes, C, g, flip);
    AADG_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int dw_wgrad(const T* x, const T* dy, float* dw, int N, int C, int H, int W, int d, float* ws, hipStream_t st) {
    DwGeom g;
    if (!dw_geom(H, W, d, Elem<T>::V, &g) || C > 65535) return AADG_E_UNSUPPORTED;
    int split = (4096 + C - 1) / C;
    const int items = (N + g.PP - 1) / g.PP * g.tiles;
    if (split > items) split = items;
    if (split > DW_MAX_SPLIT) split = DW_MAX_SPLIT;
    if (sizeof(T) == 2) {
        const uint16_t* px = reinterpret_cast<const uint16_t*>(x);
        const uint16_t* pg = reinterpret_cast<const uint16_t*>(dy);
        bool done = false;
        if (d == 1 && (W == 32 || W == 64 || W == 128)) {
            // the register-window kernel: tasks = (image, 32-row strip), 256 / (W / 8) lane groups per workgroup
            split = dw_even_split(N * ((H + DWR_ROWS - 1) / DWR_ROWS), 256 / (W / 8));
            if (W == 128) hipLaunchKernelGGL(k_dw3x3_wgrad_rows<16>, dim3(C, split), dim3(256), 0, st, px, pg, N, C, H, ws);
            else if (W == 64) hipLaunchKernelGGL(k_dw3x3_wgrad_rows<8>, dim3(C, split), dim3(256), 0, st, px, pg, N, C, H, ws);
            else hipLaunchKernelGGL(k_dw3x3_wgrad_rows<4>, dim3(C, split), dim3(256), 0, st, px, pg, N, C, H, ws);
            done = true;
        }
        if (done) {
            AADG_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_dw3x3_wgrad_final, dim3((C * 9 + 255) / 256), dim3(256), 0, st, (const float*)ws, split, C, dw);
            AADG_LAUNCH_CHECK();
            return 0;
        }
    }
    const size_t lds = (size_t)g.PP * g.lds_rows * W * sizeof(float);
    hipLaunchKernelGGL((k_dw3x3_wgrad<T>), dim3(C, split), dim3(256), lds, st, x, dy, N, C, g, ws);
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

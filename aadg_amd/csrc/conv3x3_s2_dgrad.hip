// Input gradient of a 3x3 / stride-2 / padding-1 convolution over NCHW bfloat16 tensors on the matrix cores (the first block of
// ResNet stages 2 and 3), without the zero-stuffed 4x larger problem and without a layout change:
//
//     dX[n][c][2i + py][2j + px] = sum_m sum_{taps of class (py, px)} W[m][c][kh][kw] * dY[n][m][i + di][j + dj]
//
// The forward reads in[2y + kh - 1][2x + kw - 1], so an input pixel of row parity py is touched by kh = 1 (from dY row i) when py = 0
// and by kh = 0 (row i + 1) and kh = 2 (row i) when py = 1; likewise for columns.  The four parity classes are four small stride-1
// convolutions over dY with 1, 2, 2 and 4 taps -- 9 tap products in all, the flops of the forward -- whose operands are dY at
// (i, j), (i, j + 1), (i + 1, j), (i + 1, j + 1): one staged copy of the dY rows as they lie, one shifted by a pixel (lane shuffles
// at staging, zeros at the row end), the row below at an aligned offset of WO pixels.  As in conv3x3_fwd.hip the channel-strided
// operand is read with ds_read_b64_tr_b16.  A lane holds dY-grid column j of all four classes, i.e. the output pixels (2j, 2j + 1)
// of rows 2i and 2i + 1: the two column classes of a row interleave into ONE 4-byte store per lane, 128 contiguous bytes per 32 lanes.
//
// Workgroup = 4 waves = 64 channels x 128 dY-grid pixels (2 x 2: 32 channels x 64 pixels per wave, 4 classes x 2 pixel tiles of
// accumulators); K-step = 16 dY channels; weights tap-major a9t[t][c][m] = W[m][c][kh][kw] (a copy made by the caller).
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int DG_BK = 16, DG_APITCH = DG_BK + 8, DG_BM = 64, DG_PIX = 128;

// (a compiler builtin since round 6, it was inline assembly behind a hand-placed s_waitcnt: the scheduler interleaves the reads with the
// MFMAs and counts its own waits)
typedef short dg_tr16_v4i16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 dg_tr16(const uint16_t* p) {
    const dg_tr16_v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dg_tr16_v4i16 __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(u32x2, v);
}

// X3 = true ("f32x3", see conv1x1_fwd.hip): dY / dX float32, A9 pre-split into bfloat16 (hi, lo) planes; a staged 8-pixel chunk (two
// float4 loads) is split into its halves and staged into two sets of copies; hi*hi + hi*lo + lo*hi per fragment pair, float32
// accumulation; a lane stores its pixel pair as one 8-byte float2.  ~92 KB of LDS: one workgroup per CU.
template <int WO, bool X3 = false>
struct DgCfg {
    static constexpr int ROWS = DG_PIX / WO;                        // dY rows per tile
    static constexpr int SPX = (ROWS + 1) * WO;                     // staged pixels per channel (one row below)
    static constexpr int BP = ((SPX + 127) / 128) * 128 + (X3 ? 32 : 16);    // 32 / 64 bytes (mod 256): see conv3x3_fwd.hip
    static constexpr int PL = X3 ? 2 : 1;
    static constexpr int A_EL = 9 * DG_BM * DG_APITCH, B_EL = 2 * DG_BK * BP;          // per plane
    static constexpr size_t lds_bytes = (size_t)PL * (A_EL + B_EL) * sizeof(uint16_t);
};

template <int WO, bool X3>
__global__ __launch_bounds__(256, X3 ? 1 : 2) void k_dgrad3x3_s2(const uint16_t* __restrict__ A9, const uint16_t* __restrict__ A9_lo,
                                                                 const void* __restrict__ DY_, void* __restrict__ DX_, int C, int M, int Ho,
                                                                 int tiles_c, int tiles_r, int pts, int stream) {
    using Cfg = DgCfg<WO, X3>;
    constexpr int PL = Cfg::PL;
    constexpr int ROWS = Cfg::ROWS, BP = Cfg::BP, CPR = WO / 8, SR = ROWS + 1;
    constexpr int NA = 9 * DG_BM * 2, LA = (NA + 255) / 256;
    constexpr int NB = DG_BK * SR * CPR, LB = (NB + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* As = lds;                    // [PL][9][64][DG_APITCH]
    uint16_t* Bs = lds + PL * Cfg::A_EL;   // [PL][2 copies][16][BP]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wc = wv >> 1, wp = wv & 1;                   // channel half, pixel half of the tile
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tc = q % tiles_c, pt = (q / tiles_c) * 8 + xcd;
    if (pt >= pts) return;
    const int n = pt / tiles_r, tr = pt - n * tiles_r;
    const int c0 = tc * DG_BM, i0 = tr * ROWS;
    const size_t HWo = (size_t)Ho * WO;
    const uint16_t* dyn = reinterpret_cast<const uint16_t*>(DY_) + (X3 ? 0 : (size_t)n * M * HWo);
    const float* dynf = reinterpret_cast<const float*>(DY_) + (X3 ? (size_t)n * M * HWo : 0);

    uint4 ra[PL][LA], rb[PL][LB];          // X3: rb[0] / rb[1] = pixels 0..3 / 4..7 of the chunk as float32
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int id = tid + 256 * i;
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) ra[pl][i] = make_uint4(0, 0, 0, 0);
            if (id < NA) {
                const int t = id / (DG_BM * 2), r = id - t * (DG_BM * 2), c = c0 + (r >> 1), k = k0 + (r & 1) * 8;
                if (c < C && k < M) {
                    ra[0][i] = *reinterpret_cast<const uint4*>(A9 + ((size_t)t * C + c) * M + k);
                    if (X3) ra[PL - 1][i] = *reinterpret_cast<const uint4*>(A9_lo + ((size_t)t * C + c) * M + k);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int id = tid + 256 * i;
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) rb[pl][i] = make_uint4(0, 0, 0, 0);
            if (id < NB) {
                const int cc = id / (SR * CPR), r2 = id - cc * (SR * CPR), rr = r2 / CPR, ch = r2 - rr * CPR;
                const int k = k0 + cc, ii = i0 + rr;
                if (k < M && ii < Ho) {
                    if (X3) {
                        const float* src = dynf + ((size_t)k * Ho + ii) * WO + ch * 8;
                        rb[0][i] = *reinterpret_cast<const uint4*>(src);
                        rb[PL - 1][i] = *reinterpret_cast<const uint4*>(src + 4);
                    } else {
                        rb[0][i] = *reinterpret_cast<const uint4*>(dyn + ((size_t)k * Ho + ii) * WO + ch * 8);
                    }
                }
            }
        }
    };

    f32x16 d[4][2];                        // [class 2 py + px][pixel tile]
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) d[cl][ni][r] = 0.0f;

    const int g = lane >> 5, i16 = lane & 15, gi = (lane >> 4) & 1;
    const uint16_t* a_base = As + (wc * 32 + (lane & 31)) * DG_APITCH + 8 * g;
    const uint16_t* b_base = Bs + (8 * g + (i16 >> 2)) * BP + wp * 64 + 16 * gi + 4 * (i16 & 3);

    fetch(0);
    for (int k0 = 0; k0 < M; k0 += DG_BK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int id = tid + 256 * i;
            if (id < NA) {
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) *reinterpret_cast<uint4*>(As + pl * Cfg::A_EL + (id >> 1) * DG_APITCH + (id & 1) * 8) = ra[pl][i];
            }
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int id = tid + 256 * i;
            uint4 vv[PL];
            if (X3) {
                uint2 h0, l0, h1, l1;
                const uint4 q0 = rb[0][i], q1 = rb[PL - 1][i];
                aadg_split4(make_float4(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w)), h0, l0);
                aadg_split4(make_float4(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z), __uint_as_float(q1.w)), h1, l1);
                vv[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
                vv[PL - 1] = make_uint4(l0.x, l0.y, l1.x, l1.y);
            } else {
                vv[0] = rb[0][i];
            }
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) {
                const uint4 v = vv[pl];
                uint32_t right = __shfl_down(v.x, 1, 64);      // the next chunk of the row sits in the next lane (CPR divides 64)
                if (id < NB) {
                    const int cc = id / (SR * CPR), r2 = id - cc * (SR * CPR), ch = r2 % CPR;
                    if (ch == CPR - 1) right = 0u;             // j + 1 = WO: no such dY column
                    uint16_t* dst = Bs + pl * Cfg::B_EL + cc * BP + r2 * 8;
                    *reinterpret_cast<uint4*>(dst) = v;                                                                     // dY[i][j]
                    *reinterpret_cast<uint4*>(dst + DG_BK * BP) =
                        make_uint4(__builtin_amdgcn_alignbit(v.y, v.x, 16), __builtin_amdgcn_alignbit(v.z, v.y, 16),
                                   __builtin_amdgcn_alignbit(v.w, v.z, 16), __builtin_amdgcn_alignbit(right, v.w, 16));    // dY[i][j + 1]
                }
            }
        }
        __syncthreads();
        if (k0 + DG_BK < M) fetch(k0 + DG_BK);
        bf16x8 a[PL][9];
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
#pragma unroll
            for (int t = 0; t < 9; ++t)
                a[pl][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a_base + pl * Cfg::A_EL + t * DG_BM * DG_APITCH));
        // one pixel tile at a time: the 8 transpose reads of the second tile reuse the registers of the first (16 instead of 32 live)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            u32x2 lo[PL][2][2], hi[PL][2][2];      // [plane][dj][di]
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int dj = 0; dj < 2; ++dj)
#pragma unroll
                    for (int di = 0; di < 2; ++di) {
                        const uint16_t* p = b_base + pl * Cfg::B_EL + dj * DG_BK * BP + di * WO + 32 * ni;
                        lo[pl][dj][di] = dg_tr16(p);
                        hi[pl][dj][di] = dg_tr16(p + 4 * BP);
                    }
            bf16x8 b[PL][2][2];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int dj = 0; dj < 2; ++dj)
#pragma unroll
                    for (int di = 0; di < 2; ++di)
                        b[pl][dj][di] = __builtin_bit_cast(bf16x8, make_uint4(lo[pl][dj][di].x, lo[pl][dj][di].y, hi[pl][dj][di].x, hi[pl][dj][di].y));
            // tap t = kh * 3 + kw; kh = 1 <-> (py 0, di 0), kh = 0 <-> (py 1, di 1), kh = 2 <-> (py 1, di 0); the same for kw / px / dj
            // (class, tap, dj, di) of the nine products
            constexpr int TAPS[9][4] = {{0, 4, 0, 0}, {1, 3, 1, 0}, {2, 1, 0, 1}, {3, 0, 1, 1}, {1, 5, 0, 0}, {2, 7, 0, 0}, {3, 2, 0, 1},
                                        {3, 6, 1, 0}, {3, 8, 0, 0}};
            if (X3) {
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    d[TAPS[i][0]][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1][TAPS[i][1]], b[0][TAPS[i][2]][TAPS[i][3]],
                                                                                d[TAPS[i][0]][ni], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    d[TAPS[i][0]][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][TAPS[i][1]], b[PL - 1][TAPS[i][2]][TAPS[i][3]],
                                                                                d[TAPS[i][0]][ni], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 9; ++i)
                d[TAPS[i][0]][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][TAPS[i][1]], b[0][TAPS[i][2]][TAPS[i][3]], d[TAPS[i][0]][ni],
                                                                            0, 0, 0);
        }
    }
    // C/D layout: column (dY-grid pixel) = lane & 31, row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).  Output pixel pair
    // (2j, 2j + 1) of row 2i + py = classes (py, 0), (py, 1): one 4-byte store
    const int WI = 2 * WO;
    const size_t HWi = (size_t)4 * Ho * WO;
    if (X3) {
        float* dxf = reinterpret_cast<float*>(DX_) + (size_t)n * C * HWi;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int ql = wp * 64 + 32 * ni + (lane & 31);
            const int il = ql / WO, j = ql - il * WO, i = i0 + il;
            if (i >= Ho) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (c >= C) continue;
                float* row = dxf + (size_t)c * HWi + (size_t)(2 * i) * WI + 2 * j;
                *reinterpret_cast<float2*>(row) = make_float2(d[0][ni][r], d[1][ni][r]);
                *reinterpret_cast<float2*>(row + WI) = make_float2(d[2][ni][r], d[3][ni][r]);
            }
        }
        return;
    }
    uint16_t* dxn = reinterpret_cast<uint16_t*>(DX_) + (size_t)n * C * HWi;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int ql = wp * 64 + 32 * ni + (lane & 31);   // pixel of the tile, row-major over ROWS x WO
        const int il = ql / WO, j = ql - il * WO, i = i0 + il;
        if (i >= Ho) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (c >= C) continue;
            uint16_t* row = dxn + (size_t)c * HWi + (size_t)(2 * i) * WI + 2 * j;
            const uint32_t v0 = aadg_f2bf_pk(d[0][ni][r], d[1][ni][r]), v1 = aadg_f2bf_pk(d[2][ni][r], d[3][ni][r]);
            if (stream) {                                  // dX (4x dY) too large to stay cached until the BatchNorm backward reads it
                __builtin_nontemporal_store(v0, reinterpret_cast<uint32_t*>(row));
                __builtin_nontemporal_store(v1, reinterpret_cast<uint32_t*>(row + WI));
            } else {
                *reinterpret_cast<uint32_t*>(row) = v0;
                *reinterpret_cast<uint32_t*>(row + WI) = v1;
            }
        }
    }
}

template <int WO, bool X3>
int launch(const uint16_t* A9, const uint16_t* A9_lo, const void* DY, void* DX, int N, int C, int M, int Ho, hipStream_t st) {
    using Cfg = DgCfg<WO, X3>;
    const int tiles_c = (C + DG_BM - 1) / DG_BM, tiles_r = (Ho + Cfg::ROWS - 1) / Cfg::ROWS;
    const long long pts = (long long)N * tiles_r, groups = (pts + 7) / 8;
    const long long wgs = groups * 8 * tiles_c;
    if (wgs > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dgrad3x3_s2<WO, X3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)Cfg::lds_bytes));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_dgrad3x3_s2<WO, X3>), dim3((unsigned)wgs), dim3(256), Cfg::lds_bytes, st, A9, A9_lo, DY, DX, C, M, Ho, tiles_c, tiles_r, (int)pts,
                       (size_t)N * C * 4 * Ho * WO * 2 > ((size_t)128 << 20) ? 1 : 0);
    AADG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int aadg_conv3x3s2_dgrad_supported(int C, int M, int Ho, int Wo) {
    return C > 0 && M > 0 && (M % 8) == 0 && Ho > 0 && (Wo == 32 || Wo == 64) ? 1 : 0;
}

/* dx [N, C, 2 Ho, 2 Wo] = input gradient of a 3x3 / stride-2 / padding-1 convolution from dy [N, M, Ho, Wo] and
 * a9t [9, C, M] (a9t[kh * 3 + kw][c][m] = weight[m][c][kh][kw]); all bfloat16, float32 accumulation */
extern "C" int aadg_conv3x3s2_dgrad_bf16(const void* a9t, const void* dy, void* dx, int N, int C, int M, int Ho, int Wo, void* stream) {
    if (a9t == nullptr || dy == nullptr || dx == nullptr || N <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)a9t | (uintptr_t)dy | (uintptr_t)dx) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_conv3x3s2_dgrad_supported(C, M, Ho, Wo)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (Wo == 32) return launch<32, false>((const uint16_t*)a9t, nullptr, dy, dx, N, C, M, Ho, st);
    return launch<64, false>((const uint16_t*)a9t, nullptr, dy, dx, N, C, M, Ho, st);
}

/* The same input gradient at float32 precision ("f32x3"): dy / dx float32 NCHW; a9t_hi / a9t_lo = the bfloat16 halves of a9t */
extern "C" int aadg_conv3x3s2_dgrad_f32x3(const void* a9t_hi, const void* a9t_lo, const float* dy, float* dx, int N, int C, int M, int Ho,
                                          int Wo, void* stream) {
    if (a9t_hi == nullptr || a9t_lo == nullptr || dy == nullptr || dx == nullptr || N <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)a9t_hi | (uintptr_t)a9t_lo | (uintptr_t)dy | (uintptr_t)dx) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_conv3x3s2_dgrad_supported(C, M, Ho, Wo)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (Wo == 32) return launch<32, true>((const uint16_t*)a9t_hi, (const uint16_t*)a9t_lo, dy, dx, N, C, M, Ho, st);
    return launch<64, true>((const uint16_t*)a9t_hi, (const uint16_t*)a9t_lo, dy, dx, N, C, M, Ho, st);
}

// 3x3 / stride-2 / padding-1 convolution over NCHW bfloat16 activations as an implicit GEMM on the matrix cores (the first block of
// ResNet stages 2 and 3), without a layout change of the activations:
//
//     OUT[n][m][y][x] = sum_k sum_{kh,kw} A9[kh * 3 + kw][m][k] * IN[n][k][2 y + kh - 1][2 x + kw - 1]        (zero outside the image)
//
// Same scheme as conv3x3_fwd.hip -- rows of IN go to LDS as [k][pixel], ds_read_b64_tr_b16 hands every lane 4 consecutive k of its
// pixel -- with the stride folded into the staging step: a transpose read wants the 4 x-neighbours of an OUTPUT row contiguous, i.e.
// every second input column.  The staging step therefore splits each 8-pixel chunk of an input row into its even and odd columns
// (v_perm_b32) and writes three column planes per channel: E[x] = IN[2 x] (kw = 1), O[x] = IN[2 x + 1] (kw = 2) and O shifted by one
// (IN[2 x - 1], kw = 0; the pixel of the neighbouring chunk comes through a lane shuffle, zero at x = 0 = the padding column).
// The planes hold the 2 ROWS + 1 input rows of the tile's ROWS output rows in order, so tap row kh of output row y is plane row
// 2 y + kh: an aligned offset.  All fragment reads are aligned and branch-free.
//
// Workgroup = 4 waves = 64 out channels x 128 output pixels (2 or 4 whole output rows of one image; 2 x 2 waves of 32 x 64);
// K-step = 16 input channels; the next step's global loads are in flight during the 18 MFMAs per wave.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int S2_BK = 16, S2_APITCH = S2_BK + 8, S2_BM = 64, S2_PIX = 128;

// (a compiler builtin since round 6, it was inline assembly behind a hand-placed s_waitcnt: the scheduler interleaves the reads with the
// MFMAs and counts its own waits)
typedef short s2_tr16_v4i16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 s2_tr16(const uint16_t* p) {
    const s2_tr16_v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s2_tr16_v4i16 __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(u32x2, v);
}

// X3 = true ("f32x3", see conv1x1_fwd.hip): IN / OUT float32, A9 pre-split into bfloat16 (hi, lo) planes; a staged 8-pixel chunk (two
// float4 loads) is split into its (hi, lo) halves and both go through the same even / odd staging into two sets of column planes;
// hi*hi + hi*lo + lo*hi per fragment pair, float32 accumulation.  132 KB of LDS: one workgroup per CU.
template <int WO, bool X3 = false>
struct S2Cfg {
    static constexpr int ROWS = S2_PIX / WO;                        // output rows per tile
    static constexpr int R = 2 * ROWS + 1;                          // input rows staged
    static constexpr int SPX = R * WO;                              // plane pixels per channel
    static constexpr int BP = ((SPX + 127) / 128) * 128 + (X3 ? 32 : 16);    // 32 / 64 bytes (mod 256): see conv3x3_fwd.hip
    static constexpr int PL = X3 ? 2 : 1;
    static constexpr int A_EL = 9 * S2_BM * S2_APITCH, B_EL = 3 * S2_BK * BP;          // per plane
    static constexpr size_t lds_bytes = (size_t)PL * (A_EL + B_EL) * sizeof(uint16_t);
};

template <int WO, bool X3>
__global__ __launch_bounds__(256, X3 ? 1 : 2) void k_conv3x3_s2(const uint16_t* __restrict__ A9, const uint16_t* __restrict__ A9_lo,
                                                                const void* __restrict__ IN_, void* __restrict__ OUT_, int M, int K, int Ho,
                                                                int tiles_m, int tiles_r, int pts) {
    using Cfg = S2Cfg<WO, X3>;
    constexpr int PL = Cfg::PL;
    constexpr int ROWS = Cfg::ROWS, R = Cfg::R, BP = Cfg::BP, WI = 2 * WO, CPR = WI / 8;
    constexpr int NA = 9 * S2_BM * 2, LA = (NA + 255) / 256;
    constexpr int NB = S2_BK * R * CPR, LB = (NB + 255) / 256;
    constexpr int NSTEP = WO == 64 ? 32 : 64;              // plane offset of the wave's second 32-pixel tile (same output row / next one)
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* As = lds;                    // [PL][9][64][S2_APITCH]
    uint16_t* Bs = lds + PL * Cfg::A_EL;   // [PL][3 planes: kw][16][BP]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wc = wv >> 1, wp = wv & 1;                   // channel half, pixel half of the tile
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tm = q % tiles_m, pt = (q / tiles_m) * 8 + xcd;
    if (pt >= pts) return;
    const int n = pt / tiles_r, tr = pt - n * tiles_r;
    const int m0 = tm * S2_BM, y0 = tr * ROWS;
    const int Hi = 2 * Ho;
    const uint16_t* inn = reinterpret_cast<const uint16_t*>(IN_) + (X3 ? 0 : (size_t)n * K * Hi * WI);
    const float* innf = reinterpret_cast<const float*>(IN_) + (X3 ? (size_t)n * K * Hi * WI : 0);

    uint4 ra[PL][LA], rb[PL][LB];          // X3: rb[0] / rb[1] = pixels 0..3 / 4..7 of the chunk as float32
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int id = tid + 256 * i;
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) ra[pl][i] = make_uint4(0, 0, 0, 0);
            if (id < NA) {
                const int t = id / (S2_BM * 2), r = id - t * (S2_BM * 2), m = m0 + (r >> 1), k = k0 + (r & 1) * 8;
                if (m < M && k < K) {
                    ra[0][i] = *reinterpret_cast<const uint4*>(A9 + ((size_t)t * M + m) * K + k);
                    if (X3) ra[PL - 1][i] = *reinterpret_cast<const uint4*>(A9_lo + ((size_t)t * M + m) * K + k);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int id = tid + 256 * i;
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) rb[pl][i] = make_uint4(0, 0, 0, 0);
            if (id < NB) {
                const int cc = id / (R * CPR), r2 = id - cc * (R * CPR), rr = r2 / CPR, ch = r2 - rr * CPR;
                const int k = k0 + cc, iy = 2 * y0 - 1 + rr;
                if (k < K && iy >= 0 && iy < Hi) {
                    if (X3) {
                        const float* src = innf + ((size_t)k * Hi + iy) * WI + ch * 8;
                        rb[0][i] = *reinterpret_cast<const uint4*>(src);
                        rb[PL - 1][i] = *reinterpret_cast<const uint4*>(src + 4);
                    } else {
                        rb[0][i] = *reinterpret_cast<const uint4*>(inn + ((size_t)k * Hi + iy) * WI + ch * 8);
                    }
                }
            }
        }
    };

    f32x16 d[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[ni][r] = 0.0f;

    const int g = lane >> 5, i16 = lane & 15, gi = (lane >> 4) & 1;
    const uint16_t* a_base = As + (wc * 32 + (lane & 31)) * S2_APITCH + 8 * g;
    const uint16_t* b_base = Bs + (8 * g + (i16 >> 2)) * BP + wp * 128 + 16 * gi + 4 * (i16 & 3);

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += S2_BK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int id = tid + 256 * i;
            if (id < NA) {
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) *reinterpret_cast<uint4*>(As + pl * Cfg::A_EL + (id >> 1) * S2_APITCH + (id & 1) * 8) = ra[pl][i];
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
                // even / odd columns of the 8 pixels: (p0, p2), (p4, p6) and (p1, p3), (p5, p7)
                const uint32_t e0 = __builtin_amdgcn_perm(v.y, v.x, 0x05040100u), e1 = __builtin_amdgcn_perm(v.w, v.z, 0x05040100u);
                const uint32_t o0 = __builtin_amdgcn_perm(v.y, v.x, 0x07060302u), o1 = __builtin_amdgcn_perm(v.w, v.z, 0x07060302u);
                uint32_t prev = __shfl_up(o1, 1, 64);          // the previous chunk of the row sits in the previous lane (CPR divides 64)
                if (id < NB) {
                    const int cc = id / (R * CPR), r2 = id - cc * (R * CPR), rr = r2 / CPR, ch = r2 - rr * CPR;
                    if (ch == 0) prev = 0u;                    // 2 x - 1 = -1: the padding column
                    uint16_t* dst = Bs + pl * Cfg::B_EL + cc * BP + rr * WO + 4 * ch;
                    *reinterpret_cast<uint2*>(dst) =
                        make_uint2(__builtin_amdgcn_alignbit(o0, prev, 16), __builtin_amdgcn_alignbit(o1, o0, 16));       // kw = 0: IN[2 x - 1]
                    *reinterpret_cast<uint2*>(dst + S2_BK * BP) = make_uint2(e0, e1);                                        // kw = 1: IN[2 x]
                    *reinterpret_cast<uint2*>(dst + 2 * S2_BK * BP) = make_uint2(o0, o1);                                    // kw = 2: IN[2 x + 1]
                }
            }
        }
        __syncthreads();
        if (k0 + S2_BK < K) fetch(k0 + S2_BK);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            bf16x8 a[PL][3];
            u32x2 lo[PL][3][2], hi[PL][3][2];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    a[pl][kw] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a_base + pl * Cfg::A_EL +
                                                                                          (kh * 3 + kw) * S2_BM * S2_APITCH));
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const uint16_t* p = b_base + pl * Cfg::B_EL + kw * S2_BK * BP + kh * WO + ni * NSTEP;
                        lo[pl][kw][ni] = s2_tr16(p);
                        hi[pl][kw][ni] = s2_tr16(p + 4 * BP);
                    }
                }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const bf16x8 b = __builtin_bit_cast(bf16x8, make_uint4(lo[0][kw][ni].x, lo[0][kw][ni].y, hi[0][kw][ni].x, hi[0][kw][ni].y));
                    if (X3) {
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, make_uint4(lo[PL - 1][kw][ni].x, lo[PL - 1][kw][ni].y, hi[PL - 1][kw][ni].x,
                                                                                hi[PL - 1][kw][ni].y));
                        d[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1][kw], b, d[ni], 0, 0, 0);
                        d[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][kw], bl, d[ni], 0, 0, 0);
                    }
                    d[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][kw], b, d[ni], 0, 0, 0);
                }
        }
    }
    // C/D layout: column (pixel) = lane & 31, row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); lanes p / p + 1 trade
    // registers r / r + 1 so that each stores two adjacent pixels of one channel row (4-byte stores)
    const int jj = lane & 31;
    const size_t HWo = (size_t)Ho * WO;
    if (X3) {
        float* outf = reinterpret_cast<float*>(OUT_) + (size_t)n * M * HWo;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int ql = wp * 64 + 32 * ni;
            const int y = y0 + ql / WO, x = ql % WO + jj;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (m < M && y < Ho) outf[(size_t)m * HWo + (size_t)y * WO + x] = d[ni][r];
            }
        }
        return;
    }
    const bool odd = jj & 1;
    uint16_t* outn = reinterpret_cast<uint16_t*>(OUT_) + (size_t)n * M * HWo;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int ql = wp * 64 + 32 * ni;                  // first pixel of the 32-pixel tile, row-major over ROWS x WO
        const int y = y0 + ql / WO, x = ql % WO + (jj & ~1);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float mine0 = d[ni][r], mine1 = d[ni][r + 1];
            const float give = odd ? mine0 : mine1;
            const float got = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give), 0xB1, 0xF, 0xF, true));
            const float lo2 = odd ? got : mine0, hi2 = odd ? mine1 : got;
            const int rsel = r + (odd ? 1 : 0);
            const int m = m0 + wc * 32 + (rsel & 3) + 8 * (rsel >> 2) + 4 * g;
            if (m < M && y < Ho) *reinterpret_cast<uint32_t*>(outn + (size_t)m * HWo + (size_t)y * WO + x) = aadg_f2bf_pk(lo2, hi2);
        }
    }
}

template <int WO, bool X3>
int launch(const uint16_t* A9, const uint16_t* A9_lo, const void* IN, void* OUT, int N, int M, int K, int Ho, hipStream_t st) {
    using Cfg = S2Cfg<WO, X3>;
    const int tiles_m = (M + S2_BM - 1) / S2_BM, tiles_r = (Ho + Cfg::ROWS - 1) / Cfg::ROWS;
    const long long pts = (long long)N * tiles_r, groups = (pts + 7) / 8;
    const long long wgs = groups * 8 * tiles_m;
    if (wgs > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_s2<WO, X3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)Cfg::lds_bytes));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_conv3x3_s2<WO, X3>), dim3((unsigned)wgs), dim3(256), Cfg::lds_bytes, st, A9, A9_lo, IN, OUT, M, K, Ho, tiles_m,
                       tiles_r, (int)pts);
    AADG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int aadg_conv3x3s2_nchw_supported(int M, int K, int Ho, int Wo) {
    return M > 0 && K > 0 && (K % 8) == 0 && Ho > 0 && (Wo == 32 || Wo == 64) ? 1 : 0;
}

/* out [N, M, Ho, Wo] = 3x3 / stride-2 / padding-1 convolution of in [N, K, 2 Ho, 2 Wo] with the tap-major weights a9 [9, M, K];
 * all bfloat16, float32 accumulation */
extern "C" int aadg_conv3x3s2_nchw_bf16(const void* a9, const void* in, void* out, int N, int M, int K, int Ho, int Wo, void* stream) {
    if (a9 == nullptr || in == nullptr || out == nullptr || N <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)a9 | (uintptr_t)in | (uintptr_t)out) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_conv3x3s2_nchw_supported(M, K, Ho, Wo)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (Wo == 32) return launch<32, false>((const uint16_t*)a9, nullptr, in, out, N, M, K, Ho, st);
    return launch<64, false>((const uint16_t*)a9, nullptr, in, out, N, M, K, Ho, st);
}

/* The same convolution at float32 precision ("f32x3"): in / out float32 NCHW; a9_hi / a9_lo = the bfloat16 halves of the float32 tap-major
 * weights [9, M, K] */
extern "C" int aadg_conv3x3s2_nchw_f32x3(const void* a9_hi, const void* a9_lo, const float* in, float* out, int N, int M, int K, int Ho,
                                         int Wo, void* stream) {
    if (a9_hi == nullptr || a9_lo == nullptr || in == nullptr || out == nullptr || N <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)a9_hi | (uintptr_t)a9_lo | (uintptr_t)in | (uintptr_t)out) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_conv3x3s2_nchw_supported(M, K, Ho, Wo)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (Wo == 32) return launch<32, true>((const uint16_t*)a9_hi, (const uint16_t*)a9_lo, in, out, N, M, K, Ho, st);
    return launch<64, true>((const uint16_t*)a9_hi, (const uint16_t*)a9_lo, in, out, N, M, K, Ho, st);
}

// 3x3 / stride-1 / padding = dilation convolution at float32 precision ("f32x3", see conv1x1_fwd.hip) with an LDS layout made for it:
//
//     OUT[n][m][y][x] = sum_k sum_{kh,kw} A9[kh * 3 + kw][m][k] * IN[n][k][y + (kh - 1) D][x + (kw - 1) D]      IN / OUT float32 NCHW
//
// k_conv3x3_nchw<.., X3 = true> (conv3x3_fwd.hip) keeps the bfloat16 kernel's layout -- [k][pixel] rows read with ds_read_b64_tr_b16,
// hence THREE column-shifted copies of every staged row -- twice (hi and lo planes): 132-160 KB of LDS, one workgroup per CU, one wave
// per SIMD.  At that occupancy the kernel is issue-bound and serialised: timing-only ablations on the 512-channel layer (2.55 ms) --
// a third of the MFMAs: 1.82 ms; no LDS stores of the IN tile: 1.95; no global loads: 2.08 -- say the 0.95 ms of MFMA issue hide almost
// nothing of the staging (scripts/ab/x3_ablate.sh).  What hides it in the bfloat16 kernel is the SECOND workgroup per CU.
//
// Here the staged tile is [pixel][k]: one 64-byte record per pixel = 16 input channels as four 16-byte chunks [hi k..k+3][lo k..k+3].
//   * a B fragment (8 consecutive k of ONE pixel, hi and lo) = two adjacent chunks of that pixel's record: plain ds_read_b128, any pixel
//     offset -- a tap is an address, no shifted copies, no transpose read; rows carry D zero pixels at both ends (the padding columns)
//     and the rows above / below the image stay zero (written once);
//   * the transposition happens in registers while staging: a thread loads 4 pixels x 4 channels (four float4), and the (hi, lo) split
//     of the four CHANNELS of one pixel is one chunk (aadg_split4 packs pairs across channels);
//   * pixel records sit 80 bytes apart (16 of padding): the 16 lanes of a ds_read_b128 group -- 16 consecutive pixels -- hit 16 distinct
//     bank quads (20 i mod 64 dwords), and so do the staging stores (4 chunks of a record x 4 records 320 bytes apart); a tap is a
//     compile-time offset from ONE per-lane address;
//   * weights: [tap][m] records of 64 bytes, chunk index ^ ((record >> 2) & 3) (from the pre-split planes of
//     aadg_weight_layouts_split_bf16).
// LDS: 37 KB of weights + 27-44 KB of pixels = 64-80 KB: TWO workgroups per CU.  Whole tiles only (M % 64 == 0, K % 16 == 0: every such
// layer of the backbone); other shapes take k_conv3x3_nchw<.., true>.  Input gradient: the same kernel on dY with mirrored taps.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int Q_BM = 64, Q_PIX = 256, Q_BK = 16;

template <int W, int D>
struct QCfg {
    static constexpr int ROWS = Q_PIX / W;                      // image rows per tile
    static constexpr int SR = ROWS + 2 * D;                     // staged rows
    static constexpr int PW = W + 2 * D;                        // staged row: D zero pixels | W pixels | D zero pixels
    static constexpr int NPX = SR * PW;                         // pixel records
    static constexpr int A_BYTES = 9 * Q_BM * 64, B_REC = 80, B_BYTES = NPX * B_REC;
    static constexpr size_t lds_bytes = (size_t)A_BYTES + B_BYTES;
};

// byte offset of chunk c of record r
__device__ __forceinline__ int q_chunk(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

// PRE (K <= Q_PRE_K): IN is the INPUT of the BatchNorm + ReLU in front of this convolution -- the staged values become
// max(fma(x, pre_scale[k], pre_shift[k]), 0) (conv1x1_fwd.hip, PRE); the padding stays zero: it pads the NORMALISED tensor.
constexpr int Q_PRE_K = 512;
template <int W, int D, bool STATS, bool PRE = false>
__global__ __launch_bounds__(256, 2) void k_conv3x3_x3q(const uint16_t* __restrict__ A9, const uint16_t* __restrict__ A9_lo,
                                                        const float* __restrict__ IN, float* __restrict__ OUT, int M, int K, int H,
                                                        int tiles_m, int tiles_r, int pts, double* __restrict__ stats,
                                                        const float* __restrict__ pre_scale = nullptr,
                                                        const float* __restrict__ pre_shift = nullptr) {
    using C = QCfg<W, D>;
    constexpr int ROWS = C::ROWS, SR = C::SR, PW = C::PW, MI = 2, NI = 2, QPR = W / 4;     // QPR: pixel quads per image row
    constexpr int NAU = 9 * Q_BM * 2, LA = (NAU + 255) / 256;       // weight units of 8 k (16 bytes of each plane) per K-step
    constexpr int NBI = 4 * SR * QPR, LB = (NBI + 255) / 256;       // IN items of 4 channels x 4 pixels per K-step
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsq[];
    unsigned char* As = ldsq;                      // [9 * 64 records][64 B]
    unsigned char* Bs = ldsq + C::A_BYTES;         // [NPX records][80 B]
    float* Ps = reinterpret_cast<float*>(ldsq + C::lds_bytes);      // PRE: scale [K] | shift [K] behind the tiles (2 K floats of dynamic LDS:
                                                                    // 64 channels at W = 128 keep the second workgroup of the CU)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // XCD-aware decode: the out-channel tiles of one pixel tile run on one XCD and share the IN tile through its L2
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tm = q % tiles_m, pt = (q / tiles_m) * 8 + xcd;
    if (pt >= pts) return;
    const int n = pt / tiles_r, tr = pt - n * tiles_r;
    const int m0 = tm * Q_BM, y0 = tr * ROWS;
    const size_t HW = (size_t)H * W;
    const float* inn = IN + (size_t)n * K * HW;

    // zero the pixel records once: the padding columns and the rows outside the image are never written again
    for (int i = tid; i < C::B_BYTES / 16; i += 256) reinterpret_cast<uint4*>(Bs)[i] = make_uint4(0, 0, 0, 0);
    if (PRE)
        for (int i = tid; i < K; i += 256) { Ps[i] = pre_scale[i]; Ps[K + i] = pre_shift[i]; }

    // per-thread staging descriptors (constant over the K loop)
    int a_src[LA], a_dst[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int id = min(tid + 256 * i, NAU - 1);                 // (the tail ids repeat the last unit: same data, same place)
        const int t = id / (Q_BM * 2), r = id - t * (Q_BM * 2), row = r >> 1, u = r & 1;
        a_src[i] = (t * M + m0 + row) * K + 8 * u;
        a_dst[i] = q_chunk(t * Q_BM + row, 2 * u);                  // chunks 2 u and 2 u + 1 of the record: a_dst and a_dst ^ 16
    }
    int b_src[LB], b_dst[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int id = tid + 256 * i;
        const int cq = id & 3, r2 = id >> 2, rr = r2 / QPR, pq = r2 - rr * QPR, yy = y0 - D + rr;     // lanes: chunk fastest, then pixel quad
        const bool ok = id < NBI && yy >= 0 && yy < H;
        b_src[i] = ok ? (4 * cq * H + yy) * W + 4 * pq : -1;
        b_dst[i] = (rr * PW + D + 4 * pq) * C::B_REC + 16 * cq;     // chunk cq of the record of the item's first pixel
    }

    uint4 rah[LA], ral[LA], rb[LB][4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            rah[i] = *reinterpret_cast<const uint4*>(A9 + a_src[i] + k0);
            ral[i] = *reinterpret_cast<const uint4*>(A9_lo + a_src[i] + k0);
        }
        const float* base = inn + (size_t)k0 * HW;
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const float* src = base + max(b_src[i], 0);            // (an idle item reads valid memory and is not stored)
#pragma unroll
            for (int c = 0; c < 4; ++c) rb[i][c] = *reinterpret_cast<const uint4*>(src + (size_t)c * HW);
        }
    };
    auto stage = [&](int k0) {
        float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), h4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (PRE) {                                  // the item's four channels: k0 + 4 (tid & 3) + 0..3 for every i (256 = 0 mod 4)
            s4 = *reinterpret_cast<const float4*>(Ps + k0 + 4 * (tid & 3));
            h4 = *reinterpret_cast<const float4*>(Ps + K + k0 + 4 * (tid & 3));
        }
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            if (tid + 256 * i < NAU) {
                *reinterpret_cast<uint4*>(As + a_dst[i]) = make_uint4(rah[i].x, rah[i].y, ral[i].x, ral[i].y);         // k 0..3: hi | lo
                *reinterpret_cast<uint4*>(As + (a_dst[i] ^ 16)) = make_uint4(rah[i].z, rah[i].w, ral[i].z, ral[i].w);  // k 4..7
            }
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            if (b_src[i] >= 0) {
                const float* f0 = reinterpret_cast<const float*>(&rb[i][0]);
                const float* f1 = reinterpret_cast<const float*>(&rb[i][1]);
                const float* f2 = reinterpret_cast<const float*>(&rb[i][2]);
                const float* f3 = reinterpret_cast<const float*>(&rb[i][3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {                      // pixel j of the quad: its four channels -> one chunk
                    uint2 hi, lo;
                    float4 f = make_float4(f0[j], f1[j], f2[j], f3[j]);
                    if (PRE) {
                        f.x = fmaxf(fmaf(f.x, s4.x, h4.x), 0.0f); f.y = fmaxf(fmaf(f.y, s4.y, h4.y), 0.0f);
                        f.z = fmaxf(fmaf(f.z, s4.z, h4.z), 0.0f); f.w = fmaxf(fmaf(f.w, s4.w, h4.w), 0.0f);
                    }
                    aadg_split4(f, hi, lo);
                    *reinterpret_cast<uint4*>(Bs + b_dst[i] + j * C::B_REC) = make_uint4(hi.x, hi.y, lo.x, lo.y);
                }
            }
        }
    };

    f32x16 d[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) d[mi][ni][r] = 0.0f;

    // fragment addresses (constant over the K loop).  A: record (tap, 32 mi + lane & 31), chunks 2 g and 2 g + 1; the tap adds 64 * 64
    // bytes and does not touch the swizzle bits.  B: the lane's tile pixel at tap (0, 0) -- staged row ty, record x --, chunks 2 g and
    // 2 g + 1; tap (kh, kw) adds (kh D PW + kw D) records.
    const int g = lane >> 5, l31 = lane & 31;
    int a_addr[MI], b_addr[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a_addr[mi] = q_chunk(32 * mi + l31, 2 * g);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int tp = wv * 64 + 32 * ni + l31, ty = tp / W, x = tp - ty * W;
        b_addr[ni] = (ty * PW + x) * C::B_REC + 32 * g;
    }

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += Q_BK) {
        __syncthreads();                                  // the previous step's fragment reads (and, first, the zero fill) are done
        stage(k0);
        __syncthreads();
        if (k0 + Q_BK < K) fetch(k0 + Q_BK);              // in flight during the MFMAs below
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const uint4 q0 = *reinterpret_cast<const uint4*>(As + t * Q_BM * 64 + a_addr[mi]);
                const uint4 q1 = *reinterpret_cast<const uint4*>(As + t * Q_BM * 64 + (a_addr[mi] ^ 16));
                ah[mi] = __builtin_bit_cast(bf16x8, make_uint4(q0.x, q0.y, q1.x, q1.y));
                al[mi] = __builtin_bit_cast(bf16x8, make_uint4(q0.z, q0.w, q1.z, q1.w));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int off = ((t / 3) * D * PW + (t % 3) * D) * C::B_REC;
                const uint4 q0 = *reinterpret_cast<const uint4*>(Bs + b_addr[ni] + off);
                const uint4 q1 = *reinterpret_cast<const uint4*>(Bs + b_addr[ni] + off + 16);
                bh[ni] = __builtin_bit_cast(bf16x8, make_uint4(q0.x, q0.y, q1.x, q1.y));
                bl[ni] = __builtin_bit_cast(bf16x8, make_uint4(q0.z, q0.w, q1.z, q1.w));
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    d[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], d[mi][ni], 0, 0, 0);
                    d[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], d[mi][ni], 0, 0, 0);
                    d[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], d[mi][ni], 0, 0, 0);
                }
        }
    }
    // C/D layout: column (pixel) = lane & 31, row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); the 32 lanes of a row store 128
    // contiguous bytes
    const size_t p_tile = (size_t)y0 * W + wv * 64;
    float* outf = OUT + (size_t)n * M * HW;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * g;
                const size_t p = p_tile + 32 * ni + l31;
                if (p < HW) outf[(size_t)m * HW + p] = d[mi][ni][r];
            }
    if (!STATS) return;
    // BatchNorm statistics of the output in the epilogue (the layer behind a bottleneck's 3x3 convolution is a BatchNorm): per channel sum
    // and sum of squares over the tile's pixels -> float64 atomics into stats[2 m], stats[2 m + 1], as k_conv1x1_nchw does
    // (conv1x1_fwd.hip: the halving butterfly over the 32 lanes of a row, the workgroup's four waves -- same 64 channels, 64 pixels each --
    // combined in LDS, one atomic per channel, statistic and workgroup).  Pixels beyond the image (a last tile of fewer rows) are not
    // stored and do not count.
    float v[64];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = 0.f, qq = 0.f;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const float e = (p_tile + 32 * ni + l31 < HW) ? d[mi][ni][r] : 0.0f;
                s += e; qq = fmaf(e, e, qq);
            }
            v[2 * (16 * mi + r)] = s;
            v[2 * (16 * mi + r) + 1] = qq;
        }
#define AADG_BFLY(MSK, HALF, PAT)                                                                       \
    {                                                                                                   \
        const bool up = (lane & (MSK)) != 0;                                                            \
        _Pragma("unroll") for (int i = 0; i < (HALF); ++i) {                                            \
            const float keep = up ? v[i + (HALF)] : v[i], send = up ? v[i] : v[i + (HALF)];             \
            v[i] = keep + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(send), (PAT)));     \
        }                                                                                               \
    }
    AADG_BFLY(16, 32, 0x1F | (16 << 10))
    AADG_BFLY(8, 16, 0x1F | (8 << 10))
    AADG_BFLY(4, 8, 0x1F | (4 << 10))
    AADG_BFLY(2, 4, 0x1F | (2 << 10))
    AADG_BFLY(1, 2, 0x1F | (1 << 10))
#undef AADG_BFLY
    // the item a lane is left with: bit 4 of the lane picked the upper 32 items, bit 3 the upper 16 of those, ...
    const int item = ((lane >> 4) & 1) * 32 + ((lane >> 3) & 1) * 16 + ((lane >> 2) & 1) * 8 + ((lane >> 1) & 1) * 4 + (lane & 1) * 2;
    const int rr = item >> 1, smi = rr >> 4, sr = rr & 15;
    const int row = 32 * smi + (sr & 3) + 8 * (sr >> 2) + 4 * g;
    float* red = reinterpret_cast<float*>(Bs);                // the pixel records are free once every wave has left the K loop
    __syncthreads();
    if (tid < 2 * Q_BM) red[tid] = 0.0f;
    __syncthreads();
    __hip_atomic_fetch_add(red + 2 * row, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(red + 2 * row + 1, v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (tid < 2 * Q_BM) unsafeAtomicAdd(stats + 2 * (size_t)m0 + tid, (double)red[tid]);
}

__global__ __launch_bounds__(256) void k_q_sums_init(double* __restrict__ sums, int C, double count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * C) sums[i] = 0.0;
    if (i == 2 * C) sums[i] = count;
}

template <int W, int D>
int launch_q(const uint16_t* A9, const uint16_t* A9_lo, const float* IN, float* OUT, int N, int M, int K, int H, double* stats, hipStream_t st,
             const float* pre_scale = nullptr, const float* pre_shift = nullptr) {
    using C = QCfg<W, D>;
    const int tiles_m = M / Q_BM, tiles_r = (H + C::ROWS - 1) / C::ROWS;
    const long long pts = (long long)N * tiles_r, groups = (pts + 7) / 8;
    const long long wgs = groups * 8 * tiles_m;
    if (wgs > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    static bool attr_set = false;                            // per instantiation; idempotent
    constexpr int PRE_BYTES = 2 * Q_PRE_K * (int)sizeof(float);
    if (!attr_set) {
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_x3q<W, D, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)C::lds_bytes));
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_x3q<W, D, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)C::lds_bytes));
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_x3q<W, D, true, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes + PRE_BYTES));
        attr_set = true;
    }
    if (pre_scale != nullptr && (stats == nullptr || K > Q_PRE_K)) return AADG_E_UNSUPPORTED;       // (the load transform comes with the statistics)
    if (stats != nullptr) {
        hipLaunchKernelGGL(k_q_sums_init, dim3((2 * M + 1 + 255) / 256), dim3(256), 0, st, stats, M, (double)N * (double)H * (double)W);
        AADG_LAUNCH_CHECK();
        if (pre_scale != nullptr)
            hipLaunchKernelGGL((k_conv3x3_x3q<W, D, true, true>), dim3((unsigned)wgs), dim3(256), C::lds_bytes + 2 * (size_t)K * sizeof(float), st, A9, A9_lo, IN, OUT,
                               M, K, H, tiles_m, tiles_r, (int)pts, stats, pre_scale, pre_shift);
        else
            hipLaunchKernelGGL((k_conv3x3_x3q<W, D, true>), dim3((unsigned)wgs), dim3(256), C::lds_bytes, st, A9, A9_lo, IN, OUT, M, K, H, tiles_m,
                               tiles_r, (int)pts, stats);
    } else {
        hipLaunchKernelGGL((k_conv3x3_x3q<W, D, false>), dim3((unsigned)wgs), dim3(256), C::lds_bytes, st, A9, A9_lo, IN, OUT, M, K, H, tiles_m,
                           tiles_r, (int)pts, stats);
    }
    AADG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// whole tiles only; the dispatch of aadg_conv3x3_nchw_f32x3 (conv3x3_fwd.hip) sends every other shape to k_conv3x3_nchw<.., true>
bool aadg_conv3x3_x3q_takes(int M, int K, int H, int W, int dilation) {
    return (M % Q_BM) == 0 && (K % Q_BK) == 0 && (W == 32 || W == 64 || W == 128) && (dilation == 1 || (dilation == 2 && W != 128)) &&
           (long long)9 * M * K <= 0x7FFFFFFFLL && (long long)K * H * W <= 0x7FFFFFFFLL;
}

// bn_sums (optional): float64 [2 M + 1] = (sum, sum of squares) per output channel and the element count, zeroed in here
int aadg_conv3x3_x3q(const uint16_t* a9_hi, const uint16_t* a9_lo, const float* in, float* out, int N, int M, int K, int H, int W,
                     int dilation, double* bn_sums, hipStream_t st, const float* pre_scale, const float* pre_shift) {
    if (dilation == 1) {
        if (W == 32) return launch_q<32, 1>(a9_hi, a9_lo, in, out, N, M, K, H, bn_sums, st, pre_scale, pre_shift);
        if (W == 64) return launch_q<64, 1>(a9_hi, a9_lo, in, out, N, M, K, H, bn_sums, st, pre_scale, pre_shift);
        return launch_q<128, 1>(a9_hi, a9_lo, in, out, N, M, K, H, bn_sums, st, pre_scale, pre_shift);
    }
    if (W == 32) return launch_q<32, 2>(a9_hi, a9_lo, in, out, N, M, K, H, bn_sums, st, pre_scale, pre_shift);
    return launch_q<64, 2>(a9_hi, a9_lo, in, out, N, M, K, H, bn_sums, st, pre_scale, pre_shift);
}

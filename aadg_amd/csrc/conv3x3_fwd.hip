// 3x3 / stride-1 / padding = dilation convolution over NCHW bfloat16 activations as an implicit GEMM on the matrix cores, without a
// layout change of the activations:
//
//     OUT[n][m][y][x] = sum_k sum_{kh,kw} A9[kh * 3 + kw][m][k] * IN[n][k][y + (kh - 1) D][x + (kw - 1) D]      (zero outside the image)
//
// forward: A9 = the weight, tap-major ([9][Co][Ci], a 9 x Co x Ci copy made by the caller); input gradient: the same kernel on dY
// with A9[t][c][m] = W[m][c][2 - kh][2 - kw].  The library path (MIOpen igemm_*_nhwc) transposes IN to NHWC and OUT back, and at
// small batches runs a K-split variant behind a zero-fill and in front of a cast: 125-240 us per convolution and direction at 18
// images per rank for 20-45 us of arithmetic.
//
// As in conv1x1_fwd.hip the K-strided operand (NCHW: channels are H*W apart) is read with gfx950's LDS transpose read: rows of IN go
// to LDS as they lie in memory ([k][pixel]) and ds_read_b64_tr_b16 hands every lane 4 consecutive k of ITS pixel.  The nine taps:
//   vertical   (kh): the staged region of a channel is (ROWS + 2 D) whole image rows, contiguous -- a tap row is an offset of D * W
//                    pixels in it (a multiple of 8 bytes: aligned);
//   horizontal (kw): a shift by D pixels = 2 or 4 bytes, which a transpose read (8-byte aligned blocks) cannot address.  The staging
//                    step therefore writes THREE copies of every 8-pixel chunk: as loaded, and shifted by -D / +D pixels with the
//                    pixels of the neighbouring chunks funnelled in through lane shuffles and zeros at the row ends (the padding
//                    columns).  All fragment reads are then aligned and branch-free.
// Workgroup = 4 waves = 64 x 256 tile (BM out channels x 256 pixels = 4 or 8 whole image rows of one image, 64 pixels per wave);
// K-step = 16 input channels = one MFMA K; the next step's global loads are in flight (registers) during the 36 MFMAs per wave.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int C3_BK = 16;                 // input channels per K-step
constexpr int C3_APITCH = C3_BK + 8;      // A rows: 48 bytes (conflict-free 16-byte fragment reads)
constexpr int C3_PIX = 256;               // pixels per workgroup tile

// (a compiler builtin since round 6, it was inline assembly behind a hand-placed s_waitcnt: the scheduler interleaves the reads with the
// MFMAs and counts its own waits)
typedef short lds_tr16_v4i16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 lds_tr16(const uint16_t* p) {
    const lds_tr16_v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr16_v4i16 __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(u32x2, v);
}

// X3 = true ("f32x3"): IN / OUT are float32, A9 comes pre-split as two bfloat16 planes (hi, lo).  A staged 8-pixel chunk (two float4
// loads) is split into its (hi, lo) bfloat16 halves (aadg_split4) and both halves go through the same three-copy staging into two planes
// of the same layout; every fragment pair is multiplied as hi*hi + hi*lo + lo*hi with float32 accumulation.  Both planes of both operands
// are 132-157 KB of LDS: one workgroup per CU, 108 MFMAs per wave and K-step.
template <int W, int D, int MI, bool X3 = false>
struct C3Cfg {
    static constexpr int BM = 32 * MI;                              // MI 32-row tiles per wave; the four waves share the m range
    static constexpr int ROWS = C3_PIX / W;                         // image rows per tile
    static constexpr int SPX = (ROWS + 2 * D) * W;                  // staged pixels per channel
    // pitch: 32 bytes (mod 256): the 4 rows of a transpose read hit distinct banks.  X3: 64 bytes (mod 256) -- a transpose read serves 32
    // lanes per LDS cycle (4 rows x two 16-pixel halves); with 8 banks per row step the second half of row r collides with the first
    // half of row r + 1, with 16 the 32 lanes cover the 64 banks exactly (conv1x1_fwd.hip)
    static constexpr int BP = ((SPX + 127) / 128) * 128 + (X3 ? 32 : 16);
    static constexpr int PL = X3 ? 2 : 1;                           // operand planes in LDS
    static constexpr int A_EL = 9 * 32 * MI * C3_APITCH, B_EL = 3 * C3_BK * BP;       // per plane
    static constexpr size_t lds_bytes = (size_t)PL * (A_EL + B_EL) * sizeof(uint16_t);
};

template <int W, int D, int MI, bool X3>
__global__ __launch_bounds__(256, X3 ? 1 : 2) void k_conv3x3_nchw(const uint16_t* __restrict__ A9, const uint16_t* __restrict__ A9_lo,
                                                                  const void* __restrict__ IN_, void* __restrict__ OUT_, int M, int K, int H,
                                                                  int tiles_m, int tiles_r, int pts) {
    using C = C3Cfg<W, D, MI, X3>;
    constexpr int PL = C::PL;
    constexpr int BM = 32 * MI, ROWS = C::ROWS, BP = C::BP, NI = 2, CPR = W / 8, SR = ROWS + 2 * D;
    constexpr int NA = 9 * BM * 2, LA = (NA + 255) / 256;            // 16-byte chunks of the A tile (9 taps x BM rows x 2) per K-step
    constexpr int NB = C3_BK * SR * CPR, LB = (NB + 255) / 256;      // ... of the IN tile
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* As = lds;                    // [PL][9][BM][C3_APITCH]
    uint16_t* Bs = lds + PL * C::A_EL;     // [PL][3 copies][16][BP]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // XCD-aware decode: the out-channel tiles of one pixel tile run on one XCD and share the IN tile through its L2
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tm = q % tiles_m, pt = (q / tiles_m) * 8 + xcd;
    if (pt >= pts) return;                                // the pixel tiles are padded to a multiple of 8
    const int n = pt / tiles_r, tr = pt - n * tiles_r;
    const int m0 = tm * BM, y0 = tr * ROWS;
    const size_t HW = (size_t)H * W;
    const uint16_t* inn = reinterpret_cast<const uint16_t*>(IN_) + (X3 ? 0 : (size_t)n * K * HW);
    const float* innf = reinterpret_cast<const float*>(IN_) + (X3 ? (size_t)n * K * HW : 0);

    // Per-thread staging descriptors, computed ONCE: which chunk of the weight slab / of the IN tile this thread moves in every K-step,
    // where it comes from (element offset at k0 = 0; -1: always zero -- beyond the tile, or a row above / below the image) and where it
    // goes in LDS.  Inside the K loop only `k0` is added: the divisions by SR * CPR (48, 96, ...) and the image-border tests were a
    // third of the kernel's VALU instructions when they were redone per K-step (SQ_INSTS_VALU 378 per wave and K-step for 108 MFMAs).
    int a_src[LA], a_dst[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int id = tid + 256 * i;
        const int t = id / (BM * 2), r = id - t * (BM * 2), m = m0 + (r >> 1);
        a_src[i] = (id < NA && m < M) ? (t * M + m) * K + (r & 1) * 8 : -1;               // (9 M K < 2^31: checked by the launcher)
        a_dst[i] = (id >> 1) * C3_APITCH + (id & 1) * 8;
    }
    int b_src[LB], b_dst[LB], b_cc[LB];
    bool b_first[LB], b_last[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int id = tid + 256 * i;
        const int cc = id / (SR * CPR), r2 = id - cc * (SR * CPR), rr = r2 / CPR, ch = r2 - rr * CPR, yy = y0 - D + rr;
        b_cc[i] = id < NB ? cc : C3_BK;                                                   // C3_BK: never below K - k0
        b_src[i] = (id < NB && yy >= 0 && yy < H) ? (cc * H + yy) * W + ch * 8 : -1;      // (16 H W < 2^31)
        b_dst[i] = cc * BP + r2 * 8;
        b_first[i] = ch == 0;
        b_last[i] = ch == CPR - 1;
    }
    uint4 ra[PL][LA], rb[PL][LB];          // X3: rb[0] / rb[1] = pixels 0..3 / 4..7 of the chunk as float32
    auto fetch = [&](int k0) {
        const int krem = K - k0;                                  // channels left: a chunk of channel cc is live iff cc < krem
#pragma unroll
        for (int i = 0; i < LA; ++i) {
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) ra[pl][i] = make_uint4(0, 0, 0, 0);
            if (a_src[i] >= 0 && ((tid + 256 * i) & 1) * 8 < krem) {
                ra[0][i] = *reinterpret_cast<const uint4*>(A9 + a_src[i] + k0);
                if (X3) ra[PL - 1][i] = *reinterpret_cast<const uint4*>(A9_lo + a_src[i] + k0);
            }
        }
        const size_t koff = (size_t)k0 * HW;
#pragma unroll
        for (int i = 0; i < LB; ++i) {
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) rb[pl][i] = make_uint4(0, 0, 0, 0);
            if (b_src[i] >= 0 && b_cc[i] < krem) {
                if (X3) {
                    const float* src = innf + koff + b_src[i];
                    rb[0][i] = *reinterpret_cast<const uint4*>(src);
                    rb[PL - 1][i] = *reinterpret_cast<const uint4*>(src + 4);
                } else {
                    rb[0][i] = *reinterpret_cast<const uint4*>(inn + koff + b_src[i]);
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

    const int g = lane >> 5, i16 = lane & 15, gi = (lane >> 4) & 1;
    const uint16_t* a_base = As + (lane & 31) * C3_APITCH + 8 * g;
    // transpose-read address of this lane: row (8 g + i16 / 4) of the K-step, pixel 64 wv + 16 gi + 4 (i16 % 4) of the tile (+ 32 ni)
    const uint16_t* b_base = Bs + (8 * g + (i16 >> 2)) * BP + wv * 64 + 16 * gi + 4 * (i16 & 3);

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += C3_BK) {
        __syncthreads();                                  // the previous step's fragment reads are done
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            if (tid + 256 * i < NA) {
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) *reinterpret_cast<uint4*>(As + pl * C::A_EL + a_dst[i]) = ra[pl][i];
            }
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int id = tid + 256 * i;
            // neighbouring chunks of the image row sit in the neighbouring lanes (CPR divides 64): the pixel(s) shifted in
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
                uint32_t left = __shfl_up(v.w, 1, 64), right = __shfl_down(v.x, 1, 64);
                if (id < NB) {
                    if (b_first[i]) left = 0u;                 // the padding columns
                    if (b_last[i]) right = 0u;
                    uint16_t* dst = Bs + pl * C::B_EL + b_dst[i];
                    uint4 vm, vp;                              // vm[p] = IN[p - D], vp[p] = IN[p + D]
                    if (D == 1) {
                        const uint32_t s1 = __builtin_amdgcn_alignbit(v.y, v.x, 16), s2 = __builtin_amdgcn_alignbit(v.z, v.y, 16),
                                       s3 = __builtin_amdgcn_alignbit(v.w, v.z, 16);
                        vm = make_uint4(__builtin_amdgcn_alignbit(v.x, left, 16), s1, s2, s3);
                        vp = make_uint4(s1, s2, s3, __builtin_amdgcn_alignbit(right, v.w, 16));
                    } else {
                        vm = make_uint4(left, v.x, v.y, v.z);
                        vp = make_uint4(v.y, v.z, v.w, right);
                    }
                    *reinterpret_cast<uint4*>(dst) = vm;                               // copy 0: tap kw = 0
                    *reinterpret_cast<uint4*>(dst + C3_BK * BP) = v;                   // copy 1: kw = 1
                    *reinterpret_cast<uint4*>(dst + 2 * C3_BK * BP) = vp;              // copy 2: kw = 2
                }
            }
        }
        __syncthreads();
        if (k0 + C3_BK < K) fetch(k0 + C3_BK);            // in flight during the MFMAs below
        // fragments of one tap row: 3 x (MI weight fragments + NI transpose-read pairs) per plane
        struct Frags {
            uint4 a[PL][3][MI];
            u32x2 lo[PL][3][NI], hi[PL][3][NI];
        };
        auto read_row = [&](int kh, Frags& f) {
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        f.a[pl][kw][mi] = *reinterpret_cast<const uint4*>(a_base + pl * C::A_EL + ((kh * 3 + kw) * BM + 32 * mi) * C3_APITCH);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const uint16_t* p = b_base + pl * C::B_EL + kw * C3_BK * BP + kh * D * W + 32 * ni;
                        f.lo[pl][kw][ni] = lds_tr16(p);
                        f.hi[pl][kw][ni] = lds_tr16(p + 4 * BP);
                    }
                }
        };
        auto settle = [&](Frags&) {};                          // (the reads are builtins now: the compiler waits for them itself)
        auto mfma_row = [&](const Frags& f) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const bf16x8 b = __builtin_bit_cast(bf16x8, make_uint4(f.lo[0][kw][ni].x, f.lo[0][kw][ni].y, f.hi[0][kw][ni].x, f.hi[0][kw][ni].y));
                    if (X3) {
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, make_uint4(f.lo[PL - 1][kw][ni].x, f.lo[PL - 1][kw][ni].y, f.hi[PL - 1][kw][ni].x,
                                                                                f.hi[PL - 1][kw][ni].y));
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
                            d[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[PL - 1][kw][mi]), b, d[mi][ni], 0, 0, 0);
                            d[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[0][kw][mi]), bl, d[mi][ni], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        d[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[0][kw][mi]), b, d[mi][ni], 0, 0, 0);
                }
        };
        if (X3) {
            // one wave per SIMD (both planes fill the LDS): nothing else covers the fragment reads' latency (36 b128 + 72 transpose reads
            // per tap row through one LDS pipe shared by the four waves), so the reads of tap row kh + 1 are issued before the 36 MFMAs
            // of row kh -- a second register set, free at this occupancy
            Frags f0, f1;
            read_row(0, f0);
            settle(f0);
            read_row(1, f1);
            mfma_row(f0);
            settle(f1);
            read_row(2, f0);
            mfma_row(f1);
            settle(f0);
            mfma_row(f0);
        } else {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                Frags f;
                read_row(kh, f);
                settle(f);
                mfma_row(f);
            }
        }
    }
    // C/D layout: column (pixel) = lane & 31, row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); lanes p / p + 1 trade
    // registers r / r + 1 so that each stores two adjacent pixels of one channel row (4-byte stores)
    const int jj = lane & 31;
    const size_t p_tile = (size_t)y0 * W + wv * 64;
    if (X3) {
        float* outf = reinterpret_cast<float*>(OUT_) + (size_t)n * M * HW;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * g;
                    const size_t p = p_tile + 32 * ni + jj;
                    if (m < M && p < HW) outf[(size_t)m * HW + p] = d[mi][ni][r];
                }
        return;
    }
    uint16_t* outn = reinterpret_cast<uint16_t*>(OUT_) + (size_t)n * M * HW;
    const bool odd = jj & 1;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float mine0 = d[mi][ni][r], mine1 = d[mi][ni][r + 1];
                const float give = odd ? mine0 : mine1;
                const float got = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give), 0xB1, 0xF, 0xF, true));
                const float lo2 = odd ? got : mine0, hi2 = odd ? mine1 : got;
                const int rsel = r + (odd ? 1 : 0);
                const int m = m0 + 32 * mi + (rsel & 3) + 8 * (rsel >> 2) + 4 * g;
                const size_t p = p_tile + 32 * ni + (jj & ~1);
                if (m < M && p < HW) *reinterpret_cast<uint32_t*>(outn + (size_t)m * HW + p) = aadg_f2bf_pk(lo2, hi2);
            }
}

template <int W, int D, int MI, bool X3>
int launch(const uint16_t* A9, const uint16_t* A9_lo, const void* IN, void* OUT, int N, int M, int K, int H, hipStream_t st) {
    using C = C3Cfg<W, D, MI, X3>;
    constexpr int BM = 32 * MI;
    const int tiles_m = (M + BM - 1) / BM, tiles_r = (H + C::ROWS - 1) / C::ROWS;
    const long long pts = (long long)N * tiles_r, groups = (pts + 7) / 8;
    const long long wgs = groups * 8 * tiles_m;
    if (wgs > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    if ((long long)9 * M * K > 0x7FFFFFFFLL || (long long)C3_BK * H * W > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;    // 32-bit staging offsets
    static bool attr_set = false;                            // per instantiation; idempotent
    if (!attr_set) {
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_nchw<W, D, MI, X3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)C::lds_bytes));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_conv3x3_nchw<W, D, MI, X3>), dim3((unsigned)wgs), dim3(256), C::lds_bytes, st, A9, A9_lo, IN, OUT, M, K, H, tiles_m,
                       tiles_r, (int)pts);
    AADG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int aadg_conv3x3_nchw_supported(int M, int K, int H, int W, int dilation) {
    return M > 0 && K > 0 && (K % 8) == 0 && H > 0 && (W == 32 || W == 64 || W == 128) && (dilation == 1 || dilation == 2) ? 1 : 0;
}

namespace {
template <bool X3>
int conv3x3_dispatch(const uint16_t* pa, const uint16_t* pl, const void* in, void* out, int N, int M, int K, int H, int W, int dilation,
                     hipStream_t st) {
    // Measured, not kept: a 128-channel tile (MI = 4) halves the weight-fragment reads per MFMA but needs 94 KB of LDS = one workgroup per CU:
    // 10-25 % slower.  Ablations (results wrong, timing only): a third of the weight-fragment reads, or a third of the transpose reads: no
    // change -- LDS read bandwidth is not the limit; skipping the per-step staging of the weight tile (stores): 15-18 % faster -- the serial
    // load -> store -> barrier section between two 36-MFMA bursts is.  A 32-channel tile (MI = 1, 52 KB, three workgroups per CU) to
    // interleave more of those sections: 10-40 % slower (half the MFMAs per staged IN chunk).  A 512-pixel tile (8 waves, one workgroup per
    // CU, weight tile amortised over twice the pixels, less halo): +-5 %.  Round 2: issuing the fragment reads of tap row kh + 1 before the
    // MFMAs of row kh (second register set, 120 -> 220-240 VGPRs): no change -- the two waves per SIMD already overlap LDS latency.
    // Dropping the per-step global loads (timing only): 6.5 -> 4.7 ms per step and direction (670 -> 980 TFLOP/s on 512 x 512 at 32 x 32):
    // every workgroup re-reads its 9 x 64 x K weight slab (590 KB against 390 KB of activations at K = 512) -- 4.3 TB/s out of L2 -- and
    // one K-step of MFMAs does not cover the load latency; loads two steps ahead need a second register set (> 256 VGPRs at two waves
    // per SIMD with this tile).  Next: a 128 x 256 tile on 8 waves (weights amortised over twice the MFMAs at the same LDS reads per MFMA).
    if (dilation == 1) {
        if (W == 32) return launch<32, 1, 2, X3>(pa, pl, in, out, N, M, K, H, st);
        if (W == 64) return launch<64, 1, 2, X3>(pa, pl, in, out, N, M, K, H, st);
        return launch<128, 1, 2, X3>(pa, pl, in, out, N, M, K, H, st);
    }
    if (W == 32) return launch<32, 2, 2, X3>(pa, pl, in, out, N, M, K, H, st);
    if (W == 64) return launch<64, 2, 2, X3>(pa, pl, in, out, N, M, K, H, st);
    return launch<128, 2, 2, X3>(pa, pl, in, out, N, M, K, H, st);
}
}  // namespace


/* out [N, M, H, W] = conv3x3(in [N, K, H, W]; a9 [9, M, K] tap-major), stride 1, padding = dilation; all bfloat16, float32
 * accumulation */
extern "C" int aadg_conv3x3_nchw_bf16(const void* a9, const void* in, void* out, int N, int M, int K, int H, int W, int dilation,
                                      void* stream) {
    if (a9 == nullptr || in == nullptr || out == nullptr || N <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)a9 | (uintptr_t)in | (uintptr_t)out) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_conv3x3_nchw_supported(M, K, H, W, dilation)) return AADG_E_UNSUPPORTED;
    return conv3x3_dispatch<false>((const uint16_t*)a9, nullptr, in, out, N, M, K, H, W, dilation, (hipStream_t)stream);
}

/* The same convolution at float32 precision ("f32x3"): in / out float32 NCHW; a9_hi / a9_lo = the bfloat16 (hi, lo) halves of the float32
 * tap-major weights [9, M, K] (aadg_weight_layouts_split_bf16); hi*hi + hi*lo + lo*hi on the matrix cores, float32 accumulation */
bool aadg_conv3x3_x3q_takes(int M, int K, int H, int W, int dilation);
int aadg_conv3x3_x3q(const uint16_t* a9_hi, const uint16_t* a9_lo, const float* in, float* out, int N, int M, int K, int H, int W, int dilation,
                     double* bn_sums, hipStream_t st, const float* pre_scale, const float* pre_shift);
extern "C" int aadg_conv3x3_nchw_f32x3(const void* a9_hi, const void* a9_lo, const float* in, float* out, int N, int M, int K, int H, int W,
                                       int dilation, void* stream) {
    return aadg_conv3x3_nchw_f32x3_stats(a9_hi, a9_lo, in, out, N, M, K, H, W, dilation, nullptr, stream);
}

/* the shapes whose BatchNorm statistics the convolution can take in its epilogue (the whole-tile kernel of conv3x3_x3.hip) */
extern "C" int aadg_conv3x3_f32x3_stats_supported(int M, int K, int H, int W, int dilation) {
    return aadg_conv3x3_nchw_supported(M, K, H, W, dilation) && aadg_conv3x3_x3q_takes(M, K, H, W, dilation) ? 1 : 0;
}

/* ... and, with bn_sums != NULL, the BatchNorm statistics of `out` from the epilogue (as aadg_conv1x1_nchw_f32x3_stats: float64 [2 M + 1] =
 * (sum, sum of squares) per output channel + the element count N H W, zeroed in here).  AADG_E_UNSUPPORTED for a shape outside
 * aadg_conv3x3_f32x3_stats_supported when bn_sums is given. */
extern "C" int aadg_conv3x3_nchw_f32x3_stats(const void* a9_hi, const void* a9_lo, const float* in, float* out, int N, int M, int K, int H,
                                             int W, int dilation, double* bn_sums, void* stream) {
    return aadg_conv3x3_nchw_f32x3_pre(a9_hi, a9_lo, in, out, N, M, K, H, W, dilation, nullptr, nullptr, bn_sums, stream);
}

/* ... and, with pre_scale / pre_shift [K] != NULL, `in` is the INPUT of the BatchNorm + ReLU in front of this convolution (see
 * aadg_conv1x1_nchw_f32x3_pre): max(in * pre_scale[k] + pre_shift[k], 0) on the way to LDS, zero padding as for the normalised tensor.
 * Needs bn_sums (the forward of a bottleneck's conv2), K <= 512 and a shape of aadg_conv3x3_f32x3_stats_supported. */
extern "C" int aadg_conv3x3_nchw_f32x3_pre(const void* a9_hi, const void* a9_lo, const float* in, float* out, int N, int M, int K, int H, int W,
                                           int dilation, const float* pre_scale, const float* pre_shift, double* bn_sums, void* stream) {
    if (a9_hi == nullptr || a9_lo == nullptr || in == nullptr || out == nullptr || N <= 0 || (pre_scale == nullptr) != (pre_shift == nullptr))
        return AADG_E_BADARG;
    if ((((uintptr_t)a9_hi | (uintptr_t)a9_lo | (uintptr_t)in | (uintptr_t)out) & 15u) != 0 || (((uintptr_t)bn_sums) & 7u) != 0) return AADG_E_BADARG;
    if (!aadg_conv3x3_nchw_supported(M, K, H, W, dilation)) return AADG_E_UNSUPPORTED;
#ifndef AADG_NO_X3Q
    // whole-tile shapes (every such layer of the backbone): the [pixel][k] kernel of conv3x3_x3.hip, two workgroups per CU
    if (aadg_conv3x3_x3q_takes(M, K, H, W, dilation))
        return aadg_conv3x3_x3q((const uint16_t*)a9_hi, (const uint16_t*)a9_lo, in, out, N, M, K, H, W, dilation, bn_sums, (hipStream_t)stream,
                                pre_scale, pre_shift);
#endif
    if (pre_scale != nullptr) return AADG_E_UNSUPPORTED;
    if (bn_sums != nullptr) return AADG_E_UNSUPPORTED;
    return conv3x3_dispatch<true>((const uint16_t*)a9_hi, (const uint16_t*)a9_lo, in, out, N, M, K, H, W, dilation, (hipStream_t)stream);
}

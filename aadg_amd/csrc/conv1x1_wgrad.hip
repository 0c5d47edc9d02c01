// Weight gradient of a 1x1 / stride-1 convolution over NCHW bfloat16 activations on the matrix cores:
//
//     dW[o][c] = sum_n sum_k dY[n][o][k] * X[n][c][k]          (k = pixel index, HW contiguous in both operands)
//
// The library path for this op transposes both activation tensors to NHWC first (two extra passes over tensors of
// up to 1.2 GB each at N = 144 x 512 x 512) and then runs an implicit-GEMM kernel.  In NCHW both operands are already
// K-contiguous, which is exactly the A / B fragment layout of v_mfma_f32_32x32x16_bf16 (lane = row, 8 consecutive k):
// no transpose, no im2col.  One workgroup = one (BM x BN) tile of dW and one slice of the N*HW reduction:
//
//   global --16-byte loads (128 contiguous bytes per row)--> registers --> LDS (double buffered, 144-byte row pitch)
//   LDS --ds_read_b128 fragments--> 4 x 4 MFMA 32x32x16 per wave and K-step of 64 --> float32 atomics into dW
//
// float32 accumulation; the split-K partials are combined with hardware float atomics (order-dependent in the last bit).
#include <hip/hip_bf16.h>
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int WG_BK = 64;                 // K elements per step (128 bytes per row)
constexpr int WG_PITCH = WG_BK + 8;       // LDS row pitch in elements (144 bytes: conflict-free 16-byte fragment reads)

// WR x WC waves, each an (MI x NI) block of 32 x 32 MFMA tiles of dW: workgroup tile BM x BN = (32 MI WR) x (32 NI WC).
// Per 16-wide K sub-step a wave reads MI + NI fragments from LDS for MI * NI MFMAs, and a workgroup pulls (BM + BN) * 128
// bytes per 64-wide K step through L2 for BM * BN * 128 flops: the 256 x 256 tile (8 waves of 128 x 64) halves both ratios
// relative to 128 x 128 (4 waves of 64 x 64), which is what the channel-rich late layers (>= 256 x 256 weights) are bound by.
//
// X3 = true ("f32x3"): dY / X are float32.  A K-step is still 128 bytes per row = 32 pixels; every loaded float4 becomes one 16-byte LDS
// chunk [hi0..3][lo0..3] (aadg_split4), so the 8 K-values of a fragment are the hi halves of two neighbouring chunks and their lo
// halves come with the same two reads; products hi*hi + hi*lo + lo*hi, float32 accumulation.
// EXACT (X3 only): Co / Ci whole tiles -- no null-row branches around the loads (one basic block per K-step).
// PRE (X3 + EXACT only): X is the INPUT of the BatchNorm + ReLU whose output the convolution consumed (aadg_conv1x1_nchw_f32x3_pre): the
// rows of X become max(fma(x, pre_scale[c], pre_shift[c]), 0) while they are staged -- the same expression, the same values.
template <int WR, int WC, int MI, int NI, bool X3, bool EXACT = false, bool PRE = false>
__global__ __launch_bounds__(64 * WR * WC) void k_wgrad1x1(const void* __restrict__ dY_, const void* __restrict__ X_,
                                                           float* __restrict__ acc, int Co, int Ci, int HW, int tiles, int tiles_n,
                                                           int steps_total, int steps_per_block, const float* __restrict__ pre_scale = nullptr,
                                                           const float* __restrict__ pre_shift = nullptr) {
    static_assert(!PRE || X3, "the load transform exists for float32 tensors (whole X tiles: Ci a multiple of the tile's columns)");
    typedef typename std::conditional<X3, float, uint16_t>::type elem_t;
    const elem_t* dY = reinterpret_cast<const elem_t*>(dY_);
    const elem_t* X = reinterpret_cast<const elem_t*>(X_);
    constexpr int BKE = X3 ? 32 : WG_BK;                       // K elements (pixels) per step: 128 bytes of a row either way
    constexpr int NT = 64 * WR * WC;
    constexpr int BM = 32 * MI * WR, BN = 32 * NI * WC, R = BM + BN, LPT = R * 8 / NT, LPT_A = BM * 8 / NT;
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "a staging slot is entirely dY or entirely X");
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];      // [2][R][WG_PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv / WC, wc = wv - wr * WC;
    // 1-D grid, XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the id is
    // decoded as (K-slice group, tile, xcd): all tiles that stream the SAME K-slice of dY / X run on the same XCD and share
    // those tiles of the operands through its L2 instead of re-reading them from HBM once per tile.
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tile = q % tiles, slice = (q / tiles) * 8 + xcd;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int s0 = slice * steps_per_block, s1 = min(steps_total, s0 + steps_per_block);
    if (s0 >= s1) return;
    const int spi = HW / BKE;             // steps per image

    // this thread's LPT (row, 16-byte chunk) slots of the staged tile; rows beyond Co / Ci read as zero.  Round 6: a row that does not
    // exist is LOADED from the last one that does and zeroed with a select -- a guarded load is a basic block of its own (the loads of a
    // step could not be issued together), an unconditional one is not
    const elem_t* src[LPT];
    uint32_t live = 0;                                     // bit i: slot i is a row of the tensors
    static_assert(LPT <= 32, "one bit per slot");
    const int row0 = tid >> 3, c8 = (tid & 7) * 8, cg = (tid & 7) * (BKE / 8);     // chunk offset in LDS / in the global row (elements)
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int row = row0 + (NT / 8) * i;
        if (i < LPT_A) {
            const int m = m0 + row;
            src[i] = dY + (size_t)min(m, Co - 1) * HW + cg;
            if (m < Co) live |= 1u << i;
        } else {
            const int n = n0 + row - BM;
            src[i] = X + (size_t)min(n, Ci - 1) * HW + cg;
            if (n < Ci) live |= 1u << i;
        }
    }
    const size_t stride_a = (size_t)Co * HW, stride_b = (size_t)Ci * HW;
    float xs[PRE ? LPT - LPT_A : 1], xh[PRE ? LPT - LPT_A : 1];          // PRE: scale / shift of the thread's X rows (constant over the steps)
    if (PRE) {
#pragma unroll
        for (int i = LPT_A; i < LPT; ++i) {
            const int n = n0 + row0 + (NT / 8) * i - BM;
            xs[i - LPT_A] = pre_scale[n]; xh[i - LPT_A] = pre_shift[n];
        }
    }
    uint4 stage[LPT];
    auto fetch = [&](int step) {
        const int n = step / spi, kk = (step - n * spi) * BKE;
#pragma unroll
        for (int i = 0; i < LPT; ++i)
        {
            const uint4 v = *reinterpret_cast<const uint4*>(src[i] + (size_t)n * (i < LPT_A ? stride_a : stride_b) + kk);
            stage[i] = (EXACT || ((live >> i) & 1u)) ? v : make_uint4(0, 0, 0, 0);
        }
    };

    f32x16 d[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) d[mi][ni][r] = 0.0f;

    const int a_row = wr * 32 * MI + (lane & 31), b_row = BM + wc * 32 * NI + (lane & 31), koff = (lane >> 5) * 8;
    const int st_off = row0 * WG_PITCH + c8;
    fetch(s0);
    int buf = 0;
    for (int step = s0; step < s1; ++step) {
        uint16_t* L = lds + (size_t)buf * R * WG_PITCH;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            uint4 v = stage[i];
            if (X3) {
                uint2 hi, lo;
                float4 f = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                if (PRE && i >= LPT_A) {
                    const float sc = xs[i - LPT_A], sh = xh[i - LPT_A];
                    f.x = fmaxf(fmaf(f.x, sc, sh), 0.0f); f.y = fmaxf(fmaf(f.y, sc, sh), 0.0f);
                    f.z = fmaxf(fmaf(f.z, sc, sh), 0.0f); f.w = fmaxf(fmaf(f.w, sc, sh), 0.0f);
                }
                aadg_split4(f, hi, lo);
                v = make_uint4(hi.x, hi.y, lo.x, lo.y);
            }
            *reinterpret_cast<uint4*>(L + st_off + (NT / 8) * i * WG_PITCH) = v;
        }
        __syncthreads();
        if (step + 1 < s1) fetch(step + 1);              // in flight during the MFMAs below
        if (X3) {
#pragma unroll
            for (int ks = 0; ks < BKE / 16; ++ks) {
                bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const uint16_t* p = L + (a_row + 32 * mi) * WG_PITCH + ks * 32 + 2 * koff;
                    const uint4 q0 = *reinterpret_cast<const uint4*>(p), q1 = *reinterpret_cast<const uint4*>(p + 8);
                    ah[mi] = __builtin_bit_cast(bf16x8, make_uint4(q0.x, q0.y, q1.x, q1.y));
                    al[mi] = __builtin_bit_cast(bf16x8, make_uint4(q0.z, q0.w, q1.z, q1.w));
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const uint16_t* p = L + (b_row + 32 * ni) * WG_PITCH + ks * 32 + 2 * koff;
                    const uint4 q0 = *reinterpret_cast<const uint4*>(p), q1 = *reinterpret_cast<const uint4*>(p + 8);
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
        } else
#pragma unroll
        for (int ks = 0; ks < WG_BK / 16; ++ks) {
            bf16x8 a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                a[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(L + (a_row + 32 * mi) * WG_PITCH + ks * 16 + koff));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                b[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(L + (b_row + 32 * ni) * WG_PITCH + ks * 16 + koff));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) d[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], d[mi][ni], 0, 0, 0);
        }
        buf ^= 1;
    }
    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wc * 32 * NI + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 32 * MI + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (EXACT || (m < Co && n < Ci)) unsafeAtomicAdd(acc + (size_t)m * Ci + n, d[mi][ni][r]);
            }
        }
}

// target_wgs: workgroups aimed at (every one of them ends with BM * BN float atomics: the big tile runs one per CU)
template <int WR, int WC, int MI, int NI, bool X3>
int launch(const void* dY, const void* X, float* acc, int N, int Co, int Ci, int HW, int target_wgs, hipStream_t st,
           const float* pre_scale = nullptr, const float* pre_shift = nullptr) {
    constexpr int BKE = X3 ? 32 : WG_BK;
    constexpr int BM = 32 * MI * WR, BN = 32 * NI * WC, R = BM + BN, NT = 64 * WR * WC;
    const int tiles_m = (Co + BM - 1) / BM, tiles_n = (Ci + BN - 1) / BN, tiles = tiles_m * tiles_n;
    const int steps_total = N * (HW / BKE);
    // the big tile runs ONE workgroup per CU (target 256): rounded DOWN, so that tiles x split never exceeds the chip (1280 -> 256, 5
    // tiles: 51 slices = 255 workgroups instead of 52 = 260 with four of them in a second round; no measurable change -- that layer's
    // time is its 255 x 65536 float atomics onto 327 k addresses)
    int split = target_wgs <= 256 ? (target_wgs / tiles > 0 ? target_wgs / tiles : 1) : (target_wgs + tiles - 1) / tiles;
    // K-steps per workgroup: each workgroup ends with BM x BN float atomics (the cost of several K-steps), so 32 steps when the
    // reduction is long enough to still fill the chip (N = 144: 4.66 -> 4.46 ms over the 16 backbone shapes against 8), fewer --
    // down to 8 -- when that would leave CUs idle (the 32 x 32 layers at 18 images per rank have 288 steps in all)
    int min_steps = 32;
    while (min_steps > 8 && (long long)(steps_total / min_steps) * tiles < 256) min_steps /= 2;
    if (split > steps_total / min_steps) split = steps_total / min_steps;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    const int steps_per_block = (steps_total + split - 1) / split;
    split = (steps_total + steps_per_block - 1) / steps_per_block;
    const size_t lds = (size_t)2 * R * WG_PITCH * sizeof(uint16_t);
    static bool attr_set = false;                            // per instantiation; idempotent
    if (!attr_set) {
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad1x1<WR, WC, MI, NI, X3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad1x1<WR, WC, MI, NI, X3, X3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad1x1<WR, WC, MI, NI, X3, X3, X3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad1x1<WR, WC, MI, NI, X3, false, X3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    if (pre_scale != nullptr && !(X3 && (Ci % BN) == 0)) return AADG_E_UNSUPPORTED;       // every staged X row must exist
    AADG_HIP_TRY(hipMemsetAsync(acc, 0, (size_t)Co * Ci * sizeof(float), st));
    const int slice_groups = (split + 7) / 8;                 // slices are padded to a multiple of 8 (empty ones exit at once)
    if (pre_scale != nullptr && (Co % BM) == 0)
        hipLaunchKernelGGL((k_wgrad1x1<WR, WC, MI, NI, X3, X3, X3>), dim3((unsigned)(slice_groups * tiles * 8)), dim3(NT), lds, st, dY, X, acc,
                           Co, Ci, HW, tiles, tiles_n, steps_total, steps_per_block, pre_scale, pre_shift);
    else if (pre_scale != nullptr)                          // Co is no whole number of tiles (the 8-row classifier): null dY rows, whole X tiles
        hipLaunchKernelGGL((k_wgrad1x1<WR, WC, MI, NI, X3, false, X3>), dim3((unsigned)(slice_groups * tiles * 8)), dim3(NT), lds, st, dY, X, acc,
                           Co, Ci, HW, tiles, tiles_n, steps_total, steps_per_block, pre_scale, pre_shift);
    else if (X3 && (Co % BM) == 0 && (Ci % BN) == 0)
        hipLaunchKernelGGL((k_wgrad1x1<WR, WC, MI, NI, X3, X3>), dim3((unsigned)(slice_groups * tiles * 8)), dim3(NT), lds, st, dY, X, acc, Co,
                           Ci, HW, tiles, tiles_n, steps_total, steps_per_block);
    else
    hipLaunchKernelGGL((k_wgrad1x1<WR, WC, MI, NI, X3>), dim3((unsigned)(slice_groups * tiles * 8)), dim3(NT), lds, st, dY, X, acc, Co, Ci,
                       HW, tiles, tiles_n, steps_total, steps_per_block);
    AADG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int aadg_conv1x1_wgrad_supported(int Co, int Ci, int HW) {
    return Co > 0 && Ci > 0 && HW >= WG_BK && (HW % WG_BK) == 0 ? 1 : 0;
}

namespace {
template <bool X3>
int wgrad1x1_dispatch(const void* a, const void* b, float* dweight, int N, int Co, int Ci, int HW, hipStream_t st,
                      const float* pre_scale = nullptr, const float* pre_shift = nullptr) {
    constexpr int BKE = X3 ? 32 : WG_BK;
    if (Co <= 64) return launch<1, 4, 2, 2, X3>(a, b, dweight, N, Co, Ci, HW, 1024, st, pre_scale, pre_shift);
    if (Ci <= 64) return launch<4, 1, 2, 2, X3>(a, b, dweight, N, Co, Ci, HW, 1024, st, pre_scale, pre_shift);
    // the 256 x 256 tile needs a reduction long enough for ~one workgroup per CU at >= 32 steps each
    const long long big_wgs = (long long)(Co / 256) * (Ci / 256) * ((long long)N * (HW / BKE) / 32);
    if (Co >= 256 && Ci >= 256 && (Co % 256) == 0 && (Ci % 256) == 0 && (long long)Co * Ci >= 512 * 512 && big_wgs >= 192)
        return launch<2, 4, 4, 2, X3>(a, b, dweight, N, Co, Ci, HW, 256, st, pre_scale, pre_shift);
    return launch<2, 2, 2, 2, X3>(a, b, dweight, N, Co, Ci, HW, 1024, st, pre_scale, pre_shift);
}
}  // namespace

extern "C" int aadg_conv1x1_wgrad_bf16(const void* dy, const void* x, float* dweight, int N, int Co, int Ci, int HW, void* stream) {
    if (dy == nullptr || x == nullptr || dweight == nullptr || N <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)dy | (uintptr_t)x) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_conv1x1_wgrad_supported(Co, Ci, HW) || (long long)N * (HW / WG_BK) > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    return wgrad1x1_dispatch<false>(dy, x, dweight, N, Co, Ci, HW, (hipStream_t)stream);
}

/* the shapes whose weight gradient takes pre_scale / pre_shift (aadg_conv1x1_wgrad_f32x3_pre): Ci a whole number of the X tiles of the
 * configuration the dispatch picks for (N, Co, Ci, HW) */
extern "C" int aadg_conv1x1_wgrad_f32x3_pre_supported(int N, int Co, int Ci, int HW) {
    if (N <= 0 || Co <= 0 || Ci <= 0 || HW < 32 || (HW % 32) != 0) return 0;
    if (Co <= 64) return (Ci % 256) == 0;
    if (Ci <= 64) return (Ci % 64) == 0;
    const long long big_wgs = (long long)(Co / 256) * (Ci / 256) * ((long long)N * (HW / 32) / 32);
    if (Co >= 256 && Ci >= 256 && (Co % 256) == 0 && (Ci % 256) == 0 && (long long)Co * Ci >= 512 * 512 && big_wgs >= 192) return 1;
    return (Ci % 128) == 0;
}

/* dW [Co, Ci] float32 from float32 NCHW dy / x at float32 precision ("f32x3": three bfloat16 matrix-core products per pair of
 * (hi, lo)-split operands, float32 accumulation).  HW % 32 == 0. */
extern "C" int aadg_conv1x1_wgrad_f32x3(const float* dy, const float* x, float* dweight, int N, int Co, int Ci, int HW, void* stream) {
    return aadg_conv1x1_wgrad_f32x3_pre(dy, x, dweight, N, Co, Ci, HW, nullptr, nullptr, stream);
}

/* ... and, with pre_scale / pre_shift [Ci] != NULL (ABI 10), x is the INPUT of the BatchNorm + ReLU in front of the convolution (see
 * aadg_conv1x1_nchw_f32x3_pre): its rows become max(x * pre_scale[c] + pre_shift[c], 0) while they are staged.  Whole-tile shapes only
 * (Co, Ci multiples of the tile the dispatch picks: every bottleneck of the backbone) -- AADG_E_UNSUPPORTED otherwise. */
extern "C" int aadg_conv1x1_wgrad_f32x3_pre(const float* dy, const float* x, float* dweight, int N, int Co, int Ci, int HW,
                                            const float* pre_scale, const float* pre_shift, void* stream) {
    if (dy == nullptr || x == nullptr || dweight == nullptr || N <= 0 || (pre_scale == nullptr) != (pre_shift == nullptr)) return AADG_E_BADARG;
    if ((((uintptr_t)dy | (uintptr_t)x) & 15u) != 0) return AADG_E_BADARG;
    if (Co <= 0 || Ci <= 0 || HW < 32 || (HW % 32) != 0 || (long long)N * (HW / 32) > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    return wgrad1x1_dispatch<true>(dy, x, dweight, N, Co, Ci, HW, (hipStream_t)stream, pre_scale, pre_shift);
}

// 1x1 / stride-1 convolution over NCHW bfloat16 activations as a per-image GEMM on the matrix cores:
//
//     OUT[n][m][p] = sum_k A[m][k] * IN[n][k][p]        (m: output channel, k: input channel, p: pixel, contiguous)
//
// forward: A = weight [Co][Ci];  input gradient: A = weight^T [Ci][Co] (a contiguous transposed copy made by the caller), IN = dY.
// A is K-contiguous: plain 16-byte LDS fragments.  IN is K-STRIDED (NCHW: pixels are contiguous, channels are H*W apart) --
// the layout a matrix-core B operand cannot take from row-major memory with ordinary reads.  gfx950's LDS transpose read
// solves it without touching the data: rows of IN go to LDS exactly as they lie in memory ([k][p], 16-byte copies), and
// ds_read_b64_tr_b16 hands every lane the 4 consecutive k of ITS pixel column (16 lanes read one 4 x 16 block; lane i passes
// the address of row i / 4, columns 4 (i % 4) .. + 3 and receives column i) -- two of them make the 8-k B fragment of
// v_mfma_f32_32x32x16_bf16.  No NHWC copy of the activations, no im2col, no transposing pass.
//
// Workgroup = 4 waves, tile BM x 256 pixels of one image, K-step 64 through a single LDS buffer with the next step's global
// loads in flight (registers) during the MFMAs.  Epilogue: neighbouring lanes trade one register (DPP) and store pixel pairs.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int CF_BK = 64, CF_BP = 256;
constexpr int CF_BPITCH = CF_BP + 16;           // B rows: 544 bytes = 32 (mod 256): the 4 rows of a transpose read hit distinct banks

// the gfx950 LDS transpose read as a compiler builtin (round 6; it was inline assembly with a hand-placed s_waitcnt): the scheduler moves
// the reads of the next K sub-step between the MFMAs of the current one and counts its own waits (2-3 % on the compute-bound layers)
typedef short cf_v4i16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 lds_read_tr16(const uint16_t* p) {
    const cf_v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cf_v4i16 __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(u32x2, v);
}

// WR x WC waves; wave tile (32 MI) x (32 NI); BM = 32 MI WR, BP = 32 NI WC = 256.  (f32x3: 1 x 4 waves of 128 x 64 -- few transpose reads per MFMA.)
//
// X3 = true ("f32x3"): IN / OUT are float32 and A comes pre-split as two bfloat16 planes (A = hi, A_lo = lo; csrc/weight_layouts.hip).
// A loaded float4 of IN (4 pixels of one channel) is split into (hi, lo) bfloat16 halves while it goes to LDS -- two planes of the same
// [k][pixel] layout, read with the same transpose reads -- and every fragment pair is multiplied as hi*hi + hi*lo + lo*hi (float32
// accumulation): float32-grade products at a third of the bfloat16 rate.  K-step 32 (both planes of both operands: 55 KB of LDS).
// EXACT (X3 only): M, K and HW are whole tiles / K-steps -- no bounds checks, i.e. no exec-mask branches around the twelve loads of a
// K-step (each guarded load is a basic block of its own: the loads of a step cannot be issued together across them).
// PRE (X3 + EXACT only, K <= CF_PRE_K): IN is the INPUT of a BatchNorm + ReLU whose output this convolution consumes -- the loaded
// values become max(fma(x, pre_scale[k], pre_shift[k]), 0) on their way to LDS (the BatchNorm kernels' own expression, so the mask its
// backward re-derives from x is the one applied here); the normalised tensor is never written.
constexpr int CF_PRE_K = 512;
template <int WR, int WC, int MI, int NI, bool X3, bool EXACT = false, bool PRE = false>
__global__ __launch_bounds__(256, 2) void k_conv1x1_nchw(const uint16_t* __restrict__ A, const uint16_t* __restrict__ A_lo,
                                                         const void* __restrict__ IN_, void* __restrict__ OUT_, int M, int K, int HW,
                                                         int tiles_p, int tiles_m, double* __restrict__ stats,
                                                         const float* __restrict__ pre_scale = nullptr,
                                                         const float* __restrict__ pre_shift = nullptr) {
    static_assert(!PRE || X3, "the load transform exists for float32 tensors (whole IN tiles: K % 32 == 0, HW % 256 == 0)");
    constexpr int BP = 32 * NI * WC;                                       // pixels per tile: 256 (128 for the 256-row f32x3 tile)
    static_assert(WR * WC == 4 && (BP == CF_BP || (X3 && BP == 128)), "4 waves, 256 (f32x3: or 128) pixels");
    constexpr int QPR = BP / 4;                                            // float4 chunks per staged IN row (f32x3)
    constexpr int BM = 32 * MI * WR;
    constexpr int BK = X3 ? 32 : CF_BK, APITCH = BK + 8, PL = X3 ? 2 : 1;
    constexpr int AC = BK / 8;                                             // 16-byte chunks per A row and plane
    constexpr int LA = BM * AC * PL / 256;                                 // A chunks per thread and K-step
    constexpr int LB = X3 ? BK * (BP / 4) / 256 : BK * (BP / 8) / 256;         // IN chunks (16 bytes: 4 float32 / 8 bfloat16 pixels)
    __shared__ __attribute__((aligned(16))) uint16_t As[PL * BM * APITCH];
    // B pitch.  bfloat16 kernel: 544 bytes.  X3: 576 bytes = 16 banks per row step -- a transpose read serves 32 lanes per LDS cycle (4 rows x
    // two 16-column halves): with 8 banks per row step the second half of row r collides with the first half of row r + 1
    // (SQ_LDS_BANK_CONFLICT was 36 % of SQ_LDS_IDX_ACTIVE); with 16 the 32 lanes cover the 64 banks exactly
    constexpr int BPITCH = X3 ? BP + 32 : CF_BPITCH;
    __shared__ __attribute__((aligned(16))) uint16_t Bs[PL * BK * BPITCH];
    __shared__ float Ps[PRE ? 2 * CF_PRE_K : 2];                          // PRE: scale [K] | shift [K] (visible behind the loop's first barrier)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv / WC, wc = wv - wr * WC;
    if (PRE)
        for (int i = tid; i < K; i += 256) { Ps[i] = pre_scale[i]; Ps[CF_PRE_K + i] = pre_shift[i]; }
    // grid: pixel tile fastest, then output-channel tile, then image: the workgroups that share an IN tile are neighbours
    const int tp = blockIdx.x % tiles_p, t2 = blockIdx.x / tiles_p;
    const int tm = t2 % tiles_m, n = t2 / tiles_m;
    const int m0 = tm * BM, p0 = tp * BP;
    const uint16_t* inn = reinterpret_cast<const uint16_t*>(IN_) + (size_t)n * K * HW * (X3 ? 2 : 1);
    const float* innf = reinterpret_cast<const float*>(IN_) + (size_t)n * K * HW;

    uint4 ra[LA], rb[LB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int id = tid + 256 * i, pl = id / (BM * AC), r = id - pl * (BM * AC), row = r / AC, c = (r - row * AC) * 8;
            const int m = m0 + row, k = k0 + c;
            // (not whole tiles: the element that does not exist is LOADED from the nearest one that does and zeroed with a select --
            // round 6; a guarded load is a basic block of its own and the loads of a step cannot be issued together across them)
            const uint4 va = *reinterpret_cast<const uint4*>((pl ? A_lo : A) + (size_t)(EXACT ? m : min(m, M - 1)) * K + (EXACT ? k : min(k, K - 8)));
            ra[i] = (EXACT || (m < M && k < K)) ? va : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int id = tid + 256 * i;
            if (X3) {
                const int row = id / QPR, c = (id % QPR) * 4, k = k0 + row, p = p0 + c;
                const uint4 vb = *reinterpret_cast<const uint4*>(innf + (size_t)(EXACT ? k : min(k, K - 1)) * HW + (EXACT ? p : min(p, HW - 4)));
                rb[i] = (EXACT || (k < K && p < HW)) ? vb : make_uint4(0, 0, 0, 0);
            } else {
                const int row = id >> 5, c = (id & 31) * 8, k = k0 + row, p = p0 + c;
                const uint4 vb = *reinterpret_cast<const uint4*>(inn + (size_t)min(k, K - 1) * HW + min(p, HW - 8));
                rb[i] = (k < K && p < HW) ? vb : make_uint4(0, 0, 0, 0);
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
    const uint16_t* a_base = As + (wr * 32 * MI + (lane & 31)) * APITCH + 8 * g;
    // transpose-read address of this lane: row (8 g + i16 / 4) of the K-sub-step, columns 16 gi + 4 (i16 % 4) of the N tile
    const uint16_t* b_base = Bs + (8 * g + (i16 >> 2)) * BPITCH + wc * 32 * NI + 16 * gi + 4 * (i16 & 3);

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();                                  // the previous step's fragment reads are done
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int id = tid + 256 * i, pl = id / (BM * AC), r = id - pl * (BM * AC), row = r / AC, c = (r - row * AC) * 8;
            *reinterpret_cast<uint4*>(As + pl * BM * APITCH + row * APITCH + c) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int id = tid + 256 * i;
            if (X3) {
                uint2 hi, lo;
                float4 f = make_float4(__uint_as_float(rb[i].x), __uint_as_float(rb[i].y), __uint_as_float(rb[i].z), __uint_as_float(rb[i].w));
                if (PRE) {
                    const float sc = Ps[k0 + id / QPR], sh = Ps[CF_PRE_K + k0 + id / QPR];
                    f.x = fmaxf(fmaf(f.x, sc, sh), 0.0f); f.y = fmaxf(fmaf(f.y, sc, sh), 0.0f);
                    f.z = fmaxf(fmaf(f.z, sc, sh), 0.0f); f.w = fmaxf(fmaf(f.w, sc, sh), 0.0f);
                }
                aadg_split4(f, hi, lo);
                uint16_t* dst = Bs + (id / QPR) * BPITCH + (id % QPR) * 4;
                *reinterpret_cast<uint2*>(dst) = hi;
                *reinterpret_cast<uint2*>(dst + BK * BPITCH) = lo;
            } else {
                *reinterpret_cast<uint4*>(Bs + (id >> 5) * BPITCH + (id & 31) * 8) = rb[i];
            }
        }
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK);                  // in flight during the MFMAs below
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[PL][MI], b[PL][NI];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    a[pl][mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a_base + pl * BM * APITCH + 32 * mi * APITCH + 16 * ks));
            u32x2 lo[PL][NI], hi[PL][NI];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    lo[pl][ni] = lds_read_tr16(b_base + pl * BK * BPITCH + (16 * ks) * BPITCH + 32 * ni);
                    hi[pl][ni] = lds_read_tr16(b_base + pl * BK * BPITCH + (16 * ks + 4) * BPITCH + 32 * ni);
                }
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    b[pl][ni] = __builtin_bit_cast(bf16x8, make_uint4(lo[pl][ni].x, lo[pl][ni].y, hi[pl][ni].x, hi[pl][ni].y));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (X3) {                              // the small cross terms first, the hi * hi product last
                        d[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1][mi], b[0][ni], d[mi][ni], 0, 0, 0);
                        d[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[PL - 1][ni], d[mi][ni], 0, 0, 0);
                    }
                    d[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[0][ni], d[mi][ni], 0, 0, 0);
                }
        }
    }
    // C/D layout: column (pixel) = lane & 31, row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int jj = lane & 31;
    if (X3) {
        // float32 out: the 32 lanes of a row store 128 contiguous bytes
        float* outf = reinterpret_cast<float*>(OUT_) + (size_t)n * M * HW;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wr * 32 * MI + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * g;
                    const int p = p0 + wc * 32 * NI + 32 * ni + jj;
                    if (EXACT || (m < M && p < HW)) outf[(size_t)m * HW + p] = d[mi][ni][r];
                }
        if (stats != nullptr) {
            // BatchNorm statistics of the output in the epilogue (training: the layer behind this convolution is a BatchNorm): per
            // channel sum and sum of squares over the tile's pixels -> double atomics into stats[2 m], stats[2 m + 1] (the float64
            // totals aadg_bn_sync_forward(phase 2, ...) normalises with), instead of a pass of k_bn_reduce_fwd over the whole float32
            // tensor.  Pixels beyond HW were computed from zero-filled input: they contribute 0.  A lane holds, for each of its
            // 16 MI rows, NI pixels: summed per lane, then over the 32 lanes of the row with a halving butterfly -- at step `msk` a
            // lane keeps one half of its items and hands the other half to lane ^ msk (ds_swizzle: the LDS crossbar, no memory), so
            // 32 + 16 + 8 + 4 + 2 exchanges leave every lane with the two totals of ONE row (row index = the lane's bits reversed).
            constexpr int NR = 16 * MI, NG = NR / 32;       // rows per lane half; groups of 32 rows (one butterfly each)
            static_assert(NR % 32 == 0, "the butterfly below pairs 32 rows with the 32 lanes of a half");
            float v[2 * NR];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = 0.f, q = 0.f;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) { s += d[mi][ni][r]; q = fmaf(d[mi][ni][r], d[mi][ni][r], q); }
                    v[2 * (16 * mi + r)] = s;
                    v[2 * (16 * mi + r) + 1] = q;
                }
#define AADG_BFLY(B, MSK, HALF, PAT)                                                                    \
            {                                                                                           \
                const bool up = (lane & (MSK)) != 0;                                                    \
                _Pragma("unroll") for (int i = 0; i < (HALF); ++i) {                                    \
                    const float keep = up ? v[(B) + i + (HALF)] : v[(B) + i], send = up ? v[(B) + i] : v[(B) + i + (HALF)]; \
                    v[(B) + i] = keep + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(send), (PAT))); \
                }                                                                                       \
            }
            // ds_swizzle bit mode: and_mask 0x1F | xor_mask << 10 (groups of 32 lanes); one butterfly per group of 32 rows (mi pairs)
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                AADG_BFLY(64 * q, 16, 32, 0x1F | (16 << 10))
                AADG_BFLY(64 * q, 8, 16, 0x1F | (8 << 10))
                AADG_BFLY(64 * q, 4, 8, 0x1F | (4 << 10))
                AADG_BFLY(64 * q, 2, 4, 0x1F | (2 << 10))
                AADG_BFLY(64 * q, 1, 2, 0x1F | (1 << 10))
            }
#undef AADG_BFLY
            // item kept through the steps: bit 4 of the lane picked the upper 32 items, bit 3 the upper 16 of those, ...
            const int item = ((lane >> 4) & 1) * 32 + ((lane >> 3) & 1) * 16 + ((lane >> 2) & 1) * 8 + ((lane >> 1) & 1) * 4 + (lane & 1) * 2;
            const int rr = item >> 1, mi = rr >> 4, r = rr & 15;
            // the WC waves that share a row are combined in LDS first (the operand buffers are free once every wave has left the
            // K loop): one float64 atomic per channel, statistic and WORKGROUP -- with one per wave the 36 864 atomics per address of
            // a 64-channel layer at 128 x 128 doubled that kernel's duration
            float* red = reinterpret_cast<float*>(As);
            __syncthreads();
            for (int i = tid; i < 2 * BM; i += 256) red[i] = 0.0f;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const int row = wr * 32 * MI + 32 * (mi + 2 * q) + (r & 3) + 8 * (r >> 2) + 4 * g;       // row of the workgroup's tile
                __hip_atomic_fetch_add(red + 2 * row, v[64 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(red + 2 * row + 1, v[64 * q + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();
            for (int i = tid; i < 2 * BM; i += 256)
                if (m0 + (i >> 1) < M) unsafeAtomicAdd(stats + 2 * (size_t)m0 + i, (double)red[i]);
        }
        return;
    }
    // bfloat16 out: lanes p / p + 1 trade registers r / r + 1 so that each stores two adjacent pixels of one channel row (4-byte stores)
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
                const float lo = odd ? got : mine0, hi = odd ? mine1 : got;
                const int rsel = r + (odd ? 1 : 0);
                const int m = m0 + wr * 32 * MI + 32 * mi + (rsel & 3) + 8 * (rsel >> 2) + 4 * g;
                const int p = p0 + wc * 32 * NI + 32 * ni + (jj & ~1);
                if (m < M && p < HW) *reinterpret_cast<uint32_t*>(outn + (size_t)m * HW + p) = aadg_f2bf_pk(lo, hi);
            }
}

template <int WR, int WC, int MI, int NI, bool X3>
int launch(const uint16_t* A, const uint16_t* A_lo, const void* IN, void* OUT, int N, int M, int K, int HW, hipStream_t st,
           double* stats = nullptr, const float* pre_scale = nullptr, const float* pre_shift = nullptr) {
    constexpr int BM = 32 * MI * WR, BP = 32 * NI * WC;
    if (X3 && (M % BM) == 0 && (K % 32) == 0 && (HW % BP) == 0) {          // whole tiles: the branch-free instantiation
        const long long wgs_e = (long long)N * (HW / BP) * (M / BM);
        if (wgs_e > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
        if (pre_scale != nullptr) {
            if (K > CF_PRE_K) return AADG_E_UNSUPPORTED;
            hipLaunchKernelGGL((k_conv1x1_nchw<WR, WC, MI, NI, X3, X3, X3>), dim3((unsigned)wgs_e), dim3(256), 0, st, A, A_lo, IN, OUT, M, K, HW,
                               HW / BP, M / BM, stats, pre_scale, pre_shift);
        } else {
            hipLaunchKernelGGL((k_conv1x1_nchw<WR, WC, MI, NI, X3, X3>), dim3((unsigned)wgs_e), dim3(256), 0, st, A, A_lo, IN, OUT, M, K, HW,
                               HW / BP, M / BM, stats);
        }
        AADG_LAUNCH_CHECK();
        return 0;
    }
    if (pre_scale != nullptr) {
        // M is not a whole number of tiles (the 8-row classifier): the IN tiles still are -- no zero-filled IN value that the transform would
        // turn into relu(shift)
        if (!X3 || (K % 32) != 0 || (HW % BP) != 0 || K > CF_PRE_K) return AADG_E_UNSUPPORTED;
        const long long wgs_p = (long long)N * (HW / BP) * ((M + BM - 1) / BM);
        if (wgs_p > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
        hipLaunchKernelGGL((k_conv1x1_nchw<WR, WC, MI, NI, X3, false, X3>), dim3((unsigned)wgs_p), dim3(256), 0, st, A, A_lo, IN, OUT, M, K, HW,
                           HW / BP, (M + BM - 1) / BM, stats, pre_scale, pre_shift);
        AADG_LAUNCH_CHECK();
        return 0;
    }
    const int tiles_p = (HW + BP - 1) / BP, tiles_m = (M + BM - 1) / BM;
    const long long wgs = (long long)N * tiles_p * tiles_m;
    if (wgs > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    hipLaunchKernelGGL((k_conv1x1_nchw<WR, WC, MI, NI, X3>), dim3((unsigned)wgs), dim3(256), 0, st, A, A_lo, IN, OUT, M, K, HW, tiles_p,
                       tiles_m, stats);
    AADG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int aadg_conv1x1_nchw_supported(int M, int K, int HW) {
    return M > 0 && K > 0 && (K % 8) == 0 && HW >= 8 && (HW % 8) == 0 ? 1 : 0;      // 16-byte rows of A and IN
}

/* out [N, M, HW] = a [M, K] x in [N, K, HW] per image (all bfloat16, float32 accumulation): the forward of a 1x1 convolution
 * (a = weight) or its input gradient (a = weight^T, in = dY) on NCHW tensors */
extern "C" int aadg_conv1x1_nchw_bf16(const void* a, const void* in, void* out, int N, int M, int K, int HW, void* stream) {
    if (a == nullptr || in == nullptr || out == nullptr || N <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)a | (uintptr_t)in | (uintptr_t)out) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_conv1x1_nchw_supported(M, K, HW)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const uint16_t* pa = (const uint16_t*)a;
    if (M <= 64) return launch<1, 4, 2, 2, false>(pa, nullptr, in, out, N, M, K, HW, st);             // 64 x 256 tile
    return launch<2, 2, 2, 4, false>(pa, nullptr, in, out, N, M, K, HW, st);                         // 128 x 256 tile
}

/* The same contraction at float32 precision ("f32x3"): in / out float32 NCHW; a_hi / a_lo = the bfloat16 (hi, lo) halves of the
 * float32 operand a [M, K] (aadg_weight_layouts_split_bf16); products hi*hi + hi*lo + lo*hi on the matrix cores, float32 accumulation */
namespace {
__global__ __launch_bounds__(256) void k_bn_sums_init(double* __restrict__ sums, int C, double count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * C) sums[i] = 0.0;
    if (i == 2 * C) sums[i] = count;
}
}  // namespace

extern "C" int aadg_conv1x1_nchw_f32x3(const void* a_hi, const void* a_lo, const float* in, float* out, int N, int M, int K, int HW,
                                       void* stream) {
    return aadg_conv1x1_nchw_f32x3_stats(a_hi, a_lo, in, out, N, M, K, HW, nullptr, stream);
}

/* ... and, with bn_sums != NULL, the BatchNorm statistics of `out` from the convolution's epilogue: bn_sums [2 M + 1] doubles receives
 * (sum, sum of squares) per output channel and the element count N * HW in the last entry -- the buffer aadg_bn_sync_forward(phase 2)
 * / aadg_bn_sync_backward take (one pass over the float32 output less per BatchNorm layer; the buffer is zeroed in here). */
extern "C" int aadg_conv1x1_nchw_f32x3_stats(const void* a_hi, const void* a_lo, const float* in, float* out, int N, int M, int K, int HW,
                                             double* bn_sums, void* stream) {
    return aadg_conv1x1_nchw_f32x3_pre(a_hi, a_lo, in, out, N, M, K, HW, nullptr, nullptr, bn_sums, stream);
}

/* the shapes of aadg_conv1x1_nchw_f32x3_pre: whole tiles of `in` (K % 32 == 0, HW % 256 == 0), K <= 512; any M */
extern "C" int aadg_conv1x1_f32x3_pre_supported(int M, int K, int HW) {
    return aadg_conv1x1_nchw_supported(M, K, HW) && (K % 32) == 0 && K <= CF_PRE_K && (HW % CF_BP) == 0 ? 1 : 0;
}

/* ... and, with pre_scale / pre_shift [K] != NULL (ABI 10), `in` is the INPUT of the BatchNorm + ReLU in front of this convolution: every
 * loaded value becomes max(in * pre_scale[k] + pre_shift[k], 0) on its way to the matrix cores, so the normalised tensor is never
 * written or read (scale / shift: aadg_bn_finalize_f32).  Shapes of aadg_conv1x1_f32x3_pre_supported only. */
extern "C" int aadg_conv1x1_nchw_f32x3_pre(const void* a_hi, const void* a_lo, const float* in, float* out, int N, int M, int K, int HW,
                                           const float* pre_scale, const float* pre_shift, double* bn_sums, void* stream) {
    if ((pre_scale == nullptr) != (pre_shift == nullptr)) return AADG_E_BADARG;
    if (pre_scale != nullptr && !aadg_conv1x1_f32x3_pre_supported(M, K, HW)) return AADG_E_UNSUPPORTED;
    if (a_hi == nullptr || a_lo == nullptr || in == nullptr || out == nullptr || N <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)a_hi | (uintptr_t)a_lo | (uintptr_t)in | (uintptr_t)out) & 15u) != 0 || (((uintptr_t)bn_sums) & 7u) != 0) return AADG_E_BADARG;
    if (!aadg_conv1x1_nchw_supported(M, K, HW)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const uint16_t* ph = (const uint16_t*)a_hi;
    const uint16_t* pl = (const uint16_t*)a_lo;
    if (bn_sums != nullptr) {
        hipLaunchKernelGGL(k_bn_sums_init, dim3((2 * M + 1 + 255) / 256), dim3(256), 0, st, bn_sums, M, (double)N * (double)HW);
        AADG_LAUNCH_CHECK();
    }
    if (M <= 64) return launch<1, 4, 2, 2, true>(ph, pl, in, out, N, M, K, HW, st, bn_sums, pre_scale, pre_shift);
    // 128 x 256 tile, every wave = all 128 rows x 64 pixels (round 6; it was 2 x 2 waves of 64 x 128): per 16 k a wave reads 8 A fragments
    // (plain ds_read_b128) and 4 B fragments (transpose reads) instead of 4 and 8 -- the transpose reads are the slow ones: 6-8 % on the
    // compute-bound layers (512 -> 2048 at 32 x 32, 144 images: 1.26 -> 1.16 ms; scripts/r6/c1_time.py)
    // (measured and not taken: a 256 x 128 tile of 2 x 2 such waves -- launch<2, 2, 4, 2, true>, the kernel takes 128-pixel tiles for it --
    // halves the activation bytes a workgroup stages per output: within +-3 % of this one, layer by layer)
    return launch<1, 4, 4, 2, true>(ph, pl, in, out, N, M, K, HW, st, bn_sums, pre_scale, pre_shift);
}

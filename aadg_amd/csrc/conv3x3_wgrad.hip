// Weight gradient of a 3x3 / stride-1 / "same" (padding = dilation) convolution over NCHW bfloat16 activations on the matrix cores:
//
//     dW[o][c][kh][kw] = sum_n sum_{y,x} dY[n][o][y][x] * X[n][c][y + (kh - 1) D][x + (kw - 1) D]        (zero outside the image)
//
// The library path (MIOpen igemm_wrw_*_nhwc) transposes both activations to NHWC, zero-fills a float32 workspace, runs the implicit
// GEMM with a global K split and casts the result: at 18 images per rank that is ~110-210 us per convolution for 20-45 us of
// arithmetic.  In NCHW both operands are K-contiguous (K = the pixels of an image row) -- the fragment layout of
// v_mfma_f32_32x32x16_bf16 -- exactly as for the 1x1 case (conv1x1_wgrad.hip); the nine taps only SHIFT the X operand:
//
//   vertical taps   (kh): another image row -> another slot of a ring of 2 D + 2 rows in LDS (one new row per K-step; rows outside the
//                         image are staged as zeros)
//   horizontal taps (kw): a shift by D elements inside the row.  One aligned 16-byte fragment read + the 4-byte words before and
//                         behind it give all three: D = 1 funnel-shifts by one bfloat16 (4 x v_alignbit_b32), D = 2 is a shift by one
//                         32-bit word (register renaming only).  The words beyond the row ends are zeros (the padding columns).
//
// One workgroup (4 waves, 2 x 2) = a 64 x 64 tile of (out, in) channels x all nine taps (9 x 16 accumulator registers per lane) and a
// run of consecutive image rows; K-step = one image row (W = 32 / 64 / 128 pixels).  dY rows are double buffered; the next step's rows
// are in flight (registers) during the MFMAs.  Per 16-pixel sub-step a wave issues 1 + 3 x 3 LDS reads for 9 MFMAs.
// float32 accumulation, split-K partials combined with hardware float atomics into acc[9][Co][Ci] (tap-major: the atomics of a wave
// are contiguous over the in-channel index).
#include <hip/hip_bf16.h>
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int W3_BM = 64, W3_BN = 64;
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));     // a 16-byte LDS read at 4-byte alignment

// X3 = true ("f32x3"): dY / X are float32; a loaded float4 (4 pixels) is split into (hi, lo) bfloat16 halves on its way to LDS (two planes
// of the layout below) and the tap fragments are built per plane; products hi*hi + hi*lo + lo*hi, float32 accumulation.  Both planes are
// 77-148 KB of LDS (one workgroup per CU); at W = 128 (and W = 64, D = 2) the tile is 64 x 32 channels and the two waves of a channel pair split the row's
// K-sub-steps between them.
template <int W, int D, bool X3 = false>
struct W3Cfg {
    static constexpr int PL = X3 ? 2 : 1;
    static constexpr int BN = (X3 && (W == 128 || (W == 64 && D == 2))) ? 32 : W3_BN;       // in-channel tile (both planes must fit the LDS)
    // Dilation 2 (round 6): output row y reads the input rows y - 2, y, y + 2 -- rows of its own parity only -- so the even and the odd
    // rows of an image are two independent sequences whose vertical taps are ONE row apart: the K-steps walk the even rows, then the
    // odd ones, and the ring holds 4 rows instead of 6 (32 x 32: 78 KB of LDS instead of 106 -- two workgroups per CU).  DV = the
    // ring distance of the vertical taps; the horizontal taps still shift by D pixels.
    static constexpr int DV = D == 2 ? 1 : D;
    static constexpr int R = 2 * DV + 2;            // ring slots: rows j - DV .. j + DV of the sequence in use, one being replaced
    static constexpr int PA = W + 8;                // dY row pitch (elements): 2 W + 16 bytes = 4 (mod 8) words -> conflict-free b128
    static constexpr int PX = W + 24;               // X row pitch: 8 pad | W data | 2 halo | pad; data 16-byte aligned, pitch = 4 (mod 8) words
    static constexpr int CPR = X3 ? W / 4 : W / 8;  // 16-byte global chunks per row (4 float32 / 8 bfloat16 pixels)
    static constexpr int LPA = W3_BM * CPR / 256;   // chunks per thread of a staged dY row set
    static constexpr int LPX = BN * CPR / 256;      // ... of an X row set
    static constexpr size_t lds_bytes = (size_t)PL * ((size_t)2 * W3_BM * PA + (size_t)R * BN * PX) * sizeof(uint16_t);
};

__device__ __forceinline__ bf16x8 frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return __builtin_bit_cast(bf16x8, make_uint4(a, b, c, d));
}

// PRE (X3 only): X is the INPUT of the BatchNorm + ReLU the convolution applied on load (aadg_conv3x3_nchw_f32x3_pre): its rows become
// max(fma(x, pre_scale[c], pre_shift[c]), 0) on their way to LDS; rows outside the image stay zero (the padding of the normalised tensor).
// (W = 32 in f32x3: 78 KB of LDS -- two workgroups fit a CU, and the compiler fits the on-load instantiation into 252 registers when
// told to (it took 268-296 and one wave per SIMD: 0.62 -> 0.52 ms on 256 -> 256, 144 images; the plain instantiation was at 243 already))
template <int W, int D, bool X3, bool PRE = false>
__global__ __launch_bounds__(256, (W == 128 || (X3 && W != 32)) ? 1 : 2) void k_wgrad3x3(const void* __restrict__ dY_, const void* __restrict__ X_,
                                                  float* __restrict__ acc,
                                                  int Co, int Ci, int H, int tiles, int tiles_n, int rows_total, int rows_per_block,
                                                  const float* __restrict__ pre_scale = nullptr, const float* __restrict__ pre_shift = nullptr) {
    static_assert(!PRE || X3, "the load transform exists for float32 tensors");
    using C = W3Cfg<W, D, X3>;
    typedef typename std::conditional<X3, float, uint16_t>::type elem_t;
    const elem_t* dY = reinterpret_cast<const elem_t*>(dY_);
    const elem_t* X = reinterpret_cast<const elem_t*>(X_);
    constexpr int R = C::R, PA = C::PA, PX = C::PX, LPA = C::LPA, LPX = C::LPX, CPR = C::CPR, PL = C::PL, BN = C::BN;
    constexpr int EPC = X3 ? 4 : 8;                       // pixels per 16-byte global chunk
    constexpr bool KSPLIT = BN == 32;                     // the waves (wr, 0) / (wr, 1) share a 32 x 32 tile and split the K-sub-steps
    constexpr int A_PLANE = 2 * W3_BM * PA, X_PLANE = R * BN * PX;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* dYs = lds;                                  // [PL][2][64][PA]
    uint16_t* Xs = lds + PL * A_PLANE;                    // [PL][R][BN][PX]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    // XCD-aware decode (as conv1x1_wgrad.hip): the tiles that stream the same rows of dY / X share an XCD's L2
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tile = q % tiles, slice = (q / tiles) * 8 + xcd;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * W3_BM, n0 = tn * BN;
    const int g0 = slice * rows_per_block, g1 = min(rows_total, g0 + rows_per_block);
    if (g0 >= g1) return;

    // zero the pad / halo words of every X row once (data stores never touch them)
    for (int i = tid; i < PL * R * BN; i += 256) {
        uint16_t* row = Xs + (size_t)i * PX;
        *reinterpret_cast<uint4*>(row) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(row + 8 + W) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(row + 16 + W) = make_uint4(0, 0, 0, 0);
    }

    const size_t HW = (size_t)H * W;
    auto load_dy = [&](int n, int y, uint4* st) {
#pragma unroll
        for (int i = 0; i < LPA; ++i) {
            const int id = tid + 256 * i, row = id / CPR, c = (id - row * CPR) * EPC, m = m0 + row;
            st[i] = m < Co ? *reinterpret_cast<const uint4*>(dY + ((size_t)n * Co + m) * HW + (size_t)y * W + c) : make_uint4(0, 0, 0, 0);
        }
    };
    auto load_x = [&](int n, int y, uint4* st) {             // rows outside the image: zeros
#pragma unroll
        for (int i = 0; i < LPX; ++i) {
            const int id = tid + 256 * i, row = id / CPR, c = (id - row * CPR) * EPC, ch = n0 + row;
            st[i] = (ch < Ci && y >= 0 && y < H) ? *reinterpret_cast<const uint4*>(X + ((size_t)n * Ci + ch) * HW + (size_t)y * W + c)
                                                 : make_uint4(0, 0, 0, 0);
        }
    };
    auto slot_of = [&](int y) { return (y + 2 * R) % R; };
    // one staged chunk -> LDS: bfloat16 as loaded (16 bytes), float32 split into its hi / lo halves (8 bytes into each plane)
    auto put = [&](uint16_t* dst, int plane_el, uint4 v) {
        if (X3) {
            uint2 hi, lo;
            aadg_split4(make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)), hi, lo);
            *reinterpret_cast<uint2*>(dst) = hi;
            *reinterpret_cast<uint2*>(dst + plane_el) = lo;
        } else {
            *reinterpret_cast<uint4*>(dst) = v;
        }
    };
    float xs[PRE ? LPX : 1], xh[PRE ? LPX : 1];              // PRE: scale / shift of the thread's X rows (channels n0 + row: constant)
    if (PRE) {
#pragma unroll
        for (int i = 0; i < LPX; ++i) {
            const int ch = min(n0 + (tid + 256 * i) / CPR, Ci - 1);
            xs[i] = pre_scale[ch]; xh[i] = pre_shift[ch];
        }
    }
    auto store_xj = [&](int j, int y, const uint4* st) {      // j: position in the row sequence (ring slot), y: the image row it is (or -1)
        uint16_t* base = Xs + (size_t)slot_of(j) * BN * PX;
        const bool live = y >= 0 && y < H;                    // (uniform) a row of the image: else zeros, the padding
#pragma unroll
        for (int i = 0; i < LPX; ++i) {
            const int id = tid + 256 * i, row = id / CPR, c = (id - row * CPR) * EPC;
            uint4 v = st[i];
            if (PRE && live && n0 + row < Ci) {
                const float sc = xs[i], sh = xh[i];
                v.x = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.x), sc, sh), 0.0f));
                v.y = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.y), sc, sh), 0.0f));
                v.z = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.z), sc, sh), 0.0f));
                v.w = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.w), sc, sh), 0.0f));
            }
            put(base + row * PX + 8 + c, X_PLANE, v);
        }
    };
    auto store_x = [&](int y, const uint4* st) { store_xj(y, y, st); };      // rows in image order: the sequence position is the row
    auto store_dy = [&](int buf, const uint4* st) {
        uint16_t* base = dYs + (size_t)buf * W3_BM * PA;
#pragma unroll
        for (int i = 0; i < LPA; ++i) {
            const int id = tid + 256 * i, row = id / CPR, c = (id - row * CPR) * EPC;
            put(base + row * PA + c, A_PLANE, st[i]);
        }
    };

    f32x16 d[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[t][r] = 0.0f;

    const int a_row = wr * 32 + (lane & 31), b_row = (KSPLIT ? 0 : wc * 32) + (lane & 31), koff = (lane >> 5) * 8;
    const bool hi_half = lane >= 32;
    // An image is H K-steps long (32 at the deepest stages); refilling the ring at its start used to cost 2 D + 1 dependent memory round
    // trips with nothing else in flight.  Round 6: all rows of a refill are requested at once (3x3 256 -> 256 at 32 x 32, 144 images:
    // 0.562 -> 0.446 ms; dilation 2 unchanged).  Measured and dropped: requesting the NEXT image's first rows before the MFMAs of this
    // image's last row, like any other step's rows -- 0.69 ms (the extra staged rows of registers live across the loop).
    constexpr int DV = C::DV;
    constexpr bool PAR = D == 2;                             // rows in parity order (W3Cfg): the sequences of an image are its even / odd rows
    const int Hh = (H + 1) >> 1;                             // rows of the even sequence
    uint4 sdy[LPA], sx[LPX];
    int buf = 0;
    for (int g = g0; g < g1; ++g) {
        const int n = g / H, y = g - n * H;                  // PAR: y is the position in the image's even-then-odd order
        int j = y;                                           // position in the sequence (= the row itself in image order)
        if constexpr (PAR) {
            const int par = y >= Hh ? 1 : 0;
            j = y - par * Hh;
            const int Hs = par ? H - Hh : Hh;                // rows of this sequence
            auto row_of = [&](int jj) { return (jj < 0 || jj >= Hs) ? -1 : 2 * jj + par; };      // image row of position jj (-1: padding)
            if (g == g0 || j == 0) {
                __syncthreads();                             // every wave is done with the previous sequence's rows
                uint4 fill[3][LPX];
#pragma unroll
                for (int r = 0; r < 3; ++r) load_x(n, row_of(j - 1 + r), fill[r]);
                load_dy(n, row_of(j), sdy);
#pragma unroll
                for (int r = 0; r < 3; ++r) store_xj(j - 1 + r, row_of(j - 1 + r), fill[r]);
            } else {
                store_xj(j + 1, row_of(j + 1), sx);          // fetched during the previous step
            }
            store_dy(buf, sdy);
            __syncthreads();
            if (g + 1 < g1 && j + 1 < Hs) {                  // next step's rows: in flight during the MFMAs below
                load_dy(n, row_of(j + 1), sdy);
                load_x(n, row_of(j + 2), sx);
            }
        } else {
            if (g == g0 || y == 0) {
                // (re)fill the ring for this image: rows y - D .. y + D, and dY row y
                __syncthreads();                             // every wave is done with the previous image's rows
                uint4 fill[2 * D + 1][LPX];
#pragma unroll
                for (int r = 0; r <= 2 * D; ++r) load_x(n, y - D + r, fill[r]);
                load_dy(n, y, sdy);
#pragma unroll
                for (int r = 0; r <= 2 * D; ++r) store_x(y - D + r, fill[r]);
            } else {
                store_x(y + D, sx);                          // fetched during the previous step
            }
            store_dy(buf, sdy);
            __syncthreads();
            if (g + 1 < g1 && y + 1 < H) {                   // next step's rows: in flight during the MFMAs below
                load_dy(n, y + 1, sdy);
                load_x(n, y + 1 + D, sx);
            }
        }
        const uint16_t* ab = dYs + (size_t)buf * W3_BM * PA + a_row * PA + koff;
        const uint16_t* xb[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) xb[kh] = Xs + ((size_t)slot_of(j + (kh - 1) * DV) * BN + b_row) * PX + 8 + koff;
        // The 8 pixels of a lane's fragment are followed (lanes < 32) / preceded (lanes >= 32) by the fragment of lane ^ 32 of the same
        // sub-step, and preceded / followed by that lane's fragment of the previous / next sub-step: the two neighbour words of the
        // shifted taps come from v_permlane32_swap instead of two 4-byte LDS reads per fragment (64 lanes on 16 banks: 4-way
        // conflicts -- SQ_LDS_BANK_CONFLICT was 4x the forward kernel's).  The row ends are the zero padding columns.
        constexpr int KS = W / 16;
        constexpr int KS_LO = 0, KS_N = KSPLIT ? KS / 2 : KS;      // this wave's sub-steps: [ks0, ks0 + KS_N)
        const int ks0 = KSPLIT ? wc * (KS / 2) : KS_LO;
        bf16x8 a[PL][KS_N];
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
#pragma unroll
            for (int j = 0; j < KS_N; ++j)
                a[pl][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ab + pl * A_PLANE + (ks0 + j) * 16));
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            // fragments of the three horizontal taps, per plane: f[pl][kw][j]
            bf16x8 f0[PL][KS_N], f1[PL][KS_N], f2[PL][KS_N];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) {
                if (D == 2 && KSPLIT) {
                    // dilation 2 = a shift by one 32-bit word: the shifted fragments as 4-byte-aligned 16-byte LDS reads (two ds_read2_b32
                    // each, straight into the MFMA operand registers -- no VALU); the words beyond the row ends are the zero pads.
                    // Round 6, measured against the lane-exchange path below: 0.727 vs 0.80 ms at 64 x 64 (256 -> 256, 36 images: the
                    // K-split tile, two waves share the fragments' rows), but 2.43 vs 2.31 ms at 32 x 32 (512 -> 512, 144 images) --
                    // there the register moves of the exchange cost less than 2.25x the LDS read passes (scripts/r6/w3_time.py)
#pragma unroll
                    for (int j = 0; j < KS_N; ++j) {
                        const uint16_t* p = xb[kh] + pl * X_PLANE + (ks0 + j) * 16;
                        const u32x4_a4 l = *reinterpret_cast<const u32x4_a4*>(p - 2), r = *reinterpret_cast<const u32x4_a4*>(p + 2);
                        f0[pl][j] = frag(l.x, l.y, l.z, l.w);
                        f1[pl][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
                        f2[pl][j] = frag(r.x, r.y, r.z, r.w);
                    }
                    continue;
                }
                // sub-steps ks0 - 1 .. ks0 + KS_N (clamped to the row): the neighbours' edge words come from them
                uint4 cur[KS_N + 2];
#pragma unroll
                for (int j = 0; j < KS_N + 2; ++j) {
                    const int ks = ks0 + j - 1;
                    cur[j] = (ks >= 0 && ks < KS) ? *reinterpret_cast<const uint4*>(xb[kh] + pl * X_PLANE + ks * 16) : make_uint4(0, 0, 0, 0);
                }
                uint32_t w_up[KS_N + 2], w_lo[KS_N + 2], x_up[KS_N + 2], x_lo[KS_N + 2];     // partner's last / first word, as seen by the upper / lower half
#pragma unroll
                for (int j = 0; j < KS_N + 2; ++j) {
                    const auto rw = __builtin_amdgcn_permlane32_swap(cur[j].w, cur[j].w, false, false);
                    const auto rx = __builtin_amdgcn_permlane32_swap(cur[j].x, cur[j].x, false, false);
                    w_up[j] = rw[0]; w_lo[j] = rw[1];            // [0]: lanes >= 32 hold the lower half's value; [1]: lanes < 32 the upper half's
                    x_up[j] = rx[0]; x_lo[j] = rx[1];
                }
#pragma unroll
                for (int j = 0; j < KS_N; ++j) {
                    // (an out-of-row neighbour sub-step was loaded as zeros: the padding columns)
                    const uint32_t prv = hi_half ? w_up[j + 1] : w_lo[j];
                    const uint32_t nxt = hi_half ? x_up[j + 2] : x_lo[j + 1];
                    const uint4 c = cur[j + 1];
                    if (D == 1) {
                        const uint32_t s1 = __builtin_amdgcn_alignbit(c.y, c.x, 16), s2 = __builtin_amdgcn_alignbit(c.z, c.y, 16),
                                       s3 = __builtin_amdgcn_alignbit(c.w, c.z, 16);
                        f0[pl][j] = frag(__builtin_amdgcn_alignbit(c.x, prv, 16), s1, s2, s3);            // X[p - 1]
                        f2[pl][j] = frag(s1, s2, s3, __builtin_amdgcn_alignbit(nxt, c.w, 16));            // X[p + 1]
                    } else {
                        f0[pl][j] = frag(prv, c.x, c.y, c.z);                                             // X[p - 2]
                        f2[pl][j] = frag(c.y, c.z, c.w, nxt);                                             // X[p + 2]
                    }
                    f1[pl][j] = __builtin_bit_cast(bf16x8, c);
                }
            }
#pragma unroll
            for (int j = 0; j < KS_N; ++j) {
                if (X3) {
                    d[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1][j], f0[0][j], d[kh * 3 + 0], 0, 0, 0);
                    d[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][j], f0[PL - 1][j], d[kh * 3 + 0], 0, 0, 0);
                    d[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1][j], f1[0][j], d[kh * 3 + 1], 0, 0, 0);
                    d[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][j], f1[PL - 1][j], d[kh * 3 + 1], 0, 0, 0);
                    d[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1][j], f2[0][j], d[kh * 3 + 2], 0, 0, 0);
                    d[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][j], f2[PL - 1][j], d[kh * 3 + 2], 0, 0, 0);
                }
                d[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][j], f0[0][j], d[kh * 3 + 0], 0, 0, 0);
                d[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][j], f1[0][j], d[kh * 3 + 1], 0, 0, 0);
                d[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][j], f2[0][j], d[kh * 3 + 2], 0, 0, 0);
            }
        }
        buf ^= 1;
    }
    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int nn = n0 + (KSPLIT ? 0 : wc * 32) + (lane & 31);
    if (nn < Ci) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float* at = acc + (size_t)t * Co * Ci;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < Co) unsafeAtomicAdd(at + (size_t)m * Ci + nn, d[t][r]);
            }
        }
    }
}

// ---- stride 2 (padding 1, dilation 1): dW[o][c][kh][kw] = sum dY[n][o][y][x] * X[n][c][2 y + kh - 1][2 x + kw - 1] ------------------
// K-step = one OUTPUT row (WO pixels); it reads the input rows 2 y - 1, 2 y, 2 y + 1 (two new rows per step: a ring of five).  The
// staging step splits every input row into its even and odd columns (E[x] = X[2 x], O[x] = X[2 x + 1]; two v_perm per chunk), so
// the three horizontal taps are again K-contiguous: kw = 1 -> E[x], kw = 2 -> O[x], kw = 0 -> O[x - 1] (funnel with the word before,
// zero at the row start).
// X3 = true ("f32x3"): dY / X float32, split into (hi, lo) bfloat16 halves on their way to LDS (two planes of the layout), fragments per
// plane, hi*hi + hi*lo + lo*hi; at WO = 64 the in-channel tile is 32 (both planes of the five-row ring must fit the LDS) and the two waves
// of a channel pair split the row's K-sub-steps.
template <int WO, bool X3 = false>
struct W3S2Cfg {
    static constexpr int PL = X3 ? 2 : 1;
    static constexpr int BN = (X3 && WO == 64) ? 32 : W3_BN;
    static constexpr int R = 5;
    static constexpr int PA = WO + 8;
    static constexpr int PX = 2 * WO + 24;          // 8 pad | E (WO) | 8 pad = O's left halo | O (WO) | 8 pad; pitch = 4 (mod 8) words
    static constexpr int EO = 8, OO = 16 + WO;
    static constexpr int CPA = X3 ? WO / 4 : WO / 8;                // 16-byte global chunks per dY row
    static constexpr int LPA = W3_BM * CPA / 256;                   // dY chunks per thread
    static constexpr int LPX = BN * (2 * WO / 8) / 256;             // 8-pixel chunks per thread of ONE input row
    static constexpr size_t lds_bytes = (size_t)PL * ((size_t)2 * W3_BM * PA + (size_t)R * BN * PX) * sizeof(uint16_t);
};

template <int WO, bool X3>
__global__ __launch_bounds__(256, (WO == 64 || X3) ? 1 : 2) void k_wgrad3x3_s2(const void* __restrict__ dY_, const void* __restrict__ X_,
                                                        float* __restrict__ acc,
                                                        int Co, int Ci, int Ho, int tiles, int tiles_n, int rows_total, int rows_per_block) {
    using C = W3S2Cfg<WO, X3>;
    typedef typename std::conditional<X3, float, uint16_t>::type elem_t;
    const elem_t* dY = reinterpret_cast<const elem_t*>(dY_);
    const elem_t* X = reinterpret_cast<const elem_t*>(X_);
    constexpr int R = C::R, PA = C::PA, PX = C::PX, LPA = C::LPA, LPX = C::LPX, WI = 2 * WO, CPA = C::CPA, CPX = WI / 8, PL = C::PL, BN = C::BN;
    constexpr int EPA = X3 ? 4 : 8;                       // pixels per 16-byte dY chunk
    constexpr bool KSPLIT = BN == 32;
    constexpr int A_PLANE = 2 * W3_BM * PA, X_PLANE = R * BN * PX;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* dYs = lds;                                  // [PL][2][64][PA]
    uint16_t* Xs = lds + PL * A_PLANE;                    // [PL][R][BN][PX]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tile = q % tiles, slice = (q / tiles) * 8 + xcd;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * W3_BM, n0 = tn * BN;
    const int g0 = slice * rows_per_block, g1 = min(rows_total, g0 + rows_per_block);
    if (g0 >= g1) return;
    const int Hi = 2 * Ho;
    for (int i = tid; i < PL * R * BN; i += 256) {        // pads / halos: zero once
        uint16_t* row = Xs + (size_t)i * PX;
        *reinterpret_cast<uint4*>(row) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(row + 8 + WO) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(row + 16 + 2 * WO) = make_uint4(0, 0, 0, 0);
    }
    const size_t HWo = (size_t)Ho * WO, HWi = (size_t)Hi * WI;
    auto load_dy = [&](int n, int y, uint4* st) {
#pragma unroll
        for (int i = 0; i < LPA; ++i) {
            const int id = tid + 256 * i, row = id / CPA, c = (id - row * CPA) * EPA, m = m0 + row;
            st[i] = m < Co ? *reinterpret_cast<const uint4*>(dY + ((size_t)n * Co + m) * HWo + (size_t)y * WO + c) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_dy = [&](int buf, const uint4* st) {
        uint16_t* base = dYs + (size_t)buf * W3_BM * PA;
#pragma unroll
        for (int i = 0; i < LPA; ++i) {
            const int id = tid + 256 * i, row = id / CPA, c = (id - row * CPA) * EPA;
            if (X3) {
                uint2 hi, lo;
                const uint4 v = st[i];
                aadg_split4(make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)), hi, lo);
                *reinterpret_cast<uint2*>(base + row * PA + c) = hi;
                *reinterpret_cast<uint2*>(base + A_PLANE + row * PA + c) = lo;
            } else {
                *reinterpret_cast<uint4*>(base + row * PA + c) = st[i];
            }
        }
    };
    // input row r; rows outside the image: zeros.  X3: st[2 i] / st[2 i + 1] = pixels 0..3 / 4..7 of chunk i as float32
    auto load_x = [&](int n, int r, uint4* st) {
#pragma unroll
        for (int i = 0; i < LPX; ++i) {
            const int id = tid + 256 * i, row = id / CPX, ch = id - row * CPX, c = n0 + row;
            const bool ok = c < Ci && r >= 0 && r < Hi;
            const elem_t* src = X + ((size_t)n * Ci + c) * HWi + (size_t)r * WI + ch * 8;
            if (X3) {
                st[2 * i] = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
                st[2 * i + 1] = ok ? *reinterpret_cast<const uint4*>(src + 4) : make_uint4(0, 0, 0, 0);
            } else {
                st[i] = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto slot_of = [&](int r) { return (r + 2 * R) % R; };
    auto store_x = [&](int r, const uint4* st) {             // split the 8 pixels of a chunk into 4 even + 4 odd columns
        uint16_t* base = Xs + (size_t)slot_of(r) * BN * PX;
#pragma unroll
        for (int i = 0; i < LPX; ++i) {
            const int id = tid + 256 * i, row = id / CPX, ch = id - row * CPX;
            uint4 vv[PL];
            if (X3) {
                uint2 h0, l0, h1, l1;
                const uint4 q0 = st[2 * i], q1 = st[2 * i + 1];
                aadg_split4(make_float4(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w)), h0, l0);
                aadg_split4(make_float4(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z), __uint_as_float(q1.w)), h1, l1);
                vv[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
                vv[PL - 1] = make_uint4(l0.x, l0.y, l1.x, l1.y);
            } else {
                vv[0] = st[i];
            }
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) {
                const uint4 v = vv[pl];
                const uint2 e = make_uint2(__builtin_amdgcn_perm(v.y, v.x, 0x05040100u), __builtin_amdgcn_perm(v.w, v.z, 0x05040100u));
                const uint2 o = make_uint2(__builtin_amdgcn_perm(v.y, v.x, 0x07060302u), __builtin_amdgcn_perm(v.w, v.z, 0x07060302u));
                *reinterpret_cast<uint2*>(base + pl * X_PLANE + row * PX + C::EO + 4 * ch) = e;
                *reinterpret_cast<uint2*>(base + pl * X_PLANE + row * PX + C::OO + 4 * ch) = o;
            }
        }
    };

    f32x16 d[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[t][r] = 0.0f;
    const int a_row = wr * 32 + (lane & 31), b_row = (KSPLIT ? 0 : wc * 32) + (lane & 31), koff = (lane >> 5) * 8;
    constexpr int XR = X3 ? 2 * LPX : LPX;
    uint4 sdy[LPA], sx0[XR], sx1[XR];
    int buf = 0;
    for (int g = g0; g < g1; ++g) {
        const int n = g / Ho, y = g - n * Ho;
        if (g == g0 || y == 0) {
            __syncthreads();                                 // every wave is done with the previous rows
            for (int r = 2 * y - 1; r <= 2 * y + 1; ++r) {
                load_x(n, r, sx0);
                store_x(r, sx0);
            }
            load_dy(n, y, sdy);
        } else {
            store_x(2 * y, sx0);                             // fetched during the previous step
            store_x(2 * y + 1, sx1);
        }
        store_dy(buf, sdy);
        __syncthreads();
        if (g + 1 < g1 && y + 1 < Ho) {
            load_dy(n, y + 1, sdy);
            load_x(n, 2 * y + 2, sx0);
            load_x(n, 2 * y + 3, sx1);
        }
        const uint16_t* ab = dYs + (size_t)buf * W3_BM * PA + a_row * PA + koff;
        const uint16_t* xb[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) xb[kh] = Xs + ((size_t)slot_of(2 * y + kh - 1) * BN + b_row) * PX + koff;
        constexpr int KS = WO / 16, KS_N = KSPLIT ? KS / 2 : KS;
        const int ks0 = KSPLIT ? wc * (KS / 2) : 0;
#pragma unroll
        for (int j = 0; j < KS_N; ++j) {
            const int ks = ks0 + j;
            bf16x8 a[PL];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) a[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ab + pl * A_PLANE + ks * 16));
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                bf16x8 f0[PL], f1[PL], f2[PL];
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) {
                    const uint16_t* xr = xb[kh] + pl * X_PLANE + ks * 16;
                    const uint4 e = *reinterpret_cast<const uint4*>(xr + C::EO);
                    const uint4 o = *reinterpret_cast<const uint4*>(xr + C::OO);
                    const uint32_t prv = *reinterpret_cast<const uint32_t*>(xr + C::OO - 2);
                    f0[pl] = frag(__builtin_amdgcn_alignbit(o.x, prv, 16), __builtin_amdgcn_alignbit(o.y, o.x, 16),
                                  __builtin_amdgcn_alignbit(o.z, o.y, 16), __builtin_amdgcn_alignbit(o.w, o.z, 16));       // O[x - 1]
                    f1[pl] = __builtin_bit_cast(bf16x8, e);
                    f2[pl] = __builtin_bit_cast(bf16x8, o);
                }
                if (X3) {
                    d[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1], f0[0], d[kh * 3 + 0], 0, 0, 0);
                    d[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1], f1[0], d[kh * 3 + 1], 0, 0, 0);
                    d[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1], f2[0], d[kh * 3 + 2], 0, 0, 0);
                    d[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], f0[PL - 1], d[kh * 3 + 0], 0, 0, 0);
                    d[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], f1[PL - 1], d[kh * 3 + 1], 0, 0, 0);
                    d[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], f2[PL - 1], d[kh * 3 + 2], 0, 0, 0);
                }
                d[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], f0[0], d[kh * 3 + 0], 0, 0, 0);
                d[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], f1[0], d[kh * 3 + 1], 0, 0, 0);
                d[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], f2[0], d[kh * 3 + 2], 0, 0, 0);
            }
        }
        buf ^= 1;
    }
    const int nn = n0 + (KSPLIT ? 0 : wc * 32) + (lane & 31);
    if (nn < Ci) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float* at = acc + (size_t)t * Co * Ci;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < Co) unsafeAtomicAdd(at + (size_t)m * Ci + nn, d[t][r]);
            }
        }
    }
}

template <int WO, bool X3>
int launch_s2(const void* dY, const void* X, float* acc, int N, int Co, int Ci, int Ho, hipStream_t st) {
    using C = W3S2Cfg<WO, X3>;
    const int tiles_m = (Co + W3_BM - 1) / W3_BM, tiles_n = (Ci + C::BN - 1) / C::BN, tiles = tiles_m * tiles_n;
    const long long rows_total = (long long)N * Ho;
    const long long target = 512, min_mfma = 1152;
    long long slices = (target + tiles - 1) / tiles;
    long long rpb = (rows_total + slices - 1) / slices;
    const long long min_rows = (min_mfma + (WO / 16) * 9 - 1) / ((WO / 16) * 9);
    if (rpb < min_rows) rpb = min_rows;
    slices = (rows_total + rpb - 1) / rpb;
    static bool attr_set = false;
    if (!attr_set) {
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3x3_s2<WO, X3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)C::lds_bytes));
        attr_set = true;
    }
    AADG_HIP_TRY(hipMemsetAsync(acc, 0, (size_t)9 * Co * Ci * sizeof(float), st));
    const long long slice_groups = (slices + 7) / 8;
    hipLaunchKernelGGL((k_wgrad3x3_s2<WO, X3>), dim3((unsigned)(slice_groups * tiles * 8)), dim3(256), C::lds_bytes, st, dY, X, acc, Co, Ci, Ho,
                       tiles, tiles_n, (int)rows_total, (int)rpb);
    AADG_LAUNCH_CHECK();
    return 0;
}

template <int W, int D, bool X3>
int launch(const void* dY, const void* X, float* acc, int N, int Co, int Ci, int H, hipStream_t st, const float* pre_scale = nullptr,
           const float* pre_shift = nullptr) {
    using C = W3Cfg<W, D, X3>;
    const int tiles_m = (Co + W3_BM - 1) / W3_BM, tiles_n = (Ci + C::BN - 1) / C::BN, tiles = tiles_m * tiles_n;
    const long long rows_total = (long long)N * H;
    // Every workgroup ends with 9 x 64 x 64 float atomics (the cost of ~1000 MFMAs), so: ~512 workgroups (two per CU), and never fewer
    // than 1152 MFMAs per wave (18 images per rank: 1.93 -> 1.20 ms over the backbone's 14 convolutions; 144 images: 5.5 -> 5.1)
    const long long target = 512, min_mfma = 1152;
    long long slices = (target + tiles - 1) / tiles;
    long long rpb = (rows_total + slices - 1) / slices;
    const long long min_rows = (min_mfma + (W / 16) * 9 - 1) / ((W / 16) * 9);
    if (rpb < min_rows) rpb = min_rows;
    slices = (rows_total + rpb - 1) / rpb;
    static bool attr_set = false;                            // per instantiation; idempotent
    if (!attr_set) {
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3x3<W, D, X3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)C::lds_bytes));
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3x3<W, D, X3, X3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)C::lds_bytes));
        attr_set = true;
    }
    if (pre_scale != nullptr && !X3) return AADG_E_UNSUPPORTED;
    AADG_HIP_TRY(hipMemsetAsync(acc, 0, (size_t)9 * Co * Ci * sizeof(float), st));
    const long long slice_groups = (slices + 7) / 8;          // slices are padded to a multiple of 8 (empty ones exit at once)
    if (pre_scale != nullptr)
        hipLaunchKernelGGL((k_wgrad3x3<W, D, X3, X3>), dim3((unsigned)(slice_groups * tiles * 8)), dim3(256), C::lds_bytes, st, dY, X, acc, Co, Ci,
                           H, tiles, tiles_n, (int)rows_total, (int)rpb, pre_scale, pre_shift);
    else
    hipLaunchKernelGGL((k_wgrad3x3<W, D, X3>), dim3((unsigned)(slice_groups * tiles * 8)), dim3(256), C::lds_bytes, st, dY, X, acc, Co, Ci, H,
                       tiles, tiles_n, (int)rows_total, (int)rpb);
    AADG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int aadg_conv3x3_wgrad_supported(int Co, int Ci, int H, int W, int dilation) {
    return Co > 0 && Ci > 0 && H > 0 && (W == 32 || W == 64 || W == 128) && (dilation == 1 || dilation == 2) ? 1 : 0;
}

namespace {
template <bool X3>
int wgrad3x3_dispatch(const void* a, const void* b, float* dweight9, int N, int Co, int Ci, int H, int W, int dilation, hipStream_t st,
                      const float* ps = nullptr, const float* ph = nullptr) {
    if (dilation == 1) {
        if (W == 32) return launch<32, 1, X3>(a, b, dweight9, N, Co, Ci, H, st, ps, ph);
        if (W == 64) return launch<64, 1, X3>(a, b, dweight9, N, Co, Ci, H, st, ps, ph);
        return launch<128, 1, X3>(a, b, dweight9, N, Co, Ci, H, st, ps, ph);
    }
    if (W == 32) return launch<32, 2, X3>(a, b, dweight9, N, Co, Ci, H, st, ps, ph);
    if (W == 64) return launch<64, 2, X3>(a, b, dweight9, N, Co, Ci, H, st, ps, ph);
    if (X3) return AADG_E_UNSUPPORTED;                            // (both planes of a 128-pixel ring of six rows exceed the LDS)
    return launch<128, 2, false>(a, b, dweight9, N, Co, Ci, H, st);
}
}  // namespace

// dweight9: float32 [9][Co][Ci] (tap kh * 3 + kw major); the caller permutes it into [Co][Ci][3][3]
extern "C" int aadg_conv3x3_wgrad_bf16(const void* dy, const void* x, float* dweight9, int N, int Co, int Ci, int H, int W, int dilation,
                                       void* stream) {
    if (dy == nullptr || x == nullptr || dweight9 == nullptr || N <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)dy | (uintptr_t)x) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_conv3x3_wgrad_supported(Co, Ci, H, W, dilation) || (long long)N * H > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    return wgrad3x3_dispatch<false>(dy, x, dweight9, N, Co, Ci, H, W, dilation, (hipStream_t)stream);
}

/* The same weight gradient from float32 NCHW dy / x at float32 precision ("f32x3") */
extern "C" int aadg_conv3x3_wgrad_f32x3(const float* dy, const float* x, float* dweight9, int N, int Co, int Ci, int H, int W, int dilation,
                                        void* stream) {
    return aadg_conv3x3_wgrad_f32x3_pre(dy, x, dweight9, N, Co, Ci, H, W, dilation, nullptr, nullptr, stream);
}

/* ... and, with pre_scale / pre_shift [Ci] != NULL (ABI 10), x is the INPUT of the BatchNorm + ReLU the convolution applied on load
 * (aadg_conv3x3_nchw_f32x3_pre) */
extern "C" int aadg_conv3x3_wgrad_f32x3_pre(const float* dy, const float* x, float* dweight9, int N, int Co, int Ci, int H, int W, int dilation,
                                            const float* pre_scale, const float* pre_shift, void* stream) {
    if (dy == nullptr || x == nullptr || dweight9 == nullptr || N <= 0 || (pre_scale == nullptr) != (pre_shift == nullptr)) return AADG_E_BADARG;
    if ((((uintptr_t)dy | (uintptr_t)x) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_conv3x3_wgrad_supported(Co, Ci, H, W, dilation) || (W == 128 && dilation == 2) || (long long)N * H > 0x7FFFFFFFLL)
        return AADG_E_UNSUPPORTED;
    return wgrad3x3_dispatch<true>(dy, x, dweight9, N, Co, Ci, H, W, dilation, (hipStream_t)stream, pre_scale, pre_shift);
}

/* stride 2, padding 1, dilation 1: dy [N, Co, Ho, Wo], x [N, Ci, 2 Ho, 2 Wo]; Wo in {32, 64} */
extern "C" int aadg_conv3x3s2_wgrad_supported(int Co, int Ci, int Ho, int Wo) {
    return Co > 0 && Ci > 0 && Ho > 0 && (Wo == 32 || Wo == 64) ? 1 : 0;
}
extern "C" int aadg_conv3x3s2_wgrad_bf16(const void* dy, const void* x, float* dweight9, int N, int Co, int Ci, int Ho, int Wo, void* stream) {
    if (dy == nullptr || x == nullptr || dweight9 == nullptr || N <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)dy | (uintptr_t)x) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_conv3x3s2_wgrad_supported(Co, Ci, Ho, Wo) || (long long)N * Ho > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (Wo == 32) return launch_s2<32, false>(dy, x, dweight9, N, Co, Ci, Ho, st);
    return launch_s2<64, false>(dy, x, dweight9, N, Co, Ci, Ho, st);
}
/* ... and from float32 tensors at float32 precision ("f32x3") */
extern "C" int aadg_conv3x3s2_wgrad_f32x3(const float* dy, const float* x, float* dweight9, int N, int Co, int Ci, int Ho, int Wo,
                                          void* stream) {
    if (dy == nullptr || x == nullptr || dweight9 == nullptr || N <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)dy | (uintptr_t)x) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_conv3x3s2_wgrad_supported(Co, Ci, Ho, Wo) || (long long)N * Ho > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (Wo == 32) return launch_s2<32, true>(dy, x, dweight9, N, Co, Ci, Ho, st);
    return launch_s2<64, true>(dy, x, dweight9, N, Co, Ci, Ho, st);
}

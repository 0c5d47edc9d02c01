// The ResNet stem: 7x7 / stride 2 / padding 3 convolution of a 3-channel image into 64 channels (bfloat16 NCHW in and out,
// float32 master weights), forward.  The library runs it as an NHWC implicit GEMM between three layout transposes and a
// zero-fill (~2.1 ms at N = 144 x 512^2, of which the GEMM kernel is half); with 3 input channels the problem is a thin
// GEMM that maps onto the matrix cores directly from NCHW:
//
//     Y[o][pixel] = sum_k  Wk[o][k] * Xk[k][pixel],      k = (c, kh, kw') with kw' = 0..7: seven taps behind one zero pad
//
//   * the 8 values k..k+7 an MFMA lane holds for its pixel are 8 CONSECUTIVE input columns (2j-4 .. 2j+3) of one input
//     row (c, 2i+kh-3): four 4-byte LDS reads from a bfloat16 patch, no im2col buffer;
//   * K = 21 (c, kh) rows x 8 = 168 -> 11 MFMA K-steps of 16 (two (c, kh) rows per step, the 22nd has zero weights);
//   * a workgroup (4 waves) owns 8 output rows x 64 columns: 21 x 144-column patch per channel in LDS (18 KB), each wave
//     two output rows x (2 x 2) 32x32 tiles, the 22 weight fragments per lane stay in registers across tiles (persistent
//     grid-stride loop over tiles);
//   * epilogue: neighbouring lanes (pixels j, j+1) trade one register so that every lane stores two adjacent pixels.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int ST_CO = 64, ST_KS = 11, ST_TR = 8, ST_TC = 64;       // out channels, K-steps, tile rows / columns (outputs)
constexpr int ST_PR = 2 * ST_TR + 5, ST_PC = 144;                 // patch rows per channel, patch row pitch (elements)

// 8 consecutive image elements as bfloat16: the image may still be float32 (the augmentation kernel's output) -- rounding
// it here (round to nearest even, what Tensor.to(bfloat16) does) saves the separate cast pass over the batch
template <typename TIN> __device__ __forceinline__ uint4 load8_bf16(const TIN* p);
template <> __device__ __forceinline__ uint4 load8_bf16<uint16_t>(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }
template <> __device__ __forceinline__ uint4 load8_bf16<float>(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    return make_uint4(aadg_f2bf_pk(a.x, a.y), aadg_f2bf_pk(a.z, a.w), aadg_f2bf_pk(b.x, b.y), aadg_f2bf_pk(b.z, b.w));
}

// wfrag[(ks * 2 + mt) * 64 + lane] = the A fragment (8 bfloat16) of lane for K-step ks and channel tile mt
__global__ __launch_bounds__(256) void k_stem_pack(const float* __restrict__ w, uint4* __restrict__ wfrag) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ST_KS * 2 * 64) return;
    const int lane = t & 63, mt = (t >> 6) & 1, ks = t >> 7;
    const int idx = 2 * ks + (lane >> 5), o = 32 * mt + (lane & 31);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (idx < 21) {
        const float* p = w + ((size_t)o * 21 + idx) * 7;          // w[o][c][kh][0..6], (c, kh) = idx
#pragma unroll
        for (int k = 0; k < 7; ++k) v[k + 1] = p[k];
    }
    uint2 h0, l0, h1, l1;                                           // (hi, lo) halves: the float32-precision kernel reads both planes
    aadg_split4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
    aadg_split4(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
    wfrag[t] = make_uint4(h0.x, h0.y, h1.x, h1.y);                  // = the bfloat16 rounding of the weights (what the bfloat16 kernel reads)
    wfrag[ST_KS * 2 * 64 + t] = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

// X3 = true ("f32x3", csrc/conv1x1_fwd.hip): float32 image in, float32 map out; the patch is staged as (hi, lo) bfloat16 planes, the
// hi weight fragments stay in registers, the lo ones in LDS; hi*hi + hi*lo + lo*hi per fragment pair, float32 accumulation.
template <typename TIN, bool X3>
__global__ __launch_bounds__(256) void k_stem7x7(const TIN* __restrict__ x, const uint4* __restrict__ wfrag,
                                                 void* __restrict__ y_, int H, int W, int tiles_x, int tiles_y, int total_tiles) {
    constexpr int PL = X3 ? 2 : 1, P_EL = 3 * ST_PR * ST_PC;
    __shared__ __attribute__((aligned(16))) uint16_t P[PL * P_EL];
    __shared__ __attribute__((aligned(16))) uint4 WLo[X3 ? ST_KS * 2 * 64 : 1];
    uint16_t* y = reinterpret_cast<uint16_t*>(y_);
    if (X3) {
        for (int i = threadIdx.x; i < ST_KS * 2 * 64; i += 256) WLo[i] = wfrag[ST_KS * 2 * 64 + i];
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int jj = lane & 31, g = lane >> 5;
    const int Ho = H / 2, Wo = W / 2;
    bf16x8 a[2][ST_KS];
    int koff[ST_KS];                                               // patch offset of this lane's (c, kh) row per K-step
#pragma unroll
    for (int ks = 0; ks < ST_KS; ++ks) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a[mt][ks] = __builtin_bit_cast(bf16x8, wfrag[(ks * 2 + mt) * 64 + lane]);
        const int idx = min(2 * ks + g, 20);                       // the 22nd row has zero weights: any finite data will do
        koff[ks] = ((idx / 7) * ST_PR + idx % 7) * ST_PC;
    }
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, t2 = tile / tiles_x;
        const int ty = t2 % tiles_y, n = t2 / tiles_y;
        const int i0 = ty * ST_TR, j0 = tx * ST_TC;
        const TIN* xn = x + (size_t)n * 3 * H * W;
        __syncthreads();                                           // the previous tile's readers are done with P
        // patch: input rows 2*i0 - 3 .. 2*i0 + 17, columns 2*j0 - 8 .. 2*j0 + 135 (18 chunks of 8), zero outside the image
        for (int it = tid; it < 3 * ST_PR * (ST_PC / 8); it += 256) {
            const int q = it % (ST_PC / 8), rc = it / (ST_PC / 8);
            const int pr = rc % ST_PR, c = rc / ST_PR;
            const int row = 2 * i0 - 3 + pr, col = 2 * j0 - 8 + 8 * q;
            uint4 v = make_uint4(0u, 0u, 0u, 0u), vl = make_uint4(0u, 0u, 0u, 0u);
            if (row >= 0 && row < H && col >= 0 && col < W) {
                if constexpr (X3) {
                    const float* src = reinterpret_cast<const float*>(xn) + ((size_t)c * H + row) * W + col;
                    uint2 h0, l0, h1, l1;
                    aadg_split4(*reinterpret_cast<const float4*>(src), h0, l0);
                    aadg_split4(*reinterpret_cast<const float4*>(src + 4), h1, l1);
                    v = make_uint4(h0.x, h0.y, h1.x, h1.y);
                    vl = make_uint4(l0.x, l0.y, l1.x, l1.y);
                } else {
                    v = load8_bf16<TIN>(xn + ((size_t)c * H + row) * W + col);
                }
            }
            *reinterpret_cast<uint4*>(P + (c * ST_PR + pr) * ST_PC + 8 * q) = v;
            if (X3) *reinterpret_cast<uint4*>(P + P_EL + (c * ST_PR + pr) * ST_PC + 8 * q) = vl;
        }
        __syncthreads();
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int ro = wv * 2 + rr, i = i0 + ro;
            if (i >= Ho) break;
            f32x16 d[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) d[mt][nt][r] = 0.0f;
            // tap kw' = 0 of output column j sits at patch column 2*(j - j0) + 4
            const uint32_t* Pw = reinterpret_cast<const uint32_t*>(P + 2 * ro * ST_PC + 2 * jj + 4);
#pragma unroll
            for (int ks = 0; ks < ST_KS; ++ks) {
                bf16x8 b[2], bl[2], al[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const uint32_t* p = Pw + (koff[ks] >> 1) + 32 * nt;
                    b[nt] = __builtin_bit_cast(bf16x8, make_uint4(p[0], p[1], p[2], p[3]));
                    if (X3) {
                        const uint32_t* q = p + P_EL / 2;
                        bl[nt] = __builtin_bit_cast(bf16x8, make_uint4(q[0], q[1], q[2], q[3]));
                    }
                }
                if (X3) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) al[mt] = __builtin_bit_cast(bf16x8, WLo[(ks * 2 + mt) * 64 + lane]);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if (X3) {
                            d[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], b[nt], d[mt][nt], 0, 0, 0);
                            d[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][ks], bl[nt], d[mt][nt], 0, 0, 0);
                        }
                        d[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][ks], b[nt], d[mt][nt], 0, 0, 0);
                    }
            }
            if constexpr (X3) {
                // float32 out: the 32 lanes of a channel row store 128 contiguous bytes (streaming: 2.4 GB at N = 144 x 512^2)
                float* yf = reinterpret_cast<float*>(y_) + (((size_t)n * ST_CO) * Ho + i) * Wo + j0;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int o = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * g, j = 32 * nt + jj;
                            if (j0 + j < Wo) __builtin_nontemporal_store(d[mt][nt][r], yf + (size_t)o * Ho * Wo + j);
                        }
                continue;
            }
            // C/D layout: column (pixel) = lane & 31, row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
            // Lanes j (even) and j + 1 trade registers r / r + 1: the even lane stores pixels (j, j + 1) of channel row r,
            // the odd lane those of row r + 1 -- 4-byte stores instead of 2-byte ones.
            uint16_t* yn = y + (((size_t)n * ST_CO) * Ho + i) * Wo + j0;
            const bool odd = jj & 1;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float mine0 = d[mt][nt][r], mine1 = d[mt][nt][r + 1];
                        const float give = odd ? mine0 : mine1;
                        // quad_perm [1, 0, 3, 2]: swap with the neighbouring lane inside the VALU (no LDS crossbar trip)
                        const float got = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give), 0xB1, 0xF, 0xF, true));
                        const float lo = odd ? got : mine0, hi = odd ? mine1 : got;
                        const int rsel = r + (odd ? 1 : 0);
                        const int o = 32 * mt + (rsel & 3) + 8 * (rsel >> 2) + 4 * g;
                        const int j = 32 * nt + (jj & ~1);
                        if (j0 + j < Wo)
                            __builtin_nontemporal_store(aadg_f2bf_pk(lo, hi), reinterpret_cast<uint32_t*>(yn + (size_t)o * Ho * Wo + j));
                    }
        }
    }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------
//     dW[o][t] = sum over images and output pixels of dY[o][pixel] * X[c][2i + kh - 3][2j + kw - 3],   t = (c, kh, kw)
// as an MFMA GEMM with M = 64 channels, N = 147 taps (5 tiles of 32) and K = output pixels.  The 8 K-values of a lane are
// 8 consecutive output columns j: contiguous in dY, but every second input column of X -- so the input patch is staged as
// two planes (even / odd input columns), in which the taps of consecutive j ARE consecutive; their start is only 2-byte
// aligned (it depends on the lane's kw), hence five 4-byte LDS reads + a per-lane byte funnel (v_alignbyte) per fragment.
// A workgroup owns 4 output rows x 64 columns x 32 output channels per tile (one row per wave; the two channel halves are
// separate tiles so that the 32 x 160 accumulator + fragments stay under 256 registers: two waves per SIMD), walks tiles
// persistently with the next tile's global loads in flight during the MFMAs, and ends with one LDS reduction over its waves
// + 32 x 147 float atomics.
constexpr int SW_ROWS = 4, SW_PX = 64, SW_CH = 32, SW_DPITCH = 72, SW_XROWS = 2 * SW_ROWS + 5, SW_XPITCH = 72, SW_NT = 5;
constexpr int SW_DY_ELEMS = SW_ROWS * SW_CH * SW_DPITCH, SW_X_ELEMS = 2 * 3 * SW_XROWS * SW_XPITCH;
constexpr int SW_RED_FLOATS = SW_CH * 32 * SW_NT;
constexpr int SW_DY_ITEMS = SW_ROWS * SW_CH * (SW_PX / 8) / 256, SW_X_CHUNKS = 3 * SW_XROWS * 18, SW_X_ITEMS = (SW_X_CHUNKS + 255) / 256;

// X3 = true ("f32x3"): x and dy float32; both are split into (hi, lo) bfloat16 planes on their way to LDS, the fragments are built per
// plane, and every product is hi*hi + hi*lo + lo*hi with float32 accumulation.
template <typename TIN, bool X3>
__global__ __launch_bounds__(256, 2) void k_stem7x7_wgrad(const TIN* __restrict__ x, const void* __restrict__ dy_,
                                                       float* __restrict__ dw, int H, int W, int tiles_x, int tiles_y, int total_tiles) {
    constexpr int PL = X3 ? 2 : 1;
    constexpr int OP_BYTES = PL * (SW_DY_ELEMS + SW_X_ELEMS) * 2;
    constexpr int LDS_BYTES = OP_BYTES > SW_RED_FLOATS * 4 ? OP_BYTES : SW_RED_FLOATS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    uint16_t* dyL = reinterpret_cast<uint16_t*>(lds_raw);                 // [plane][row][channel][SW_DPITCH]
    uint16_t* xL = dyL + PL * SW_DY_ELEMS;                                // [plane][parity][c][patch row][SW_XPITCH], index = m + 2
    const uint16_t* dy = reinterpret_cast<const uint16_t*>(dy_);
    const float* dyf = reinterpret_cast<const float*>(dy_);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = lane >> 5;
    const int Ho = H / 2, Wo = W / 2;
    // this lane's tap per N tile: patch offset (elements) of K-value 0 of K-step 0 for output row 0
    int toff[SW_NT];
#pragma unroll
    for (int nt = 0; nt < SW_NT; ++nt) {
        const int t = min(32 * nt + (lane & 31), 146);                    // taps >= 147 are never stored: any finite data
        const int c = t / 49, rem = t - c * 49, kh = rem / 7, kw = rem - kh * 7;
        const int par = (kw & 1) ? 0 : 1;                                 // odd kw <-> even input column
        toff[nt] = ((par * 3 + c) * SW_XROWS + kh + 2 * wv) * SW_XPITCH + ((kw + 1) >> 1) + 2 + 8 * g;
    }
    f32x16 d[SW_NT];
#pragma unroll
    for (int nt = 0; nt < SW_NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[nt][r] = 0.0f;

    // tile = (image, row block, column block, channel half); the staged chunks of a tile live in registers until the LDS is free
    uint4 rdy[PL * SW_DY_ITEMS], rx[PL * SW_X_ITEMS];            // X3: [2 k] / [2 k + 1] = pixels 0..3 / 4..7 of chunk k as float32
#define AADG_SW_ISSUE(tile_)                                                                                               \
    do {                                                                                                                   \
        const int half_ = (tile_) & 1, t1_ = (tile_) >> 1;                                                                 \
        const int tx_ = t1_ % tiles_x, t2_ = t1_ / tiles_x;                                                                \
        const int ty_ = t2_ % tiles_y, n_ = t2_ / tiles_y;                                                                 \
        const int i0_ = ty_ * SW_ROWS, j0_ = tx_ * SW_PX;                                                                  \
        const TIN* xn_ = x + (size_t)n_ * 3 * H * W;                                                                  \
        const size_t dyo_ = ((size_t)n_ * ST_CO + half_ * SW_CH) * Ho * Wo;                                               \
        _Pragma("unroll") for (int k_ = 0; k_ < SW_DY_ITEMS; ++k_) {                                                       \
            const int it_ = tid + 256 * k_, q_ = it_ & 7, o_ = (it_ >> 3) & (SW_CH - 1), r_ = it_ >> 8;                    \
            const int i_ = i0_ + r_, j_ = j0_ + 8 * q_;                                                                    \
            const bool ok_ = i_ < Ho && j_ < Wo;                                                                           \
            const size_t e_ = dyo_ + ((size_t)o_ * Ho + i_) * Wo + j_;                                                     \
            if (X3) {                                                                                                      \
                rdy[PL * k_] = ok_ ? *reinterpret_cast<const uint4*>(dyf + e_) : make_uint4(0u, 0u, 0u, 0u);               \
                rdy[PL * k_ + PL - 1] = ok_ ? *reinterpret_cast<const uint4*>(dyf + e_ + 4) : make_uint4(0u, 0u, 0u, 0u);  \
            } else {                                                                                                       \
                rdy[k_] = ok_ ? *reinterpret_cast<const uint4*>(dy + e_) : make_uint4(0u, 0u, 0u, 0u);                     \
            }                                                                                                              \
        }                                                                                                                  \
        _Pragma("unroll") for (int k_ = 0; k_ < SW_X_ITEMS; ++k_) {                                                        \
            const int it_ = tid + 256 * k_, q_ = it_ % 18, rc_ = it_ / 18;                                                 \
            const int pr_ = rc_ % SW_XROWS, c_ = rc_ / SW_XROWS;                                                           \
            const int row_ = 2 * i0_ - 3 + pr_, col_ = 2 * j0_ - 8 + 8 * q_;                                               \
            const bool okx_ = it_ < SW_X_CHUNKS && row_ >= 0 && row_ < H && col_ >= 0 && col_ < W;                         \
            if (X3) {                                                                                                      \
                const float* sx_ = reinterpret_cast<const float*>(xn_) + ((size_t)c_ * H + row_) * W + col_;               \
                rx[PL * k_] = okx_ ? *reinterpret_cast<const uint4*>(sx_) : make_uint4(0u, 0u, 0u, 0u);                    \
                rx[PL * k_ + PL - 1] = okx_ ? *reinterpret_cast<const uint4*>(sx_ + 4) : make_uint4(0u, 0u, 0u, 0u);       \
            } else {                                                                                                       \
                rx[k_] = make_uint4(0u, 0u, 0u, 0u);                                                                       \
                if (okx_) rx[k_] = load8_bf16<TIN>(xn_ + ((size_t)c_ * H + row_) * W + col_);                              \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)

    int tile = blockIdx.x;
    if (tile < total_tiles) AADG_SW_ISSUE(tile);
    for (; tile < total_tiles; tile += gridDim.x) {
        const int i0 = (((tile >> 1) / tiles_x) % tiles_y) * SW_ROWS;
        __syncthreads();                                                  // the previous tile's readers are done with the LDS
        // a chunk's two float4 -> its (hi, lo) bfloat16 halves, 8 pixels each
        auto halves = [](uint4 q0, uint4 q1, uint4& hi, uint4& lo) {
            uint2 h0, l0, h1, l1;
            aadg_split4(make_float4(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w)), h0, l0);
            aadg_split4(make_float4(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z), __uint_as_float(q1.w)), h1, l1);
            hi = make_uint4(h0.x, h0.y, h1.x, h1.y);
            lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
        };
#pragma unroll
        for (int k = 0; k < SW_DY_ITEMS; ++k) {
            const int it = tid + 256 * k, q = it & 7, o = (it >> 3) & (SW_CH - 1), r = it >> 8;
            uint4 vh = rdy[PL * k], vl = make_uint4(0u, 0u, 0u, 0u);
            if (X3) halves(rdy[PL * k], rdy[PL * k + PL - 1], vh, vl);
            *reinterpret_cast<uint4*>(dyL + (r * SW_CH + o) * SW_DPITCH + 8 * q) = vh;
            if (X3) *reinterpret_cast<uint4*>(dyL + SW_DY_ELEMS + (r * SW_CH + o) * SW_DPITCH + 8 * q) = vl;
        }
        // input patch rows 2*i0 - 3 .. 2*i0 + 9, columns 2*j0 - 8 .. 2*j0 + 135, split into even / odd column planes:
        // chunk q (8 columns) -> plane indices 4q .. 4q + 3 of both planes
#pragma unroll
        for (int k = 0; k < SW_X_ITEMS; ++k) {
            const int it = tid + 256 * k;
            if (it < SW_X_CHUNKS) {
                const int q = it % 18, rc = it / 18;
                const int pr = rc % SW_XROWS, c = rc / SW_XROWS;
                uint4 vv[PL];
                vv[0] = rx[PL * k];
                if (X3) halves(rx[PL * k], rx[PL * k + PL - 1], vv[0], vv[PL - 1]);
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) {
                    const uint4 v = vv[pl];
                    const uint2 ev = make_uint2((v.x & 0xFFFFu) | (v.y << 16), (v.z & 0xFFFFu) | (v.w << 16));
                    const uint2 od = make_uint2((v.x >> 16) | (v.y & 0xFFFF0000u), (v.z >> 16) | (v.w & 0xFFFF0000u));
                    *reinterpret_cast<uint2*>(xL + pl * SW_X_ELEMS + ((0 * 3 + c) * SW_XROWS + pr) * SW_XPITCH + 4 * q) = ev;
                    *reinterpret_cast<uint2*>(xL + pl * SW_X_ELEMS + ((1 * 3 + c) * SW_XROWS + pr) * SW_XPITCH + 4 * q) = od;
                }
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < total_tiles) AADG_SW_ISSUE(tile + (int)gridDim.x);     // in flight during the MFMAs below
        if (i0 + wv < Ho) {
            const uint16_t* A = dyL + (wv * SW_CH + (lane & 31)) * SW_DPITCH + 8 * g;
#pragma unroll
            for (int ks = 0; ks < SW_PX / 16; ++ks) {
                bf16x8 a[PL];
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) a[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(A + pl * SW_DY_ELEMS + 16 * ks));
#pragma unroll
                for (int nt = 0; nt < SW_NT; ++nt) {
                    const int s = toff[nt] + 16 * ks;                                // first element of this lane's 8 K-values
                    const uint32_t sh = (uint32_t)(s & 1) * 2u;                      // byte shift inside the first dword
                    bf16x8 b[PL];
#pragma unroll
                    for (int pl = 0; pl < PL; ++pl) {
                        const uint32_t* p = reinterpret_cast<const uint32_t*>(xL + pl * SW_X_ELEMS) + (s >> 1);
                        const uint32_t w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4];
                        b[pl] = __builtin_bit_cast(bf16x8, make_uint4(__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
                                                                      __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w4, w3, sh)));
                    }
                    if (X3) {
                        d[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1], b[0], d[nt], 0, 0, 0);
                        d[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[PL - 1], d[nt], 0, 0, 0);
                    }
                    d[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], d[nt], 0, 0, 0);
                }
            }
        }
    }
#undef AADG_SW_ISSUE
    // reduce the four waves in LDS, then one float atomic per weight and workgroup.  The channel half is the same for every
    // tile of a workgroup (the grid is even, tiles alternate halves)
    const int half = blockIdx.x & 1;
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds_raw);                       // [32][160]
    for (int i = tid; i < SW_RED_FLOATS; i += 256) red[i] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < SW_NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = (r & 3) + 8 * (r >> 2) + 4 * g, t = 32 * nt + (lane & 31);
            __hip_atomic_fetch_add(&red[o * (32 * SW_NT) + t], d[nt][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    __syncthreads();
    for (int i = tid; i < SW_CH * 147; i += 256) {
        const int o = i / 147, t = i - o * 147;
        unsafeAtomicAdd(dw + (size_t)(half * SW_CH + o) * 147 + t, red[o * (32 * SW_NT) + t]);
    }
}

}  // namespace

extern "C" int aadg_stem_conv7x7_supported(int H, int W) {
    return H >= 2 && W >= 16 && (H % 2) == 0 && (W % 16) == 0 ? 1 : 0;      // output width a multiple of 8: aligned 16-byte patch chunks
}

extern "C" size_t aadg_stem_conv7x7_workspace_bytes(void) { return (size_t)2 * ST_KS * 2 * 64 * sizeof(uint4); }   // hi + lo weight fragments

/* y [N, 64, H/2, W/2] (bfloat16) = conv2d(bfloat16(x [N, 3, H, W]), weight [64, 3, 7, 7] (float32), stride 2, padding 3);
 * x_dtype 0: x is float32 and is rounded to bfloat16 on load, 1: x is bfloat16 */
extern "C" int aadg_stem_conv7x7_bf16(const void* x, int x_dtype, const float* weight, void* y, int N, int H, int W, void* ws,
                                      size_t ws_bytes, void* stream) {
    if (x == nullptr || weight == nullptr || y == nullptr || ws == nullptr || N <= 0 || (x_dtype != 0 && x_dtype != 1)) return AADG_E_BADARG;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)ws) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_stem_conv7x7_supported(H, W)) return AADG_E_UNSUPPORTED;
    if (ws_bytes < aadg_stem_conv7x7_workspace_bytes()) return AADG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int Ho = H / 2, Wo = W / 2;
    const int tiles_x = (Wo + ST_TC - 1) / ST_TC, tiles_y = (Ho + ST_TR - 1) / ST_TR;
    const long long total = (long long)N * tiles_x * tiles_y;
    if (total > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_stem_pack, dim3((ST_KS * 2 * 64 + 255) / 256), dim3(256), 0, st, weight, reinterpret_cast<uint4*>(ws));
    AADG_LAUNCH_CHECK();
    const int grid = (int)(total < 512 ? total : 512);              // persistent: the 2 workgroups a CU holds (241 registers per lane), weight fragments loaded once each
    if (x_dtype == 0)
        hipLaunchKernelGGL((k_stem7x7<float, false>), dim3(grid), dim3(256), 0, st, (const float*)x, (const uint4*)ws, y, H, W, tiles_x,
                           tiles_y, (int)total);
    else
        hipLaunchKernelGGL((k_stem7x7<uint16_t, false>), dim3(grid), dim3(256), 0, st, (const uint16_t*)x, (const uint4*)ws, y, H, W,
                           tiles_x, tiles_y, (int)total);
    AADG_LAUNCH_CHECK();
    return 0;
}

/* The same convolution at float32 precision ("f32x3", see aadg_conv1x1_nchw_f32x3): x [N, 3, H, W] and y [N, 64, H/2, W/2] float32 */
extern "C" int aadg_stem_conv7x7_f32x3(const float* x, const float* weight, float* y, int N, int H, int W, void* ws, size_t ws_bytes,
                                       void* stream) {
    if (x == nullptr || weight == nullptr || y == nullptr || ws == nullptr || N <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)ws) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_stem_conv7x7_supported(H, W)) return AADG_E_UNSUPPORTED;
    if (ws_bytes < aadg_stem_conv7x7_workspace_bytes()) return AADG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int Ho = H / 2, Wo = W / 2;
    const int tiles_x = (Wo + ST_TC - 1) / ST_TC, tiles_y = (Ho + ST_TR - 1) / ST_TR;
    const long long total = (long long)N * tiles_x * tiles_y;
    if (total > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_stem_pack, dim3((ST_KS * 2 * 64 + 255) / 256), dim3(256), 0, st, weight, reinterpret_cast<uint4*>(ws));
    AADG_LAUNCH_CHECK();
    const int grid = (int)(total < 512 ? total : 512);
    hipLaunchKernelGGL((k_stem7x7<float, true>), dim3(grid), dim3(256), 0, st, x, (const uint4*)ws, (void*)y, H, W, tiles_x, tiles_y,
                       (int)total);
    AADG_LAUNCH_CHECK();
    return 0;
}

/* dweight [64, 3, 7, 7] (float32, overwritten) = gradient of aadg_stem_conv7x7_bf16 w.r.t. its weight, from x [N, 3, H, W]
 * (x_dtype as in the forward) and dy [N, 64, H/2, W/2] (bfloat16); float32 accumulation, partial sums combined with float atomics. */
extern "C" int aadg_stem_conv7x7_wgrad_bf16(const void* x, int x_dtype, const void* dy, float* dweight, int N, int H, int W, void* stream) {
    if (x == nullptr || dy == nullptr || dweight == nullptr || N <= 0 || (x_dtype != 0 && x_dtype != 1)) return AADG_E_BADARG;
    if ((((uintptr_t)x | (uintptr_t)dy) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_stem_conv7x7_supported(H, W)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int Ho = H / 2, Wo = W / 2;
    const int tiles_x = (Wo + SW_PX - 1) / SW_PX, tiles_y = (Ho + SW_ROWS - 1) / SW_ROWS;
    const long long total = (long long)N * tiles_x * tiles_y * 2;          // x 2 channel halves (tile parity)
    if (total > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    AADG_HIP_TRY(hipMemsetAsync(dweight, 0, (size_t)ST_CO * 147 * sizeof(float), st));
    const int grid = (int)(total < 1024 ? total : 1024);                   // even: a workgroup keeps one channel half
    if (x_dtype == 0)
        hipLaunchKernelGGL((k_stem7x7_wgrad<float, false>), dim3(grid), dim3(256), 0, st, (const float*)x, dy, dweight, H, W,
                           tiles_x, tiles_y, (int)total);
    else
        hipLaunchKernelGGL((k_stem7x7_wgrad<uint16_t, false>), dim3(grid), dim3(256), 0, st, (const uint16_t*)x, dy, dweight, H, W,
                           tiles_x, tiles_y, (int)total);
    AADG_LAUNCH_CHECK();
    return 0;
}

/* ... and from float32 x [N, 3, H, W] / dy [N, 64, H/2, W/2] at float32 precision ("f32x3") */
extern "C" int aadg_stem_conv7x7_wgrad_f32x3(const float* x, const float* dy, float* dweight, int N, int H, int W, void* stream) {
    if (x == nullptr || dy == nullptr || dweight == nullptr || N <= 0) return AADG_E_BADARG;
    if ((((uintptr_t)x | (uintptr_t)dy) & 15u) != 0) return AADG_E_BADARG;
    if (!aadg_stem_conv7x7_supported(H, W)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int Ho = H / 2, Wo = W / 2;
    const int tiles_x = (Wo + SW_PX - 1) / SW_PX, tiles_y = (Ho + SW_ROWS - 1) / SW_ROWS;
    const long long total = (long long)N * tiles_x * tiles_y * 2;
    if (total > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    AADG_HIP_TRY(hipMemsetAsync(dweight, 0, (size_t)ST_CO * 147 * sizeof(float), st));
    const int grid = (int)(total < 1024 ? total : 1024);
    hipLaunchKernelGGL((k_stem7x7_wgrad<float, true>), dim3(grid), dim3(256), 0, st, x, (const void*)dy, dweight, H, W, tiles_x, tiles_y,
                       (int)total);
    AADG_LAUNCH_CHECK();
    return 0;
}

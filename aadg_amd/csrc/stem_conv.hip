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
    wfrag[t] = make_uint4(aadg_f2bf_pk(v[0], v[1]), aadg_f2bf_pk(v[2], v[3]), aadg_f2bf_pk(v[4], v[5]), aadg_f2bf_pk(v[6], v[7]));
}

__global__ __launch_bounds__(256) void k_stem7x7(const uint16_t* __restrict__ x, const uint4* __restrict__ wfrag,
                                                 uint16_t* __restrict__ y, int H, int W, int tiles_x, int tiles_y, int total_tiles) {
    __shared__ __attribute__((aligned(16))) uint16_t P[3 * ST_PR * ST_PC];
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
        const uint16_t* xn = x + (size_t)n * 3 * H * W;
        __syncthreads();                                           // the previous tile's readers are done with P
        // patch: input rows 2*i0 - 3 .. 2*i0 + 17, columns 2*j0 - 8 .. 2*j0 + 135 (18 chunks of 8), zero outside the image
        for (int it = tid; it < 3 * ST_PR * (ST_PC / 8); it += 256) {
            const int q = it % (ST_PC / 8), rc = it / (ST_PC / 8);
            const int pr = rc % ST_PR, c = rc / ST_PR;
            const int row = 2 * i0 - 3 + pr, col = 2 * j0 - 8 + 8 * q;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (row >= 0 && row < H && col >= 0 && col < W) v = *reinterpret_cast<const uint4*>(xn + ((size_t)c * H + row) * W + col);
            *reinterpret_cast<uint4*>(P + (c * ST_PR + pr) * ST_PC + 8 * q) = v;
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
                bf16x8 b[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const uint32_t* p = Pw + (koff[ks] >> 1) + 32 * nt;
                    b[nt] = __builtin_bit_cast(bf16x8, make_uint4(p[0], p[1], p[2], p[3]));
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) d[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][ks], b[nt], d[mt][nt], 0, 0, 0);
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
                            *reinterpret_cast<uint32_t*>(yn + (size_t)o * Ho * Wo + j) = aadg_f2bf_pk(lo, hi);
                    }
        }
    }
}

}  // namespace

extern "C" int aadg_stem_conv7x7_supported(int H, int W) {
    return H >= 2 && W >= 16 && (H % 2) == 0 && (W % 16) == 0 ? 1 : 0;      // output width a multiple of 8: aligned 16-byte patch chunks
}

extern "C" size_t aadg_stem_conv7x7_workspace_bytes(void) { return (size_t)ST_KS * 2 * 64 * sizeof(uint4); }

/* y [N, 64, H/2, W/2] (bfloat16) = conv2d(x [N, 3, H, W] (bfloat16), weight [64, 3, 7, 7] (float32), stride 2, padding 3) */
extern "C" int aadg_stem_conv7x7_bf16(const void* x, const float* weight, void* y, int N, int H, int W, void* ws, size_t ws_bytes,
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
    const int grid = (int)(total < 512 ? total : 512);              // persistent: the 2 workgroups a CU holds (241 registers per lane), weight fragments loaded once each
    hipLaunchKernelGGL(k_stem7x7, dim3(grid), dim3(256), 0, st, (const uint16_t*)x, (const uint4*)ws, (uint16_t*)y, H, W, tiles_x,
                       tiles_y, (int)total);
    AADG_LAUNCH_CHECK();
    return 0;
}

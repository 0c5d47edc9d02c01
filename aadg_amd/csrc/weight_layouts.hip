// Per-step re-layout of the float32 master weights of the own convolutions, all layers in ONE launch:
//     plain [Co][Ci][T]   bfloat16 cast (library fallbacks, shape reference)
//     fwd   [T][Co][Ci]   tap-major: the A operand of k_conv3x3_nchw / k_conv3x3_s2 (and, T = 1, of k_conv1x1_nchw = plain)
//     bwd   [T'][Ci][Co]  tap-major and transposed: the A operand of the same kernels run on dY (input gradient); T' = T - 1 - t
//                         (mirrored taps) for the stride-1 3x3 convolution, T' = t for k_dgrad3x3_s2 and for 1x1
// Before: one `weight.to(bfloat16)` multi-tensor copy plus ~70 `permute(...).contiguous()` launches of 5-7 us per step (one per
// own convolution and direction), a cost that does not shrink with the per-rank batch.
// Workgroup = one 32 x 32 (Co x Ci) tile of one layer with all its taps (32 x 256 for 1x1 weights), through LDS so that all three
// outputs are written in 64-byte runs.
// SPLIT (the "f32x3" convolutions): every output holds TWO bfloat16 planes -- hi = bf16(w) at the buffer's start and lo = bf16(w - hi)
// Co * Ci * T elements behind it -- the pre-split A operands of the float32-precision kernels.
#include "common.h"

namespace {

constexpr int WL_T = 32, WL_MAXT = 9, WL_C1 = 256;

// T, NO, NC > 0: compile-time taps / tile extents (full 32 x 32 tiles of 1x1 and 3x3 weights: index arithmetic by constants);
// 0: run-time values (edge tiles, other tap counts)
template <int TT, int FULL, bool SPLIT>
__device__ __forceinline__ void wl_tile(const aadg_wl_item& it, int o0, int c0, uint32_t (*L)[WL_T * WL_MAXT + 2]) {
    const int Co = it.Co, Ci = it.Ci, T = TT > 0 ? TT : it.taps;
    const int ncmax = T == 1 ? WL_C1 : WL_T;                   // in channels per tile: 256 for 1x1 weights (the LDS row holds 288 values)
    const int no = FULL ? WL_T : min(WL_T, Co - o0), nc = FULL ? (TT == 1 ? WL_C1 : WL_T) : min(ncmax, Ci - c0);
    const int run = nc * T;                                    // contiguous source elements per out channel of the tile
    const float* w = (const float*)it.w;
    uint16_t* plain = (uint16_t*)it.plain;
    const size_t plane = (size_t)Co * Ci * T;                  // SPLIT: the lo plane of every output lies one plane behind its hi plane
    for (int e = threadIdx.x; e < no * run; e += 256) {
        const int o = e / run, r = e - o * run;
        const size_t g = ((size_t)(o0 + o) * Ci + c0) * T + r;
        uint16_t v, vl = 0;
        if (SPLIT) aadg_split1(w[g], v, vl); else v = (uint16_t)aadg_f2bf_bits(w[g]);
        L[o][r] = (uint32_t)v | ((uint32_t)vl << 16);
        if (plain != nullptr) {
            plain[g] = v;
            if (SPLIT) plain[plane + g] = vl;
        }
    }
    __syncthreads();
    uint16_t* fwd = (uint16_t*)it.fwd;
    uint16_t* bwd = (uint16_t*)it.bwd;
    const int per_tap = no * nc;
    if (fwd != nullptr)
        for (int e = threadIdx.x; e < T * per_tap; e += 256) {
            const int t = e / per_tap, r = e - t * per_tap, o = r / nc, c = r - o * nc;
            const uint32_t v = L[o][c * T + t];
            const size_t g = ((size_t)t * Co + o0 + o) * Ci + c0 + c;
            fwd[g] = (uint16_t)v;
            if (SPLIT) fwd[plane + g] = (uint16_t)(v >> 16);
        }
    if (bwd != nullptr)
        for (int e = threadIdx.x; e < T * per_tap; e += 256) {
            const int t = e / per_tap, r = e - t * per_tap, c = r / no, o = r - c * no;
            const int tt = it.flip ? T - 1 - t : t;
            const uint32_t v = L[o][c * T + t];
            const size_t g = ((size_t)tt * Ci + c0 + c) * Co + o0 + o;
            bwd[g] = (uint16_t)v;
            if (SPLIT) bwd[plane + g] = (uint16_t)(v >> 16);
        }
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void k_weight_layouts(const aadg_wl_item* __restrict__ items, const int32_t* __restrict__ tiles) {
    __shared__ uint32_t L[WL_T][WL_T * WL_MAXT + 2];
    const int item = tiles[3 * blockIdx.x], o0 = tiles[3 * blockIdx.x + 1], c0 = tiles[3 * blockIdx.x + 2];
    const aadg_wl_item it = items[item];
    const bool full = it.Co - o0 >= WL_T && it.Ci - c0 >= (it.taps == 1 ? WL_C1 : WL_T);
    if (full && it.taps == 9) wl_tile<9, 1, SPLIT>(it, o0, c0, L);
    else if (full && it.taps == 1) wl_tile<1, 1, SPLIT>(it, o0, c0, L);
    else wl_tile<0, 0, SPLIT>(it, o0, c0, L);
}

}  // namespace

/* items: DEVICE array of aadg_wl_item; tiles: DEVICE int32 [n_tiles][3] = (item, first out channel, first in channel) of every
 * tile of every item: 32 out channels x 32 in channels, 32 x 256 for taps == 1 (the caller enumerates them once).  taps <= 9. */
extern "C" int aadg_weight_layouts_bf16(const aadg_wl_item* items, const int32_t* tiles, int n_tiles, void* stream) {
    if (items == nullptr || tiles == nullptr || n_tiles < 0) return AADG_E_BADARG;
    if (n_tiles == 0) return 0;
    hipLaunchKernelGGL(k_weight_layouts<false>, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, items, tiles);
    AADG_LAUNCH_CHECK();
    return 0;
}

/* The same, with every output buffer holding two bfloat16 planes of Co * Ci * taps elements each: hi = bf16(w), then lo = bf16(w - hi) */
extern "C" int aadg_weight_layouts_split_bf16(const aadg_wl_item* items, const int32_t* tiles, int n_tiles, void* stream) {
    if (items == nullptr || tiles == nullptr || n_tiles < 0) return AADG_E_BADARG;
    if (n_tiles == 0) return 0;
    hipLaunchKernelGGL(k_weight_layouts<true>, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, items, tiles);
    AADG_LAUNCH_CHECK();
    return 0;
}

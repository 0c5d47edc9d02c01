// Live uint8 augmentation path on gfx950 (a2-a7; kernels K6/K7 + the a3 ops).
//
// Replaces, for a whole batch of (sample, policy) units at once, the reference's DataLoader-worker
// chain  Policy ops (data/basic.py:70-167 via Pillow) -> DGRandomScaleCrop (data/transform.py:97-135)
// -> Normalize_dg (:149-172) -> ToTensor (:217-236) -> train_dg_collate_fn (:323-340).
//
// All arithmetic is integer / Pillow fixed point, restated from the behaviour of Pillow's C core
// (Resample.c, Blend.c, Filter.c, ImageOps.py) and bit-exact with it; the only floating point is
//   * the resampling coefficients (double, evaluated in Pillow's operation order; this file is
//     compiled with -ffp-contract=off so no product is fused),
//   * Image.blend's float32 expression, and
//   * the final u8/127.5-1 (a correctly rounded float32 division, tabulated per block).
//
// Two data flows share the statistics / LUT / table kernels:
//
//  FUSED (units that are not down-scaled, <= 2 Sharpness ops, 4-aligned sizes -- every unit of the optic
//  configs): nothing but the final tensors is written to HBM.
//   k_hist        stage 0: per-channel histograms + sum of L of the raw source image
//   k_hist_fused  stage k>=1: rebuilds the image after k ops tile by tile in LDS and histograms it
//   k_lut         the 3x256 byte LUT of every LUT-class op (7 of the 10 ops), kept per stage
//   k_tables      Pillow BILINEAR coefficient tables + NEAREST index tables for the crop window
//   k_fused       per 256x16 output tile: source patch (+halo) -> LDS as RGBX words (12-byte vector loads),
//                 the whole op chain applied in LDS (LUT / Color / Cutout in place, Sharpness ping-pong),
//                 horizontal fixed-point pass -> LDS, vertical pass from LDS (ds_read_b128), pad, /127.5-1 via
//                 an LDS table, CHW float4 stores, mask gather -> multilabel float4 stores.
//
//  STAGED (everything else; also the eager single-op entry point): u8 intermediates in the workspace.
//   k_hist / k_lut / k_apply per op stage, then k_tables and k_final (gathers from global memory).
#include <stddef.h>
#include <stdlib.h>

#include <vector>

#include "common.h"

namespace {

// the 5 float32 output planes of a unit are written once and read next by the backbone, 0.9 GB later: streaming (non-temporal)
// stores keep them from displacing the source patches other tiles of the unit are about to read (k_fused3: 228 -> 186 us per
// 168-unit launch, the whole call 292 -> 250 us)
// ... when the batch's output is larger than a cache can hold until the backbone reads it (a per-rank batch of 21 units, 110 MB,
// is not: streaming it cost the tile kernel and the stem convolution behind it ~3 %).  The flag rides in the `dataset` argument.
constexpr int AUG_STREAM_OUT = 0x100;
constexpr size_t AUG_STREAM_BYTES = (size_t)128 << 20;
__device__ __forceinline__ void store_out4(float* p, float4 v, bool stream) { aadg_store_out(p, v, stream); }


constexpr int KMAX = 8;           // max taps per output pixel (scale factor >= 1/3)
constexpr int HIST_STRIDE = AADG_HIST_STRIDE;  // 768 bins + u64 L-sum + pad (u32 words)
constexpr int TAB_STRIDE = 2 * KMAX + 4;  // ints per crop position: xmin,xk[KMAX],ymin,yk[KMAX],xnn,ynn
constexpr int PRECISION_BITS = 22;

constexpr int FT_W = 256;          // fused tile: output columns (one float4 per lane and row)
constexpr int FT_H = 16;           // fused tile: output rows
constexpr int PATCH_CAP = 6144;    // LDS pixels (u32) per patch buffer: 24 KiB, two buffers per block
constexpr int MAX_SHARP = 2;       // Sharpness ops a fused unit may chain (1-pixel halo each)

struct UnitRef {
    const aadg_unit* units;
    aadg_unit single;
    int use_single;
    int allow_fused;
};
__device__ __forceinline__ const aadg_unit& pick(const UnitRef& r, int u) {
    return r.use_single ? r.single : r.units[u];
}

__device__ __forceinline__ bool op_needs_stats(int op) {
    return op == AADG_OP_AUTOCONTRAST || op == AADG_OP_EQUALIZE || op == AADG_OP_CONTRAST;
}
__device__ __forceinline__ bool op_is_lut(int op) {
    return op <= AADG_OP_CONTRAST || op == AADG_OP_BRIGHTNESS;  // 0..5 and 7
}

// Statistics by push-forward (round 2): when every op before op k is a per-channel byte map (op_is_lut) and op k needs only the
// per-channel histograms (AutoContrast, Equalize -- Contrast needs the mean of L, which mixes the channels), the histogram of the image
// after k ops is the push-forward of the RAW image's histogram through the composed maps: hist_k[c][L_{k-1}(...L_0(v))] += hist_0[c][v].
// k_lut does that from the stage-0 histogram and the earlier stages' LUTs -- no pixel pass (k_hist_fused) for such a stage.  Fused-flow
// units only (the staged flow materialises its intermediate images and histograms them directly).  Mirrored by launch_hints() on the host.
// (the helpers below read op[0..3] / farg[0..3] at CONSTANT indices and select: a run-time index made every call a chain of dependent
//  scalar loads, each with its own wait, in front of the first pixel load of a workgroup)
__device__ __forceinline__ int unit_op_sel(const aadg_unit& un, int j) {
    const int a = un.op[0], b = un.op[1], c = un.op[2], d = un.op[3];
    return j == 0 ? a : (j == 1 ? b : (j == 2 ? c : d));
}
__device__ __forceinline__ bool stats_by_pushforward(const aadg_unit& un, int k, bool fused_flow) {
    if (!fused_flow || k < 1 || k >= un.n_ops) return false;
    const int opk = unit_op_sel(un, k);
    bool all_lut = true;
#pragma unroll
    for (int j = 0; j < AADG_MAX_OPS; ++j) all_lut = all_lut && (j >= k || op_is_lut(un.op[j]));
    return (opk == AADG_OP_AUTOCONTRAST || opk == AADG_OP_EQUALIZE) && all_lut;
}
__device__ __forceinline__ bool any_pushforward(const aadg_unit& un, bool fused_flow) {
    bool any = false;
#pragma unroll
    for (int k = 1; k < AADG_MAX_OPS; ++k) any = any || stats_by_pushforward(un, k, fused_flow);
    return any;
}

// does this unit take the fused (LDS-resident) data flow?  Must agree across all kernels of a call.
__device__ __forceinline__ int sharp_count(const aadg_unit& un, int upto) {
    int s = 0;
#pragma unroll
    for (int k = 0; k < AADG_MAX_OPS; ++k) s += (k < upto && un.op[k] == AADG_OP_SHARPNESS && un.farg[k] != 1.0f) ? 1 : 0;
    return s;
}
// number of Sharpness stencils of a unit, all four op slots read unconditionally (independent scalar loads, one wait)
__device__ __forceinline__ int sharp_count4(const aadg_unit& un, int n_ops) {
    int s = 0;
#pragma unroll
    for (int k = 0; k < AADG_MAX_OPS; ++k) s += (k < n_ops && un.op[k] == AADG_OP_SHARPNESS && un.farg[k] != 1.0f) ? 1 : 0;
    return s;
}
// data flow of a unit: 0 = staged, 1 = fused "UP" tile (no down-scaling: <= 2 taps per axis), 2 = fused
// "GENERIC" tile (down-scaling by at most 2x on either axis: <= 5 taps)
enum { FLOW_STAGED = 0, FLOW_UP = 1, FLOW_GENERIC = 2 };
__device__ __forceinline__ int unit_flow(bool allow, const aadg_unit& un, int Hs, int Ws, int crop) {
    if (!allow || (Ws & 3) || (crop & 3)) return FLOW_STAGED;
    if (sharp_count(un, un.n_ops) > MAX_SHARP) return FLOW_STAGED;
    if (un.scaled_w >= Ws && un.scaled_h >= Hs) return FLOW_UP;
    if (2 * un.scaled_w >= Ws && 2 * un.scaled_h >= Hs) return FLOW_GENERIC;
    return FLOW_STAGED;
}
__device__ __forceinline__ bool unit_fusable(bool allow, const aadg_unit& un, int Hs, int Ws, int crop) {
    return unit_flow(allow, un, Hs, Ws, crop) != FLOW_STAGED;
}
__device__ __forceinline__ bool unit_fusable(const UnitRef& ur, const aadg_unit& un, int Hs, int Ws, int crop) {
    return unit_fusable(ur.allow_fused != 0, un, Hs, Ws, crop);
}

__device__ __forceinline__ uint32_t rgb2l(uint32_t r, uint32_t g, uint32_t b) {
    return (19595u * r + 38470u * g + 7471u * b + 0x8000u) >> 16;
}

// Image.blend(degenerate, image, alpha) for one byte; float32, unfused (Blend.c)
__device__ __forceinline__ uint32_t blend_px(int deg, int img, float alpha, bool /*interp*/) {
    // Pillow: 0 <= alpha <= 1 truncates t, otherwise t <= 0 -> 0, t >= 255 -> 255, else truncate.  One clamp serves both:
    // for alpha in [0, 1] the float32 t already lies between deg and img (rounding is monotonic), so it is a no-op there.
    const float t = __fadd_rn((float)deg, __fmul_rn(alpha, (float)(img - deg)));
    return (uint32_t)(int)fminf(fmaxf(t, 0.0f), 255.0f);
}

struct Bufs {
    const uint8_t* pool;
    uint8_t* buf0;
    uint8_t* buf1;
    size_t img_bytes;
};
// image a unit reads at stage k (k == n_ops: the finished image)
__device__ __forceinline__ const uint8_t* stage_input(const Bufs& b, const aadg_unit& un, int u, int k) {
    if (k == 0) return b.pool + (size_t)un.src * b.img_bytes;
    return ((k - 1) & 1 ? b.buf1 : b.buf0) + (size_t)u * b.img_bytes;
}

// ------------------------------------------------------------------------------------------------
// k_hist: grid (chunks, N), 256 threads.  Thread = groups of 4 pixels (12 bytes, 3 dword loads).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void hist_pixels(const uint8_t* __restrict__ in, int npix, int bx, int nbx, uint32_t* gh, uint32_t (*sh)[768]);
__device__ __forceinline__ void hist_body(const Bufs& bufs, const UnitRef& ur, const int* __restrict__ ulist, int stage, int npix, int Hs,
                                          int Ws, int crop, uint32_t* hist, int bx, int by, int nbx, uint32_t (*sh)[768]) {
    const int u = ulist != nullptr ? ulist[by] : by;      // ulist: the units whose op `stage` needs statistics
    const aadg_unit& un = pick(ur, u);
    const bool fused_flow = unit_fusable(ur, un, Hs, Ws, crop);
    if (stage == 0) {
        // the raw image's histogram: for a statistics op in slot 0, and as the source of later stages' push-forward
        if (!(un.n_ops > 0 && op_needs_stats(un.op[0])) && !any_pushforward(un, fused_flow)) return;
    } else {
        if (un.n_ops <= stage || !op_needs_stats(un.op[stage])) return;
        if (fused_flow) return;                                   // k_hist_fused / push-forward cover those
    }
    hist_pixels(stage_input(bufs, un, u, stage), npix, bx, nbx, hist + (size_t)u * HIST_STRIDE, sh);
}
// histogram + L-sum of chunk bx of nbx of one HWC image, added to gh[HIST_STRIDE]
__device__ __forceinline__ void hist_pixels(const uint8_t* __restrict__ in, int npix, int bx, int nbx, uint32_t* gh, uint32_t (*sh)[768]) {
    const int tid = threadIdx.x, wv = tid >> 6;
    for (int i = tid; i < 4 * 768; i += 256) (&sh[0][0])[i] = 0;
    __syncthreads();
    uint32_t* h = sh[wv];
    unsigned long long lsum = 0;
    const bool vec = ((npix & 3) == 0) && ((((uintptr_t)in) & 3) == 0);
    if (vec) {
        const int ngroups = npix >> 2;
        const uint32_t* p32 = reinterpret_cast<const uint32_t*>(in);
        for (int g = bx * 256 + tid; g < ngroups; g += nbx * 256) {
            uint32_t a = p32[3 * g], b = p32[3 * g + 1], c = p32[3 * g + 2];
            uint32_t r0 = a & 255, g0 = (a >> 8) & 255, b0 = (a >> 16) & 255, r1 = a >> 24;
            uint32_t g1 = b & 255, b1 = (b >> 8) & 255, r2 = (b >> 16) & 255, g2 = b >> 24;
            uint32_t b2 = c & 255, r3 = (c >> 8) & 255, g3 = (c >> 16) & 255, b3 = c >> 24;
            atomicAdd(&h[r0], 1u); atomicAdd(&h[r1], 1u); atomicAdd(&h[r2], 1u); atomicAdd(&h[r3], 1u);
            atomicAdd(&h[256 + g0], 1u); atomicAdd(&h[256 + g1], 1u); atomicAdd(&h[256 + g2], 1u); atomicAdd(&h[256 + g3], 1u);
            atomicAdd(&h[512 + b0], 1u); atomicAdd(&h[512 + b1], 1u); atomicAdd(&h[512 + b2], 1u); atomicAdd(&h[512 + b3], 1u);
            lsum += rgb2l(r0, g0, b0) + rgb2l(r1, g1, b1) + rgb2l(r2, g2, b2) + rgb2l(r3, g3, b3);
        }
    } else {
        for (int p = bx * 256 + tid; p < npix; p += nbx * 256) {
            uint32_t r = in[3 * (size_t)p], g = in[3 * (size_t)p + 1], b = in[3 * (size_t)p + 2];
            atomicAdd(&h[r], 1u); atomicAdd(&h[256 + g], 1u); atomicAdd(&h[512 + b], 1u);
            lsum += rgb2l(r, g, b);
        }
    }
    lsum = wave_sum(lsum);
    __syncthreads();
    for (int i = tid; i < 768; i += 256) {
        uint32_t v = sh[0][i] + sh[1][i] + sh[2][i] + sh[3][i];
        if (v) atomicAdd(&gh[i], v);
    }
    if ((tid & 63) == 0 && lsum) atomicAdd(reinterpret_cast<unsigned long long*>(gh + 768), lsum);
}
__global__ __launch_bounds__(256) void k_hist(Bufs bufs, UnitRef ur, const int* __restrict__ ulist, int stage, int npix, int Hs, int Ws,
                                              int crop, uint32_t* hist) {
    __shared__ uint32_t sh[4][768];
    hist_body(bufs, ur, ulist, stage, npix, Hs, Ws, crop, hist, blockIdx.x, blockIdx.y, gridDim.x, sh);
}

// grid (chunks, P): histograms of the pool images themselves (aadg_pool_histograms_u8)
__global__ __launch_bounds__(256) void k_pool_hist(const uint8_t* __restrict__ pool, size_t img_bytes, int npix, uint32_t* hist) {
    __shared__ uint32_t sh[4][768];
    hist_pixels(pool + (size_t)blockIdx.y * img_bytes, npix, blockIdx.x, gridDim.x, hist + (size_t)blockIdx.y * HIST_STRIDE, sh);
}

// ------------------------------------------------------------------------------------------------
// k_lut: grid N, 256 threads; thread i owns LUT entry i of each channel.
// ------------------------------------------------------------------------------------------------
// pool_hist != nullptr: the raw image's statistics come from the caller's per-pool-image cache (row un.src) instead of this call's
// stage-0 pixel pass (row u of hist0)
__device__ __forceinline__ void lut_body(const UnitRef& ur, int stage, int N, int u, int npix, int Hs, int Ws, int crop,
                                         const uint32_t* hist0, const uint32_t* hist, const uint32_t* pool_hist, uint8_t* lut) {
    const aadg_unit& un = pick(ur, u);
    if (un.n_ops <= stage) return;
    const int op = un.op[stage];
    if (!op_is_lut(op)) return;
    const int i = threadIdx.x;
    uint8_t* L = lut + ((size_t)stage * N + u) * 768;
    const uint32_t* raw = pool_hist != nullptr ? pool_hist + (size_t)un.src * HIST_STRIDE : hist0 + (size_t)u * HIST_STRIDE;
    const uint32_t* gh = stage == 0 ? raw : hist + (size_t)u * HIST_STRIDE;      // this stage's pixel-pass histogram
    __shared__ __attribute__((aligned(16))) uint32_t pushed[768];
    if (stats_by_pushforward(un, stage, unit_fusable(ur, un, Hs, Ws, crop))) {
        // histogram after `stage` per-channel maps = the raw histogram pushed through them
        for (int t = i; t < 768; t += 256) pushed[t] = 0u;
        __syncthreads();
        for (int c = 0; c < 3; ++c) {
            const uint32_t cnt = raw[256 * c + i];
            int v = i;
            for (int j = 0; j < stage; ++j) v = lut[((size_t)j * N + u) * 768 + 256 * c + v];
            if (cnt) atomicAdd(&pushed[256 * c + v], cnt);
        }
        __syncthreads();
        gh = pushed;
    }
    if (op == AADG_OP_INVERT) {
        for (int c = 0; c < 3; ++c) L[256 * c + i] = (uint8_t)(255 - i);
    } else if (op == AADG_OP_SOLARIZE) {
        const uint8_t v = i < un.iarg[stage] ? (uint8_t)i : (uint8_t)(255 - i);
        for (int c = 0; c < 3; ++c) L[256 * c + i] = v;
    } else if (op == AADG_OP_POSTERIZE) {
        const uint8_t v = (uint8_t)(i & ~((1u << (8 - un.iarg[stage])) - 1u));
        for (int c = 0; c < 3; ++c) L[256 * c + i] = v;
    } else if (op == AADG_OP_BRIGHTNESS || op == AADG_OP_CONTRAST) {
        const float alpha = un.farg[stage];
        int deg = 0;
        if (op == AADG_OP_CONTRAST) {
            const unsigned long long ls = *reinterpret_cast<const unsigned long long*>(gh + 768);
            deg = (int)((double)ls / (double)npix + 0.5);  // int(ImageStat mean + 0.5)
        }
        uint8_t v;
        if (alpha == 1.0f) v = (uint8_t)i;
        else if (alpha == 0.0f) v = (uint8_t)deg;
        else v = (uint8_t)blend_px(deg, i, alpha, alpha >= 0.0f && alpha <= 1.0f);
        for (int c = 0; c < 3; ++c) L[256 * c + i] = v;
    } else {  // AUTOCONTRAST / EQUALIZE: wave c owns channel c, lane l its bins 4l .. 4l + 3 -- no barriers (the 256-thread scan
              // needed 48 of them per map: 8 us per launch, most of k_luts_tables)
        const int c = i >> 6, lane = i & 63;
        if (c < 3) {
            const uint4 h4 = *reinterpret_cast<const uint4*>(gh + 256 * c + 4 * lane);
            const uint32_t h[4] = {h4.x, h4.y, h4.z, h4.w};
            int lo = 256, hi = -1, nnz = 0;
            uint32_t tot = 0;
#pragma unroll
            for (int t = 3; t >= 0; --t) if (h[t]) lo = 4 * lane + t;
#pragma unroll
            for (int t = 0; t < 4; ++t) if (h[t]) { hi = 4 * lane + t; ++nnz; tot += h[t]; }
            uint32_t incl = tot;                           // inclusive scan of the lane totals
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            unsigned long long s_sum = tot;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo = min(lo, __shfl_xor(lo, o, 64));
                hi = max(hi, __shfl_xor(hi, o, 64));
                nnz += __shfl_xor(nnz, o, 64);
                s_sum += __shfl_xor(s_sum, o, 64);
            }
            const int s_lo = lo, s_hi = hi, s_nnz = nnz;
            uint32_t v4 = 0;
            if (op == AADG_OP_AUTOCONTRAST) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int bin = 4 * lane + t;
                    int ix = bin;
                    if (s_hi > s_lo) {
                        const double scale = 255.0 / (double)(s_hi - s_lo);
                        const double offset = (double)(-s_lo) * scale;
                        ix = (int)((double)bin * scale + offset);
                        ix = ix < 0 ? 0 : (ix > 255 ? 255 : ix);
                    }
                    v4 |= (uint32_t)ix << (8 * t);
                }
            } else {
                const int hl = s_hi < 0 ? 0 : s_hi;
                const uint32_t mine = (hl & 3) == 0 ? h[0] : ((hl & 3) == 1 ? h[1] : ((hl & 3) == 2 ? h[2] : h[3]));
                const unsigned long long last = __shfl(mine, hl >> 2, 64);
                const unsigned long long step = s_nnz > 1 ? (s_sum - last) / 255ull : 0ull;
                uint32_t excl = incl - tot;                // exclusive prefix of bin 4 * lane
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    uint32_t v = (uint32_t)(4 * lane + t);
                    if (step) {
                        const unsigned long long q = (step / 2 + (unsigned long long)excl) / step;
                        v = (uint32_t)(q > 255ull ? 255ull : q);
                    }
                    v4 |= v << (8 * t);
                    excl += h[t];
                }
            }
            *reinterpret_cast<uint32_t*>(L + 256 * c + 4 * lane) = v4;
        }
    }
}
// grid N (ulist == nullptr) or the length of ulist: one unit's stage-`stage` byte map per workgroup
__global__ __launch_bounds__(256) void k_lut(UnitRef ur, int stage, int N, const int* __restrict__ ulist, int npix, int Hs, int Ws, int crop,
                                             const uint32_t* hist0, const uint32_t* hist, const uint32_t* pool_hist, uint8_t* lut) {
    lut_body(ur, stage, N, ulist != nullptr ? ulist[blockIdx.x] : (int)blockIdx.x, npix, Hs, Ws, crop, hist0, hist, pool_hist, lut);
}

// ------------------------------------------------------------------------------------------------
// k_apply: grid (chunks, N), 256 threads; one op, u8 HWC -> u8 HWC.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_apply(Bufs bufs, UnitRef ur, int stage, int H, int W, int crop,
                                               const uint8_t* lut, uint8_t* out_override) {
    const int u = blockIdx.y;
    const aadg_unit& un = pick(ur, u);
    if (un.n_ops <= stage) return;
    if (unit_fusable(ur, un, H, W, crop)) return;
    const int op = un.op[stage];
    const uint8_t* in = stage_input(bufs, un, u, stage);
    uint8_t* out = out_override ? out_override : ((stage & 1) ? bufs.buf1 : bufs.buf0) + (size_t)u * bufs.img_bytes;
    const int tid = threadIdx.x;
    const int npix = H * W;
    __shared__ uint8_t sl[768];
    if (op_is_lut(op)) {
        const uint8_t* L = lut + ((size_t)stage * gridDim.y + u) * 768;
        for (int i = tid; i < 768; i += 256) sl[i] = L[i];
        __syncthreads();
    }
    const float alpha = un.farg[stage];
    const bool interp = alpha >= 0.0f && alpha <= 1.0f;
    const bool vec = ((W & 3) == 0) && ((((uintptr_t)in) & 3) == 0) && ((((uintptr_t)out) & 3) == 0);
    const int rx0 = un.rect[stage][0], ry0 = un.rect[stage][1], rx1 = un.rect[stage][2], ry1 = un.rect[stage][3];

    // per-pixel functor on (r,g,b) at (y,x) -> packed; neighbours fetched from `in` for Sharpness
    auto px_op = [&](int y, int x, uint32_t& r, uint32_t& g, uint32_t& b) {
        if (op_is_lut(op)) {
            r = sl[r]; g = sl[256 + g]; b = sl[512 + b];
        } else if (op == AADG_OP_COLOR) {
            if (alpha != 1.0f) {
                const int l = (int)rgb2l(r, g, b);
                if (alpha == 0.0f) { r = g = b = (uint32_t)l; }
                else { r = blend_px(l, (int)r, alpha, interp); g = blend_px(l, (int)g, alpha, interp); b = blend_px(l, (int)b, alpha, interp); }
            }
        } else if (op == AADG_OP_CUTOUT) {
            if (x >= rx0 && x <= rx1 && y >= ry0 && y <= ry1) { r = g = b = 127u; }
        } else if (op == AADG_OP_SHARPNESS) {
            if (alpha != 1.0f) {
                uint32_t d[3] = {r, g, b};
                if (y > 0 && x > 0 && y < H - 1 && x < W - 1) {
                    uint32_t s[3] = {4 * r, 4 * g, 4 * b};
                    for (int dy = -1; dy <= 1; ++dy) {
                        const uint8_t* row = in + ((size_t)(y + dy) * W + (x - 1)) * 3;
#pragma unroll
                        for (int k = 0; k < 9; ++k) s[k % 3] += row[k];
                    }
                    d[0] = (s[0] + 6) / 13; d[1] = (s[1] + 6) / 13; d[2] = (s[2] + 6) / 13;  // trunc(sum/13 + 0.5)
                }
                if (alpha == 0.0f) { r = d[0]; g = d[1]; b = d[2]; }
                else { r = blend_px((int)d[0], (int)r, alpha, interp); g = blend_px((int)d[1], (int)g, alpha, interp); b = blend_px((int)d[2], (int)b, alpha, interp); }
            }
        }
    };

    if (vec) {
        const int ngroups = npix >> 2;
        const uint32_t* p32 = reinterpret_cast<const uint32_t*>(in);
        uint32_t* o32 = reinterpret_cast<uint32_t*>(out);
        for (int gi = blockIdx.x * 256 + tid; gi < ngroups; gi += gridDim.x * 256) {
            uint32_t a = p32[3 * gi], b = p32[3 * gi + 1], c = p32[3 * gi + 2];
            uint32_t R[4] = {a & 255, a >> 24, (b >> 16) & 255, (c >> 8) & 255};
            uint32_t G[4] = {(a >> 8) & 255, b & 255, b >> 24, (c >> 16) & 255};
            uint32_t B[4] = {(a >> 16) & 255, (b >> 8) & 255, c & 255, c >> 24};
            const int p0 = gi << 2;
            const int y = p0 / W, x0 = p0 - y * W;
#pragma unroll
            for (int k = 0; k < 4; ++k) px_op(y, x0 + k, R[k], G[k], B[k]);
            o32[3 * gi] = R[0] | (G[0] << 8) | (B[0] << 16) | (R[1] << 24);
            o32[3 * gi + 1] = G[1] | (B[1] << 8) | (R[2] << 16) | (G[2] << 24);
            o32[3 * gi + 2] = B[2] | (R[3] << 8) | (G[3] << 16) | (B[3] << 24);
        }
    } else {
        for (int p = blockIdx.x * 256 + tid; p < npix; p += gridDim.x * 256) {
            uint32_t r = in[3 * (size_t)p], g = in[3 * (size_t)p + 1], b = in[3 * (size_t)p + 2];
            const int y = p / W, x = p - y * W;
            px_op(y, x, r, g, b);
            out[3 * (size_t)p] = (uint8_t)r; out[3 * (size_t)p + 1] = (uint8_t)g; out[3 * (size_t)p + 2] = (uint8_t)b;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_tables: grid N, 256 threads.  Coefficient / index tables for the crop window of one unit.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}

// first source tap of output index xx (the xmin of Pillow's precompute_coeffs); same double expressions as
// bilinear_coeffs below, so the two always agree
__device__ __forceinline__ int bilinear_xmin(int inSize, int outSize, int xx) {
    if (inSize == outSize) return xx;
    const double scale = (double)inSize / (double)outSize;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double center = 0.0 + ((double)xx + 0.5) * scale;
    const int xmin = (int)(center - support + 0.5);
    return xmin < 0 ? 0 : xmin;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for output index `xx` (of outSize) from inSize
__device__ void bilinear_coeffs(int inSize, int outSize, int xx, int* xmin_out, int* k /*KMAX*/) {
    if (inSize == outSize) {  // Image.resize does not resample an unchanged axis
        *xmin_out = xx;
        k[0] = 1 << PRECISION_BITS;
        for (int t = 1; t < KMAX; ++t) k[t] = 0;
        return;
    }
    double scale = (double)inSize / (double)outSize;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double center = 0.0 + ((double)xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > inSize) xmax = inSize;
    int n = xmax - xmin;
    if (n > KMAX) n = KMAX;  // host validates scale >= 1/3, never taken
    double kd[KMAX];
    double ww = 0.0;
    for (int t = 0; t < KMAX; ++t) {
        double w = 0.0;
        if (t < n) { w = bilinear_filter(((double)(t + xmin) - center + 0.5) * ss); ww += w; }
        kd[t] = w;
    }
    for (int t = 0; t < KMAX; ++t) {
        double v = kd[t];
        if (t < n && ww != 0.0) v = v / ww;
        k[t] = t < n ? (int)(0.5 + v * (double)(1 << PRECISION_BITS)) : 0;
    }
    *xmin_out = xmin;
}

// the BILINEAR tap tables of unit u (first tap + KMAX fixed-point coefficients per output column / row of the crop window)
__device__ __forceinline__ void tables_coeff_body(const UnitRef& ur, int Hs, int Ws, int crop, int* tab, int u) {
    const aadg_unit& un = pick(ur, u);
    int* base = tab + (size_t)u * crop * TAB_STRIDE;
    int* xmin = base;
    int* xk = xmin + crop;
    int* ymin = xk + (size_t)crop * KMAX;
    int* yk = ymin + crop;
    const int w = un.scaled_w, h = un.scaled_h;
    const int ox = un.crop_x - un.pad, oy = un.crop_y - un.pad;
    for (int i = threadIdx.x; i < 2 * crop; i += 256) {
        const bool isx = i < crop;
        const int o = isx ? i : i - crop;
        const int s = o + (isx ? ox : oy);
        const int outSize = isx ? w : h, inSize = isx ? Ws : Hs;
        int k[KMAX];
        int mn = -1;
        if (s >= 0 && s < outSize) bilinear_coeffs(inSize, outSize, s, &mn, k);
        else for (int t = 0; t < KMAX; ++t) k[t] = 0;
        (isx ? xmin : ymin)[o] = mn;
        int* kk = (isx ? xk : yk) + (size_t)o * KMAX;
        for (int t = 0; t < KMAX; ++t) kk[t] = k[t];
    }
}
// The NEAREST index tables of unit u (labels).  ImagingScaleAffine accumulates xo += a0 in double, SEQUENTIALLY: the rounding of the
// running sum decides exact ties, so the index is NOT floor((i + 0.5) * in / out).  Rounds 1-3 let one lane per axis walk the sum
// (~1000 dependent additions: 13 / 25 us for 512 / 1024-pixel crops, the long pole of the tables launch).
// Round 4: the same values WITHOUT the serial chain.  xo_{i+1} = fl(xo_i + a0) is a rounding per step, but inside one binade
// [2^e, 2^(e+1)) every xo is a multiple of u = 2^(e-52) and a0 = q u + r is fixed, so fl() adds the SAME multiple of u at every step:
// q or q + 1 by r against u / 2 -- and for r = u / 2 exactly (a tie; common: a0's last mantissa bit) round-to-even makes the first
// result even, after which the increment is constant too (q even: q; q odd: q + 1).  So per binade: two real additions give
// xo_{i+1}, xo_{i+2} (the steady increment d = xo_{i+2} - xo_{i+1}, exact), the rest of the binade is the arithmetic sequence
// m1 + j dm in units of u (integers below 2^53), and the step that leaves the binade is again a real addition.  <= 14 binades
// between a0 / 2 and 2048: one lane per axis builds <= 32 segments (lo, hi, m1, dm, shift) in ~1 us, then every lane reads its own
// index off them -- instead of ~1000 dependent additions (13 / 25 us for 512 / 1024-pixel crops: the long pole of k_luts_tables).
// Equal to the sequential walk for every index of every size pair tried on the host (2423 pairs: all output sizes in [in / 3, 3 in]
// for in = 33, 64, 96 and every 7th for in = 300 .. 2048) and bit-exact against the oracle's walk in the GPU tests.
constexpr int NN_SEGS = 40;
struct NnSegs {
    int lo[NN_SEGS], hi[NN_SEGS], sh[NN_SEGS], n;
    long long m1[NN_SEGS], dm[NN_SEGS];
};
__device__ __forceinline__ int f64_exponent(double x) { return (int)((__double_as_longlong(x) >> 52) & 0x7FF) - 1023; }
__device__ __forceinline__ double f64_pow2(int e) { return __longlong_as_double((long long)(e + 1023) << 52); }
__device__ __forceinline__ long long f64_mantissa(double x) { return (__double_as_longlong(x) & ((1ll << 52) - 1)) | (1ll << 52); }
// segments covering source indices [0, lim) of the walk for inSize -> outSize (inSize != outSize)
__device__ void nn_build_segments(int inSize, int outSize, int lim, NnSegs& S) {
    const double a0 = (double)inSize / (double)outSize;
    double x = 0.0 + a0 * 0.5;
    int i = 0, n = 0;
    auto single = [&](int idx, double v) {
        S.lo[n] = S.hi[n] = idx; S.m1[n] = f64_mantissa(v); S.dm[n] = 0; S.sh[n] = 52 - f64_exponent(v); ++n;
    };
    while (i < lim && n + 2 <= NN_SEGS) {
        const int e = f64_exponent(x);
        const double top = f64_pow2(e + 1);
        single(i, x);
        const double x1 = x + a0;
        if (x1 >= top) { i += 1; x = x1; continue; }
        const double x2 = x1 + a0;
        if (x2 >= top) { single(i + 1, x1); i += 2; x = x2; continue; }
        const long long m1 = f64_mantissa(x1);
        const long long dm = (long long)((x2 - x1) * f64_pow2(52 - e));           // exact: a multiple of u below 2^(e+1)
        const long long room = ((1ll << 53) - 1) - m1;
        long long j = (long long)((double)room * __builtin_amdgcn_rcp((double)dm));      // approximate quotient, put right below
        j = j < 0 ? 0 : j;
        while (j * dm > room) --j;
        while ((j + 1) * dm <= room) ++j;
        S.lo[n] = i + 1; S.hi[n] = i + 1 + (int)j; S.m1[n] = m1; S.dm[n] = dm; S.sh[n] = 52 - e; ++n;
        const double x_last = (double)(m1 + j * dm) * f64_pow2(e - 52);           // exact
        x = x_last + a0;
        i += 2 + (int)j;
    }
    S.n = n;
}
__device__ __forceinline__ void tables_nn_body_closed(const UnitRef& ur, int Hs, int Ws, int crop, int* tab, int u, NnSegs* segs /* 2, LDS */) {
    const aadg_unit& un = pick(ur, u);
    int* base = tab + (size_t)u * crop * TAB_STRIDE;
    int* xnn = base + 2 * (crop + (size_t)crop * KMAX);
    int* ynn = xnn + crop;
    const int w = un.scaled_w, h = un.scaled_h;
    const int ox = un.crop_x - un.pad, oy = un.crop_y - un.pad;
    const int tid = threadIdx.x;
    if (tid == 0 || tid == 64) {
        const bool isx = tid == 0;
        const int outSize = isx ? w : h, inSize = isx ? Ws : Hs, off = isx ? ox : oy;
        segs[isx ? 0 : 1].n = 0;
        if (inSize != outSize) nn_build_segments(inSize, outSize, min(off + crop, outSize), segs[isx ? 0 : 1]);
    }
    __syncthreads();
    // waves 0-1 write the x table, waves 2-3 the y table.  Segments outer, indices inner: a segment's five words are read once (one LDS
    // round trip), then the 128 lanes of the axis stride over the part of the crop window it covers.  (Indices outer -- every lane
    // scanning the list for its own index -- is a chain of dependent LDS reads per index: 16 us at 1024 pixels, slower than the walk.)
    const int ax = tid >> 7, l = tid & 127;
    const NnSegs& S = segs[ax];
    const int outSize = ax == 0 ? w : h, inSize = ax == 0 ? Ws : Hs, off = ax == 0 ? ox : oy;
    int* out = ax == 0 ? xnn : ynn;
    const int lim = min(off + crop, outSize), start = max(off, 0);      // live source indices: [start, lim)
    for (int o = l; o < crop; o += 128) {                               // everything outside the scaled image: no label
        const int sidx = o + off;
        if (sidx < start || sidx >= lim) out[o] = -1;
        else if (inSize == outSize) out[o] = sidx;                      // a0 == 1: xo = s + 0.5 exactly
    }
    if (inSize == outSize) return;
    const int nseg = S.n;
    const bool before = (tid & 63) < nseg && S.hi[tid & 63] < start;
    const int first = __popcll(__ballot(before));                       // the first segment that reaches into the window
    for (int t = first; t < nseg; ++t) {
        const int lo = S.lo[t], hi = S.hi[t], sh = S.sh[t];
        const long long m1 = S.m1[t], dm = S.dm[t];
        const int a = max(lo, start), bnd = min(hi, lim - 1);
        for (int sidx = a + l; sidx <= bnd; sidx += 128) {
            const long long m = m1 + (long long)(sidx - lo) * dm;
            const int xin = sh >= 64 ? 0 : (int)(m >> sh);               // sh > 52: values below 1
            out[sidx - off] = xin < inSize ? xin : -1;
        }
        if (hi >= lim - 1) break;
    }
}
__device__ __forceinline__ void tables_body(const UnitRef& ur, int Hs, int Ws, int crop, int* tab, int u) {
    __shared__ NnSegs segs[2];
    tables_coeff_body(ur, Hs, Ws, crop, tab, u);
    tables_nn_body_closed(ur, Hs, Ws, crop, tab, u, segs);
}
__global__ __launch_bounds__(256) void k_tables(UnitRef ur, int Hs, int Ws, int crop, int* tab) {
    tables_body(ur, Hs, Ws, crop, tab, blockIdx.x);
}
// The coefficient / index tables depend on the unit records only, the stage-0 histograms on the records and the source images:
// one launch for both (blocks [0, chunks * nstat): histogram chunks; the next N blocks: one unit's tables each) instead of two
// latency-bound kernels back to back.
__global__ __launch_bounds__(256) void k_hist_tables(Bufs bufs, UnitRef ur, const int* __restrict__ ulist, int nstat, int chunks, int npix,
                                                     int Hs, int Ws, int crop, uint32_t* hist, int* tab) {
    __shared__ uint32_t sh[4][768];
    const int b = blockIdx.x, nh = chunks * nstat;
    if (b < nh) hist_body(bufs, ur, ulist, 0, npix, Hs, Ws, crop, hist, b % chunks, b / chunks, chunks, sh);
    else tables_body(ur, Hs, Ws, crop, tab, b - nh);
}

// With the pool's histograms cached by the caller, the stage-0 byte maps depend on the unit records and the cache only -- like the
// tables: one launch for both (blocks [0, N): one unit's stage-0 map each; the next N: its tables).
__global__ __launch_bounds__(256) void k_lut_tables(UnitRef ur, int N, int npix, int Hs, int Ws, int crop, const uint32_t* pool_hist,
                                                    uint8_t* lut, int* tab) {
    const int b = blockIdx.x;
    if (b < N) lut_body(ur, 0, N, b, npix, Hs, Ws, crop, pool_hist, pool_hist, pool_hist, lut);
    else tables_body(ur, Hs, Ws, crop, tab, b - N);
}

// The same with EVERY stage's map that does not wait for a pixel pass (blocks [0, N): unit u, stages in order -- a later stage's
// push-forward reads the earlier maps this workgroup has just written).  A stage >= 1 whose statistics need a pixel pass is left
// to the late chain of aadg_aug_u8_forward_ex2 (and so is everything behind it: those units are re-done there).  Blocks [N, 2N):
// unit b - N's tap tables; [2N, 3N): its NEAREST tables (a serial walk: own workgroup so that the taps do not wait behind it).
__global__ __launch_bounds__(256) void k_luts_tables(UnitRef ur, int N, int max_ops, int npix, int Hs, int Ws, int crop,
                                                     const uint32_t* pool_hist, uint8_t* lut, int* tab, uint32_t* hist_zero) {
    __shared__ NnSegs nn[2];
    const int b = blockIdx.x;
    if (b >= 2 * N) {
        tables_nn_body_closed(ur, Hs, Ws, crop, tab, b - 2 * N, nn);
        return;
    }
    if (b >= N) {
        tables_coeff_body(ur, Hs, Ws, crop, tab, b - N);
        return;
    }
    if (hist_zero != nullptr)                  // the pixel-pass histograms of slots 1 .. max_ops - 1 accumulate with atomics: clear this unit's rows
        for (int k = 1; k < max_ops; ++k)
            for (int t = threadIdx.x; t < HIST_STRIDE; t += 256) hist_zero[((size_t)k * N + b) * HIST_STRIDE + t] = 0u;
    const aadg_unit& un = pick(ur, b);
    const bool fused_flow = unit_fusable(ur, un, Hs, Ws, crop);
    for (int k = 0; k < max_ops && k < un.n_ops; ++k) {
        if (k > 0 && op_needs_stats(un.op[k]) && !stats_by_pushforward(un, k, fused_flow)) continue;
        lut_body(ur, k, N, b, npix, Hs, Ws, crop, pool_hist, pool_hist, pool_hist, lut);
        __threadfence_block();
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// k_final: grid (ceil(crop/256), ceil(crop/ROWS_PER_BLOCK), N), 256 threads.
// A wave owns 256 consecutive output columns of one row (4 per lane -> one float4 store per plane).
// ------------------------------------------------------------------------------------------------
constexpr int FIN_ROWS = 16;

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void k_final(Bufs bufs, const uint8_t* masks, UnitRef ur, int Hs, int Ws, int crop,
                                               int dataset_in, const int* tab, float* out_img, float* out_lbl) {
    const int dataset = dataset_in & 0xFF;
    const bool stream_out = (dataset_in & AUG_STREAM_OUT) != 0;
    const int u = blockIdx.z;
    const aadg_unit& un = pick(ur, u);
    if (unit_fusable(ur, un, Hs, Ws, crop)) return;
    const uint8_t* img = stage_input(bufs, un, u, un.n_ops);
    const uint8_t* msk = masks + (size_t)un.src * Hs * Ws;
    const int K = dataset == AADG_DATASET_OPTIC ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int* base = tab + (size_t)u * crop * TAB_STRIDE;
    const int* xmin_t = base;
    const int* xk_t = xmin_t + crop;
    const int* ymin_t = xk_t + (size_t)crop * KMAX;
    const int* yk_t = ymin_t + crop;
    const int* xnn_t = yk_t + (size_t)crop * KMAX;
    const int* ynn_t = xnn_t + crop;

    __shared__ float lutf[256];
    lutf[tid] = __fsub_rn(__fdiv_rn((float)tid, 127.5f), 1.0f);  // np.float32: x /= 127.5; x -= 1.0
    __syncthreads();

    const int x0 = blockIdx.x * 256 + lane * 4;
    if (x0 >= crop) return;
    const int nvalid = crop - x0 < 4 ? crop - x0 : 4;
    // number of live taps of this unit (uniform per unit)
    const int nxt = un.scaled_w == Ws ? 1 : KMAX;
    const int nyt = un.scaled_h == Hs ? 1 : KMAX;

    int xm[4], xn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        xm[i] = i < nvalid ? xmin_t[x0 + i] : -1;
        xn[i] = i < nvalid ? xnn_t[x0 + i] : -1;
    }
    const size_t plane = (size_t)crop * crop;
    float* oi = out_img + (size_t)u * 3 * plane;
    float* ol = out_lbl + (size_t)u * K * plane;
    const bool vec = (crop & 3) == 0;

    const int yb = blockIdx.y * FIN_ROWS;
    for (int yy = wv; yy < FIN_ROWS; yy += 4) {
        const int y = yb + yy;
        if (y >= crop) break;
        const int ym = ymin_t[y];
        const int yn = ynn_t[y];
        int acc[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = acc[i][2] = 1 << (PRECISION_BITS - 1);
        if (ym >= 0) {
            for (int v = 0; v < nyt; ++v) {
                const int ky = yk_t[(size_t)y * KMAX + v];
                if (ky == 0) continue;
                int row = ym + v;
                row = row < Hs ? row : Hs - 1;
                const uint8_t* rp = img + (size_t)row * Ws * 3;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (xm[i] < 0) continue;
                    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
                    for (int t = 0; t < nxt; ++t) {
                        const int kx = xk_t[(size_t)(x0 + i) * KMAX + t];
                        if (kx == 0) continue;
                        int col = xm[i] + t;
                        col = col < Ws ? col : Ws - 1;
                        const uint8_t* pp = rp + (size_t)col * 3;
                        s0 += (int)pp[0] * kx; s1 += (int)pp[1] * kx; s2 += (int)pp[2] * kx;
                    }
                    acc[i][0] += clip8(s0) * ky; acc[i][1] += clip8(s1) * ky; acc[i][2] += clip8(s2) * ky;
                }
            }
        }
        float o[3][4];
        float l0[4], l1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool inside = ym >= 0 && xm[i] >= 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c][i] = lutf[inside ? clip8(acc[i][c]) : 0];
            uint32_t m = 0;
            if (yn >= 0 && xn[i] >= 0) m = msk[(size_t)yn * Ws + xn[i]];
            if (dataset == AADG_DATASET_OPTIC) { l0[i] = m <= 50 ? 1.0f : 0.0f; l1[i] = m <= 200 ? 1.0f : 0.0f; }
            else { l0[i] = m != 0 ? 1.0f : 0.0f; l1[i] = 0.0f; }
        }
        const size_t off = (size_t)y * crop + x0;
        if (vec) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                store_out4(oi + c * plane + off, make_float4(o[c][0], o[c][1], o[c][2], o[c][3]), stream_out);
            store_out4(ol + off, make_float4(l0[0], l0[1], l0[2], l0[3]), stream_out);
            if (K == 2) store_out4(ol + plane + off, make_float4(l1[0], l1[1], l1[2], l1[3]), stream_out);
        } else {
            for (int i = 0; i < nvalid; ++i) {
                for (int c = 0; c < 3; ++c) oi[c * plane + off + i] = o[c][i];
                ol[off + i] = l0[i];
                if (K == 2) ol[plane + off + i] = l1[i];
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Fused data flow: LDS patch builder shared by k_hist_fused and k_fused.
//
// Loads source rows [r_lo, r_hi) x cols [c_lo, c_hi) (c_lo, c_hi multiples of 4, region already includes the
// halo and is clipped to the image) as one RGBX word per pixel, then applies ops [0, nops) in LDS.
// Pixels within `sharp_count` of a patch edge that is not an image edge are NOT valid afterwards.
// Returns the buffer (A or B) that holds the result.  Ends with a __syncthreads().
// ------------------------------------------------------------------------------------------------
struct __attribute__((packed)) UnalignedU32 { uint32_t v; };
// three dwords of a pixel group (12 bytes, 4-byte aligned)
struct __attribute__((packed, aligned(4))) U32x3 { uint32_t x, y, z; };
__device__ __forceinline__ uint32_t blend3(uint32_t deg, uint32_t img, float alpha, bool interp) {
    const uint32_t r = blend_px((int)(deg & 255), (int)(img & 255), alpha, interp);
    const uint32_t g = blend_px((int)((deg >> 8) & 255), (int)((img >> 8) & 255), alpha, interp);
    const uint32_t b = blend_px((int)((deg >> 16) & 255), (int)((img >> 16) & 255), alpha, interp);
    return r | (g << 8) | (b << 16);
}

// A word of the (uniform) unit record at an index known only at run time, through the scalar cache.  The compiler reads `un.op[j]` with a
// run-time j behind a barrier as a VECTOR load with a full wait (it no longer treats the record as invariant there): one L2 round trip per
// field, in series, in front of every tile's op chain.  The record was read by the prologue's scalar loads, so these hit the scalar cache.
__device__ __forceinline__ uint32_t unit_word(const aadg_unit& un, uint32_t byte_off) {
    uint32_t v;
    const aadg_unit* p = &un;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p), "s"(byte_off) : "memory");
    return v;
}
__device__ __forceinline__ int unit_op(const aadg_unit& un, int j) { return (int)unit_word(un, 8u + 4u * (uint32_t)j); }
__device__ __forceinline__ int unit_iarg(const aadg_unit& un, int j) { return (int)unit_word(un, 24u + 4u * (uint32_t)j); }
__device__ __forceinline__ float unit_farg(const aadg_unit& un, int j) { return __uint_as_float(unit_word(un, 40u + 4u * (uint32_t)j)); }
static_assert(offsetof(aadg_unit, op) == 8 && offsetof(aadg_unit, iarg) == 24 && offsetof(aadg_unit, farg) == 40 && offsetof(aadg_unit, rect) == 56,
              "unit_word offsets follow include/aadg_hip.h");

// ops whose per-channel byte map is staged in LDS (768 bytes per stage, built by k_lut): the three statistics ops,
// Brightness and Solarize.  Invert / Posterize are one integer instruction per pixel; Color mixes channels and
// Cutout depends on the position; Sharpness is not pointwise and is handled by the caller.
__device__ __forceinline__ bool needs_lds_lut(int op) {
    return op_needs_stats(op) || op == AADG_OP_BRIGHTNESS || op == AADG_OP_SOLARIZE;
}
__device__ __forceinline__ bool is_stencil(const aadg_unit& un, int j) {
    return unit_op(un, j) == AADG_OP_SHARPNESS && unit_farg(un, j) != 1.0f;
}

// One pointwise op applied to a set of pixels.  The op is uniform per workgroup: the dispatch happens ONCE here and
// `each(f)` then runs the branch-free per-pixel function f(p, y, x) over the caller's pixels (registers, unrolled, or an
// LDS region) -- no per-pixel switch, no per-pixel checks of uniform parameters.
template <typename Each>
__device__ __forceinline__ void dispatch_op(const aadg_unit& un, int j, const uint8_t* sl_all, Each each) {
    const int op = unit_op(un, j);
    switch (op) {
        case AADG_OP_INVERT:
            each([](uint32_t p, int, int) { return ~p & 0xFFFFFFu; });
            break;
        case AADG_OP_POSTERIZE: {
            const uint32_t m = (0xFFu & ~((1u << (8 - unit_iarg(un, j))) - 1u)) * 0x010101u;
            each([m](uint32_t p, int, int) { return p & m; });
            break;
        }
        case AADG_OP_SOLARIZE: case AADG_OP_AUTOCONTRAST: case AADG_OP_EQUALIZE: case AADG_OP_CONTRAST: case AADG_OP_BRIGHTNESS: {
            const uint8_t* sl = sl_all + j * 768;
            each([sl](uint32_t p, int, int) {
                return (uint32_t)sl[p & 255u] | ((uint32_t)sl[256 + ((p >> 8) & 255u)] << 8) | ((uint32_t)sl[512 + ((p >> 16) & 255u)] << 16);
            });
            break;
        }
        case AADG_OP_COLOR: {
            const float alpha = unit_farg(un, j);
            each([alpha](uint32_t p, int, int) {
                const uint32_t l = rgb2l(p & 255u, (p >> 8) & 255u, (p >> 16) & 255u);
                return blend3(l * 0x010101u, p, alpha, true);
            });
            break;
        }
        case AADG_OP_CUTOUT: {
            const uint32_t ro = 56u + 16u * (uint32_t)j;
            const int x0 = (int)unit_word(un, ro), y0 = (int)unit_word(un, ro + 4), x1 = (int)unit_word(un, ro + 8), y1 = (int)unit_word(un, ro + 12);
            each([=](uint32_t p, int y, int x) { return (x >= x0 && x <= x1 && y >= y0 && y <= y1) ? 0x7F7F7Fu : p; });
            break;
        }
        default:
            break;      // Sharpness with factor 1 and unknown ids: identity
    }
}

// one pointwise op on a single pixel (staged / histogram helpers; the tile kernels use dispatch_op over many pixels)
__device__ __forceinline__ uint32_t pointwise_op(const aadg_unit& un, int j, uint32_t p, int y, int x, const uint8_t* sl_all) {
    uint32_t r = p;
    dispatch_op(un, j, sl_all, [&](auto f) { r = f(p, y, x); });
    return r;
}

// The Sharpness passes of a patch (ops [j0, nops) with op j0 a stencil, the pointwise ops between / behind the stencils applied by the
// lane that has just written the pixel): `cur` holds rows [r_lo, r_lo + ph) x columns [c_lo, c_lo + pw) after ops [0, j0), `oth` is the
// ping-pong buffer.  Returns the buffer that holds the result; every pass ends with a barrier.
__device__ __forceinline__ uint32_t* patch_stencil_passes(const aadg_unit& un, int nops, int j0, int Hs, int Ws, int r_lo, int c_lo, int ph, int pw,
                                                          uint32_t* cur, uint32_t* oth, const uint8_t* sl_all) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // Sharpness passes: wave <-> (64-column chunk, row segment), a lane walks DOWN its column with the horizontal 3-sums of the
    // previous / current / next row in registers: 3 LDS reads and ~50 instructions per pixel instead of 9 and ~80 (each pixel as the
    // centre of its own 3 x 3 gather).  Narrow patches split the rows so that all four waves have work.
    // Work items: (64-column chunk, row segment) pairs dealt round-robin to the four waves.  A patch is typically 256 + 8 columns wide
    // (the halo, rounded to groups of 4): its fifth chunk holds 8 live columns, and as a chunk of its own it DOUBLED the stencil time
    // of the wave that got it (5 items on 4 waves; the wave-0 SIMD of a CU then carries twice the instructions of the others --
    // the statistics pass of a stencil tile took ~20 us).  A last chunk of <= 32 columns is therefore folded: its lanes are
    // (column, row segment) pairs -- 8 columns x 8 segments, 16 x 4 or 32 x 2 -- so that it costs rows / 8 .. rows / 2 (+ 2 halo rows).
    const int nfull = pw >> 6, wlast = pw & 63;
    const bool fold = wlast > 0 && wlast <= 32 && nfull > 0;
    const int nch = fold ? nfull : (pw + 63) >> 6;
    const int nseg = nch >= 3 ? 1 : (nch == 2 ? 2 : 4);
    const int seg_rows = (ph + nseg - 1) / nseg;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    const int fshift = wlast <= 8 ? 3 : (wlast <= 16 ? 4 : 5);          // folded chunk: log2 of its padded width
    const int frows = (ph + (64 >> fshift) - 1) >> (6 - fshift);        // rows per lane segment
    auto my_items = [&](auto body) {              // body(column of this lane, first row, end row) for the items of this wave
        int it = 0;
        for (int sg = 0; sg < nseg; ++sg)
            for (int c = 0; c < nch; ++c, ++it)
                if ((it & 3) == wvs) {
                    const int ra = sg * seg_rows, rb = min(ph, ra + seg_rows);
                    if (ra < rb) body(64 * c + lane, ra, rb);
                }
        if (fold && (it & 3) == wvs) {
            const int ra = (lane >> fshift) * frows, rb = min(ph, ra + frows);
            body(64 * nfull + (lane & ((1 << fshift) - 1)), ra, rb);         // ra >= rb: nothing to do for this lane
        }
    };
    while (j0 < nops) {                           // op j0 is a Sharpness stencil
        const float alpha = unit_farg(un, j0);
        int j1 = j0 + 1;
        while (j1 < nops && !is_stencil(un, j1)) ++j1;
        my_items([&](int col, int ra, int rb) {
            const bool act = col < pw && ra < rb;
            ra = min(ra, ph - 1);                                             // idle lanes of a folded chunk read in range
            const int cc = min(col, pw - 1), cm = max(cc - 1, 0), cp = min(cc + 1, pw - 1);
            const int x = c_lo + cc;
            const bool col_in = x > 0 && x < Ws - 1 && cc > 0 && cc < pw - 1;   // ImageFilter.SMOOTH copies the 1-pixel image border
            uint32_t h0rb = 0, h0g = 0, h1rb, h1g, h2rb = 0, h2g = 0, p1, p2 = 0;
            auto hrow = [&](int r, uint32_t& hrb, uint32_t& hg, uint32_t& pc) {
                const uint32_t* rp = cur + r * pw;
                const uint32_t q0 = rp[cm], q1 = rp[cc], q2 = rp[cp];
                pc = q1;
                hrb = (q0 & 0xFF00FFu) + (q1 & 0xFF00FFu) + (q2 & 0xFF00FFu);
                hg = ((q0 >> 8) & 255u) + ((q1 >> 8) & 255u) + ((q2 >> 8) & 255u);
            };
            if (ra > 0) { uint32_t t; hrow(ra - 1, h0rb, h0g, t); }
            hrow(ra, h1rb, h1g, p1);
            for (int r = ra; r < rb; ++r) {
                if (r + 1 < ph) hrow(r + 1, h2rb, h2g, p2);
                const int y = r_lo + r;
                const bool in = col_in && y > 0 && y < Hs - 1 && r > 0 && r < ph - 1;
                const uint32_t srb = h0rb + h1rb + h2rb + 4u * (p1 & 0xFF00FFu), sg = h0g + h1g + h2g + 4u * ((p1 >> 8) & 255u);   // centre weight 5 = 4 + 1
                // (x + 6) / 13 for x <= 13 * 255: a 24-bit multiply by ceil(2^16 / 13) and a shift (exact for x + 6 <= 3321:
                // the error term x * 10 / (13 * 65536) stays below 1/13); a 32-bit division is several quarter-rate multiplies
                const uint32_t cr = (uint32_t)__mul24((int)((srb & 0xFFFFu) + 6u), 5042) >> 16, cb = (uint32_t)__mul24((int)((srb >> 16) + 6u), 5042) >> 16,
                               cg = (uint32_t)__mul24((int)(sg + 6u), 5042) >> 16;
                const uint32_t d = in ? (cr | (cg << 8) | (cb << 16)) : p1;
                if (act) oth[r * pw + col] = blend3(d, p1, alpha, true);
                h0rb = h1rb; h0g = h1g; h1rb = h2rb; h1g = h2g; p1 = p2;
            }
        });
        // the pointwise ops that follow run on the pixels this lane has just written: no barrier needed
        for (int j = j0 + 1; j < j1; ++j)
            dispatch_op(un, j, sl_all, [&](auto f) {
                my_items([&](int col, int ra, int rb) {
                    if (col < pw)
                        for (int r = ra; r < rb; ++r) {
                            const int i = r * pw + col;
                            oth[i] = f(oth[i], r_lo + r, c_lo + col);
                        }
                });
            });
        uint32_t* t = cur; cur = oth; oth = t;
        __syncthreads();
        j0 = j1;
    }
    return cur;
}

// Loads the patch (12-byte vector loads, all issued up front), applies the leading pointwise ops while the
// pixels are still in registers, stores RGBX words to LDS; every Sharpness op then costs one LDS ping-pong
// pass, after which the pointwise ops that follow it run in place on the lane's own pixels.
// WMODE: how columns >= 256 of a patch wider than 256 pixels are fetched: 0 = never wider; 1 = a second group of registers per lane
// (lanes 0..2 use it); 2 = a small trailing pass by the first 4 * rows threads (3 registers instead of 3 * NR)
template <int NR, int WMODE>
__device__ __forceinline__ uint32_t* build_patch(const aadg_unit& un, int nops, const uint8_t* __restrict__ src, int Hs, int Ws,
                                 int r_lo, int r_hi, int c_lo, int c_hi, uint32_t* A, uint32_t* B,
                                 const uint8_t* __restrict__ lut, size_t lut_stage_stride, int u, uint8_t* sl_all) {
    const int tid = threadIdx.x;
    // No load of the prologue is conditional per lane: addresses are clamped into the patch / the byte-map slots (every stage's slot exists
    // in the workspace) and what an idle lane fetched is dropped at the LDS store.  Conditional loads became branches with a wait at every
    // join -- a chain of round trips in front of every tile (the horizontal pass of the down-scaling units: 78 -> 54 us per chunk).
    uint32_t lreg[AADG_MAX_OPS];
    bool any_lut = false;
    {
        const uint32_t* lp = reinterpret_cast<const uint32_t*>(lut + (size_t)u * 768) + min(tid, 191);
#pragma unroll
        for (int j = 0; j < AADG_MAX_OPS; ++j) {
            lreg[j] = lp[j * (lut_stage_stride >> 2)];
            any_lut = any_lut || (j < nops && needs_lds_lut(un.op[j]));
        }
    }
    const int pw = c_hi - c_lo, ph = r_hi - r_lo, q4 = pw >> 2;
    // thread <-> (row = wave + 4k, group of 4 pixels = lane (+64)): no integer divisions.  All loads of this
    // thread are issued before the first use.
    const int lane = tid & 63, wv = tid >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    // NR = rows per wave (4 * NR >= patch rows); WIDE = patch may be wider than 256 pixels
    constexpr bool WIDE = WMODE != 0;
    constexpr int NW = WMODE == 1 ? NR : 1;
    uint32_t ra[NR], rb[NR], rc[NR], rd[NW], re[NW], rf[NW];
    const bool wide = WIDE && pw > 256;                         // uniform
    const bool has2 = WMODE == 1 && lane + 64 < q4;
    // WMODE 2: thread t < 4 * rows fetches group 64 + (t & 3) of row t >> 2
    const int trow = tid >> 2, tgrp = 64 + (tid & 3);
    const bool tail = WMODE == 2 && trow < ph && tgrp < q4;
    uint32_t wa = 0, wb = 0, wc = 0;
    const int row_bytes = Ws * 3;
    const uint8_t* blk = src + ((size_t)r_lo * Ws + c_lo) * 3;  // uniform
    const uint32_t goff = 12u * (uint32_t)min(lane, q4 - 1), goff2 = 12u * (uint32_t)min(lane + 64, q4 - 1);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const uint8_t* rowp = blk + (size_t)min(wvu + 4 * k, ph - 1) * row_bytes;      // uniform
        const U32x3 v = *reinterpret_cast<const U32x3*>(rowp + goff);
        ra[k] = v.x; rb[k] = v.y; rc[k] = v.z;
        if (WMODE == 1) {
            rd[k % NW] = re[k % NW] = rf[k % NW] = 0;
            if (wide) {
                const U32x3 v2 = *reinterpret_cast<const U32x3*>(rowp + goff2);
                rd[k % NW] = v2.x; re[k % NW] = v2.y; rf[k % NW] = v2.z;
            }
        }
    }
    if (WMODE == 2 && wide) {
        const U32x3 v = *reinterpret_cast<const U32x3*>(blk + (uint32_t)min(trow, ph - 1) * (uint32_t)row_bytes + 12u * (uint32_t)min(tgrp, q4 - 1));
        wa = v.x; wb = v.y; wc = v.z;
    }
    if (any_lut) {
        if (tid < 192) {
#pragma unroll
            for (int j = 0; j < AADG_MAX_OPS; ++j) reinterpret_cast<uint32_t*>(sl_all + j * 768)[tid] = lreg[j];
        }
        __syncthreads();
    }
    int j0 = 0;                                   // leading pointwise segment [0, j0)
    while (j0 < nops && !is_stencil(un, j0)) ++j0;
    // unpack to RGBX pixels held in registers, apply the leading pointwise ops OP BY OP (one uniform dispatch per op for
    // all of this thread's pixels), store to LDS
    uint32_t px[NR][4];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const uint32_t a = ra[k], b = rb[k], c = rc[k];
        px[k][0] = a & 0xFFFFFFu; px[k][1] = (a >> 24) | ((b & 0xFFFFu) << 8);
        px[k][2] = (b >> 16) | ((c & 0xFFu) << 16); px[k][3] = c >> 8;
    }
    for (int j = 0; j < j0; ++j)
        dispatch_op(un, j, sl_all, [&](auto f) {
#pragma unroll
            for (int k = 0; k < NR; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) px[k][t] = f(px[k][t], r_lo + wv + 4 * k, c_lo + 4 * lane + t);
        });
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int row = wv + 4 * k;
        if (row < ph && lane < q4)
            *reinterpret_cast<uint4*>(&A[row * pw + 4 * lane]) = make_uint4(px[k][0], px[k][1], px[k][2], px[k][3]);
    }
    if (tail)                      // stored raw, the leading ops run on them in LDS below
        *reinterpret_cast<uint4*>(&A[trow * pw + 4 * tgrp]) =
            make_uint4(wa & 0xFFFFFFu, (wa >> 24) | ((wb & 0xFFFFu) << 8), (wb >> 16) | ((wc & 0xFFu) << 16), wc >> 8);
    if (WMODE == 1 && has2) {      // patches wider than 256 pixels: lanes 0..2 own a second group per row (rare, small);
#pragma unroll                     // stored raw, the leading ops run on them in LDS below
        for (int k = 0; k < NR; ++k) {
            const int row = wv + 4 * k;
            if (row < ph) {
                const uint32_t a = rd[k % NW], b = re[k % NW], c = rf[k % NW];
                *reinterpret_cast<uint4*>(&A[row * pw + 4 * (lane + 64)]) =
                    make_uint4(a & 0xFFFFFFu, (a >> 24) | ((b & 0xFFFFu) << 8), (b >> 16) | ((c & 0xFFu) << 16), c >> 8);
            }
        }
    }
    uint32_t* cur = A;
    uint32_t* oth = B;
    __syncthreads();
    // in-place pointwise ops [ja, jb) on the lane's own pixels (row = wave + 4k, column = c0 + lane + 64c) of an LDS buffer
    auto lds_ops = [&](uint32_t* buf, int ja, int jb, int c0) {
        for (int j = ja; j < jb; ++j)
            dispatch_op(un, j, sl_all, [&](auto f) {
                for (int row = wv; row < ph; row += 4)
#pragma unroll 2
                    for (int col = c0 + lane; col < pw; col += 64) {
                        const int i = row * pw + col;
                        buf[i] = f(buf[i], r_lo + row, c_lo + col);
                    }
            });
    };
    if (WIDE && pw > 256 && j0 > 0) {             // uniform per workgroup
        lds_ops(cur, 0, j0, 256);
        __syncthreads();
    }
    cur = patch_stencil_passes(un, nops, j0, Hs, Ws, r_lo, c_lo, ph, pw, cur, oth, sl_all);
    return cur;
}

// k_hist_fused: grid (ceil(Ws/256), ceil(Hs/16), units of the stage's statistics list); statistics of the image after `stage` ops
// (stage >= 1) for the "late" units -- the ones whose statistics can neither come from the pool cache nor be pushed forward.
//
// Round 4.  The round-2/3 kernel rebuilt every 256 x 16 tile in LDS through build_patch and counted all three channels of every pixel
// with LDS atomics whatever the op needed: 131 us per 168-unit batch at 1024 x 1024 (0.7 TB/s), 29 us at 512 x 512.  What the op in
// slot `stage` reads decides the work now (uniform per workgroup):
//   Contrast      only the sum of L (ImageStat mean of convert('L'))      -> per-lane sums, no histogram
//   AutoContrast  only the smallest / largest value present per channel   -> per-lane min / max; the workgroup marks those two bins
//                 (k_lut's AutoContrast branch reads nothing else of the histogram)
//   Equalize      the per-channel histograms                               -> LDS atomics into 4 interleaved copies per bin
// and what stands in front of it decides the data flow:
//   no Sharpness stencil among ops [0, stage) (pointwise ops only: byte maps, Color, Cutout; 80 % of the late units, Contrast behind
//   a byte map alone is 45 %): the tile is STREAMED -- 12-byte loads straight from the source image, the ops applied in registers,
//   nothing staged in LDS, no barrier between load and count;
//   with a stencil: the LDS patch (build_patch), as before.
enum { ST_CONTRAST = 0, ST_AUTOCONTRAST = 1, ST_EQUALIZE = 2 };
constexpr int HF_COPIES = 2;      // interleaved sub-histograms: lanes of a wave that hit one bin spread over 2 banks / addresses

template <int KIND>
struct StatAcc {
    uint32_t lsum;
    uint32_t lo_r, lo_g, lo_b, hi_r, hi_g, hi_b;
    uint32_t* sh;
    __device__ __forceinline__ StatAcc(uint32_t* sh_) : lsum(0), lo_r(255), lo_g(255), lo_b(255), hi_r(0), hi_g(0), hi_b(0),
                                                        sh(sh_ + (threadIdx.x & (HF_COPIES - 1))) {}
    __device__ __forceinline__ void add(uint32_t p) {
        const uint32_t r = p & 255u, g = (p >> 8) & 255u, b = (p >> 16) & 255u;
        if (KIND == ST_CONTRAST) lsum += rgb2l(r, g, b);
        else if (KIND == ST_AUTOCONTRAST) {
            lo_r = min(lo_r, r); hi_r = max(hi_r, r); lo_g = min(lo_g, g); hi_g = max(hi_g, g); lo_b = min(lo_b, b); hi_b = max(hi_b, b);
        } else {
            atomicAdd(&sh[r * HF_COPIES], 1u); atomicAdd(&sh[(256 + g) * HF_COPIES], 1u); atomicAdd(&sh[(512 + b) * HF_COPIES], 1u);
        }
    }
};
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}
// the workgroup's result -> this unit's row of the stage's statistics (rows are zeroed by k_luts_tables).  `red`: 32 words of LDS.
template <int KIND>
__device__ __forceinline__ void stat_flush(StatAcc<KIND>& acc, uint32_t* sh, uint32_t* red, uint32_t* gh, bool any) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (KIND == ST_CONTRAST) {
        const unsigned long long ws = wave_sum((unsigned long long)acc.lsum);
        if (lane == 0) reinterpret_cast<unsigned long long*>(red)[wv] = ws;
        __syncthreads();
        if (tid == 0) {
            const unsigned long long* r64 = reinterpret_cast<const unsigned long long*>(red);
            const unsigned long long t = r64[0] + r64[1] + r64[2] + r64[3];
            if (t) atomicAdd(reinterpret_cast<unsigned long long*>(gh + 768), t);
        }
    } else if (KIND == ST_AUTOCONTRAST) {
        uint32_t v[6] = {acc.lo_r, acc.lo_g, acc.lo_b, 255u - acc.hi_r, 255u - acc.hi_g, 255u - acc.hi_b};     // six minima
#pragma unroll
        for (int t = 0; t < 6; ++t) v[t] = wave_min_u32(v[t]);
        if (lane == 0)
#pragma unroll
            for (int t = 0; t < 6; ++t) red[wv * 6 + t] = v[t];
        __syncthreads();
        if (tid < 6 && any) {
            const uint32_t m = min(min(red[tid], red[6 + tid]), min(red[12 + tid], red[18 + tid]));
            const int c = tid % 3;
            const uint32_t bin = tid < 3 ? m : 255u - m;
            atomicOr(&gh[256 * c + bin], 1u);           // a marker: AutoContrast's map depends on the occupied extremes only
        }
    } else {
        __syncthreads();
        for (int i = tid; i < 768; i += 256) {
            const uint2 c2 = *reinterpret_cast<const uint2*>(&sh[i * HF_COPIES]);
            const uint32_t v = c2.x + c2.y;
            if (v) atomicAdd(&gh[i], v);
        }
    }
}

// A workgroup walks tiles blockIdx.x, blockIdx.x + G, ... of its unit (G = gridDim.x) with its counts in registers / LDS and adds them
// to the unit's row ONCE: the first version flushed per tile, and the 64 (512 x 512) / 256 (1024 x 1024) workgroups of a unit then
// queued on the same few words of its row -- device-scope atomics on one address complete ~0.35 us apart, which WAS the kernel's
// duration (29 / 94 us).
//
// streamed tiles: rows [ry0, ry1) x columns [cx0, cx1) (cx0, cx1 multiples of 4), ops [0, stage) all pointwise; the next tile's loads
// are in flight while the current tile is counted
struct StatRegs { uint32_t a[4], b[4], c[4]; };
__device__ __forceinline__ void stat_tile_bounds(int t, int tx, int Hs, int Ws, int& ry0, int& ry1, int& cx0, int& cx1) {
    const int by = t / tx, bx = t - by * tx;
    ry0 = by * 16; ry1 = min(ry0 + 16, Hs);
    cx0 = bx * 256; cx1 = min(cx0 + 256, Ws);
}
__device__ __forceinline__ void stat_load_tile(StatRegs& r, const uint8_t* __restrict__ src, int Ws, int ry0, int ry1, int cx0, int cx1) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q4 = (cx1 - cx0) >> 2, ph = ry1 - ry0;
    // addresses clamped into the tile: no load is conditional (what an idle lane fetched is not counted)
    const uint8_t* blk = src + ((size_t)ry0 * Ws + cx0) * 3;        // uniform
    const uint32_t goff = 12u * (uint32_t)min(lane, q4 - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const U32x3 v = *reinterpret_cast<const U32x3*>(blk + (size_t)min(wv + 4 * k, ph - 1) * (Ws * 3) + goff);
        r.a[k] = v.x; r.b[k] = v.y; r.c[k] = v.z;
    }
}
// the byte maps of stages [0, AADG_MAX_OPS) of unit u -> LDS (every stage's slot exists in the workspace: unconditional loads)
__device__ __forceinline__ void stat_stage_luts(const uint8_t* __restrict__ lut, size_t lut_stage_stride, int u, uint8_t* sl_all) {
    const int tid = threadIdx.x;
    const uint32_t* lp = reinterpret_cast<const uint32_t*>(lut + (size_t)u * 768) + min(tid, 191);
    uint32_t lreg[AADG_MAX_OPS];
#pragma unroll
    for (int j = 0; j < AADG_MAX_OPS; ++j) lreg[j] = lp[j * (lut_stage_stride >> 2)];
    if (tid < 192) {
#pragma unroll
        for (int j = 0; j < AADG_MAX_OPS; ++j) reinterpret_cast<uint32_t*>(sl_all + j * 768)[tid] = lreg[j];
    }
}
template <int KIND>
__device__ __forceinline__ void stat_stream_unit(const aadg_unit& un, int stage, const uint8_t* __restrict__ src, int Hs, int Ws,
                                                 const uint8_t* __restrict__ lut, size_t lut_stage_stride, int u,
                                                 uint8_t* sl_all, uint32_t* sh, uint32_t* red, uint32_t* gh, int t0, int G, int tend) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tx = (Ws + 255) >> 8, ntiles = min(tend, tx * ((Hs + 15) >> 4));        // tiles t0, t0 + G, ... below tend
    int t = t0;
    if (t >= ntiles) return;
    int ry0, ry1, cx0, cx1;
    stat_tile_bounds(t, tx, Hs, Ws, ry0, ry1, cx0, cx1);
    StatRegs cur;
    stat_load_tile(cur, src, Ws, ry0, ry1, cx0, cx1);
    stat_stage_luts(lut, lut_stage_stride, u, sl_all);
    if (KIND == ST_EQUALIZE)
        for (int i = tid; i < 768 * HF_COPIES; i += 256) sh[i] = 0;
    __syncthreads();
    StatAcc<KIND> acc(sh);
    while (true) {
        const int tn = t + G;
        int ny0 = 0, ny1 = 0, nx0 = 0, nx1 = 0;
        StatRegs nxt;
        if (tn < ntiles) {                                   // uniform
            stat_tile_bounds(tn, tx, Hs, Ws, ny0, ny1, nx0, nx1);
            stat_load_tile(nxt, src, Ws, ny0, ny1, nx0, nx1);
        }
        uint32_t px[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t a = cur.a[k], b = cur.b[k], c = cur.c[k];
            px[k][0] = a & 0xFFFFFFu; px[k][1] = (a >> 24) | ((b & 0xFFFFu) << 8);
            px[k][2] = (b >> 16) | ((c & 0xFFu) << 16); px[k][3] = c >> 8;
        }
        for (int j = 0; j < stage; ++j)
            dispatch_op(un, j, sl_all, [&](auto f) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) px[k][q] = f(px[k][q], ry0 + wv + 4 * k, cx0 + 4 * lane + q);
            });
        const int q4 = (cx1 - cx0) >> 2, ph = ry1 - ry0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (wv + 4 * k < ph && lane < q4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc.add(px[k][q]);
            }
        if (tn >= ntiles) break;
        t = tn; ry0 = ny0; ry1 = ny1; cx0 = nx0; cx1 = nx1; cur = nxt;
    }
    stat_flush<KIND>(acc, sh, red, gh, true);
}

// tiles behind a Sharpness stencil: the image after `stage` ops is rebuilt in LDS (halo of one pixel per stencil)
template <int KIND>
__device__ __forceinline__ void stat_patch_unit(const aadg_unit& un, int stage, int s, const uint8_t* __restrict__ src, int Hs, int Ws,
                                                uint32_t* A, uint32_t* B, const uint8_t* __restrict__ lut,
                                                size_t lut_stage_stride, int u, uint8_t* sl_all, uint32_t* sh, uint32_t* red, uint32_t* gh,
                                                int t0, int G, int tend) {
    const int tid = threadIdx.x;
    const int tx = (Ws + 255) >> 8, ntiles = min(tend, tx * ((Hs + 15) >> 4));
    if (t0 >= ntiles) return;
    if (KIND == ST_EQUALIZE)
        for (int i = tid; i < 768 * HF_COPIES; i += 256) sh[i] = 0;      // build_patch's barriers order this before the counting
    StatAcc<KIND> acc(sh);
    for (int t = t0; t < ntiles; t += G) {
        int ry0, ry1, cx0, cx1;
        stat_tile_bounds(t, tx, Hs, Ws, ry0, ry1, cx0, cx1);
        const int r_lo = max(0, ry0 - s), r_hi = min(Hs, ry1 + s);
        const int c_lo = max(0, cx0 - s) & ~3, c_hi = min(Ws, (cx1 + s + 3) & ~3);
        if (t != t0) __syncthreads();                         // the previous tile's readers are done with A / B
        const uint32_t* cur = build_patch<6, 1>(un, stage, src, Hs, Ws, r_lo, r_hi, c_lo, c_hi, A, B, lut, lut_stage_stride, u, sl_all);
        const int pw = c_hi - c_lo, rw = cx1 - cx0, n = (ry1 - ry0) * rw;
        const uint32_t* origin = cur + (ry0 - r_lo) * pw + (cx0 - c_lo);
        if (KIND == ST_EQUALIZE) {
            // Neighbouring pixels of a smooth image fall into the same bins, and LDS atomics of one wave on one address serialise: the
            // lanes of a wave take pixels 97 positions apart (97 is prime: a bijection of [0, n) unless 97 divides n); the running index
            // advances by (256 * stride) mod n -- no division per pixel; a full-width tile splits it by a shift
            const int stride = (n % 97) ? 97 : 1;
            const int step = (int)((256u * (unsigned)stride) % (unsigned)n);
            int i = (int)(((unsigned)tid * (unsigned)stride) % (unsigned)n);
            if (rw == 256) {
#pragma unroll 4
                for (int i0 = tid; i0 < n; i0 += 256, i += step, i -= i >= n ? n : 0) acc.add(origin[(i >> 8) * pw + (i & 255)]);
            } else {
                for (int i0 = tid; i0 < n; i0 += 256, i += step, i -= i >= n ? n : 0) {
                    const int row = i / rw;
                    acc.add(origin[row * pw + (i - row * rw)]);
                }
            }
        } else {
            // sums / extremes: plain row-major reads (conflict-free), wave <-> rows wave, wave + 4, ...
            const int lane = tid & 63, wv = tid >> 6;
            for (int row = wv; row < ry1 - ry0; row += 4)
#pragma unroll 4
                for (int col = lane; col < rw; col += 64) acc.add(origin[row * pw + col]);
        }
    }
    stat_flush<KIND>(acc, sh, red, gh, true);
}

// A 256 x 64 block behind exactly ONE Sharpness stencil (the common late case: op 0 = Sharpness, op 1 reads statistics), without LDS:
// wave <-> 16 rows of the block, lane <-> 4 consecutive pixels; the wave walks DOWN with the horizontal 3-sums of the previous /
// current / next row in registers (the arithmetic of build_patch's stencil pass, term for term), the neighbours of a lane's outer
// pixels come from the adjacent lanes (shuffles), those of the strip's outer columns from one extra 3-byte load per row.  The
// pointwise ops in front of the stencil are applied as the rows are loaded, the ones behind it to the stencil's outputs, and the
// result goes straight into the counts -- no patch, no barrier, 18 row loads per 16 rows instead of a 20-row patch per 16-row tile
// with one of four waves doing two of its five column chunks.
template <int KIND>
__device__ __forceinline__ void stat_strip_block(const aadg_unit& un, int stage, int js, const uint8_t* __restrict__ src, int Hs, int Ws,
                                                 int bx, int by, const uint8_t* __restrict__ lut, size_t lut_stage_stride, int u,
                                                 uint8_t* sl_all, uint32_t* sh, uint32_t* red, uint32_t* gh, bool block_wg, int t0,
                                                 int stride, int tend) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (!block_wg && t0 >= tend) return;
    stat_stage_luts(lut, lut_stage_stride, u, sl_all);
    if (KIND == ST_EQUALIZE)
        for (int i = tid; i < 768 * HF_COPIES; i += 256) sh[i] = 0;
    __syncthreads();
    StatAcc<KIND> acc(sh);
    const int tx = (Ws + 255) >> 8;
    // a block workgroup: ONE pass, wave <-> 16 rows of the 256 x 64 block; a walker (a caller's list that did not put this unit among the
    // stencil units: correct, only slower): its 16-row tiles one after the other, wave <-> 4 rows
    for (int tile = t0; tile < tend; tile += stride) {
    const int tby = tile / tx, tbx = tile - tby * tx;
    const int c0 = (block_wg ? bx : tbx) * 256, x0 = c0 + 4 * lane;
    const int r0 = block_wg ? by * 64 + wv * 16 : tby * 16 + wv * 4, r1 = min(Hs, r0 + (block_wg ? 16 : 4));
    const bool live = x0 < Ws;                                   // Ws % 4 == 0: a lane's four pixels are inside together
    const bool edge_l = lane == 0 && x0 > 0, edge_r = live && (lane == 63 || x0 + 4 >= Ws) && x0 + 4 < Ws;
    const float alpha = unit_farg(un, js);
    // One source row = three loads per lane: its four pixels (12 bytes) and the dwords that hold the pixel to their left / right (only the
    // strip's outer lanes use them; the others read a word of their own pixels: no load is conditional).  The rows are fetched FOUR AHEAD
    // of the walk: a wave that waited for each row before it asked for the next spent 18 memory round trips per 16 rows.
    struct RowRaw { U32x3 m; uint32_t l, r; };
    const int x0c = min(x0, Ws - 4);
    const int loff = edge_l ? -4 : 0, roff = edge_r ? 11 : 8;
    auto fetch_row = [&](RowRaw& R, int y) {
        const uint8_t* rowp = src + ((size_t)min(max(y, 0), Hs - 1) * Ws + x0c) * 3;
        R.m = *reinterpret_cast<const U32x3*>(rowp);
        R.l = reinterpret_cast<const UnalignedU32*>(rowp + loff)->v;
        R.r = reinterpret_cast<const UnalignedU32*>(rowp + roff)->v;
    };
    // the lane's four pixels after ops [0, js), the pixel to their left and to their right
    auto decode_row = [&](const RowRaw& R, int y, uint32_t (&p)[4], uint32_t& pl, uint32_t& pr) {
        const uint32_t a = R.m.x, b = R.m.y, c = R.m.z;
        uint32_t el = R.l >> 8, er = R.r >> 8;
        p[0] = a & 0xFFFFFFu; p[1] = (a >> 24) | ((b & 0xFFFFu) << 8);
        p[2] = (b >> 16) | ((c & 0xFFu) << 16); p[3] = c >> 8;
        for (int j = 0; j < js; ++j)
            dispatch_op(un, j, sl_all, [&](auto f) {
#pragma unroll
                for (int t = 0; t < 4; ++t) p[t] = f(p[t], y, x0 + t);
                el = f(el, y, x0 - 1);
                er = f(er, y, x0 + 4);
            });
        const uint32_t from_l = (uint32_t)__shfl_up((int)p[3], 1, 64), from_r = (uint32_t)__shfl_down((int)p[0], 1, 64);
        pl = edge_l ? el : from_l;
        pr = edge_r ? er : from_r;
    };
    auto hsums = [&](const uint32_t (&p)[4], uint32_t pl, uint32_t pr, uint32_t (&hrb)[4], uint32_t (&hg)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t q0 = t == 0 ? pl : p[t - 1], q1 = p[t], q2 = t == 3 ? pr : p[t + 1];
            hrb[t] = (q0 & 0xFF00FFu) + (q1 & 0xFF00FFu) + (q2 & 0xFF00FFu);
            hg[t] = ((q0 >> 8) & 255u) + ((q1 >> 8) & 255u) + ((q2 >> 8) & 255u);
        }
    };
    if (r0 < r1) {                                               // uniform per wave
        // events e = 0 .. ne - 1 <-> source rows r0 - 1 + e (rows outside the image: a clamped row whose values the border rule never uses);
        // event e >= 2 completes the 3 x 3 neighbourhood of row r0 + e - 2
        const int ne = r1 - r0 + 2;
        uint32_t h0rb[4] = {0, 0, 0, 0}, h0g[4] = {0, 0, 0, 0}, h1rb[4] = {0, 0, 0, 0}, h1g[4] = {0, 0, 0, 0}, h2rb[4], h2g[4];
        uint32_t p1[4] = {0, 0, 0, 0}, p2[4], pl, pr;
        RowRaw buf[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) fetch_row(buf[q], r0 - 1 + q);
        for (int e0 = 0; e0 < ne; e0 += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q;
                if (e < ne) {                                    // uniform
                    const int yn = r0 - 1 + e;
                    decode_row(buf[q], yn, p2, pl, pr);
                    fetch_row(buf[q], yn + 4);
                    hsums(p2, pl, pr, h2rb, h2g);
                    if (e >= 2) {
                        const int y = yn - 1;
                        uint32_t o[4];
                        const bool row_in = y > 0 && y < Hs - 1;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int x = x0 + t;
                            const bool in = row_in && x > 0 && x < Ws - 1;   // ImageFilter.SMOOTH copies the 1-pixel image border
                            const uint32_t srb = h0rb[t] + h1rb[t] + h2rb[t] + 4u * (p1[t] & 0xFF00FFu), sg = h0g[t] + h1g[t] + h2g[t] + 4u * ((p1[t] >> 8) & 255u);
                            const uint32_t cr = (uint32_t)__mul24((int)((srb & 0xFFFFu) + 6u), 5042) >> 16, cb = (uint32_t)__mul24((int)((srb >> 16) + 6u), 5042) >> 16,
                                           cg = (uint32_t)__mul24((int)(sg + 6u), 5042) >> 16;
                            const uint32_t d = in ? (cr | (cg << 8) | (cb << 16)) : p1[t];
                            o[t] = blend3(d, p1[t], alpha, true);
                        }
                        for (int j = js + 1; j < stage; ++j)
                            dispatch_op(un, j, sl_all, [&](auto f) {
#pragma unroll
                                for (int t = 0; t < 4; ++t) o[t] = f(o[t], y, x0 + t);
                            });
                        if (live) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) acc.add(o[t]);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) { h0rb[t] = h1rb[t]; h0g[t] = h1g[t]; h1rb[t] = h2rb[t]; h1g[t] = h2g[t]; p1[t] = p2[t]; }
                }
            }
        }
    }
    if (block_wg) break;
    }
    stat_flush<KIND>(acc, sh, red, gh, true);
}

constexpr int HF_PATCH = 5280;        // 20 rows (16 + two stencil halos) x 264 columns (256 + halos, rounded to groups of 4)
// 1-D grid.  The first n_sten entries of the list are units with a stencil in front of the op (aadg_aug_u8_plan puts them first): each
// 256 x 64 block of theirs is a workgroup of its own (workgroups [0, n_sten * blocks): the long work is dispatched first) -- one stencil:
// the register walk above; more than one: the block's four 16-row tiles through the LDS patch --; the other units get G workgroups each
// that walk their 256 x 16 tiles with stride G.  (All units handled like the second kind: a stencil unit's 256 tiles behind ~30
// workgroups were the long pole of the launch -- 129 us against 94 us per 1024 x 1024 batch.)
// PATCH: the instantiation for slots k >= 2 carries the two LDS patch buffers (two stencils in front of the op need k >= 2); slot 1 -- the
// only late slot of the reference's configurations (CONTROLLER.L = 2) -- runs the instantiation without them: 9 KB of LDS instead of
// 51, i.e. the streamed tiles are not held to three workgroups per CU by buffers they never touch.
template <bool PATCH>
__global__ __launch_bounds__(256) void k_hist_fused(const uint8_t* __restrict__ pool, const aadg_unit* __restrict__ units,
                                                    const int* __restrict__ ulist, int n_sten, int G, int stage, int Hs, int Ws, int crop,
                                                    const uint8_t* __restrict__ lut, size_t lut_stage_stride, uint32_t* hist) {
    const int tx = (Ws + 255) >> 8, ntiles = tx * ((Hs + 15) >> 4), nblocks = tx * ((Hs + 63) >> 6);
    int id = blockIdx.x, slot, t0, stride, tend = ntiles, bx = 0, by = 0;
    const bool block_wg = id < n_sten * nblocks;
    if (block_wg) {
        slot = id / nblocks;
        const int blk = id - slot * nblocks;
        by = blk / tx; bx = blk - by * tx;
        t0 = by * 4 * tx + bx; stride = tx; tend = min(ntiles, t0 + 4 * tx);       // the block's four 16-row tiles
    } else {
        id -= n_sten * nblocks;
        slot = id / G; t0 = id - slot * G; stride = G; slot += n_sten;
    }
    const int u = ulist != nullptr ? ulist[slot] : slot;          // ulist: the units whose op `stage` needs statistics
    const aadg_unit& un = units[u];
    // the head of the record in ONE batch of scalar loads, every predicate from those registers, one exit
    const int n_ops = un.n_ops, u_src = un.src, sw = un.scaled_w, shh = un.scaled_h;
    int ops[AADG_MAX_OPS];
    bool sten[AADG_MAX_OPS];
#pragma unroll
    for (int k = 0; k < AADG_MAX_OPS; ++k) {                    // (no short circuit: farg[k] is loaded whatever op[k] is)
        ops[k] = un.op[k];
        const float fk = un.farg[k];
        sten[k] = (ops[k] == AADG_OP_SHARPNESS) & (fk != 1.0f);
    }
    const int op = stage == 0 ? ops[0] : (stage == 1 ? ops[1] : (stage == 2 ? ops[2] : ops[3]));
    int sc_all = 0, s = 0, js = 0;
    bool lut_before = true;
#pragma unroll
    for (int k = AADG_MAX_OPS - 1; k >= 0; --k) {
        sc_all += (k < n_ops && sten[k]) ? 1 : 0;
        s += (k < stage && sten[k]) ? 1 : 0;
        js = (k < stage && sten[k]) ? k : js;                   // the first stencil's slot
        lut_before = lut_before && (k >= stage || op_is_lut(ops[k]));
    }
    const bool fusable = !(Ws & 3) && !(crop & 3) && sc_all <= MAX_SHARP && 2 * sw >= Ws && 2 * shh >= Hs;      // unit_flow(...) != FLOW_STAGED
    const bool pushfwd = stage >= 1 && (op == AADG_OP_AUTOCONTRAST || op == AADG_OP_EQUALIZE) && lut_before;      // stats_by_pushforward: k_lut derives it
    if (n_ops <= stage || !op_needs_stats(op) || !fusable || pushfwd) return;
    __shared__ __attribute__((aligned(16))) uint32_t A[PATCH ? HF_PATCH : 4];
    __shared__ __attribute__((aligned(16))) uint32_t B[PATCH ? HF_PATCH : 4];
    __shared__ __attribute__((aligned(16))) uint8_t sl[AADG_MAX_OPS * 768];
    __shared__ __attribute__((aligned(16))) uint32_t red[32];
    __shared__ __attribute__((aligned(16))) uint32_t shx[768 * HF_COPIES];      // Equalize: accumulates over the workgroup's tiles
    const uint8_t* src = pool + (size_t)u_src * Hs * Ws * 3;
    uint32_t* gh = hist + (size_t)u * HIST_STRIDE;
    if (s == 0) {
        if (op == AADG_OP_CONTRAST) stat_stream_unit<ST_CONTRAST>(un, stage, src, Hs, Ws, lut, lut_stage_stride, u, sl, shx, red, gh, t0, stride, tend);
        else if (op == AADG_OP_AUTOCONTRAST) stat_stream_unit<ST_AUTOCONTRAST>(un, stage, src, Hs, Ws, lut, lut_stage_stride, u, sl, shx, red, gh, t0, stride, tend);
        else stat_stream_unit<ST_EQUALIZE>(un, stage, src, Hs, Ws, lut, lut_stage_stride, u, sl, shx, red, gh, t0, stride, tend);
    } else if (s == 1) {
        if (op == AADG_OP_CONTRAST) stat_strip_block<ST_CONTRAST>(un, stage, js, src, Hs, Ws, bx, by, lut, lut_stage_stride, u, sl, shx, red, gh, block_wg, t0, stride, tend);
        else if (op == AADG_OP_AUTOCONTRAST) stat_strip_block<ST_AUTOCONTRAST>(un, stage, js, src, Hs, Ws, bx, by, lut, lut_stage_stride, u, sl, shx, red, gh, block_wg, t0, stride, tend);
        else stat_strip_block<ST_EQUALIZE>(un, stage, js, src, Hs, Ws, bx, by, lut, lut_stage_stride, u, sl, shx, red, gh, block_wg, t0, stride, tend);
    } else if (PATCH) {
        if (op == AADG_OP_CONTRAST) stat_patch_unit<ST_CONTRAST>(un, stage, s, src, Hs, Ws, A, B, lut, lut_stage_stride, u, sl, shx, red, gh, t0, stride, tend);
        else if (op == AADG_OP_AUTOCONTRAST) stat_patch_unit<ST_AUTOCONTRAST>(un, stage, s, src, Hs, Ws, A, B, lut, lut_stage_stride, u, sl, shx, red, gh, t0, stride, tend);
        else stat_patch_unit<ST_EQUALIZE>(un, stage, s, src, Hs, Ws, A, B, lut, lut_stage_stride, u, sl, shx, red, gh, t0, stride, tend);
    }
}
// launch of the statistics pass of one stage: nstat list entries, the first n_sten of them stencil units (n_sten = nstat: no list
// order known -- every unit's tiles get their own workgroups)
static inline int launch_hist_fused(const uint8_t* pool, const aadg_unit* units, const int* ulist, int nstat, int n_sten, int stage, int Hs,
                                    int Ws, int crop, const uint8_t* lut, size_t lut_stage_stride, uint32_t* hist, hipStream_t st) {
    const int ntiles = ((Ws + 255) / 256) * ((Hs + 15) / 16), nblocks = ((Ws + 255) / 256) * ((Hs + 63) / 64);
    const int n_rest = nstat - n_sten;
    // the other units: enough workgroups to fill what the stencil blocks leave of the chip's 768 slots (3 per CU), and never more than
    // 8 tiles per workgroup (the walkers must not outlast the stencil blocks -- with 8 workgroups per unit, 32 tiles each at
    // 1024 x 1024, they did: 149 us per batch)
    int G = 1;
    if (n_rest > 0) {
        const long long room = (stage >= 2 ? 768 : 1536) - (long long)n_sten * nblocks;       // 3 / 6 workgroups per CU
        G = (int)((room > 0 ? room : 0) / n_rest);
        const int g_min = (ntiles + 7) / 8;
        G = G < g_min ? g_min : G;
        G = G > ntiles ? ntiles : G;
    }
    const long long grid = (long long)n_sten * nblocks + (long long)n_rest * G;
    if (grid <= 0 || grid > 0x7FFFFFFFll) return AADG_E_BADARG;
    if (stage >= 2)
        hipLaunchKernelGGL(k_hist_fused<true>, dim3((unsigned)grid), dim3(256), 0, st, pool, units, ulist, n_sten, G, stage, Hs, Ws, crop, lut, lut_stage_stride, hist);
    else
        hipLaunchKernelGGL(k_hist_fused<false>, dim3((unsigned)grid), dim3(256), 0, st, pool, units, ulist, n_sten, G, stage, Hs, Ws, crop, lut, lut_stage_stride, hist);
    AADG_LAUNCH_CHECK();
    return 0;
}

// np.float32(v) / 127.5 - 1.0 for an integer 0 <= v <= 255, bit-exact without a division: one Newton step on
// the reciprocal product is correctly rounded for all 256 inputs (checked exhaustively, tests + DESIGN.md).
__device__ __forceinline__ float normalise_u8(int v) {
    const float x = (float)v;
    const float r = 0.00784313725490196f;                 // float32(1 / 127.5)
    const float q0 = __fmul_rn(x, r);
    const float e = __fmaf_rn(-q0, 127.5f, x);
    return __fsub_rn(__fmaf_rn(e, r, q0), 1.0f);
}

// ------------------------------------------------------------------------------------------------
// Down-scaling ("generic") units: any axis scale in [1/2, 1) (Pillow's antialiased BILINEAR: up to 5 taps when shrinking by 2).
// ------------------------------------------------------------------------------------------------
constexpr int GT_TAPS = 5;

__device__ __forceinline__ int axis_taps(int inSize, int outSize) {
    if (inSize == outSize) return 1;
    if (outSize > inSize) return 2;
    // down-scaling by at most 2: Pillow allocates ksize = 2 * ceil(support) + 1 = 5 coefficients per output, but only
    // xmax - xmin = floor(c + s + 0.5) - floor(c - s + 0.5) <= floor(2 s) + 1 of them are used (support s = inSize / outSize): 3 while
    // s < 1.5, 4 while s < 2.  The comparisons are strict (s at least 1 / outSize below the bound), so the rounding of the double
    // expressions cannot add a tap; exactly s = 2 keeps all five slots.
    if (3 * outSize > 2 * inSize) return 3;
    if (2 * outSize > inSize) return 4;
    return GT_TAPS;
}

// ------------------------------------------------------------------------------------------------
// GENERIC units in two passes (round 3): Pillow's own order -- the horizontal pass over every source row the crop window needs,
// rounded to uint8, then the vertical pass -- as two streaming kernels around a packed RGBX intermediate in the workspace
// (`hbuf`: [slot][source row][output column] uint32, <= 4 B per source-row x output-column, written once and read ~nty/scale times
// out of L2 / the Infinity Cache).
//   * k_gen_hpass: tile = GH_CB output columns x GH_ROWS source rows.  The op chain runs in the LDS patch (build_patch, shared with
//     the other tile kernels: 256-column patches, so every lane of the patch loader has work), a thread owns an output column and
//     walks down the patch rows.  No vertical halo: a source row is resampled exactly once (the one-pass tile resamples the
//     (2*16+4)/32 rows of a 16-row output tile, stages 36 % lane-idle 144-column patches, and 30 % of its workgroups are padding).
//   * k_gen_vpass: wave <-> output row, lane <-> 4 consecutive columns; the <= 5 taps are 16-byte loads of the intermediate, the row
//     tables are scalar loads; normalise through the LDS table, NEAREST mask gather, multilabel planes, streaming 16-byte stores;
//     rows / columns of the padding are constant stores.
// (The one-pass tile of rounds 1-2, k_fused_generic, is gone since round 4; its measurements are in DESIGN.md section 4.)
// ------------------------------------------------------------------------------------------------
constexpr int GH_CB = 128;             // output columns per horizontal-pass tile: <= 2 * 128 + 5 source columns (+ halo, alignment) <= 272
constexpr int GH_NR = 4;               // patch rows per wave of build_patch: 4 * GH_NR >= GH_PATCH_ROWS
constexpr int GH_PATCH_ROWS = 16;      // patch rows; a tile covers 16 - 2 * (stencils of the unit) source rows.  Measured on a fixed 144-unit
                                       // 1024 x 1024 mix (plain / stencil launch): 22 rows 144 / 93 us, 16 rows 141 / 86 us (20 KB of LDS: 7
                                       // workgroups per CU), 12 rows 149 / 93 us
constexpr int GH_CAP = GH_PATCH_ROWS * 272 + 8;      // patch words (272 = widest patch) + the slack the last row's taps may read
template <bool SHARP> struct GhRows { static constexpr int value = SHARP ? GH_PATCH_ROWS - 2 * MAX_SHARP : GH_PATCH_ROWS; };   // fewest rows per tile (grid size)

// coefficient k (<= 2^22) -> 4k as an unsigned 24-bit multiplier.  4k = 2^24 only for a single tap of weight one (k1 <= 1):
// with 2^24 - 1 the sum is c0 * 2^24 + (2^23 + 4 c1 k1 - c0), whose byte 3 is still c0 because 0 < 2^23 + 4 c1 k1 - c0 < 2^24.
__device__ __forceinline__ uint32_t prescale4(int k) {
    const uint32_t q = (uint32_t)k << 2;
    return q > 0xFFFFFFu ? 0xFFFFFFu : q;
}

// one output of the horizontal pass: coefficients scaled by 4 (prescale4), so that byte 3 of each 32-bit sum IS the rounded channel
// value ((2^21 + sum c k) >> 22, never above 255 for non-negative coefficients that sum to 2^22) and two v_perm_b32 pack the word
template <int NT>
__device__ __forceinline__ uint32_t hpass_px(const uint32_t* rowp, const uint32_t* hk) {
    uint32_t s0 = 1u << 23, s1 = s0, s2 = s0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint32_t p = rowp[t];                          // taps beyond the patch have zero coefficients: whatever word lies there
                                                             // (the next row, or the 8 words of slack behind the buffer) is multiplied by 0
        s0 += __umul24(p & 255u, hk[t]);
        s1 += __umul24((p >> 8) & 255u, hk[t]);
        s2 += __umul24((p >> 16) & 255u, hk[t]);
    }
    const uint32_t rg = __builtin_amdgcn_perm(s1, s0, 0x0c0c0703u);      // [s0.b3, s1.b3, 0, 0]
    return __builtin_amdgcn_perm(s2, rg, 0x0c070100u);                  // [rg.b0, rg.b1, s2.b3, 0]
}

// ---- plain units (no stencil): GH_NB consecutive row blocks per workgroup, software-pipelined ----
// A horizontal-pass workgroup is a chain of dependent round trips -- unit record -> tables -> source rows -> LDS -> stores -- of ~8 us for
// 2048 outputs, and a CU already holds 7 of them (28 of 32 waves): the kernel ran at a third of what its traffic allows.  The plain variant
// therefore walks GH_NB row blocks of its column tile: the unit record, the tables, the taps and the byte maps are read once, and the source
// rows of block i + 1 are in flight (registers) while block i goes through LDS -- plain loads survive __syncthreads().
constexpr int GH_NB = 1;
struct GhRegs { uint32_t a[GH_NR], b[GH_NR], c[GH_NR], wa, wb, wc; };

// the loads of one row block: rows [0, ph) x pixel groups [0, q4) of the block at blk (uniform pointer: first row, column c_lo);
// thread <-> (row = wave + 4k, group = lane), the groups beyond 64 (patches wider than 256 pixels: uniform `wide`) by thread
// t <-> (row t >> 2, group 64 + (t & 3)).  Addresses are clamped into the block: no load is conditional per lane; a uniform base and a
// 32-bit lane offset per load.  Three dwords per pixel group (typed as one 12-byte vector).
__device__ __forceinline__ void gh_fetch(GhRegs& R, const uint8_t* __restrict__ blk, int row_bytes, int ph, int q4, bool wide) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t goff = 12u * (uint32_t)min(lane, q4 - 1);
#pragma unroll
    for (int k = 0; k < GH_NR; ++k) {
        const uint8_t* rowp = blk + (size_t)min(wv + 4 * k, ph - 1) * row_bytes;      // uniform
        const U32x3 v = *reinterpret_cast<const U32x3*>(rowp + goff);
        R.a[k] = v.x; R.b[k] = v.y; R.c[k] = v.z;
    }
    if (wide) {
        const uint32_t toff = (uint32_t)min(tid >> 2, ph - 1) * (uint32_t)row_bytes + 12u * (uint32_t)min(64 + (tid & 3), q4 - 1);
        const U32x3 v = *reinterpret_cast<const U32x3*>(blk + toff);
        R.wa = v.x; R.wb = v.y; R.wc = v.z;
    }
}

// registers of a fetched block -> RGBX pixels of the lane (px: rows wave + 4k, pt: the group beyond column 256)
__device__ __forceinline__ void gh_unpack(const GhRegs& R, uint32_t (&px)[GH_NR][4], uint32_t (&pt)[4]) {
#pragma unroll
    for (int k = 0; k < GH_NR; ++k) {
        const uint32_t a = R.a[k], b = R.b[k], c = R.c[k];
        px[k][0] = a & 0xFFFFFFu; px[k][1] = (a >> 24) | ((b & 0xFFFFu) << 8);
        px[k][2] = (b >> 16) | ((c & 0xFFu) << 16); px[k][3] = c >> 8;
    }
    pt[0] = R.wa & 0xFFFFFFu; pt[1] = (R.wa >> 24) | ((R.wb & 0xFFFFu) << 8);
    pt[2] = (R.wb >> 16) | ((R.wc & 0xFFu) << 16); pt[3] = R.wc >> 8;
}

// pixels -> leading pointwise ops [0, nops) -> LDS patch A[ph][pw]; ends with a barrier
__device__ __forceinline__ void gh_stage(uint32_t (&px)[GH_NR][4], const uint32_t (&pt)[4], const aadg_unit& un, int nops, int r0, int ph,
                                         int c_lo, int pw, uint32_t* A, const uint8_t* sl_all) {
    const int tid = threadIdx.x, lane = tid & 63, q4 = pw >> 2;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int j = 0; j < nops; ++j)
        dispatch_op(un, j, sl_all, [&](auto f) {
#pragma unroll
            for (int k = 0; k < GH_NR; ++k) {
#pragma unroll
                for (int t = 0; t < 4; ++t) px[k][t] = f(px[k][t], r0 + wv + 4 * k, c_lo + 4 * lane + t);
                __builtin_amdgcn_sched_barrier(0);              // one row's byte-map reads in flight at a time (registers)
            }
        });
    uint32_t* Al = A + 4 * lane;
#pragma unroll
    for (int k = 0; k < GH_NR; ++k) {
        const int row = wv + 4 * k;                             // uniform
        if (row < ph && lane < q4) *reinterpret_cast<uint4*>(Al + row * pw) = make_uint4(px[k][0], px[k][1], px[k][2], px[k][3]);
    }
    const bool wide = pw > 256;                                // uniform
    if (wide) {
        const int trow = tid >> 2, tgrp = 64 + (tid & 3);
        if (trow < ph && tgrp < q4)                            // stored raw, the ops run on them in LDS below
            *reinterpret_cast<uint4*>(&A[trow * pw + 4 * tgrp]) = make_uint4(pt[0], pt[1], pt[2], pt[3]);
    }
    __syncthreads();
    if (wide && nops > 0) {
        for (int j = 0; j < nops; ++j)
            dispatch_op(un, j, sl_all, [&](auto f) {
                for (int row = wv; row < ph; row += 4)
                    for (int col = 256 + lane; col < pw; col += 64) {
                        const int i = row * pw + col;
                        A[i] = f(A[i], r0 + row, c_lo + col);
                    }
            });
        __syncthreads();
    }
}

template <bool SHARP>
__device__ __forceinline__ void gen_hpass_pipe(const uint8_t* __restrict__ pool, const aadg_unit* __restrict__ units,
                                               const int* __restrict__ order, int slot, int bx, int by0, int Hs, int Ws, int crop,
                                               const int* __restrict__ tab, const uint8_t* __restrict__ lut,
                                               size_t lut_stage_stride, uint32_t* __restrict__ hbuf, int hslot0, uint32_t* A, uint32_t* Bs,
                                               uint8_t* sl) {
    const int u = order != nullptr ? order[slot] : slot;
    const aadg_unit& un = units[u];
    // the record's fields in one batch of scalar loads
    const int n_ops = un.n_ops, w = un.scaled_w, h = un.scaled_h, u_pad = un.pad, u_cx = un.crop_x, u_cy = un.crop_y, u_src = un.src;
    const int sc = sharp_count4(un, n_ops);
    if ((Ws & 3) || (crop & 3) || sc > MAX_SHARP || (w >= Ws && h >= Hs) || 2 * w < Ws || 2 * h < Hs) return;      // not a FLOW_GENERIC unit
    if ((sc > 0) != SHARP) return;                              // the other variant owns this unit
    if (!SHARP && h >= Hs && Ws >= 8) return;                   // shrinks the width only: k_fused3w's unit (one pass)
    const int s = SHARP ? sc : 0;
    const int ROWS = GH_PATCH_ROWS - 2 * s;                     // the patch holds the block's rows + the stencils' halo
    const int tid = threadIdx.x;
    const int* base = tab + (size_t)u * crop * TAB_STRIDE;
    const int* xmin_t = base;
    const int* xk_t = xmin_t + crop;
    const int* ymin_t = xk_t + (size_t)crop * KMAX;
    const int ox = u_cx - u_pad, oy = u_cy - u_pad;
    const int ufx = max(0, -ox), ulx = min(crop - 1, w - 1 - ox);
    const int ufy = max(0, -oy), uly = min(crop - 1, h - 1 - oy);
    if (ufx > ulx || ufy > uly) return;
    const int x0 = bx * GH_CB;
    const int fx = max(x0, ufx), lx = min(x0 + GH_CB - 1, ulx);
    if (fx > lx) return;
    const int ntx = axis_taps(Ws, w), nty = axis_taps(Hs, h);
    // one batch of scalar loads: the row range of the unit, the column range of the tile
    const int t_rlo = ymin_t[ufy], t_rhi = ymin_t[uly], t_clo = xmin_t[fx], t_chi = xmin_t[lx];
    // the byte maps of the LUT-class ops (every stage's slot exists in the workspace: no load is conditional) and the taps of the
    // thread's output column: issued before the first wait
    uint32_t lreg[AADG_MAX_OPS];
    {
        const uint32_t* lp = reinterpret_cast<const uint32_t*>(lut + (size_t)u * 768) + min(tid, 191);
#pragma unroll
        for (int j = 0; j < AADG_MAX_OPS; ++j) lreg[j] = lp[j * (lut_stage_stride >> 2)];
    }
    // horizontal pass: thread <-> (output column hc, row parity hg: uniform per wave)
    const int hc = tid & (GH_CB - 1);
    const int hg = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int xh = x0 + hc;
    const bool col_ok = xh >= fx && xh <= lx;
    const int xc = min(max(xh, fx), lx);
    int hxm = xmin_t[xc];
    int kraw[GT_TAPS];
#pragma unroll
    for (int t = 0; t < GT_TAPS; ++t) kraw[t] = xk_t[(size_t)xc * KMAX + t];
    const int ur_lo = t_rlo, ur_hi = min(Hs, t_rhi + nty);
    const int r_first = ur_lo + by0 * GH_NB * ROWS;
    if (r_first >= ur_hi) return;
    // patch columns: the taps' range + the stencils' halo, clipped to the image, in groups of 4
    const int c_lo = max(0, t_clo - s) & ~3, c_hi4 = min(Ws, (min(Ws, t_chi + ntx) + s + 3) & ~3);
    const int pw = c_hi4 - c_lo, q4 = pw >> 2;
    const bool wide = pw > 256;
    const int row_bytes = Ws * 3;
    const uint8_t* img = pool + (size_t)u_src * Hs * row_bytes + (size_t)c_lo * 3;     // uniform: row 0 of the source, column c_lo
    // patch rows of the block that starts at source row r: [max(0, r - s), min(Hs, min(ur_hi, r + ROWS) + s))
    auto blk_lo = [&](int r) { return max(0, r - s); };
    auto blk_hi = [&](int r) { return min(Hs, min(ur_hi, r + ROWS) + s); };
    GhRegs R;
    R.wa = R.wb = R.wc = 0;
    gh_fetch(R, img + (size_t)blk_lo(r_first) * row_bytes, row_bytes, blk_hi(r_first) - blk_lo(r_first), q4, wide);
    uint32_t hk[GT_TAPS];
#pragma unroll
    for (int t = 0; t < GT_TAPS; ++t) hk[t] = t < ntx ? prescale4(kraw[t]) : 0u;
    hxm -= c_lo;                                                // LDS column of the first tap
    if (tid < 192) {
#pragma unroll
        for (int j = 0; j < AADG_MAX_OPS; ++j) reinterpret_cast<uint32_t*>(sl + j * 768)[tid] = lreg[j];
    }
    int j0 = n_ops;                                             // leading pointwise segment [0, j0)
    if (SHARP) {
        j0 = 0;
        while (j0 < n_ops && !is_stencil(un, j0)) ++j0;
    }
    // (the intermediate holds ONE chunk of list slots: slot s of a chunk that starts at hslot0 lives at index s - hslot0)
    uint32_t* hrow = hbuf + ((size_t)(slot - hslot0) * Hs + r_first) * crop + x0;      // uniform: row r_lo of the intermediate, column x0
    __syncthreads();                                            // the byte maps are in LDS
#pragma unroll 1
    for (int r_lo = r_first, i = 0; i < GH_NB && r_lo < ur_hi; ++i, r_lo += ROWS) {
        const int nrows = min(ROWS, ur_hi - r_lo);
        const int p_lo = blk_lo(r_lo), ph = blk_hi(r_lo) - p_lo;
        uint32_t px[GH_NR][4], pt[4];
        gh_unpack(R, px, pt);
        // the next block's rows are in flight while this one goes through LDS (plain loads survive the barriers)
        const int r_nx = r_lo + ROWS;
        if (i + 1 < GH_NB && r_nx < ur_hi) gh_fetch(R, img + (size_t)blk_lo(r_nx) * row_bytes, row_bytes, blk_hi(r_nx) - blk_lo(r_nx), q4, wide);
        gh_stage(px, pt, un, j0, p_lo, ph, c_lo, pw, A, sl);
        const uint32_t* cur = A;
        if (SHARP) cur = patch_stencil_passes(un, n_ops, j0, Hs, Ws, p_lo, c_lo, ph, pw, A, Bs, sl);
        if (col_ok) {
            const uint32_t* colp = cur + (r_lo - p_lo) * pw + hxm;
#define AADG_HPASS_ROWS(NT)                                                                                              \
    _Pragma("unroll 2") for (int rr = hg; rr < nrows; rr += 2) (hrow + (size_t)rr * crop)[hc] = hpass_px<NT>(colp + rr * pw, hk)
            switch (ntx) {                                        // uniform per unit
                case 1: AADG_HPASS_ROWS(1); break;
                case 2: AADG_HPASS_ROWS(2); break;
                case 3: AADG_HPASS_ROWS(3); break;
                case 4: AADG_HPASS_ROWS(4); break;
                default: AADG_HPASS_ROWS(GT_TAPS); break;
            }
#undef AADG_HPASS_ROWS
        }
        hrow += (size_t)ROWS * crop;
        __syncthreads();                                        // the patch is free for the next block
    }
}

template <bool SHARP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SHARP ? 4 : 5))) void k_gen_hpass(const uint8_t* __restrict__ pool, const aadg_unit* __restrict__ units,
                                                   const int* __restrict__ order, int slot0, int Hs, int Ws, int crop,
                                                   const int* __restrict__ tab, const uint8_t* __restrict__ lut,
                                                   size_t lut_stage_stride, uint32_t* __restrict__ hbuf, int hslot0) {
    __shared__ __attribute__((aligned(16))) uint32_t A[GH_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t Bs[SHARP ? GH_CAP : 4];
    __shared__ __attribute__((aligned(16))) uint8_t sl[AADG_MAX_OPS * 768];
    gen_hpass_pipe<SHARP>(pool, units, order, slot0 + blockIdx.z, blockIdx.x, blockIdx.y, Hs, Ws, crop, tab, lut, lut_stage_stride, hbuf, hslot0, A, Bs, sl);
}

constexpr int GV_ROWS = 16;            // output rows per vertical-pass workgroup: 4 consecutive rows per wave
constexpr int GV_G = 4;                // rows whose loads are issued together (2: 250 us, 4: 245 us; 8 or 32 rows per workgroup: 267 / 282 us)

// GV_G consecutive output rows of a lane's 4 columns: all loads of these rows (intermediate rows at clamped indices, the mask window)
// are issued before the first one is used -- a wave has nothing else to cover their latency with
struct VRow { int y, vym, vyn; int vk[GT_TAPS]; };

template <int NT>
__device__ __forceinline__ void vpass_pair(const VRow (&row)[GV_G], int nrows_live, const uint32_t* __restrict__ hcol, const uint8_t* __restrict__ msk,
                                           int Hs, int Ws, int crop, const int* xm, const int* xn, bool win_ok, bool any_col, bool optic, int K,
                                           const float* lutf, float* oi, float* ol, size_t plane, int xq, bool stream_out) {
    const float padv = -1.0f;
    bool any_row = false, any_msk = false;                                     // uniform
#pragma unroll
    for (int j = 0; j < GV_G; ++j) { any_row = any_row || row[j].vym >= 0; any_msk = any_msk || row[j].vyn >= 0; }
    any_row = any_row && any_col;
    uint4 hv[GV_G][NT];
    uint32_t mlo[GV_G], mhi[GV_G];
#pragma unroll
    for (int j = 0; j < GV_G; ++j) mlo[j] = mhi[j] = 0u;
    if (any_row) {
#pragma unroll
        for (int j = 0; j < GV_G; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int hr = min(max(row[j].vym, 0) + t, Hs - 1);           // zero coefficient beyond the last source row
                hv[j][t] = *reinterpret_cast<const uint4*>(hcol + (size_t)hr * crop);
            }
    }
    // NEAREST mask bytes: no axis shrinks by more than 2, so the 4 source columns of a lane lie within 8 consecutive bytes -- two
    // unaligned dword loads instead of four byte gathers (a lane at the row's end, or with a pad column, takes the byte loads below)
    const int mbase = min(max(xn[0], 0), Ws - 8);
    if (any_msk) {
#pragma unroll
        for (int j = 0; j < GV_G; ++j) {
            const uint8_t* mrow = msk + (size_t)max(row[j].vyn, 0) * Ws + mbase;
            mlo[j] = reinterpret_cast<const UnalignedU32*>(mrow)->v;
            mhi[j] = reinterpret_cast<const UnalignedU32*>(mrow + 4)->v;
        }
    }
#pragma unroll
    for (int j = 0; j < GV_G; ++j) {
        if (j >= nrows_live) break;                                            // uniform: the last rows of the image
        const VRow& rw = row[j];
        uint32_t mv[4] = {0u, 0u, 0u, 0u};
        if (rw.vyn >= 0) {
            if (win_ok) {
                const unsigned long long win = ((unsigned long long)mhi[j] << 32) | mlo[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) mv[i] = (uint32_t)(win >> (8 * (xn[i] - mbase))) & 255u;
            } else {
                const uint8_t* mrow = msk + (size_t)rw.vyn * Ws;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (xn[i] >= 0) mv[i] = mrow[xn[i]];
            }
        }
        float o[3][4];
        if (rw.vym >= 0 && any_col) {
            int acc[4][3];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = acc[i][2] = 1 << (PRECISION_BITS - 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint32_t a[4] = {hv[j][t].x, hv[j][t].y, hv[j][t].z, hv[j][t].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][0] += __mul24((int)(a[i] & 255), rw.vk[t]);
                    acc[i][1] += __mul24((int)((a[i] >> 8) & 255), rw.vk[t]);
                    acc[i][2] += __mul24((int)((a[i] >> 16) & 255), rw.vk[t]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) o[c][i] = xm[i] >= 0 ? lutf[clip8(acc[i][c])] : padv;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[0][i] = o[1][i] = o[2][i] = padv;
        }
        float l0[4], l1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t m = mv[i];
            if (optic) { l0[i] = m <= 50 ? 1.0f : 0.0f; l1[i] = m <= 200 ? 1.0f : 0.0f; }
            else { l0[i] = m != 0 ? 1.0f : 0.0f; l1[i] = 0.0f; }
        }
        const size_t off = (size_t)rw.y * crop + xq;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            store_out4(oi + c * plane + off, make_float4(o[c][0], o[c][1], o[c][2], o[c][3]), stream_out);
        store_out4(ol + off, make_float4(l0[0], l0[1], l0[2], l0[3]), stream_out);
        if (K == 2) store_out4(ol + plane + off, make_float4(l1[0], l1[1], l1[2], l1[3]), stream_out);
    }
}

__global__ __launch_bounds__(256) void k_gen_vpass(const uint8_t* __restrict__ masks, const aadg_unit* __restrict__ units,
                                                   const int* __restrict__ order, int slot0, int Hs, int Ws, int crop, int dataset_in,
                                                   const int* __restrict__ tab, const uint32_t* __restrict__ hbuf,
                                                   float* __restrict__ out_img, float* __restrict__ out_lbl) {
    // (slot0 = the first list slot of this launch's chunk = index 0 of the intermediate)
    const int dataset = dataset_in & 0xFF;
    const bool stream_out = (dataset_in & AUG_STREAM_OUT) != 0;
    const int slot = slot0 + blockIdx.z;
    const int u = order != nullptr ? order[slot] : slot;
    const aadg_unit& un = units[u];
    if (unit_flow(true, un, Hs, Ws, crop) != FLOW_GENERIC) return;
    if (un.scaled_h >= Hs && Ws >= 8 && sharp_count4(un, un.n_ops) == 0) return;      // k_fused3w's unit
    __shared__ float lutf[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // uniform: the row tables become scalar loads
    lutf[tid] = normalise_u8(tid);
    __syncthreads();
    const int K = dataset == AADG_DATASET_OPTIC ? 2 : 1;
    const int* base = tab + (size_t)u * crop * TAB_STRIDE;
    const int* xmin_t = base;
    const int* xk_t = xmin_t + crop;
    const int* ymin_t = xk_t + (size_t)crop * KMAX;
    const int* yk_t = ymin_t + crop;
    const int* xnn_t = yk_t + (size_t)crop * KMAX;
    const int* ynn_t = xnn_t + crop;
    const int xq = blockIdx.x * 256 + 4 * lane;
    if (xq >= crop) return;
    const int ybase = blockIdx.y * GV_ROWS + wv * (GV_ROWS / 4);
    if (ybase >= crop) return;
    // the row tables of the wave's rows: one batch of scalar loads (rows beyond the image: the last row's entries, not stored)
    VRow rows[GV_ROWS / 4];
#pragma unroll
    for (int r = 0; r < GV_ROWS / 4; ++r) {
        const int y = min(ybase + r, crop - 1);
        rows[r].y = y;
        rows[r].vym = ymin_t[y];
        rows[r].vyn = ynn_t[y];
#pragma unroll
        for (int t = 0; t < GT_TAPS; ++t) rows[r].vk[t] = yk_t[(size_t)y * KMAX + t];
    }
    const int4 xm4 = *reinterpret_cast<const int4*>(xmin_t + xq), xn4 = *reinterpret_cast<const int4*>(xnn_t + xq);
    const int xm[4] = {xm4.x, xm4.y, xm4.z, xm4.w}, xn[4] = {xn4.x, xn4.y, xn4.z, xn4.w};
    const int nty = axis_taps(Hs, un.scaled_h);
    const uint8_t* msk = masks + (size_t)un.src * Hs * Ws;
    const uint32_t* hcol = hbuf + (size_t)(slot - slot0) * Hs * crop + xq;
    const bool win_ok = xn[0] >= 0 && xn[3] >= 0 && xn[1] >= xn[0] && xn[2] >= xn[0] && xn[3] >= xn[0] && xn[1] - xn[0] < 8 && xn[2] - xn[0] < 8 &&
                        xn[3] - xn[0] < 8 && xn[0] + 8 <= Ws;
    // uniform: false for a wave that lies in the pad columns
    const bool any_col = __any((xm[0] >= 0) | (xm[1] >= 0) | (xm[2] >= 0) | (xm[3] >= 0)) != 0;
    const size_t plane = (size_t)crop * crop;
    float* oi = out_img + (size_t)u * 3 * plane;
    float* ol = out_lbl + (size_t)u * K * plane;
    const bool optic = dataset == AADG_DATASET_OPTIC;
#pragma unroll
    for (int pr = 0; pr < GV_ROWS / 4 / GV_G; ++pr) {
        const int live = min(GV_G, crop - (ybase + GV_G * pr));
        if (live <= 0) break;
        const VRow (&pair)[GV_G] = reinterpret_cast<const VRow (&)[GV_G]>(rows[GV_G * pr]);
#define AADG_VPASS(NT) vpass_pair<NT>(pair, live, hcol, msk, Hs, Ws, crop, xm, xn, win_ok, any_col, optic, K, lutf, oi, ol, plane, xq, stream_out)
        switch (nty) {                                        // uniform per unit
            case 1: AADG_VPASS(1); break;
            case 2: AADG_VPASS(2); break;
            case 3: AADG_VPASS(3); break;
            case 4: AADG_VPASS(4); break;
            default: AADG_VPASS(GT_TAPS); break;
        }
#undef AADG_VPASS
    }
}

// ------------------------------------------------------------------------------------------------
// k_fused3 (round 2): the UP tile (round 1: k_fused) with a leaner instruction stream and a smaller footprint.
//   * 39 KiB of LDS (an 18 KiB patch + a 17 KiB buffer for the horizontally resampled rows) and <= 128 VGPRs: 4 workgroups per
//     CU instead of 3.  Units that chain Sharpness stencils (19 %) need a halo and a ping-pong buffer: their tile is processed
//     as two 8-row halves that fit the same two buffers;
//   * the prologue has no conditional load: indices are clamped into the tables, the loads are issued back to back and
//     validity is applied to the values (conditional loads became branches with a wait at every join: +6 us per tile);
//   * both fixed-point passes multiply by coefficients scaled by 4 (clamped to 2^24 - 1, which leaves every result unchanged,
//     see prescale4): the rounded 8-bit result then IS byte 3 of the 32-bit sum, so the horizontal pass packs its three
//     channels with two v_perm_b32 and the vertical pass turns the sum into an LDS table address with one SDWA shift;
//   * the mask bytes of a lane's 4 columns lie within 4 consecutive source bytes (scale >= 1): one dword load per row
//     instead of four byte loads;
//   * 32-bit offsets from uniform base pointers for the stores.
// Arithmetic and results are those of the round-1 tile (bit-exact with Pillow).
// ------------------------------------------------------------------------------------------------
constexpr int PATCH_CAP_PLAIN = 4608;     // 17 rows x 264 columns: a 256 x 16 tile's patch without a stencil halo
constexpr int HBUF_ROWS = FT_H + 1;       // source rows a 16-row tile touches when no axis shrinks


template <bool SHARP>
__device__ __forceinline__ void fused3_body(const uint8_t* __restrict__ pool, const uint8_t* __restrict__ masks,
                                            const aadg_unit* __restrict__ units, int Hs, int Ws, int crop, int dataset_in,
                                            const int* __restrict__ tab, const uint8_t* __restrict__ lut, size_t lut_stage_stride,
                                            float* __restrict__ out_img, float* __restrict__ out_lbl, int u, int half,
                                            uint32_t* A, uint32_t* B, uint8_t* sl, float* lutf, int bx, int by) {
    const int dataset = dataset_in & 0xFF;
    const bool stream_out = (dataset_in & AUG_STREAM_OUT) != 0;
    const aadg_unit& un = units[u];
    // every field of the record the tile geometry needs, read in ONE batch of scalar loads before the first branch (loads left
    // behind an early return come back one by one, each with its own wait)
    const int n_ops = un.n_ops;
    const int w = un.scaled_w, h = un.scaled_h;
    const int u_pad = un.pad, u_cx = un.crop_x, u_cy = un.crop_y, u_src = un.src;
    const int sc_all = sharp_count4(un, n_ops);
    if ((Ws & 3) || (crop & 3) || sc_all > MAX_SHARP || w < Ws || h < Hs) return;   // unit_flow(...) != FLOW_UP
    if ((sc_all > 0) != SHARP) return;                      // the other part of the grid owns this unit
    const int sc = SHARP ? sc_all : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    lutf[tid] = normalise_u8(tid);

    const int K = dataset == AADG_DATASET_OPTIC ? 2 : 1;
    const uint32_t plane = (uint32_t)crop * (uint32_t)crop;
    float* oi = out_img + (size_t)u * 3 * plane;            // uniform bases, 32-bit lane offsets
    float* ol = out_lbl + (size_t)u * K * plane;
    const int x0 = bx * FT_W, x1 = min(x0 + FT_W, crop);
    // A unit that chains Sharpness stencils needs a halo and a ping-pong buffer: its tiles are 8 rows high (two workgroups per
    // 16-row tile), which fits the same two buffers (13 x 268 patch words) instead of a second, larger LDS layout.
    constexpr int TROWS = SHARP ? FT_H / 2 : FT_H;
    const int y0 = by * FT_H + half * TROWS, y1 = min(y0 + TROWS, crop);
    if (y0 >= crop) return;
    const int ox = u_cx - u_pad, oy = u_cy - u_pad;
    const int fx = max(x0, -ox), lx = min(x1 - 1, w - 1 - ox);
    const int xq = x0 + 4 * lane;                           // vertical pass: lane <-> 4 consecutive columns
    const bool col_ok = xq < crop;
    constexpr int RPW = TROWS / 4;                          // rows per wave
    // rows [ya, yb) of the tile that lie entirely in the zero padding: image = normalise(0), mask = 0
    auto pad_rows = [&](int ya, int yb) {
        if (!col_ok) return;
        const float lab0 = dataset == AADG_DATASET_OPTIC ? 1.0f : 0.0f;      // optic: 0 <= 50 -> cup & disc; vessel: 0 -> background
        const float4 m1 = make_float4(-1.0f, -1.0f, -1.0f, -1.0f), lb = make_float4(lab0, lab0, lab0, lab0);
        for (int y = ya + wv; y < yb; y += 4) {
            const uint32_t off = (uint32_t)y * (uint32_t)crop + (uint32_t)xq;
            store_out4(oi + off, m1, stream_out);
            store_out4(oi + plane + off, m1, stream_out);
            store_out4(oi + 2 * (size_t)plane + off, m1, stream_out);
            store_out4(ol + off, lb, stream_out);
            if (K == 2) store_out4(ol + plane + off, make_float4(1.0f, 1.0f, 1.0f, 1.0f), stream_out);
        }
    };
    const int* base = tab + (size_t)u * crop * TAB_STRIDE;
    const int* xmin_t = base;
    const int* xk_t = xmin_t + crop;
    const int* ymin_t = xk_t + (size_t)crop * KMAX;
    const int* yk_t = ymin_t + crop;
    const int* xnn_t = yk_t + (size_t)crop * KMAX;
    const int* ynn_t = xnn_t + crop;
    const bool resx = w != Ws, resy = h != Hs;              // an unchanged axis is not resampled (one tap of weight one)

    // ---- every table entry of the tile in ONE batch of loads, in front of the first branch that depends on one.  Every index is clamped
    //      into the table, so no load is conditional (no control flow, no wait between the loads); validity is applied to the values ----
    const int ya = y0, yb = y1;
    const int fy = max(ya, -oy), ly = min(yb - 1, h - 1 - oy);
    const int cl = crop - 1;
    const int t_clo = xmin_t[min(max(fx, 0), cl)], t_chi = xmin_t[min(max(lx, 0), cl)];       // scalar loads
    const int t_rlo = ymin_t[min(max(fy, 0), cl)], t_rhi = ymin_t[min(max(ly, 0), cl)];
    int vym[RPW], vyn[RPW];
    uint32_t vk0[RPW], vk1[RPW];
    // row tables: uniform per wave, but read as VECTOR loads (index from the lane's own id): as scalar loads they came back one by one --
    // the compiler had no SGPRs left to hold twelve results and put a wait behind each -- five L2 round trips in series per tile
    const int wvv = tid >> 6;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int yc = min(ya + wvv + 4 * r, yb - 1);
        vym[r] = ymin_t[yc];
        vyn[r] = ynn_t[yc];
        const int2 ky = *reinterpret_cast<const int2*>(yk_t + (size_t)yc * KMAX);
        vk0[r] = prescale4(ky.x); vk1[r] = prescale4(ky.y);
    }
    const int xh = x0 + tid;                                // horizontal pass: thread <-> output column
    const int xhc = min(max(min(max(xh, fx), lx), 0), cl);
    int hxm = xmin_t[xhc];
    const int2 kk = *reinterpret_cast<const int2*>(xk_t + (size_t)xhc * KMAX);
    const int4 xn4 = *reinterpret_cast<const int4*>(xnn_t + min(xq, crop - 4));
    if (fx > lx || fy > ly) { pad_rows(y0, y1); return; }   // no valid column / row in this tile
    const uint32_t hk0 = prescale4(kk.x), hk1 = prescale4(kk.y);
    if (xh < fx || xh > lx) hxm = -1;                       // pad column
    const int xn[4] = {xn4.x, xn4.y, xn4.z, xn4.w};
    const uint8_t* msk = masks + (size_t)u_src * Hs * Ws;
    const uint8_t* src = pool + (size_t)u_src * Hs * Ws * 3;
    const uint32_t lbl_t0 = dataset == AADG_DATASET_OPTIC ? 50u : 0u;
    const bool lbl_flip = dataset != AADG_DATASET_OPTIC;
    const char* lutb = reinterpret_cast<const char*>(lutf);

    {
        // ---- patch -> LDS with the op chain applied, then the horizontal pass into the other buffer ----
        const int r_lo = t_rlo;
        const int nty = resy ? 2 : 1, ntx = resx ? 2 : 1;
        const int r_hi = min(Hs, t_rhi + nty);
        const int c_hi = min(Ws, t_chi + ntx);
        const int r_lo_h = max(0, r_lo - sc), r_hi_h = min(Hs, r_hi + sc);
        const int c_lo_h = max(0, t_clo - sc) & ~3, c_hi_h = min(Ws, (c_hi + sc + 3) & ~3);
        const int pw = c_hi_h - c_lo_h;
        const uint32_t* cur = build_patch<5, 2>(un, n_ops, src, Hs, Ws, r_lo_h, r_hi_h, c_lo_h, c_hi_h, A, B, lut, lut_stage_stride, u, sl);
        uint32_t* Hbuf = cur == A ? B : A;
        // mask: one (unaligned) dword per row holds the bytes of this lane's 4 columns (they lie within 4 consecutive source
        // bytes because no axis shrinks); issued after the patch loads, consumed after the horizontal pass
        uint32_t mw[RPW], msh[4], mkeep[4];
        {
            int mbase = -1;
#pragma unroll
            for (int i = 3; i >= 0; --i) mbase = xn[i] >= 0 ? xn[i] : mbase;      // first valid column of the lane
            const bool lane_has = mbase >= 0;
            mbase = min(max(mbase, 0), Ws - 4);                                    // keep the 4-byte window inside the row
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                msh[i] = xn[i] >= 0 ? 8u * (uint32_t)(xn[i] - mbase) : 0u;
                mkeep[i] = (xn[i] >= 0 && lane_has) ? 255u : 0u;
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r)                                          // unconditional: pad rows read row 0, result unused
                mw[r] = reinterpret_cast<const UnalignedU32*>(msk + (size_t)max(vyn[r], 0) * Ws + mbase)->v;
        }
        const int nrows = r_hi - r_lo;
        uint32_t* hout = Hbuf + tid;
        if (hxm >= 0) {
            const uint32_t* col = cur + (r_lo - r_lo_h) * pw + (hxm - c_lo_h);
            if (resx) {
                const int d1 = hk1 ? 1 : 0;                 // no second tap (right image edge): re-read the first
#pragma unroll 4
                for (int rr = 0; rr < nrows; ++rr) {
                    const uint32_t p0 = col[rr * pw], p1 = col[rr * pw + d1];
                    // byte 3 of each sum = the rounded channel value ((2^21 + c0 k0 + c1 k1) >> 22 with k scaled by 4)
                    const uint32_t s0 = (1u << 23) + __umul24(p0 & 255u, hk0) + __umul24(p1 & 255u, hk1);
                    const uint32_t s1 = (1u << 23) + __umul24((p0 >> 8) & 255u, hk0) + __umul24((p1 >> 8) & 255u, hk1);
                    const uint32_t s2 = (1u << 23) + __umul24((p0 >> 16) & 255u, hk0) + __umul24((p1 >> 16) & 255u, hk1);
                    const uint32_t rg = __builtin_amdgcn_perm(s1, s0, 0x0c0c0703u);      // [s0.b3, s1.b3, 0, 0]
                    hout[rr * FT_W] = __builtin_amdgcn_perm(s2, rg, 0x0c070100u);          // [rg.b0, rg.b1, s2.b3, 0]
                }
            } else {
#pragma unroll 4
                for (int rr = 0; rr < nrows; ++rr) hout[rr * FT_W] = col[rr * pw];
            }
        } else {
            for (int rr = 0; rr < nrows; ++rr) hout[rr * FT_W] = 0u;
        }
        __syncthreads();

        // ---- vertical pass + normalise + store: wave <-> output rows ya + wv, + 4, ...; lane <-> 4 consecutive columns ----
        if (col_ok) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int y = ya + wv + 4 * r;
                if (y >= yb) break;
                const int ym = vym[r];
                float o[3][4];
                if (ym >= 0) {
                    const uint32_t ky0 = vk0[r], ky1 = vk1[r];
                    const uint4 h0 = *reinterpret_cast<const uint4*>(Hbuf + (ym - r_lo) * FT_W + 4 * lane);
                    const uint32_t a0[4] = {h0.x, h0.y, h0.z, h0.w};
                    if (resy) {
                        const uint4 h1 = *reinterpret_cast<const uint4*>(Hbuf + (ym + (ky1 ? 1 : 0) - r_lo) * FT_W + 4 * lane);
                        const uint32_t a1[4] = {h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const uint32_t v = (1u << 23) + __umul24((a0[i] >> (8 * c)) & 255u, ky0) + __umul24((a1[i] >> (8 * c)) & 255u, ky1);
                                o[c][i] = *reinterpret_cast<const float*>(lutb + ((v >> 24) << 2));
                            }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int c = 0; c < 3; ++c) o[c][i] = *reinterpret_cast<const float*>(lutb + (((a0[i] >> (8 * c)) & 255u) << 2));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[0][i] = o[1][i] = o[2][i] = -1.0f;      // pad rows: normalise(0)
                }
                float l0[4], l1[4];
                const uint32_t mrow = vyn[r] >= 0 ? mw[r] : 0u;      // pad rows: mask 0
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t m = (mrow >> msh[i]) & mkeep[i];
                    l0[i] = ((m <= lbl_t0) != lbl_flip) ? 1.0f : 0.0f;
                    l1[i] = m <= 200u ? 1.0f : 0.0f;
                }
                const uint32_t off = (uint32_t)y * (uint32_t)crop + (uint32_t)xq;
                store_out4(oi + off, make_float4(o[0][0], o[0][1], o[0][2], o[0][3]), stream_out);
                store_out4(oi + plane + off, make_float4(o[1][0], o[1][1], o[1][2], o[1][3]), stream_out);
                store_out4(oi + 2 * (size_t)plane + off, make_float4(o[2][0], o[2][1], o[2][2], o[2][3]), stream_out);
                store_out4(ol + off, make_float4(l0[0], l0[1], l0[2], l0[3]), stream_out);
                if (K == 2) store_out4(ol + plane + off, make_float4(l1[0], l1[1], l1[2], l1[3]), stream_out);
            }
        }
    }
}

// one z-slice of k_fused3's grid: the plain units' tiles, then two 8-row workgroups per 16-row tile of a stencil unit
__device__ __forceinline__ void fused3_slice(const uint8_t* __restrict__ pool, const uint8_t* __restrict__ masks,
                                             const aadg_unit* __restrict__ units, const int* __restrict__ order,
                                             const int* __restrict__ order_sharp, int n_plain, int Hs, int Ws, int crop, int dataset,
                                             const int* __restrict__ tab, const uint8_t* __restrict__ lut, size_t lut_stage_stride,
                                             float* __restrict__ out_img, float* __restrict__ out_lbl, int bx, int by, int z,
                                             uint32_t* A, uint32_t* B, uint8_t* sl, float* lutf) {
    if (z < n_plain) {
        const int u = order != nullptr ? order[z] : z;
        fused3_body<false>(pool, masks, units, Hs, Ws, crop, dataset, tab, lut, lut_stage_stride, out_img, out_lbl, u, 0, A, B, sl, lutf, bx, by);
    } else {
        const int zz = z - n_plain;
        const int u = order_sharp != nullptr ? order_sharp[zz >> 1] : (zz >> 1);
        fused3_body<true>(pool, masks, units, Hs, Ws, crop, dataset, tab, lut, lut_stage_stride, out_img, out_lbl, u, zz & 1, A, B, sl, lutf, bx, by);
    }
}

// grid (ceil(crop/256), ceil(crop/16), n_plain + 2 n_sharp): the first n_plain z-slices run the plain body on unit order[z],
// the rest the Sharpness body, two 8-row workgroups per unit and 16-row tile.  `order` lists the unit indices grouped by
// class (the caller classifies on the host: `order` = the plain units, `order_sharp` = the Sharpness units); without it every
// unit gets a slice of each kind and the wrong kind returns after
// reading the unit record (a returning workgroup still occupies a slot with its 39 KiB of LDS for about a microsecond).
__global__ __launch_bounds__(256) void k_fused3(const uint8_t* __restrict__ pool, const uint8_t* __restrict__ masks,
                                                const aadg_unit* __restrict__ units, const int* __restrict__ order,
                                                const int* __restrict__ order_sharp, int n_plain, int Hs, int Ws, int crop, int dataset,
                                                const int* __restrict__ tab, const uint8_t* __restrict__ lut,
                                                size_t lut_stage_stride, float* __restrict__ out_img, float* __restrict__ out_lbl) {
    __shared__ __attribute__((aligned(16))) uint32_t A[PATCH_CAP_PLAIN];
    __shared__ __attribute__((aligned(16))) uint32_t B[HBUF_ROWS * FT_W];
    __shared__ __attribute__((aligned(16))) uint8_t sl[AADG_MAX_OPS * 768];
    __shared__ float lutf[256];
    fused3_slice(pool, masks, units, order, order_sharp, n_plain, Hs, Ws, crop, dataset, tab, lut, lut_stage_stride, out_img, out_lbl,
                 blockIdx.x, blockIdx.y, blockIdx.z, A, B, sl, lutf);
}

// ------------------------------------------------------------------------------------------------
// k_fused3w (round 5): ONE pass for the down-scaling units that shrink the WIDTH only (scaled height >= source height, no Sharpness
// stencil: 40 % of the RVS pipeline's down-scaling units).  Their vertical pass is the up-scaling one (<= 2 taps, 17 source rows per
// 16-row tile), so the tile is k_fused3's with 128 output columns: the <= 2 * 128 + 5 source columns (+ alignment) x 17 rows of the
// patch fit k_fused3's patch buffer, the horizontally resampled rows its second buffer -- the same 39 KiB of LDS, 4 workgroups per
// CU -- and the unit no longer writes and re-reads the RGBX intermediate of the two passes (4 bytes x source rows x output columns).
//   horizontal pass: thread <-> (output column, row parity), <= 5 taps (hpass_px, the two-pass kernel's arithmetic);
//   vertical pass:   half-wave <-> output row, lane <-> 4 columns, <= 2 taps (k_fused3's arithmetic); the 4 NEAREST mask bytes of a
//                    lane lie within 8 consecutive source bytes (the axis shrinks by at most 2).
// Results are those of k_gen_hpass + k_gen_vpass (Pillow's order: the horizontal pass rounded to uint8 first), bit for bit.
// ------------------------------------------------------------------------------------------------
constexpr int FW_W = 128;

__device__ __forceinline__ bool unit_wonly(const aadg_unit& un, int Hs, int Ws, int crop) {     // generic, plain, shrinks the width only
    return Ws >= 8 && unit_flow(true, un, Hs, Ws, crop) == FLOW_GENERIC && un.scaled_h >= Hs && sharp_count4(un, un.n_ops) == 0;
}

__global__ __launch_bounds__(256) void k_fused3w(const uint8_t* __restrict__ pool, const uint8_t* __restrict__ masks,
                                                 const aadg_unit* __restrict__ units, const int* __restrict__ order, int Hs, int Ws, int crop,
                                                 int dataset_in, const int* __restrict__ tab, const uint8_t* __restrict__ lut,
                                                 size_t lut_stage_stride, float* __restrict__ out_img, float* __restrict__ out_lbl) {
    __shared__ __attribute__((aligned(16))) uint32_t A[PATCH_CAP_PLAIN];
    __shared__ __attribute__((aligned(16))) uint32_t B[HBUF_ROWS * FT_W];
    __shared__ __attribute__((aligned(16))) uint8_t sl[AADG_MAX_OPS * 768];
    __shared__ float lutf[256];
    const int bx = blockIdx.x, by = blockIdx.y;
    const int u = order != nullptr ? order[blockIdx.z] : (int)blockIdx.z;
    const int dataset = dataset_in & 0xFF;
    const bool stream_out = (dataset_in & AUG_STREAM_OUT) != 0;
    const aadg_unit& un = units[u];
    const int n_ops = un.n_ops;
    const int w = un.scaled_w, h = un.scaled_h;
    const int u_pad = un.pad, u_cx = un.crop_x, u_cy = un.crop_y, u_src = un.src;
    const int sc_all = sharp_count4(un, n_ops);
    if ((Ws & 3) || (crop & 3) || Ws < 8 || sc_all != 0 || w >= Ws || 2 * w < Ws || h < Hs) return;      // not a width-only unit (unit_wonly)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    lutf[tid] = normalise_u8(tid);

    const int K = dataset == AADG_DATASET_OPTIC ? 2 : 1;
    const uint32_t plane = (uint32_t)crop * (uint32_t)crop;
    float* oi = out_img + (size_t)u * 3 * plane;
    float* ol = out_lbl + (size_t)u * K * plane;
    const int x0 = bx * FW_W, x1 = min(x0 + FW_W, crop);
    const int y0 = by * FT_H, y1 = min(y0 + FT_H, crop);
    if (y0 >= crop || x0 >= crop) return;
    const int ox = u_cx - u_pad, oy = u_cy - u_pad;
    const int fx = max(x0, -ox), lx = min(x1 - 1, w - 1 - ox);
    const int lh = lane & 31, rsub = lane >> 5;             // vertical pass: half-wave <-> row, lane of the half <-> 4 consecutive columns
    const int xq = x0 + 4 * lh;
    const bool col_ok = xq < x1;
    constexpr int RPW = FT_H / 8;                           // row pairs per wave
    auto pad_rows = [&](int ya, int yb) {
        if (!col_ok) return;
        const float lab0 = dataset == AADG_DATASET_OPTIC ? 1.0f : 0.0f;
        const float4 m1 = make_float4(-1.0f, -1.0f, -1.0f, -1.0f), lb = make_float4(lab0, lab0, lab0, lab0);
        for (int y = ya + 2 * wv + rsub; y < yb; y += 8) {
            const uint32_t off = (uint32_t)y * (uint32_t)crop + (uint32_t)xq;
            store_out4(oi + off, m1, stream_out);
            store_out4(oi + plane + off, m1, stream_out);
            store_out4(oi + 2 * (size_t)plane + off, m1, stream_out);
            store_out4(ol + off, lb, stream_out);
            if (K == 2) store_out4(ol + plane + off, make_float4(1.0f, 1.0f, 1.0f, 1.0f), stream_out);
        }
    };
    const int* base = tab + (size_t)u * crop * TAB_STRIDE;
    const int* xmin_t = base;
    const int* xk_t = xmin_t + crop;
    const int* ymin_t = xk_t + (size_t)crop * KMAX;
    const int* yk_t = ymin_t + crop;
    const int* xnn_t = yk_t + (size_t)crop * KMAX;
    const int* ynn_t = xnn_t + crop;
    const bool resy = h != Hs;

    // ---- every table entry of the tile in one batch of loads (clamped indices: no load is conditional) ----
    const int ya = y0, yb = y1;
    const int fy = max(ya, -oy), ly = min(yb - 1, h - 1 - oy);
    const int cl = crop - 1;
    const int t_clo = xmin_t[min(max(fx, 0), cl)], t_chi = xmin_t[min(max(lx, 0), cl)];
    const int t_rlo = ymin_t[min(max(fy, 0), cl)], t_rhi = ymin_t[min(max(ly, 0), cl)];
    int vym[RPW], vyn[RPW];
    uint32_t vk0[RPW], vk1[RPW];
    const int wvv = tid >> 6;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int yc = min(ya + 2 * (wvv + 4 * r) + rsub, yb - 1);
        vym[r] = ymin_t[yc];
        vyn[r] = ynn_t[yc];
        const int2 ky = *reinterpret_cast<const int2*>(yk_t + (size_t)yc * KMAX);
        vk0[r] = prescale4(ky.x); vk1[r] = prescale4(ky.y);
    }
    const int hc = tid & (FW_W - 1);                        // horizontal pass: thread <-> (output column, row parity: uniform per wave)
    const int hg = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int xh = x0 + hc;
    const int xhc = min(max(min(max(xh, fx), lx), 0), cl);
    int hxm = xmin_t[xhc];
    int kraw[GT_TAPS];
#pragma unroll
    for (int t = 0; t < GT_TAPS; ++t) kraw[t] = xk_t[(size_t)xhc * KMAX + t];
    const int4 xn4 = *reinterpret_cast<const int4*>(xnn_t + min(xq, crop - 4));
    if (fx > lx || fy > ly) { pad_rows(y0, y1); return; }   // no valid column / row in this tile
    const int ntx = axis_taps(Ws, w);
    uint32_t hk[GT_TAPS];
#pragma unroll
    for (int t = 0; t < GT_TAPS; ++t) hk[t] = t < ntx ? prescale4(kraw[t]) : 0u;
    if (xh < fx || xh > lx) hxm = -1;                       // pad column
    const int xn[4] = {xn4.x, xn4.y, xn4.z, xn4.w};
    const uint8_t* msk = masks + (size_t)u_src * Hs * Ws;
    const uint8_t* src = pool + (size_t)u_src * Hs * Ws * 3;
    const uint32_t lbl_t0 = dataset == AADG_DATASET_OPTIC ? 50u : 0u;
    const bool lbl_flip = dataset != AADG_DATASET_OPTIC;
    const char* lutb = reinterpret_cast<const char*>(lutf);

    // ---- patch -> LDS with the op chain applied ----
    const int r_lo = t_rlo;
    const int nty = resy ? 2 : 1;
    const int r_hi = min(Hs, t_rhi + nty);
    const int c_hi = min(Ws, t_chi + ntx);
    const int c_lo_h = t_clo & ~3, c_hi_h = min(Ws, (c_hi + 3) & ~3);
    const int pw = c_hi_h - c_lo_h;
    const uint32_t* cur = build_patch<5, 2>(un, n_ops, src, Hs, Ws, r_lo, r_hi, c_lo_h, c_hi_h, A, B, lut, lut_stage_stride, u, sl);
    uint32_t* Hbuf = cur == A ? B : A;
    // mask: the bytes of this lane's 4 columns lie within 8 consecutive source bytes: two (unaligned) dwords per row
    uint32_t mlo[RPW], mhi[RPW], msh[4], mkeep[4];
    {
        int mbase = -1;
#pragma unroll
        for (int i = 3; i >= 0; --i) mbase = xn[i] >= 0 ? xn[i] : mbase;      // first valid column of the lane
        const bool lane_has = mbase >= 0;
        mbase = min(max(mbase, 0), Ws - 8);                                    // keep the 8-byte window inside the row
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sh = xn[i] - mbase;
            const bool ok = xn[i] >= 0 && lane_has && sh >= 0 && sh < 8;
            msh[i] = ok ? 8u * (uint32_t)sh : 0u;
            mkeep[i] = ok ? 255u : 0u;
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {                                        // unconditional: pad rows read row 0, result unused
            const uint8_t* mrow = msk + (size_t)max(vyn[r], 0) * Ws + mbase;
            mlo[r] = reinterpret_cast<const UnalignedU32*>(mrow)->v;
            mhi[r] = reinterpret_cast<const UnalignedU32*>(mrow + 4)->v;
        }
    }
    // ---- horizontal pass into the other buffer: rows hg, hg + 2, ... of the patch ----
    const int nrows = r_hi - r_lo;
    uint32_t* hout = Hbuf + hc;
    if (hxm >= 0) {
        const uint32_t* colp = cur + (hxm - c_lo_h);
#define AADG_FW_ROWS(NT) _Pragma("unroll 2") for (int rr = hg; rr < nrows; rr += 2) hout[rr * FW_W] = hpass_px<NT>(colp + rr * pw, hk)
        switch (ntx) {                                        // uniform per unit
            case 3: AADG_FW_ROWS(3); break;
            case 4: AADG_FW_ROWS(4); break;
            default: AADG_FW_ROWS(GT_TAPS); break;
        }
#undef AADG_FW_ROWS
    } else {
        for (int rr = hg; rr < nrows; rr += 2) hout[rr * FW_W] = 0u;
    }
    __syncthreads();

    // ---- vertical pass + normalise + store ----
    if (col_ok) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int y = ya + 2 * (wv + 4 * r) + rsub;
            if (y >= yb) continue;
            const int ym = vym[r];
            float o[3][4];
            if (ym >= 0) {
                const uint32_t ky0 = vk0[r], ky1 = vk1[r];
                const uint4 h0 = *reinterpret_cast<const uint4*>(Hbuf + (ym - r_lo) * FW_W + 4 * lh);
                const uint32_t a0[4] = {h0.x, h0.y, h0.z, h0.w};
                if (resy) {
                    const uint4 h1 = *reinterpret_cast<const uint4*>(Hbuf + (ym + (ky1 ? 1 : 0) - r_lo) * FW_W + 4 * lh);
                    const uint32_t a1[4] = {h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const uint32_t v = (1u << 23) + __umul24((a0[i] >> (8 * c)) & 255u, ky0) + __umul24((a1[i] >> (8 * c)) & 255u, ky1);
                            o[c][i] = *reinterpret_cast<const float*>(lutb + ((v >> 24) << 2));
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int c = 0; c < 3; ++c) o[c][i] = *reinterpret_cast<const float*>(lutb + (((a0[i] >> (8 * c)) & 255u) << 2));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (xq + i < fx || xq + i > lx) o[0][i] = o[1][i] = o[2][i] = -1.0f;      // pad columns: normalise(0)
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[0][i] = o[1][i] = o[2][i] = -1.0f;              // pad rows
            }
            float l0[4], l1[4];
            const unsigned long long mrow = vyn[r] >= 0 ? (((unsigned long long)mhi[r] << 32) | mlo[r]) : 0ull;      // pad rows: mask 0
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t m = (uint32_t)(mrow >> msh[i]) & mkeep[i];
                l0[i] = ((m <= lbl_t0) != lbl_flip) ? 1.0f : 0.0f;
                l1[i] = m <= 200u ? 1.0f : 0.0f;
            }
            const uint32_t off = (uint32_t)y * (uint32_t)crop + (uint32_t)xq;
            store_out4(oi + off, make_float4(o[0][0], o[0][1], o[0][2], o[0][3]), stream_out);
            store_out4(oi + plane + off, make_float4(o[1][0], o[1][1], o[1][2], o[1][3]), stream_out);
            store_out4(oi + 2 * (size_t)plane + off, make_float4(o[2][0], o[2][1], o[2][2], o[2][3]), stream_out);
            store_out4(ol + off, make_float4(l0[0], l0[1], l0[2], l0[3]), stream_out);
            if (K == 2) store_out4(ol + plane + off, make_float4(l1[0], l1[1], l1[2], l1[3]), stream_out);
        }
    }
}

// list slots per chunk of the two-pass generic flow: as many as keep the chunk's intermediate (Hs x crop words per slot) within 128 MB, half of
// the Infinity Cache -- 32 slots at 1024 x 1024 (measured there, 86 generic units of 168: all at once 0.914 ms for the batch's tile kernels,
// chunks of 8 / 16 / 24 / 32 / 48 / 64 slots 0.971 / 0.909 / 0.870 / 0.862 / 0.910 / 0.931 ms).  aadg_aug_lists.gen_chunk (ABI 7) asks for
// FEWER slots per chunk (tests: chunk boundaries).  The workspace holds one chunk's intermediate.
int gen_chunk(int Hs, int crop) {
    const size_t per_slot = (size_t)Hs * crop * 4;
    const size_t n = ((size_t)128 << 20) / (per_slot ? per_slot : 1);
    return n < 4 ? 4 : (n > 4096 ? 4096 : (int)n);
}

struct WsLayout {
    size_t hist, lut, tab, buf0, buf1, hbuf, total;
};
WsLayout ws_layout(int N, int Hs, int Ws, int crop) {
    WsLayout l;
    size_t o = 0;
    l.hist = o; o = aadg_align_up(o + (size_t)AADG_MAX_OPS * N * HIST_STRIDE * 4, 256);     // one histogram set per op stage
    l.lut = o;  o = aadg_align_up(o + (size_t)AADG_MAX_OPS * N * 768, 256);
    l.tab = o;  o = aadg_align_up(o + (size_t)N * crop * TAB_STRIDE * 4, 256);
    const size_t img = (size_t)Hs * Ws * 3;
    l.buf0 = o; o = aadg_align_up(o + (size_t)N * img, 256);
    l.buf1 = o; o = aadg_align_up(o + (size_t)N * img, 256);
    // horizontally resampled rows of the down-scaling units (k_gen_hpass -> k_gen_vpass): [unit slot][Hs][crop] packed RGBX
    l.hbuf = o; o = aadg_align_up(o + (size_t)(N < gen_chunk(Hs, crop) ? N : gen_chunk(Hs, crop)) * Hs * crop * 4, 256);      // one chunk of generic units
    l.total = o;
    return l;
}

int chunks_for(int npix) {
    int c = (npix / 4 + 256 * 8 - 1) / (256 * 8);  // ~8 groups of 4 pixels per thread
    return c < 1 ? 1 : (c > 1024 ? 1024 : c);
}

// hints from a caller that has the unit records on the host (all bits set = unknown, launch everything)
constexpr int HINT_FUSED = 1, HINT_STAGED = 2, HINT_GENERIC = 4;

// The tile kernels of a call: k_fused3 over the up-scaling units (np plain slices + 2 per stencil unit; order_up == NULL: every unit is
// offered every kind of slice), and for the ng down-scaling ("generic") units the horizontal pass (units without a stencil, then the
// last ng_sharp units of the list with the stencil's ping-pong buffer) + the vertical pass.  chunk_req: list slots per chunk of the
// two passes (0 = gen_chunk(); never more than that: the workspace holds one default chunk).
int launch_tiles(const uint8_t* pool, const uint8_t* masks, const aadg_unit* units, const int* order_up, int np, int ns, const int* order_gen,
                 int ng, int ng_sharp, int Hs, int Ws, int crop, int dsk, const int* tab, const uint8_t* lut, size_t lut_stage_stride,
                 uint32_t* hbuf, float* out_img, float* out_lbl, hipStream_t st, int chunk_req = 0, int ng_wonly = 0,
                 const int* order_sharp_override = nullptr) {
    const int gx = (crop + FT_W - 1) / FT_W, gy = (crop + FT_H - 1) / FT_H, gz = np + 2 * ns;
    // (order_sharp_override: a sub-range of the caller's lists -- forward_cached launches the units that wait for no statistics pass apart
    // from the late ones)
    const int* order_sharp = order_sharp_override != nullptr ? order_sharp_override : order_up != nullptr ? order_up + np : nullptr;
    // with the caller's list the stencil units are the last ng_sharp entries; without one every unit is offered to both variants
    const bool listed = order_gen != nullptr;
    const int n0 = listed ? ng - ng_sharp : ng, s1 = listed ? ng - ng_sharp : 0;
    const int hx = (crop + GH_CB - 1) / GH_CB, hy0 = (Hs + GhRows<false>::value - 1) / GhRows<false>::value;
    // (Measured and dropped: k_fused3's tiles and the horizontal-pass tiles interleaved in ONE launch, so that the store-bound and the
    // issue-bound workgroups share the CUs: 611 us against 376 + 176 us one after the other at 1024 x 1024 -- both are issue-heavy.)
    if (gz > 0) {
        hipLaunchKernelGGL(k_fused3, dim3(gx, gy, gz), dim3(256), 0, st, pool, masks, units, order_up, order_sharp, np, Hs, Ws, crop, dsk, tab, lut,
                           lut_stage_stride, out_img, out_lbl);
        AADG_LAUNCH_CHECK();
    }
    if (ng <= 0) return 0;
    // the width-only units (ABI 9: the FIRST ng_wonly slots of the caller's list; without a list every slot is offered): one pass
    const int nw = listed ? ng_wonly : 0;
    if ((listed ? nw : ng) > 0 && Ws >= 8) {
        hipLaunchKernelGGL(k_fused3w, dim3((crop + FW_W - 1) / FW_W, gy, listed ? nw : ng), dim3(256), 0, st, pool, masks, units, order_gen, Hs, Ws, crop,
                           dsk, tab, lut, lut_stage_stride, out_img, out_lbl);
        AADG_LAUNCH_CHECK();
    }
    // The generic units go through the two passes in chunks of gen_chunk() list slots that share ONE slice of the intermediate: what the
    // horizontal pass writes is read back by the vertical pass while it is still in the Infinity Cache (256 MB; 4 MB per unit at
    // 1024 x 1024), instead of after the whole batch's intermediate has gone to HBM and come back.
    const int chunk = (chunk_req > 0 && chunk_req < gen_chunk(Hs, crop)) ? chunk_req : gen_chunk(Hs, crop);
    const int hy1 = (Hs + GhRows<true>::value - 1) / GhRows<true>::value;
    for (int a = nw; a < ng; a += chunk) {
        const int b = min(ng, a + chunk);
        // listed: plain units are slots [0, n0), stencil units [s1, ng); unlisted: every slot is offered to both variants
        const int p0 = a, p1 = min(b, n0);
        const int q0 = max(a, s1), q1 = b;
        if (p1 > p0) {
            hipLaunchKernelGGL(k_gen_hpass<false>, dim3(hx, (hy0 + GH_NB - 1) / GH_NB, p1 - p0), dim3(256), 0, st, pool, units, order_gen, p0, Hs, Ws, crop, tab, lut,
                               lut_stage_stride, hbuf, a);
            AADG_LAUNCH_CHECK();
        }
        if (q1 > q0) {
            hipLaunchKernelGGL(k_gen_hpass<true>, dim3(hx, (hy1 + GH_NB - 1) / GH_NB, q1 - q0), dim3(256), 0, st, pool, units, order_gen, q0, Hs, Ws, crop, tab, lut,
                               lut_stage_stride, hbuf, a);
            AADG_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(k_gen_vpass, dim3((crop + 255) / 256, (crop + GV_ROWS - 1) / GV_ROWS, b - a), dim3(256), 0, st, masks, units, order_gen, a,
                           Hs, Ws, crop, dsk, tab, hbuf, out_img, out_lbl);
        AADG_LAUNCH_CHECK();
    }
    return 0;
}

// statistics + LUT (+ staged apply) for stages [0, max_ops)
int run_stages(const Bufs& bufs, const UnitRef& ur, int N, int Hs, int Ws, int crop, int max_ops, uint8_t* ws8,
               const WsLayout& L, uint8_t* out_override, int classes, int stats_mask, hipStream_t st,
               const aadg_aug_lists* lists = nullptr, int* tab = nullptr, bool* tables_done = nullptr,
               const aadg_aug_lists* all_lists = nullptr) {
    const int npix = Hs * Ws;
    uint32_t* hist0 = reinterpret_cast<uint32_t*>(ws8 + L.hist);          // stage k's histograms: hist0 + k * N * HIST_STRIDE
    const size_t hist_stage = (size_t)N * HIST_STRIDE;
    // the raw images' statistics from the caller's cache: no stage-0 pixel pass
    const uint32_t* pool_hist = all_lists != nullptr ? all_lists->pool_hist : nullptr;
    bool zero = false;                                   // some pixel-pass histogram kernel of this call accumulates into hist
    for (int k = 0; k < max_ops; ++k) {
        const int nstat = lists != nullptr ? lists->n_stat[k] : N;
        if ((stats_mask & (1 << k)) && nstat > 0 && !(k == 0 && pool_hist != nullptr)) zero = true;
    }
    if (zero) AADG_HIP_TRY(hipMemsetAsync(hist0, 0, (size_t)max_ops * hist_stage * 4, st));
    uint8_t* lut = ws8 + L.lut;
    const size_t lut_stage_stride = (size_t)N * 768;
    const dim3 g(chunks_for(npix), N);
    for (int k = 0; k < max_ops; ++k) {
        // work list of the statistics kernels of this stage (caller's, else every unit is offered and the others return)
        const int* ulist = lists != nullptr ? lists->stat_units[k] : nullptr;
        const int nstat = ulist != nullptr ? lists->n_stat[k] : N;
        uint32_t* hist = hist0 + (size_t)k * hist_stage;
        if (k == 0 && pool_hist != nullptr) {
            if (tab != nullptr) {
                hipLaunchKernelGGL(k_lut_tables, dim3(2 * N), dim3(256), 0, st, ur, N, npix, Hs, Ws, crop, pool_hist, lut, tab);
                AADG_LAUNCH_CHECK();
                *tables_done = true;
                if (classes & HINT_STAGED) {
                    hipLaunchKernelGGL(k_apply, g, dim3(256), 0, st, bufs, ur, k, Hs, Ws, crop, lut, out_override);
                    AADG_LAUNCH_CHECK();
                }
                continue;
            }
        } else if ((stats_mask & (1 << k)) && nstat > 0) {
            if (k == 0 && tab != nullptr) {
                // stage-0 histograms and the resampling tables in one launch
                hipLaunchKernelGGL(k_hist_tables, dim3(g.x * nstat + N), dim3(256), 0, st, bufs, ur, ulist, nstat, (int)g.x, npix, Hs, Ws,
                                   crop, hist, tab);
                AADG_LAUNCH_CHECK();
                *tables_done = true;
            } else if (k == 0 || (classes & HINT_STAGED)) {
                hipLaunchKernelGGL(k_hist, dim3(g.x, nstat), dim3(256), 0, st, bufs, ur, ulist, k, npix, Hs, Ws, crop, hist);
                AADG_LAUNCH_CHECK();
            }
            if (k > 0 && (classes & (HINT_FUSED | HINT_GENERIC))) {
                const int n_sten = (ulist != nullptr && lists->n_stat_stencil[k] >= 0 && lists->n_stat_stencil[k] <= nstat) ? lists->n_stat_stencil[k] : nstat;
                const int rc = launch_hist_fused(bufs.pool, ur.units, ulist, nstat, n_sten, k, Hs, Ws, crop, lut, lut_stage_stride, hist, st);
                if (rc) return rc;
            }
        }
        hipLaunchKernelGGL(k_lut, dim3(N), dim3(256), 0, st, ur, k, N, (const int*)nullptr, npix, Hs, Ws, crop, (const uint32_t*)hist0,
                           (const uint32_t*)(hist0 + (size_t)k * hist_stage), pool_hist, lut);
        AADG_LAUNCH_CHECK();
        if (classes & HINT_STAGED) {
            hipLaunchKernelGGL(k_apply, g, dim3(256), 0, st, bufs, ur, k, Hs, Ws, crop, lut, out_override);
            AADG_LAUNCH_CHECK();
        }
    }
    return 0;
}

}  // namespace

// The call with the pool's statistics cached and the caller's late list (no staged units).  The first launch builds the tables and every
// byte map that waits for no pixel pass (all of them for most units).  What is left is the late units' chain -- per slot k >= 1: histogram
// pass over the image after k ops, then the maps of the late units -- and the tiles:
//     k_luts_tables | k_hist_fused(1) k_lut(1, late) ... | k_fused3 (up-scaling units) | k_gen_hpass + k_gen_vpass (down-scaling units, in chunks)
// (Round 5 measured the chain and the late units' tiles on a second, highest-priority stream and dropped it: 268 instead of 278 us per
// 168-unit call, against a second kernel name and a tile-kernel duration that no longer says how fast the tile kernel is.  Round 6 took it
// up again in the form below -- the planner lists the units that wait for nothing first, the fork is off whenever the caller times the
// tile kernel -- because the chain had become the largest part of the call outside the tile kernel: whole 512^2 call 0.566 -> 0.60-0.63.)
// the `dataset` argument of the kernels that write the outputs: + AUG_STREAM_OUT when the batch's outputs exceed AUG_STREAM_BYTES
static inline int aug_dataset_arg(int dataset, int N, int crop) {
    const size_t K = dataset == AADG_DATASET_OPTIC ? 2 : 1;
    return dataset | ((size_t)N * (3 + K) * crop * crop * sizeof(float) > AUG_STREAM_BYTES ? AUG_STREAM_OUT : 0);
}

// helper stream + events of forward_cached's fork (one per process and device: one process drives one GPU); nullptr if they cannot be made
struct AugFork {
    int device;
    hipStream_t helper;
    hipEvent_t tables, chain, done;
};
AugFork* aug_fork() {
    static AugFork f = {-1, nullptr, nullptr, nullptr, nullptr};
    static bool failed = false;
    int dev = 0;
    if (failed || hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (f.device == dev) return &f;
    if (f.device >= 0) return nullptr;                                  // made for another device: keep to one stream there
    if (hipStreamCreateWithFlags(&f.helper, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&f.tables, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&f.chain, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&f.done, hipEventDisableTiming) != hipSuccess) {
        failed = true;
        return nullptr;
    }
    f.device = dev;
    return &f;
}

int forward_cached(const uint8_t* pool, const uint8_t* masks, const UnitRef& ur, int N, int Hs, int Ws, int max_ops, int crop, int dataset,
                   float* out_img, float* out_lbl, uint8_t* ws8, const WsLayout& L, hipStream_t st, int classes, int stats_mask,
                   void* ev_before, void* ev_after, const aadg_aug_lists& ls) {
    const int n_plain = ls.n_plain, n_sharp = ls.n_sharp, n_generic = ls.n_generic, n_late = ls.n_late;
    if (n_late < 0 || n_late > N) return AADG_E_BADARG;
    const int dsk = aug_dataset_arg(dataset, N, crop);
    const int npix = Hs * Ws;
    uint32_t* hist0 = reinterpret_cast<uint32_t*>(ws8 + L.hist);
    const size_t hist_stage = (size_t)N * HIST_STRIDE;
    uint8_t* lut = ws8 + L.lut;
    const size_t lut_stage_stride = (size_t)N * 768;
    int* tab = reinterpret_cast<int*>(ws8 + L.tab);
    bool pixel_pass = false;
    for (int k = 1; k < max_ops; ++k) pixel_pass |= ((stats_mask >> k) & 1) && ls.n_stat[k] > 0;
    const bool chain = n_late > 0 && max_ops > 1;
    hipLaunchKernelGGL(k_luts_tables, dim3(3 * N), dim3(256), 0, st, ur, N, max_ops, npix, Hs, Ws, crop, ls.pool_hist, lut, tab,
                       (pixel_pass && chain) ? hist0 : (uint32_t*)nullptr);
    AADG_LAUNCH_CHECK();
    // Round 6 (ABI 12): the late units' chain -- histogram pass(es), their byte maps -- and their tiles run on a helper stream BESIDE the tile
    // kernel of the units that wait for nothing (the planner lists them first inside the plain and the Sharpness class:
    // n_plain_early / n_sharp_early; a zero-initialised struct says "none": one stream).  The early tiles read the tables and the early units' byte maps only, all complete behind
    // k_luts_tables; the chain writes the late units' histograms and byte maps only.  The down-scaling units (any of them may be late)
    // follow on the caller's stream behind the chain.  Not while the caller times the tile kernel (ev_before / ev_after): one stream then.
    const int npl = n_plain - ls.n_plain_early, nsl = n_sharp - ls.n_sharp_early;      // the late units close their class
    const int n_early = ls.n_plain_early + ls.n_sharp_early;
    AugFork* fk = (chain && ev_before == nullptr && ev_after == nullptr && n_early > 0)
                      ? aug_fork() : nullptr;
    hipStream_t cs = fk != nullptr ? fk->helper : st;                  // the chain's stream
    if (fk != nullptr) {
        AADG_HIP_TRY(hipEventRecord(fk->tables, st));
        AADG_HIP_TRY(hipStreamWaitEvent(fk->helper, fk->tables, 0));
    }
    if (chain)
        for (int k = 1; k < max_ops; ++k) {
            uint32_t* hist = hist0 + (size_t)k * hist_stage;
            if (((stats_mask >> k) & 1) && ls.n_stat[k] > 0) {
                const int rc = launch_hist_fused(pool, ur.units, ls.stat_units[k], ls.n_stat[k], ls.n_stat_stencil[k], k, Hs, Ws, crop, lut,
                                                 lut_stage_stride, hist, cs);
                if (rc) return rc;
            }
            hipLaunchKernelGGL(k_lut, dim3(n_late), dim3(256), 0, cs, ur, k, N, ls.late_units, npix, Hs, Ws, crop, (const uint32_t*)hist0,
                               (const uint32_t*)hist, ls.pool_hist, lut);
            AADG_LAUNCH_CHECK();
        }
    if (ev_before) AADG_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_before), st));
    if (fk == nullptr) {
        const int rc = launch_tiles(pool, masks, ur.units, ls.order, n_plain, n_sharp, ls.order + n_plain + n_sharp, n_generic, ls.n_generic_sharp, Hs, Ws,
                                    crop, dsk, tab, lut, lut_stage_stride, reinterpret_cast<uint32_t*>(ws8 + L.hbuf), out_img, out_lbl, st, ls.gen_chunk, ls.n_generic_wonly);
        if (rc) return rc;
    } else {
        uint32_t* hbuf = reinterpret_cast<uint32_t*>(ws8 + L.hbuf);
        // helper: the chain is complete here
        if (n_generic > 0) AADG_HIP_TRY(hipEventRecord(fk->chain, fk->helper));
        // caller's stream: the early units' tiles (plain [0, n_plain - npl), Sharpness [0, n_sharp - nsl) of their class lists)
        int rc = launch_tiles(pool, masks, ur.units, ls.order, n_plain - npl, n_sharp - nsl, nullptr, 0, 0, Hs, Ws, crop, dsk, tab, lut, lut_stage_stride,
                              hbuf, out_img, out_lbl, st, 0, 0, ls.order + n_plain);
        if (rc) return rc;
        // helper: the late units' tiles
        if (npl + nsl > 0) {
            rc = launch_tiles(pool, masks, ur.units, ls.order + (n_plain - npl), npl, nsl, nullptr, 0, 0, Hs, Ws, crop, dsk, tab, lut, lut_stage_stride,
                              hbuf, out_img, out_lbl, fk->helper, 0, 0, ls.order + n_plain + (n_sharp - nsl));
            if (rc) return rc;
        }
        AADG_HIP_TRY(hipEventRecord(fk->done, fk->helper));
        // caller's stream: the down-scaling units behind the chain
        if (n_generic > 0) {
            AADG_HIP_TRY(hipStreamWaitEvent(st, fk->chain, 0));
            rc = launch_tiles(pool, masks, ur.units, nullptr, 0, 0, ls.order + n_plain + n_sharp, n_generic, ls.n_generic_sharp, Hs, Ws, crop, dsk, tab, lut,
                              lut_stage_stride, hbuf, out_img, out_lbl, st, ls.gen_chunk, ls.n_generic_wonly);
            if (rc) return rc;
        }
        AADG_HIP_TRY(hipStreamWaitEvent(st, fk->done, 0));               // join: the call is complete on the caller's stream
    }
    if (ev_after) AADG_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_after), st));
    return 0;
}

// Host-side planning of one aadg_aug_u8_forward_ex2 call (no GPU work): checks the unit records against what the kernels rely on and
// derives the work lists of `aadg_aug_lists` from them -- the same rules as unit_flow() / stats_by_pushforward() above, restated on the
// host records (aadg_amd/_lib.py: launch_plan is the Python statement of it; tests/test_abi_cpu.py compares the two).
extern "C" int aadg_aug_u8_plan(const aadg_unit* units, int N, int P, int Hs, int Ws, int crop, int32_t* order, int32_t* stat_units,
                                int32_t* late_units, int32_t* summary) {
    if (units == nullptr || order == nullptr || stat_units == nullptr || late_units == nullptr || summary == nullptr) return AADG_E_BADARG;
    if (N <= 0 || P <= 0 || Hs <= 0 || Ws <= 0 || crop <= 0) return AADG_E_BADARG;
    const bool tiles_ok = !((Ws & 3) || (crop & 3));
    int n_cls[6] = {0, 0, 0, 0, 0, 0}, n_stat[AADG_MAX_OPS], n_sten[AADG_MAX_OPS], n_late = 0, max_ops = 0, classes = 0, stats_mask = 0;
    int n_cls_late[2] = {0, 0};                 // ABI 12: late units of the plain / Sharpness up-scaling classes (listed last in their class)
    for (int k = 0; k < AADG_MAX_OPS; ++k) n_stat[k] = n_sten[k] = 0;
    // pass 1: validation, class and statistics lists (late_units doubles as the per-unit class until the counting sort below)
    for (int i = 0; i < N; ++i) {
        const aadg_unit& u = units[i];
        if (u.src < 0 || u.src >= P || u.n_ops < 0 || u.n_ops > AADG_MAX_OPS) return AADG_E_BADARG;
        if (u.scaled_w < 1 || u.scaled_h < 1 || (long long)u.scaled_w * 3 < Ws || (long long)u.scaled_h * 3 < Hs) return AADG_E_BADARG;
        int sharp = 0;
        int sharp_before[AADG_MAX_OPS];             // Sharpness stencils among ops [0, k)
        for (int k = 0; k < AADG_MAX_OPS; ++k) sharp_before[k] = 0;
        for (int k = 0; k < u.n_ops; ++k) {
            const int op = u.op[k];
            if (op < 0 || op >= AADG_OP_COUNT) return AADG_E_BADARG;
            if (op == AADG_OP_CUTOUT && (u.rect[k][0] < 0 || u.rect[k][1] < 0 || u.rect[k][2] >= Ws || u.rect[k][3] >= Hs)) return AADG_E_BADARG;
            if (op == AADG_OP_POSTERIZE && (u.iarg[k] < 0 || u.iarg[k] > 8)) return AADG_E_BADARG;
            sharp_before[k] = sharp;
            if (op == AADG_OP_SHARPNESS && u.farg[k] != 1.0f) ++sharp;
        }
        if (u.n_ops > max_ops) max_ops = u.n_ops;
        const bool ok = tiles_ok && sharp <= MAX_SHARP;
        const bool up = ok && u.scaled_w >= Ws && u.scaled_h >= Hs;
        const bool generic = ok && !up && 2 * (long long)u.scaled_w >= Ws && 2 * (long long)u.scaled_h >= Hs;
        // generic without a stencil: the units that shrink the width only first (ABI 9: k_fused3w's list), then the others
        const bool wonly = generic && sharp == 0 && u.scaled_h >= Hs && Ws >= 8;
        const int cls = up ? (sharp == 0 ? 0 : 1) : generic ? (sharp == 0 ? (wonly ? 2 : 3) : 4) : 5;
        classes |= up ? HINT_FUSED : generic ? HINT_GENERIC : HINT_STAGED;
        ++n_cls[cls];
        // statistics: a pixel pass per op that needs the image's statistics, unless the histogram can be pushed forward from the raw
        // image's (AutoContrast / Equalize in a slot k >= 1 behind per-channel byte maps only, tile-flow units); the raw histogram is
        // then the source: slot 0 gets the pass
        bool prefix_lut = true, any_push = false, late = false;
        bool pass[AADG_MAX_OPS];
        for (int k = 0; k < AADG_MAX_OPS; ++k) {
            const bool live = k < u.n_ops;
            const int op = u.op[k];
            const bool needs = live && (op == AADG_OP_AUTOCONTRAST || op == AADG_OP_EQUALIZE || op == AADG_OP_CONTRAST);
            const bool push = k >= 1 && live && (op == AADG_OP_AUTOCONTRAST || op == AADG_OP_EQUALIZE) && prefix_lut && (up || generic);
            pass[k] = needs && !push;
            any_push = any_push || push;
            prefix_lut = prefix_lut && (op == AADG_OP_AUTOCONTRAST || op == AADG_OP_INVERT || op == AADG_OP_EQUALIZE || op == AADG_OP_SOLARIZE ||
                                        op == AADG_OP_POSTERIZE || op == AADG_OP_CONTRAST || op == AADG_OP_BRIGHTNESS);
        }
        pass[0] = pass[0] || any_push;
        for (int k = 0; k < AADG_MAX_OPS; ++k)
            if (pass[k]) {
                stat_units[(size_t)k * N + n_stat[k]++] = i;
                if (sharp_before[k] > 0) ++n_sten[k];
                stats_mask |= 1 << k;
                late = late || k >= 1;
            }
        late_units[i] = cls | (late ? 8 : 0);
        if (late && cls < 2) ++n_cls_late[cls];
    }
    // slot k's list: the units with a Sharpness stencil in front of the op FIRST (k_hist_fused gives each of their tiles a workgroup of
    // its own), the others behind them; both parts keep the ascending unit order (stable partition; mixed lists are rare)
    for (int k = 1; k < AADG_MAX_OPS; ++k) {
        if (n_sten[k] == 0 || n_sten[k] == n_stat[k]) continue;
        int32_t* row = stat_units + (size_t)k * N;
        std::vector<int32_t> rest;
        rest.reserve((size_t)(n_stat[k] - n_sten[k]));
        int w = 0;
        for (int t = 0; t < n_stat[k]; ++t) {
            const aadg_unit& u = units[row[t]];
            int sb = 0;
            for (int j = 0; j < k; ++j) sb += (u.op[j] == AADG_OP_SHARPNESS && u.farg[j] != 1.0f) ? 1 : 0;
            if (sb > 0) row[w++] = row[t];
            else rest.push_back(row[t]);
        }
        for (size_t t = 0; t < rest.size(); ++t) row[w++] = rest[t];
    }
    // pass 2: stable counting sort by class; the late list in place (its write position never passes the read position)
    int off[6];
    off[0] = 0;
    for (int c = 1; c < 6; ++c) off[c] = off[c - 1] + n_cls[c - 1];
    int off_late[2] = {off[0] + n_cls[0] - n_cls_late[0], off[1] + n_cls[1] - n_cls_late[1]};
    for (int i = 0; i < N; ++i) {
        const int v = late_units[i], c = v & 7;
        if (c < 2 && (v & 8)) order[off_late[c]++] = i;                  // the late units of a class behind its early ones, both in unit order
        else order[off[c]++] = i;
        if (v & 8) late_units[n_late++] = i;
    }
    summary[0] = n_cls[0]; summary[1] = n_cls[1]; summary[2] = n_cls[2] + n_cls[3] + n_cls[4]; summary[3] = n_cls[4];
    summary[4] = n_late; summary[5] = classes; summary[6] = stats_mask; summary[7] = max_ops;
    for (int k = 0; k < AADG_MAX_OPS; ++k) summary[8 + k] = n_stat[k];
    for (int k = 0; k < AADG_MAX_OPS; ++k) summary[8 + AADG_MAX_OPS + k] = n_sten[k];
    summary[8 + 2 * AADG_MAX_OPS] = n_cls[2];
    summary[9 + 2 * AADG_MAX_OPS] = n_cls[0] - n_cls_late[0];           // ABI 12: the units of the class that wait for no statistics pass
    summary[10 + 2 * AADG_MAX_OPS] = n_cls[1] - n_cls_late[1];
    return 0;
}

extern "C" int aadg_pool_histograms_u8(const uint8_t* pool, int P, int Hs, int Ws, uint32_t* hist, void* stream) {
    if (pool == nullptr || hist == nullptr || P <= 0 || Hs <= 0 || Ws <= 0) return AADG_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    AADG_HIP_TRY(hipMemsetAsync(hist, 0, (size_t)P * HIST_STRIDE * 4, st));
    hipLaunchKernelGGL(k_pool_hist, dim3(chunks_for(Hs * Ws), P), dim3(256), 0, st, pool, (size_t)Hs * Ws * 3, Hs * Ws, hist);
    AADG_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t aadg_aug_u8_workspace_bytes(int N, int Hs, int Ws, int crop) {
    if (N <= 0 || Hs <= 0 || Ws <= 0 || crop < 0) return 0;
    return ws_layout(N, Hs, Ws, crop).total;
}

extern "C" int aadg_aug_u8_forward_ex2(const uint8_t* pool, const uint8_t* masks, int P, int Hs, int Ws,
                                       const aadg_unit* units, int N, int max_ops, int crop, int dataset,
                                       float* out_img, float* out_lbl, void* ws, size_t ws_bytes, void* stream,
                                       int classes_hint, int stats_mask_hint, void* ev_before_final, void* ev_after_final,
                                       const aadg_aug_lists* lists) {
    const int32_t* order = lists ? lists->order : nullptr;
    const int n_plain = lists ? lists->n_plain : 0, n_sharp = lists ? lists->n_sharp : 0, n_generic = lists ? lists->n_generic : 0;
    if (lists != nullptr)
        for (int k = 0; k < AADG_MAX_OPS; ++k)
            if (lists->n_stat[k] < 0 || lists->n_stat[k] > N || (lists->n_stat[k] > 0 && lists->stat_units[k] == nullptr) ||
                lists->n_stat_stencil[k] < 0 || lists->n_stat_stencil[k] > lists->n_stat[k]) return AADG_E_BADARG;
    if (lists != nullptr && lists->gen_chunk < 0) return AADG_E_BADARG;
    if (!pool || !masks || !units || !out_img || !out_lbl || !ws) return AADG_E_BADARG;
    if (P <= 0 || Hs <= 0 || Ws <= 0 || N <= 0 || crop <= 0) return AADG_E_BADARG;
    if (max_ops < 0 || max_ops > AADG_MAX_OPS) return AADG_E_BADARG;
    if (dataset != AADG_DATASET_OPTIC && dataset != AADG_DATASET_VESSEL) return AADG_E_BADARG;
    if ((size_t)N * (size_t)crop > (size_t)1 << 28) return AADG_E_BADARG;
    if (order != nullptr && (n_plain < 0 || n_sharp < 0 || n_generic < 0 || (long long)n_plain + n_sharp + n_generic > N)) return AADG_E_BADARG;
    if (order != nullptr && (lists->n_generic_sharp < 0 || lists->n_generic_sharp > n_generic)) return AADG_E_BADARG;
    if (order != nullptr && (lists->n_generic_wonly < 0 || lists->n_generic_wonly > n_generic - lists->n_generic_sharp)) return AADG_E_BADARG;
    if (order != nullptr && (lists->n_plain_early < 0 || lists->n_plain_early > n_plain || lists->n_sharp_early < 0 || lists->n_sharp_early > n_sharp))
        return AADG_E_BADARG;                                     // ABI 12
    const WsLayout L = ws_layout(N, Hs, Ws, crop);
    const int dsk = aug_dataset_arg(dataset, N, crop);
    if (ws_bytes < L.total) return AADG_E_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    uint8_t* ws8 = reinterpret_cast<uint8_t*>(ws);
    // pool images and workspace images are both densely packed (Hs*Ws*3 bytes each)
    Bufs bufs{pool, ws8 + L.buf0, ws8 + L.buf1, (size_t)Hs * Ws * 3};
    UnitRef ur;
    ur.units = units;
    ur.use_single = 0;
    ur.allow_fused = 1;
    ::memset(&ur.single, 0, sizeof(ur.single));
    int classes = classes_hint & (HINT_FUSED | HINT_STAGED | HINT_GENERIC);
    if (classes == 0) classes = HINT_FUSED | HINT_STAGED | HINT_GENERIC;
    if ((Ws & 3) || (crop & 3)) classes = HINT_STAGED;        // unit_fusable() is false for every unit
    int stats_mask = stats_mask_hint < 0 ? 0xF : stats_mask_hint;
    if (stats_mask & ~1) stats_mask |= 1;      // a later stage's statistics may be pushed forward from the raw image's histogram
    int* tab = reinterpret_cast<int*>(ws8 + L.tab);
    if (lists != nullptr && lists->pool_hist != nullptr && lists->late_units != nullptr && order != nullptr &&
        lists->stat_units[0] != nullptr && !(classes & HINT_STAGED))
        return forward_cached(pool, masks, ur, N, Hs, Ws, max_ops, crop, dataset, out_img, out_lbl, ws8, L, st, classes, stats_mask,
                              ev_before_final, ev_after_final, *lists);
    bool tables_done = false;
    int rc = run_stages(bufs, ur, N, Hs, Ws, crop, max_ops, ws8, L, nullptr, classes, stats_mask, st,
                        (lists != nullptr && lists->stat_units[0] != nullptr) ? lists : nullptr, tab, &tables_done, lists);
    if (rc) return rc;
    if (!tables_done) {
        hipLaunchKernelGGL(k_tables, dim3(N), dim3(256), 0, st, ur, Hs, Ws, crop, tab);
        AADG_LAUNCH_CHECK();
    }
    if (ev_before_final) AADG_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_before_final), st));
    const int chunk_req = lists ? lists->gen_chunk : 0;
    if (classes & HINT_FUSED) {
        // with the caller's class lists: one z-slice per plain unit, two per Sharpness unit, the down-scaling units' passes over their
        // list; without: every unit is offered every kind of slice
        const int np = order ? n_plain : N, ns = order ? n_sharp : N;
        const int ng = (classes & HINT_GENERIC) ? (order ? n_generic : N) : 0;
        const int rc2 = launch_tiles(pool, masks, units, order, np, ns, order ? order + n_plain + n_sharp : nullptr, ng,
                                     lists ? lists->n_generic_sharp : 0, Hs, Ws, crop, dsk, tab, ws8 + L.lut, (size_t)N * 768,
                                     reinterpret_cast<uint32_t*>(ws8 + L.hbuf), out_img, out_lbl, st, chunk_req, lists ? lists->n_generic_wonly : 0);
        if (rc2) return rc2;
    } else if (classes & HINT_GENERIC) {
        const int ng = order ? n_generic : N;
        const int rc2 = launch_tiles(pool, masks, units, nullptr, 0, 0, order ? order + n_plain + n_sharp : nullptr, ng, lists ? lists->n_generic_sharp : 0,
                                     Hs, Ws, crop, dsk, tab, ws8 + L.lut, (size_t)N * 768, reinterpret_cast<uint32_t*>(ws8 + L.hbuf), out_img, out_lbl,
                                     st, chunk_req, lists ? lists->n_generic_wonly : 0);
        if (rc2) return rc2;
    }
    if (classes & HINT_STAGED) {
        const dim3 g((crop + 255) / 256, (crop + FIN_ROWS - 1) / FIN_ROWS, N);
        hipLaunchKernelGGL(k_final, g, dim3(256), 0, st, bufs, masks, ur, Hs, Ws, crop, dsk, tab, out_img, out_lbl);
        AADG_LAUNCH_CHECK();
    }
    if (ev_after_final) AADG_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_after_final), st));
    return 0;
}

extern "C" int aadg_aug_u8_forward_ex(const uint8_t* pool, const uint8_t* masks, int P, int Hs, int Ws,
                                      const aadg_unit* units, int N, int max_ops, int crop, int dataset,
                                      float* out_img, float* out_lbl, void* ws, size_t ws_bytes, void* stream,
                                      int classes_hint, int stats_mask_hint, void* ev_before_final, void* ev_after_final) {
    return aadg_aug_u8_forward_ex2(pool, masks, P, Hs, Ws, units, N, max_ops, crop, dataset, out_img, out_lbl, ws, ws_bytes, stream,
                                   classes_hint, stats_mask_hint, ev_before_final, ev_after_final, nullptr);
}

extern "C" int aadg_aug_u8_forward(const uint8_t* pool, const uint8_t* masks, int P, int Hs, int Ws,
                                   const aadg_unit* units, int N, int max_ops, int crop, int dataset,
                                   float* out_img, float* out_lbl, void* ws, size_t ws_bytes, void* stream) {
    return aadg_aug_u8_forward_ex(pool, masks, P, Hs, Ws, units, N, max_ops, crop, dataset, out_img, out_lbl, ws,
                                  ws_bytes, stream, 0, -1, nullptr, nullptr);
}

extern "C" int aadg_op_u8(const uint8_t* in, uint8_t* out, int H, int W, int op, int iarg, float farg,
                          const int32_t* rect_host, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !out || !ws || H <= 0 || W <= 0) return AADG_E_BADARG;
    if (op < 0 || op >= AADG_OP_COUNT) return AADG_E_BADARG;
    if (in == out) return AADG_E_BADARG;
    const WsLayout L = ws_layout(1, H, W, 0);
    if (ws_bytes < L.total) return AADG_E_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    uint8_t* ws8 = reinterpret_cast<uint8_t*>(ws);
    Bufs bufs{in, ws8 + L.buf0, ws8 + L.buf1, (size_t)H * W * 3};
    UnitRef ur;
    ur.units = nullptr;
    ur.use_single = 1;
    ur.allow_fused = 0;
    ::memset(&ur.single, 0, sizeof(ur.single));
    ur.single.src = 0;
    ur.single.n_ops = 1;
    ur.single.op[0] = op;
    ur.single.iarg[0] = iarg;
    ur.single.farg[0] = farg;
    for (int i = 0; i < 4; ++i) ur.single.rect[0][i] = rect_host ? rect_host[i] : (i < 2 ? 0 : -1);
    ur.single.scaled_w = W;
    ur.single.scaled_h = H;
    return run_stages(bufs, ur, 1, H, W, 0, 1, ws8, L, out, HINT_STAGED, 1, st);
}

// Training / inference BatchNorm2d over NCHW planes fused with the activation (ReLU / ReLU6) and the residual add
// that follow it in the segmentation backbone -- the consumer of the augmentation batch.  At
// N = 144 x 512 x 512 the 62 BatchNorm layers of DeepLabV3+/ResNet-50 move ~21 GB per pass over the activations;
// the library kernels they replace launch one workgroup per channel (64 workgroups for the stem on a 256-CU part).
// Here every phase is a streaming pass with 16-byte accesses and a (channel, split) grid:
//
//   k_bn_reduce_fwd    per (c, split): sum x, sum x^2                       -> partials (float2)
//   k_bn_apply         every wave combines its channel's partials (float64): mean, invstd, scale = w*invstd, shift = b - mean*scale;
//                      y = act(x * scale + shift (+ residual)); the workgroup of image 0 stores saved / running statistics
//   k_bn_reduce_bwd    per (c, split): sum g, sum g*xhat with g = (dy + further gradients) * act'(.)  (optionally writes g)
//   k_bn_dx            every wave combines the partials into the coefficients of dx = a*g + b*x + c0; image 0 stores dweight, dbias
//
// Two launches per layer and direction: the per-channel finalisation lives inside the elementwise passes (see bn_combine).
// Arithmetic = torch.nn.functional.batch_norm (biased variance for normalisation, unbiased for running_var),
// float32 accumulation, partial sums combined in float64.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

constexpr int BN_MAX_SPLIT = 64;
// grid of the reduction kernels.  (split, C) -- the channel in blockIdx.y -- made neighbouring workgroups read the SAME channel of
// images 16 MB apart; (C, split) makes them read neighbouring channels of one image: one contiguous sweep of the tensor (round 6)
#ifndef BN_GRID_C_FAST
#define BN_GRID_C_FAST 1
#endif
#if BN_GRID_C_FAST
#define BN_RED_C blockIdx.x
#define BN_RED_S blockIdx.y
#define BN_RED_NS gridDim.y
#define BN_RED_GRID(split, C) dim3((C), (split))
#else
#define BN_RED_C blockIdx.y
#define BN_RED_S blockIdx.x
#define BN_RED_NS gridDim.x
#define BN_RED_GRID(split, C) dim3((split), (C))
#endif

// ---- 16-byte vectors of T ---------------------------------------------------------------------------------
// Outputs larger than this are written with streaming stores: they would not survive in the 256 MB Infinity Cache until their
// consumer reads them and only displace what the kernel is reading (+0.4 % per step at 144 rows); smaller ones (all of a per-rank
// batch of 18 rows) stay cacheable -- there the consumer does find them (streaming everything cost 1 % at 18 rows).
constexpr size_t BN_STREAM_BYTES = (size_t)128 << 20;
constexpr int BN_STREAM = 0x100;          // rides in the run-time activation argument of the elementwise kernels
static inline bool bn_stream(size_t bytes) { return bytes > BN_STREAM_BYTES; }

template <typename T> struct Pack;
template <> struct Pack<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const float4 t = aadg_load_stream(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void load_keep(const float* p, float* v) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* v, bool stream) {
        aadg_store_out(p, make_float4(v[0], v[1], v[2], v[3]), stream);
    }
    static __device__ __forceinline__ float load1(const float* p) { return *p; }
    static __device__ __forceinline__ void store1(float* p, float v) { *p = v; }
    static __device__ __forceinline__ float round(float v) { return v; }
};
template <> struct Pack<__hip_bfloat16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const __hip_bfloat16* p, float* v) {
        // every activation / gradient stream of these kernels is read once per pass: streaming (non-temporal) loads keep them from
        // flushing L2 / the Infinity Cache on their way through (-1.4 ... -1.8 ms of 98.7 per step at 144 rows, -0.1 of 18.8 at 18 rows)
        const uint4 t = aadg_load_stream(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
    static __device__ __forceinline__ void load_keep(const __hip_bfloat16* p, float* v) {      // a stream the next kernel reads again
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
    static __device__ __forceinline__ void store(__hip_bfloat16* p, const float* v, bool stream) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = aadg_f2bf_pk(v[2 * i], v[2 * i + 1]);
        aadg_store_out(p, make_uint4(w[0], w[1], w[2], w[3]), stream);
    }
    static __device__ __forceinline__ float load1(const __hip_bfloat16* p) {
        return __uint_as_float((uint32_t)(*reinterpret_cast<const uint16_t*>(p)) << 16);
    }
    static __device__ __forceinline__ void store1(__hip_bfloat16* p, float v) {
        *reinterpret_cast<uint16_t*>(p) = (uint16_t)aadg_f2bf_bits(v);
    }
    static __device__ __forceinline__ float round(float v) { return __uint_as_float(aadg_f2bf_pk(v, v) & 0xFFFF0000u); }
};

// activation on the value the forward STORED (rounded to T): the backward re-derives the mask from the same value
__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == AADG_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == AADG_ACT_RELU6) return fminf(fmaxf(v, 0.0f), 6.0f);
    return v;
}
__device__ __forceinline__ bool act_open(float v, int act) {   // derivative is 1 (else 0)
    if (act == AADG_ACT_RELU) return v > 0.0f;
    if (act == AADG_ACT_RELU6) return v > 0.0f && v < 6.0f;
    return true;
}

struct BnWs {      // float offsets into the workspace
    size_t partial;   // [C][split] float2
    size_t scale, shift, ca, cb, cc;   // [C] each
    size_t total;
};
__host__ __device__ inline BnWs bn_ws(int C) {
    BnWs w;
    size_t o = 0;
    w.partial = o; o += (size_t)C * BN_MAX_SPLIT * 2;
    w.scale = o; o += C;
    w.shift = o; o += C;
    w.ca = o; o += C;
    w.cb = o; o += C;
    w.cc = o; o += C;
    w.total = o;
    return w;
}

// work split of one channel: N strips of `len` vectors (len = HW / VEC); a block owns pieces p = split, split + S, ...
// where a piece is up to `plen` consecutive vectors of one strip.
struct Pieces {
    int per_strip;   // pieces per strip
    int plen;        // vectors per piece (multiple of the block size)
    int total;       // N * per_strip
};
inline Pieces make_pieces(int N, int len, int threads) {
    Pieces p;
    const int span = threads * 4;                          // 4 vectors in flight per thread
    p.per_strip = (len + span - 1) / span;
    p.plen = span;
    p.total = N * p.per_strip;
    return p;
}
inline int pick_threads(int len) { return len >= 256 ? 256 : (len > 128 ? 256 : (len > 64 ? 128 : 64)); }
inline int pick_split(int C, int pieces) {
    int s = (4096 + C - 1) / C;                            // aim at >= 4096 workgroups
    if (s > pieces) s = pieces;
    if (s > BN_MAX_SPLIT) s = BN_MAX_SPLIT;
    return s < 1 ? 1 : s;
}

__device__ __forceinline__ float2 block_sum2(float a, float b) {
    __shared__ float red[2][4];
    a = wave_sum(a); b = wave_sum(b);
    const int wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = a; red[1][wv] = b; }
    __syncthreads();
    float2 r = make_float2(0.f, 0.f);
    for (int i = 0; i < nw; ++i) { r.x += red[0][i]; r.y += red[1][i]; }
    return r;
}

// The per-channel finalisation (combine the <= 64 partial sums in float64, derive the coefficients) used to be two tiny
// kernels per BatchNorm layer and direction: ~190 launches of ~5 us per step, a fixed cost that does not shrink with the
// per-rank batch.  Every wave of the elementwise kernels now redoes it for its own channel: 64 lanes load one partial each,
// a float64 butterfly leaves the same totals in every lane (same order in every workgroup of the channel: identical
// coefficients), and the workgroup of image 0 writes what has to be stored (saved / running statistics, dweight, dbias).
// split < 0: synchronised statistics -- `partial` holds the all-reduced float64 totals [C][2] (no partials to combine)
__device__ __forceinline__ void bn_combine(const float* __restrict__ partial, int c, int split, double* s, double* q) {
    if (split < 0) {
        const double* t = reinterpret_cast<const double*>(partial);
        *s = t[2 * c]; *q = t[2 * c + 1];
        return;
    }
    const int lane = threadIdx.x & 63;
    double a = 0.0, b = 0.0;
    if (lane < split) {
        const float2 p = reinterpret_cast<const float2*>(partial)[(size_t)c * BN_MAX_SPLIT + lane];
        a = (double)p.x; b = (double)p.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    *s = a; *q = b;
}
// scale / shift of channel c from the SAVED float32 statistics: the forward derives them the same way, so the backward
// recomputes bit-identical pre-activations
__device__ __forceinline__ void bn_scale_shift_of(const float* __restrict__ weight, const float* __restrict__ bias, float mean, float invstd,
                                                  int c, float* sc, float* sh) {
    const float w = weight != nullptr ? weight[c] : 1.0f, b = bias != nullptr ? bias[c] : 0.0f;
    *sc = w * invstd;
    *sh = fmaf(-mean, *sc, b);
}

constexpr int BN_MAX_EXTRA = 6;
template <typename T> struct BnExtra {      // further gradients of the same output (one per additional consumer)
    const T* p[BN_MAX_EXTRA];
    int n;
};

// ---- reductions: grid (split, C) ---------------------------------------------------------------------------
// Everything that does not change inside a launch (activation kind, where the activation mask comes from, the number of
// summed gradients, whether g is written back) is a template parameter: the loop bodies are then straight-line code, the
// `#pragma unroll 4` really puts four vectors per operand in flight, and there is no per-element branch.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_bn_reduce_fwd(const T* __restrict__ x, int C, int len, int per_strip, int plen, int total,
                                                       float* __restrict__ partial) {
    const int c = BN_RED_C, S = BN_RED_NS;
    const size_t strip_elems = (size_t)len * VEC;
    float s0 = 0.f, s1 = 0.f;
    for (int p = BN_RED_S; p < total; p += S) {
        const int n = p / per_strip, part = p - n * per_strip;
        const size_t base = ((size_t)n * C + c) * strip_elems;
        const int j1 = min(len, (part + 1) * plen);
#pragma unroll 4
        for (int j = part * plen + threadIdx.x; j < j1; j += blockDim.x) {
            float xv[VEC];
            if (VEC == 1) xv[0] = Pack<T>::load1(x + base + j); else Pack<T>::load_keep(x + base + (size_t)j * VEC, xv);
#pragma unroll
            for (int i = 0; i < VEC; ++i) { s0 += xv[i]; s1 = fmaf(xv[i], xv[i], s1); }
        }
    }
    const float2 r = block_sum2(s0, s1);
    if (threadIdx.x == 0) reinterpret_cast<float2*>(partial)[(size_t)c * BN_MAX_SPLIT + BN_RED_S] = r;
}

// backward: g = (dy + extra gradients) * act'(v), v = the value the forward stored; sums g and g * (x - mean) * invstd;
// writes g to dres when asked to.
//   MK 0: no activation (every element open)     MK 1: the forward's bit mask (one byte per 16-byte vector)
//   MK 2: generic -- v read from y when given, else recomputed from x; `act` is a run-time value (scalar / odd shapes)
//   MK 3 / 4: ReLU / ReLU6 with v recomputed from x exactly as the forward computed it (BatchNorm without a fused residual)
//   NE  : number of extra gradients (-1 = more.n at run time);  DRES: 0 / 1 (-1 = dres may or may not be null)
// DUAL (ABI 10): the residual of this BatchNorm is the output of a SECOND BatchNorm without activation (a projection shortcut normalised
// on load, aadg_bn_forward_res_affine_f32): its gradient is this layer's masked gradient g, so its backward sums -- sum g (the same) and
// sum g * xhat2 -- are taken in the same pass over g: one more read (x2) instead of a reduction pass of its own (two reads).
template <typename T> struct BnRed2 { const T* x2; const float* mean2; const float* invstd2; float* partial2; };
template <typename T, int VEC, int MK, int NE, int DRES, bool DUAL = false>
__global__ __launch_bounds__(256) void k_bn_reduce_bwd(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ dy,
                                                       BnExtra<T> more, const float* __restrict__ pconst, const uint8_t* __restrict__ mask,
                                                       T* __restrict__ dres, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ weight, const float* __restrict__ bias, int act_rt,
                                                       int C, int len, int per_strip, int plen, int total, float* __restrict__ partial,
                                                       long long dy_img_stride, BnRed2<T> d2 = BnRed2<T>()) {
    static_assert(MK == 2 || VEC > 1, "the specialised variants are vector-only");
    const int act = act_rt & 0xFF;
    const bool stream = (act_rt & BN_STREAM) != 0;
    const int c = BN_RED_C, S = BN_RED_NS;
    const size_t strip_elems = (size_t)len * VEC;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const float mu = mean[c], is = invstd[c];
    float mu2 = 0.f, is2 = 0.f;
    if (DUAL) { mu2 = d2.mean2[c]; is2 = d2.invstd2[c]; }
    float sc = 0.f, sh = 0.f;
    if (MK >= 2) bn_scale_shift_of(weight, bias, mu, is, c, &sc, &sh);
    const bool write_g = DRES < 0 ? dres != nullptr : DRES != 0;
    for (int p = BN_RED_S; p < total; p += S) {
        const int n = p / per_strip, part = p - n * per_strip;
        const size_t strip = (size_t)n * C + c;
        const size_t base = strip * strip_elems;
        const int j1 = min(len, (part + 1) * plen);
        // dy may be a channel slice of a wider gradient (the backward of a concatenation): image stride given by the caller
        const T* dyp = dy + (dy_img_stride > 0 ? (size_t)n * (size_t)dy_img_stride + (size_t)c * strip_elems : base);
        // a consumer whose gradient is constant over each plane (a global average pool) hands in one value per plane
        const float pc = pconst != nullptr ? pconst[strip] : 0.0f;
#pragma unroll 4
        for (int j = part * plen + threadIdx.x; j < j1; j += blockDim.x) {
            const size_t off = base + (size_t)j * VEC;
            float xv[VEC], gv[VEC], yv[VEC];
            if (VEC == 1) { xv[0] = Pack<T>::load1(x + off); gv[0] = Pack<T>::load1(dyp + j); }
            else { Pack<T>::load_keep(x + off, xv); Pack<T>::load_keep(dyp + (size_t)j * VEC, gv); }
            uint32_t mbits = 0;
            if (MK == 1) mbits = mask[strip * len + j];
            float x2v[VEC];
            if (DUAL) {
                if (VEC == 1) x2v[0] = Pack<T>::load1(d2.x2 + off); else Pack<T>::load_keep(d2.x2 + off, x2v);
            }
            if (MK == 2 && y != nullptr) {
                if (VEC == 1) yv[0] = Pack<T>::load1(y + off); else Pack<T>::load(y + off, yv);
            }
            // the output had several consumers: their gradients are summed here, not by separate elementwise passes
            if (NE >= 0) {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    float g2[VEC];
                    Pack<T>::load(more.p[e] + off, g2);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) gv[i] += g2[i];
                }
            } else {
                for (int e = 0; e < more.n; ++e) {
                    float g2[VEC];
                    if (VEC == 1) g2[0] = Pack<T>::load1(more.p[e] + off); else Pack<T>::load(more.p[e] + off, g2);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) gv[i] += g2[i];
                }
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                gv[i] += pc;
                bool open = true;
                if (MK == 1) open = (mbits >> i) & 1u;
                if (MK == 2) open = act_open(y != nullptr ? yv[i] : Pack<T>::round(fmaf(xv[i], sc, sh)), act);
                if (MK == 3) open = act_open(Pack<T>::round(fmaf(xv[i], sc, sh)), AADG_ACT_RELU);
                if (MK == 4) open = act_open(Pack<T>::round(fmaf(xv[i], sc, sh)), AADG_ACT_RELU6);
                const float g = open ? gv[i] : 0.0f;
                gv[i] = g;
                s0 += g;
                s1 = fmaf(g, (xv[i] - mu) * is, s1);
                if (DUAL) s2 = fmaf(g, (x2v[i] - mu2) * is2, s2);
            }
            if (write_g) {
                if (VEC == 1) Pack<T>::store1(dres + off, gv[0]); else Pack<T>::store(dres + off, gv, stream);
            }
        }
    }
    const float2 r = block_sum2(s0, s1);
    if (threadIdx.x == 0) reinterpret_cast<float2*>(partial)[(size_t)c * BN_MAX_SPLIT + BN_RED_S] = r;
    if (DUAL) {
        __syncthreads();                                          // block_sum2's scratch is read by every thread
        const float2 r2 = block_sum2(s0, s2);
        if (threadIdx.x == 0) reinterpret_cast<float2*>(d2.partial2)[(size_t)c * BN_MAX_SPLIT + BN_RED_S] = r2;
    }
}

// inference: scale / shift from the running statistics
__global__ __launch_bounds__(256) void k_bn_scale_shift(int C, const float* __restrict__ weight, const float* __restrict__ bias,
                                                        const float* __restrict__ mean, const float* __restrict__ var,
                                                        float eps, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    bn_scale_shift_of(weight, bias, mean[c], 1.0f / sqrtf(var[c] + eps), c, scale + c, shift + c);
}

// ---- synchronised statistics (data-parallel ranks): the reduction's partial sums leave the device-local
// workspace as float64 totals, are summed over the ranks by the caller (one all-reduce of 2C + 1 doubles), and the elementwise
// kernels read the totals (bn_combine with split < 0) and the element count straight from that buffer.
__global__ __launch_bounds__(64) void k_bn_pack(const float* __restrict__ partial, int split, int C, double count_local,
                                                double* __restrict__ sums, double* __restrict__ count_out,
                                                float* __restrict__ dweight, float* __restrict__ dbias) {
    const int c = blockIdx.x;
    double a, b;
    bn_combine(partial, c, split, &a, &b);
    if (threadIdx.x == 0) {
        sums[2 * c] = a;
        sums[2 * c + 1] = b;
        // backward: the parameter gradients are the LOCAL sums (DDP averages them over the ranks like every other gradient)
        if (dweight != nullptr) dweight[c] = (float)b;
        if (dbias != nullptr) dbias[c] = (float)a;
        if (c == 0 && count_out != nullptr) *count_out = count_local;
    }
}

// ---- elementwise passes: grid (N*C strips, pieces per strip) ------------------------------------------------
// ACT >= 0: the activation is a compile-time constant (vector path); ACT < 0: the run-time `act` (scalar path).
// FIN: training -- the statistics of this channel are finalised here from the reduction's partial sums (BnFin); else
// scale / shift come from the arrays k_bn_scale_shift filled (inference).
struct BnFin {
    const float* partial; int split; double count;
    const float* weight; const float* bias;
    float* running_mean; float* running_var; float momentum, eps;
    float* save_mean; float* save_invstd;
    const double* count_dev;      // synchronised statistics: the all-reduced element count lives on the device (else null)
};
template <typename T, int VEC, int ACT, bool RES, bool MASK, bool FIN>
__global__ __launch_bounds__(256) void k_bn_apply(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                  uint8_t* __restrict__ mask, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, BnFin fin, int act_rt, int C, int len, int plen,
                                                  long long y_img_stride, const float* __restrict__ res_scale = nullptr,
                                                  const float* __restrict__ res_shift = nullptr) {
    // res_scale / res_shift (RES only, ABI 10): `res` is the INPUT of the projection shortcut's BatchNorm (no activation) and is
    // normalised here, r = fma(res, res_scale[c], res_shift[c]) -- what that BatchNorm's own pass would have stored, never written
    static_assert(!MASK || VEC > 1, "one mask byte per 16-byte vector");
    const int act = ACT >= 0 ? ACT : (act_rt & 0xFF);
    const bool stream = (act_rt & BN_STREAM) != 0;
    const int strip = blockIdx.x, c = strip % C;
    float sc, sh;
    if (FIN) {
        double s, q;
        bn_combine(fin.partial, c, fin.split, &s, &q);
        const double count = fin.count_dev != nullptr ? *fin.count_dev : fin.count;
        const double m = s / count;
        double var = q / count - m * m;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)fin.eps));
        bn_scale_shift_of(fin.weight, fin.bias, (float)m, is, c, &sc, &sh);
        if (strip == c && blockIdx.y == 0 && threadIdx.x == 0) {      // image 0, first piece: the one writer of this channel
            fin.save_mean[c] = (float)m;
            fin.save_invstd[c] = is;
            if (fin.running_mean != nullptr) {
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                fin.running_mean[c] = (1.0f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)m;
                fin.running_var[c] = (1.0f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
            }
        }
    } else {
        sc = scale[c]; sh = shift[c];
    }
    const bool raff = RES && res_scale != nullptr;
    float rsc = 1.0f, rsh = 0.0f;
    if (raff) { rsc = res_scale[c]; rsh = res_shift[c]; }
    const size_t base = (size_t)strip * len * VEC;
    // y may be a channel slice of a wider tensor (written straight into a concatenation buffer): image stride given by the caller
    T* yp = y + (y_img_stride > 0 ? (size_t)(strip / C) * (size_t)y_img_stride + (size_t)c * len * VEC : base);
    const int j1 = min(len, ((int)blockIdx.y + 1) * plen);
#pragma unroll 4
    for (int j = blockIdx.y * plen + threadIdx.x; j < j1; j += blockDim.x) {
        const size_t off = base + (size_t)j * VEC;
        float v[VEC], r[VEC];
        if (VEC == 1) v[0] = Pack<T>::load1(x + off); else Pack<T>::load(x + off, v);
        if (RES) {
            if (VEC == 1) r[0] = Pack<T>::load1(res + off); else Pack<T>::load(res + off, r);
            if (raff) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) r[i] = Pack<T>::round(fmaf(r[i], rsc, rsh));
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float t = fmaf(v[i], sc, sh);
            if (RES) t = Pack<T>::round(t) + r[i];                // same roundings as bn (stored in T) followed by add
            v[i] = act_fwd(t, act);
        }
        if (MASK) {       // one byte per vector: bit i = act'(stored output i) != 0, read back by the backward
            uint32_t bits = 0;
#pragma unroll
            for (int i = 0; i < VEC; ++i) bits |= act_open(Pack<T>::round(v[i]), act) ? (1u << i) : 0u;
            mask[(size_t)strip * len + j] = (uint8_t)bits;
        }
        if (VEC == 1) Pack<T>::store1(yp + j, v[0]); else Pack<T>::store(yp + (size_t)j * VEC, v, stream);
    }
}

// ACT = activation whose mask is re-derived from x exactly as the forward did (0: none, or `dy` already holds the masked
// gradient g = dres); ACT < 0: run-time `act_rt`.  The channel's coefficients of dx = a*g + b*x + c0 are finalised here from
// the backward reduction's partial sums; the workgroup of image 0 stores dweight / dbias.
// DUAL: the second BatchNorm of k_bn_reduce_bwd<.., DUAL> -- its input gradient dx2 = a2 g + b2 x2 + c2 from the same g, in the same pass
template <typename T> struct BnDx2 {
    const T* x2; T* dx2; const float* partial2; const float* weight2; const float* mean2; const float* invstd2; float* dweight2; float* dbias2;
};
template <typename T, int VEC, int ACT, bool DUAL = false>
__global__ __launch_bounds__(256) void k_bn_dx(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                               const float* __restrict__ partial, int split, double count_host,
                                               const double* __restrict__ count_dev,
                                               const float* __restrict__ weight, const float* __restrict__ bias,
                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                               float* __restrict__ dweight, float* __restrict__ dbias, int act_rt, int C, int len,
                                               int plen, long long dy_img_stride, BnDx2<T> d2 = BnDx2<T>()) {
    const int act = ACT >= 0 ? ACT : (act_rt & 0xFF);
    const bool stream = (act_rt & BN_STREAM) != 0;
    const int strip = blockIdx.x, c = strip % C;
    double sg, sgx;
    bn_combine(partial, c, split, &sg, &sgx);
    const double count = count_dev != nullptr ? *count_dev : count_host;
    const float mu_f = mean[c], is_f = invstd[c];
    const double w = weight != nullptr ? (double)weight[c] : 1.0;
    const double is = (double)is_f, mu = (double)mu_f;
    const double ad = w * is;                             // dx = a*g - a*sg/n - a*is*sgx/n * (x - mu)
    const double bd = -ad * is * sgx / count;
    const float a = (float)ad, b = (float)bd, c0 = (float)(-ad * sg / count - bd * mu);
    if (strip == c && blockIdx.y == 0 && threadIdx.x == 0) {
        if (dweight != nullptr) dweight[c] = (float)sgx;
        if (dbias != nullptr) dbias[c] = (float)sg;
    }
    float sc = 0.f, sh = 0.f;
    if (ACT != 0) bn_scale_shift_of(weight, bias, mu_f, is_f, c, &sc, &sh);
    float a2 = 0.f, b2 = 0.f, c2 = 0.f;
    if (DUAL) {
        double sg2, sgx2;
        bn_combine(d2.partial2, c, split, &sg2, &sgx2);
        const double w2 = d2.weight2 != nullptr ? (double)d2.weight2[c] : 1.0;
        const double is2 = (double)d2.invstd2[c], mu2 = (double)d2.mean2[c];
        const double ad2 = w2 * is2, bd2 = -ad2 * is2 * sgx2 / count;
        a2 = (float)ad2; b2 = (float)bd2; c2 = (float)(-ad2 * sg2 / count - bd2 * mu2);
        if (strip == c && blockIdx.y == 0 && threadIdx.x == 0) {
            if (d2.dweight2 != nullptr) d2.dweight2[c] = (float)sgx2;
            if (d2.dbias2 != nullptr) d2.dbias2[c] = (float)sg2;
        }
    }
    const size_t base = (size_t)strip * len * VEC;
    const T* dyp = dy + (dy_img_stride > 0 ? (size_t)(strip / C) * (size_t)dy_img_stride + (size_t)c * len * VEC : base);
    const int j1 = min(len, ((int)blockIdx.y + 1) * plen);
#pragma unroll 4
    for (int j = blockIdx.y * plen + threadIdx.x; j < j1; j += blockDim.x) {
        const size_t off = base + (size_t)j * VEC;
        float xv[VEC], gv[VEC], x2v[VEC];
        if (VEC == 1) { xv[0] = Pack<T>::load1(x + off); gv[0] = Pack<T>::load1(dyp + j); }
        else { Pack<T>::load(x + off, xv); Pack<T>::load(dyp + (size_t)j * VEC, gv); }
        if (DUAL) {
            if (VEC == 1) x2v[0] = Pack<T>::load1(d2.x2 + off); else Pack<T>::load(d2.x2 + off, x2v);
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float g = gv[i];
            if (act != AADG_ACT_NONE && !act_open(Pack<T>::round(fmaf(xv[i], sc, sh)), act)) g = 0.0f;
            xv[i] = fmaf(a, g, fmaf(b, xv[i], c0));
            if (DUAL) x2v[i] = fmaf(a2, g, fmaf(b2, x2v[i], c2));
        }
        if (VEC == 1) Pack<T>::store1(dx + off, xv[0]); else Pack<T>::store(dx + off, xv, stream);
        if (DUAL) {
            if (VEC == 1) Pack<T>::store1(d2.dx2 + off, x2v[0]); else Pack<T>::store(d2.dx2 + off, x2v, stream);
        }
    }
}

struct Shape {
    int N, C, HW, vec, len, threads;
    Pieces pc;
    int split;
};
template <typename T>
inline bool make_shape(int N, int C, int HW, const void* a, const void* b, const void* c, const void* d, Shape* s) {
    if (N <= 0 || C <= 0 || HW <= 0 || (long long)N * C > 0x7FFFFFFFLL) return false;
    constexpr int V = Pack<T>::N;
    const bool aligned = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15u) == 0;
    s->N = N; s->C = C; s->HW = HW;
    s->vec = (HW % V == 0 && aligned) ? V : 1;
    s->len = HW / s->vec;
    s->threads = pick_threads(s->len);
    s->pc = make_pieces(N, s->len, s->threads);
    s->split = pick_split(C, s->pc.total);
    return s->pc.per_strip <= 65535;
}

template <typename T>
int bn_forward(const T* x, const T* res, T* y, uint8_t* mask, const float* weight, const float* bias, float* rmean, float* rvar, float momentum,
               float eps, int act, int training, int N, int C, int HW, float* save_mean, float* save_invstd, float* ws,
               long long y_img_stride, hipStream_t st, int phase = 0, double* sums = nullptr, const float* res_scale = nullptr,
               const float* res_shift = nullptr) {
    // phase 0: everything on this device.  Synchronised statistics (training): phase 1 = local sums -> `sums` [2C + 1] doubles
    // (the last one is this rank's element count); the caller all-reduces `sums`; phase 2 = normalise with the totals.
    const int stream_flag = bn_stream((size_t)N * C * HW * sizeof(T)) ? BN_STREAM : 0;
    Shape s;
    if (!make_shape<T>(N, C, HW, x, res, y, nullptr, &s)) return AADG_E_BADARG;
    if (y_img_stride != 0 && (y_img_stride < (long long)C * HW || (s.vec > 1 && (y_img_stride % s.vec) != 0))) return AADG_E_BADARG;
    if (mask != nullptr && s.vec == 1) return AADG_E_BADARG;        // the bit mask exists for the vector path only (aadg_bn_mask_bytes)
    const BnWs L = bn_ws(C);
    float* scale = ws + L.scale;
    float* shift = ws + L.shift;
    const dim3 blk(s.threads);
    int split = s.split;
    const double* count_dev = nullptr;
    if (training && phase != 2) {
        const dim3 grid = BN_RED_GRID(s.split, C);
        if (s.vec > 1)
            hipLaunchKernelGGL((k_bn_reduce_fwd<T, Pack<T>::N>), grid, blk, 0, st, x, C, s.len, s.pc.per_strip, s.pc.plen, s.pc.total, ws + L.partial);
        else
            hipLaunchKernelGGL((k_bn_reduce_fwd<T, 1>), grid, blk, 0, st, x, C, s.len, s.pc.per_strip, s.pc.plen, s.pc.total, ws + L.partial);
        AADG_LAUNCH_CHECK();
        if (phase == 1) {
            hipLaunchKernelGGL(k_bn_pack, dim3(C), dim3(64), 0, st, (const float*)(ws + L.partial), s.split, C, (double)N * (double)HW,
                               sums, sums + 2 * (size_t)C, (float*)nullptr, (float*)nullptr);
            AADG_LAUNCH_CHECK();
            return 0;
        }
    } else if (training) {
        split = -1;                                             // the elementwise kernel reads the float64 totals directly
        count_dev = sums + 2 * (size_t)C;
    } else {
        hipLaunchKernelGGL(k_bn_scale_shift, dim3((C + 255) / 256), dim3(256), 0, st, C, weight, bias, (const float*)rmean,
                           (const float*)rvar, eps, scale, shift);
        AADG_LAUNCH_CHECK();
    }
    const dim3 grid(N * C, s.pc.per_strip);
    const BnFin fin = {split < 0 ? reinterpret_cast<const float*>(sums) : ws + L.partial, split, (double)N * (double)HW, weight, bias, rmean,
                       rvar, momentum, eps, save_mean, save_invstd, count_dev};
#define AADG_BN_APPLY(VEC_, ACT_, RES_, MASK_, FIN_) \
    hipLaunchKernelGGL((k_bn_apply<T, VEC_, ACT_, RES_, MASK_, FIN_>), grid, blk, 0, st, x, res, y, mask, (const float*)scale, (const float*)shift, fin, act | stream_flag, C, s.len, s.pc.plen, y_img_stride, res_scale, res_shift)
#define AADG_BN_APPLY_ACT(ACT_)                                                          \
    do {                                                                                 \
        if (!training) { if (res != nullptr) AADG_BN_APPLY(Pack<T>::N, ACT_, true, false, false); else AADG_BN_APPLY(Pack<T>::N, ACT_, false, false, false); } \
        else if (res != nullptr) { if (mask != nullptr) AADG_BN_APPLY(Pack<T>::N, ACT_, true, true, true); else AADG_BN_APPLY(Pack<T>::N, ACT_, true, false, true); } \
        else { if (mask != nullptr) AADG_BN_APPLY(Pack<T>::N, ACT_, false, true, true); else AADG_BN_APPLY(Pack<T>::N, ACT_, false, false, true); }              \
    } while (0)
    if (s.vec == 1) {
        if (training) { if (res != nullptr) AADG_BN_APPLY(1, -1, true, false, true); else AADG_BN_APPLY(1, -1, false, false, true); }
        else { if (res != nullptr) AADG_BN_APPLY(1, -1, true, false, false); else AADG_BN_APPLY(1, -1, false, false, false); }
    } else if (act == AADG_ACT_RELU) AADG_BN_APPLY_ACT(AADG_ACT_RELU);
    else if (act == AADG_ACT_RELU6) AADG_BN_APPLY_ACT(AADG_ACT_RELU6);
    else AADG_BN_APPLY_ACT(AADG_ACT_NONE);
#undef AADG_BN_APPLY_ACT
#undef AADG_BN_APPLY
    AADG_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int bn_backward(const T* x, const T* y, const uint8_t* mask, const T* dy, const void* const* dy_extra, int n_extra, const float* pconst, const float* weight, const float* bias, const float* mean, const float* invstd,
                int act, T* dx, T* dres, float* dweight, float* dbias, int N, int C, int HW, float* ws, long long dy_img_stride,
                hipStream_t st, int phase = 0, double* sums = nullptr, const double* count_dev = nullptr, const BnDx2<T>* dual = nullptr,
                float* ws2 = nullptr) {
    // phase 0: everything on this device.  Synchronised statistics: phase 1 = masked gradient (dres) + local sums -> `sums`
    // [2C] doubles and the LOCAL dweight / dbias; the caller all-reduces `sums`; phase 2 = dx from the totals and the forward's
    // all-reduced element count (`count_dev`).
    const int stream_flag = bn_stream((size_t)N * C * HW * sizeof(T)) ? BN_STREAM : 0;
    Shape s;
    if (!make_shape<T>(N, C, HW, x, y, dy, dx, &s) || (((uintptr_t)dres & 15u) && s.vec > 1)) return AADG_E_BADARG;
    if (dy_img_stride != 0 && (dy_img_stride < (long long)C * HW || (s.vec > 1 && (dy_img_stride % s.vec) != 0))) return AADG_E_BADARG;
    if (mask != nullptr && s.vec == 1) return AADG_E_BADARG;
    const BnWs L = bn_ws(C);
    const dim3 blk(s.threads);
    BnExtra<T> more = {};
    for (int e = 0; e < n_extra; ++e) {
        if (dy_extra[e] == nullptr || (((uintptr_t)dy_extra[e]) & 15u)) return AADG_E_BADARG;
        more.p[e] = (const T*)dy_extra[e];
    }
    more.n = n_extra;
    int split = s.split;
    if (dual != nullptr) {
        // this layer's residual is a second BatchNorm normalised on load: both backward passes serve both layers
        // (phase 1 / 2, round 6: synchronised statistics -- `sums` [4C] doubles = this layer's (sum g, sum g x^) per channel, then the
        // shortcut layer's (sum g, sum g x2^); the caller all-reduces them between the two phases)
        if (s.vec == 1 || mask == nullptr || dres == nullptr || act == AADG_ACT_NONE || ws2 == nullptr) return AADG_E_UNSUPPORTED;
        if (phase != 2) {
            const BnRed2<T> r2 = {dual->x2, dual->mean2, dual->invstd2, ws2 + L.partial};
            const dim3 rgrid = BN_RED_GRID(s.split, C);
#define AADG_BN_REDUCE_DUAL(NE_)                                                                                                      \
    hipLaunchKernelGGL((k_bn_reduce_bwd<T, Pack<T>::N, 1, NE_, 1, true>), rgrid, blk, 0, st, x, y, dy, more, pconst, mask, dres, mean,    \
                       invstd, weight, bias, act | stream_flag, C, s.len, s.pc.per_strip, s.pc.plen, s.pc.total, ws + L.partial,          \
                       dy_img_stride, r2)
            if (n_extra == 0) AADG_BN_REDUCE_DUAL(0);
            else if (n_extra == 1) AADG_BN_REDUCE_DUAL(1);
            else AADG_BN_REDUCE_DUAL(-1);
#undef AADG_BN_REDUCE_DUAL
            AADG_LAUNCH_CHECK();
            if (phase == 1) {
                hipLaunchKernelGGL(k_bn_pack, dim3(C), dim3(64), 0, st, (const float*)(ws + L.partial), s.split, C, 0.0, sums,
                                   (double*)nullptr, dweight, dbias);
                hipLaunchKernelGGL(k_bn_pack, dim3(C), dim3(64), 0, st, (const float*)(ws2 + L.partial), s.split, C, 0.0,
                                   sums + 2 * (size_t)C, (double*)nullptr, dual->dweight2, dual->dbias2);
                AADG_LAUNCH_CHECK();
                return 0;
            }
        }
        BnDx2<T> d2 = *dual;
        d2.partial2 = phase == 2 ? reinterpret_cast<float*>(sums + 2 * (size_t)C) : ws2 + L.partial;
        if (phase == 2) { split = -1; dweight = nullptr; dbias = nullptr; d2.dweight2 = nullptr; d2.dbias2 = nullptr; }
        const dim3 dgrid(N * C, s.pc.per_strip);
        hipLaunchKernelGGL((k_bn_dx<T, Pack<T>::N, AADG_ACT_NONE, true>), dgrid, blk, 0, st, x, (const T*)dres, dx,
                           phase == 2 ? reinterpret_cast<const float*>(sums) : (const float*)(ws + L.partial), split,
                           (double)N * (double)HW, count_dev, weight, bias, mean, invstd, dweight,
                           dbias, AADG_ACT_NONE | stream_flag, C, s.len, s.pc.plen, 0LL, d2);
        AADG_LAUNCH_CHECK();
        return 0;
    }
    if (phase == 2) {
        split = -1;                                             // k_bn_dx reads the float64 totals directly
        dweight = nullptr; dbias = nullptr;                     // written by phase 1 (local sums)
    } else {
        const dim3 grid = BN_RED_GRID(s.split, C);
#define AADG_BN_REDUCE_BWD(VEC_, MK_, NE_, DRES_)                                                                                        \
    hipLaunchKernelGGL((k_bn_reduce_bwd<T, VEC_, MK_, NE_, DRES_>), grid, blk, 0, st, x, y, dy, more, pconst, mask, dres, mean, invstd,  \
                       weight, bias, act | stream_flag, C, s.len, s.pc.per_strip, s.pc.plen, s.pc.total, ws + L.partial, dy_img_stride)
#define AADG_BN_REDUCE_BWD_MK(MK_)                                                                     \
    do {                                                                                               \
        if (n_extra == 0) { if (dres != nullptr) AADG_BN_REDUCE_BWD(Pack<T>::N, MK_, 0, 1); else AADG_BN_REDUCE_BWD(Pack<T>::N, MK_, 0, 0); } \
        else if (n_extra == 1) AADG_BN_REDUCE_BWD(Pack<T>::N, MK_, 1, 1);   /* a residual block's output: trunk + identity branch */          \
        else AADG_BN_REDUCE_BWD(Pack<T>::N, MK_, -1, 1);                                                \
    } while (0)
        if (s.vec == 1) AADG_BN_REDUCE_BWD(1, 2, -1, -1);
        else if (act == AADG_ACT_NONE) AADG_BN_REDUCE_BWD_MK(0);
        else if (mask != nullptr) AADG_BN_REDUCE_BWD_MK(1);
        else if (y == nullptr && n_extra == 0 && dres == nullptr && act == AADG_ACT_RELU) AADG_BN_REDUCE_BWD(Pack<T>::N, 3, 0, 0);
        else if (y == nullptr && n_extra == 0 && dres == nullptr && act == AADG_ACT_RELU6) AADG_BN_REDUCE_BWD(Pack<T>::N, 4, 0, 0);
        else AADG_BN_REDUCE_BWD(Pack<T>::N, 2, -1, -1);
#undef AADG_BN_REDUCE_BWD_MK
#undef AADG_BN_REDUCE_BWD
        AADG_LAUNCH_CHECK();
        if (phase == 1) {
            hipLaunchKernelGGL(k_bn_pack, dim3(C), dim3(64), 0, st, (const float*)(ws + L.partial), s.split, C, 0.0, sums,
                               (double*)nullptr, dweight, dbias);
            AADG_LAUNCH_CHECK();
            return 0;
        }
    }
    {
        // when the masked gradient was materialised (dres), the last pass reads it instead of re-deriving the mask
        const T* g = dres != nullptr ? (const T*)dres : dy;
        const int g_ready = dres != nullptr ? 1 : 0;
        const dim3 grid(N * C, s.pc.per_strip);
        const int act_dx = g_ready ? AADG_ACT_NONE : act;
#define AADG_BN_DX(VEC_, ACT_)                                                                                                  \
    hipLaunchKernelGGL((k_bn_dx<T, VEC_, ACT_>), grid, blk, 0, st, x, g, dx,                                                       \
                       split < 0 ? reinterpret_cast<const float*>(sums) : (const float*)(ws + L.partial), split,                     \
                       (double)N * (double)HW, count_dev, weight, bias, mean, invstd, dweight, dbias, act_dx | stream_flag, C, s.len, s.pc.plen, g_ready ? 0LL : dy_img_stride)
        if (s.vec == 1) AADG_BN_DX(1, -1);
        else if (act_dx == AADG_ACT_RELU) AADG_BN_DX(Pack<T>::N, AADG_ACT_RELU);
        else if (act_dx == AADG_ACT_RELU6) AADG_BN_DX(Pack<T>::N, AADG_ACT_RELU6);
        else AADG_BN_DX(Pack<T>::N, AADG_ACT_NONE);
#undef AADG_BN_DX
        AADG_LAUNCH_CHECK();
    }
    return 0;
}

// ---- training BatchNorm + ReLU + MaxPool2d(3, 2, 1) in one pass (the ResNet stem) --------------------------------------
// The normalised map (the largest activation of the network: 64 channels at half resolution) is never written: one lane =
// 4 pooled outputs of one row, computed from 3 rows x 9 inputs that are normalised and rectified on the fly; the arg-max is
// stored as its window position (one byte per output), exactly as aadg_maxpool3x3s2_forward does, so the pooling backward is
// that kernel's.  A wave (256 outputs) lies inside one plane (Ho * Wo a multiple of 256), so bn_combine serves it.
template <typename T>
__global__ __launch_bounds__(256) void k_bn_relu_maxpool(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx,
                                                         BnFin fin, int C, int H, int W, int Ho, int Wo) {
    const int w4 = Wo / 4, per_plane = Ho * w4;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;      // grid sized exactly: planes * per_plane is a multiple of 256
    const long long plane = q / per_plane;
    const int rem = (int)(q - plane * per_plane);
    const int i = rem / w4, cg = rem - i * w4;
    const int c = (int)(plane % C);
    double s, sq;
    bn_combine(fin.partial, c, fin.split, &s, &sq);
    const double m = s / fin.count;
    double var = sq / fin.count - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)fin.eps));
    float sc, sh;
    bn_scale_shift_of(fin.weight, fin.bias, (float)m, is, c, &sc, &sh);
    if (plane == c && rem == 0) {                            // image 0, first lane of the plane: the one writer of this channel
        fin.save_mean[c] = (float)m;
        fin.save_invstd[c] = is;
        if (fin.running_mean != nullptr) {
            const double unbiased = fin.count > 1.0 ? var * fin.count / (fin.count - 1.0) : var;
            fin.running_mean[c] = (1.0f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)m;
            fin.running_var[c] = (1.0f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
        }
    }
    const T* px = x + (size_t)plane * H * W;
    float out[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    uint32_t at[4] = {4u, 4u, 4u, 4u};                       // the centre is always inside the image
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int r = 2 * i - 1 + a;
        if (r < 0 || r >= H) continue;
        const T* row = px + (size_t)r * W + 8 * cg;
        float v[9];
        Pack<T>::load(row, v + 1);                           // VEC = 8 for bfloat16; float32 takes the two-load path below
        if (Pack<T>::N == 4) Pack<T>::load(row + 4, v + 5);
        v[0] = cg > 0 ? Pack<T>::load1(row - 1) : 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = fmaxf(Pack<T>::round(fmaf(v[k], sc, sh)), 0.0f);   // what BatchNorm + ReLU would have stored
        if (cg == 0) v[0] = -INFINITY;                       // left padding never wins
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float e = v[2 * k + b];
                if (e > out[k]) { out[k] = e; at[k] = 3 * a + b; }
            }
    }
    const size_t o = (size_t)plane * Ho * Wo + (size_t)i * Wo + 4 * cg;
    if (Pack<T>::N == 8) {
        *reinterpret_cast<uint2*>(y + o) = make_uint2(aadg_f2bf_pk(out[0], out[1]), aadg_f2bf_pk(out[2], out[3]));
    } else {
        *reinterpret_cast<float4*>(y + o) = make_float4(out[0], out[1], out[2], out[3]);
    }
    *reinterpret_cast<uint32_t*>(idx + o) = at[0] | (at[1] << 8) | (at[2] << 16) | (at[3] << 24);
}

// ---- backward of BatchNorm + ReLU + MaxPool2d(3, 2, 1) --------------------------------------------------------------------
// The gradient of the normalised map is the pooling gather: input (r, c) receives dy of the <= 4 pooled outputs whose 3x3
// window covers it and whose stored arg-max is its window position.  Instead of materialising that map (the size of the
// stem activation) and reading it back twice, both BatchNorm backward passes rebuild their 8-column vector of it from the
// pooled gradient and the one-byte index.  bfloat16, W a multiple of 8 (one vector = 8 columns of one row).
__device__ __forceinline__ void pool_gather8(const uint8_t* __restrict__ idx, const __hip_bfloat16* __restrict__ dyp, int r, int cg,
                                             int Ho, int Wo, float* acc) {
    const int j0 = 4 * cg;
    const bool has5 = j0 + 4 < Wo;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
    for (int sft = 0; sft < 2; ++sft) {
        const int i = (r + sft) >> 1;
        const int a = r - 2 * i + 1;                         // window row of input row r in output row i
        if ((sft == 1 && !(r & 1)) || i >= Ho) continue;
        const size_t o = (size_t)i * Wo + j0;
        const uint32_t w = *reinterpret_cast<const uint32_t*>(idx + o);
        const uint2 t = *reinterpret_cast<const uint2*>(dyp + o);
        float gq[5] = {__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xFFFF0000u), __uint_as_float(t.y << 16),
                       __uint_as_float(t.y & 0xFFFF0000u), 0.0f};
        uint32_t p[5] = {w & 255u, (w >> 8) & 255u, (w >> 16) & 255u, w >> 24, 255u};
        if (has5) { p[4] = idx[o + 4]; gq[4] = Pack<__hip_bfloat16>::load1(dyp + o + 4); }
        const uint32_t base = 3u * (uint32_t)a;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = k >> 1;
            if (p[j] == base + 1u + (uint32_t)(k & 1)) acc[k] += gq[j];
            if ((k & 1) && p[j + 1] == base) acc[k] += gq[j + 1];
        }
    }
    // the unfused path stores this sum as bfloat16 before BatchNorm reads it: same rounding
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = Pack<__hip_bfloat16>::round(acc[k]);
}

// float32 (round 5): one vector = 4 columns 4 cg .. 4 cg + 3 of row r = the centre / right columns of the windows of outputs 2 cg and
// 2 cg + 1 and the left column of output 2 cg + 2's
__device__ __forceinline__ void pool_gather_vec(const uint8_t* __restrict__ idx, const float* __restrict__ dyp, int r, int cg, int Ho, int Wo,
                                                float* acc) {
    const int j0 = 2 * cg;
    const bool has3 = j0 + 2 < Wo;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = 0.f;
#pragma unroll
    for (int sft = 0; sft < 2; ++sft) {
        const int i = (r + sft) >> 1;
        const int a = r - 2 * i + 1;                         // window row of input row r in output row i
        if ((sft == 1 && !(r & 1)) || i >= Ho) continue;
        const size_t o = (size_t)i * Wo + j0;
        const uint32_t w = *reinterpret_cast<const uint16_t*>(idx + o);
        const float2 t = *reinterpret_cast<const float2*>(dyp + o);
        float gq[3] = {t.x, t.y, 0.0f};
        uint32_t p[3] = {w & 255u, w >> 8, 255u};
        if (has3) { p[2] = idx[o + 2]; gq[2] = dyp[o + 2]; }
        const uint32_t base = 3u * (uint32_t)a;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = k >> 1;
            if (p[j] == base + 1u + (uint32_t)(k & 1)) acc[k] += gq[j];
            if ((k & 1) && p[j + 1] == base) acc[k] += gq[j + 1];
        }
    }
}
__device__ __forceinline__ void pool_gather_vec(const uint8_t* __restrict__ idx, const __hip_bfloat16* __restrict__ dyp, int r, int cg, int Ho,
                                                int Wo, float* acc) {
    pool_gather8(idx, dyp, r, cg, Ho, Wo, acc);
}

// grid (split, C), as k_bn_reduce_bwd with MK = 3 (ReLU mask recomputed from x)
template <typename T>
__global__ __launch_bounds__(256) void k_bn_pool_reduce_bwd(const T* __restrict__ x, const uint8_t* __restrict__ idx,
                                                            const T* __restrict__ dyp, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ weight,
                                                            const float* __restrict__ bias, int C, int H, int W, int Ho, int Wo, int len,
                                                            int per_strip, int plen, int total, float* __restrict__ partial) {
    constexpr int V = Pack<T>::N;
    const int c = BN_RED_C, S = BN_RED_NS, w8 = W / V;
    float s0 = 0.f, s1 = 0.f;
    const float mu = mean[c], is = invstd[c];
    float sc, sh;
    bn_scale_shift_of(weight, bias, mu, is, c, &sc, &sh);
    for (int p = BN_RED_S; p < total; p += S) {
        const int n = p / per_strip, part = p - n * per_strip;
        const size_t strip = (size_t)n * C + c;
        const T* px = x + strip * (size_t)len * V;
        const uint8_t* pi = idx + strip * (size_t)Ho * Wo;
        const T* pg = dyp + strip * (size_t)Ho * Wo;
        const int j1 = min(len, (part + 1) * plen);
#pragma unroll 2
        for (int j = part * plen + threadIdx.x; j < j1; j += blockDim.x) {
            float xv[V], gv[V];
            Pack<T>::load(px + (size_t)j * V, xv);
            const int r = j / w8, cg = j - r * w8;
            pool_gather_vec(pi, pg, r, cg, Ho, Wo, gv);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const bool open = Pack<T>::round(fmaf(xv[i], sc, sh)) > 0.0f;
                const float g = open ? gv[i] : 0.0f;
                s0 += g;
                s1 = fmaf(g, (xv[i] - mu) * is, s1);
            }
        }
    }
    const float2 rr = block_sum2(s0, s1);
    if (threadIdx.x == 0) reinterpret_cast<float2*>(partial)[(size_t)c * BN_MAX_SPLIT + BN_RED_S] = rr;
}

// grid (N * C strips, pieces), as k_bn_dx with ACT = ReLU
template <typename T>
__global__ __launch_bounds__(256) void k_bn_pool_dx(const T* __restrict__ x, const uint8_t* __restrict__ idx,
                                                    const T* __restrict__ dyp, T* __restrict__ dx,
                                                    const float* __restrict__ partial, int split, double count,
                                                    const float* __restrict__ weight, const float* __restrict__ bias,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    float* __restrict__ dweight, float* __restrict__ dbias, int C, int H, int W, int Ho,
                                                    int Wo, int len, int plen, int stream_out) {
    constexpr int V = Pack<T>::N;
    const bool stream = stream_out != 0;
    const int strip = blockIdx.x, c = strip % C, w8 = W / V;
    double sg, sgx;
    bn_combine(partial, c, split, &sg, &sgx);
    const float mu_f = mean[c], is_f = invstd[c];
    const double wd = weight != nullptr ? (double)weight[c] : 1.0;
    const double is = (double)is_f, mu = (double)mu_f;
    const double ad = wd * is;
    const double bd = -ad * is * sgx / count;
    const float a = (float)ad, b = (float)bd, c0 = (float)(-ad * sg / count - bd * mu);
    if (strip == c && blockIdx.y == 0 && threadIdx.x == 0) {
        if (dweight != nullptr) dweight[c] = (float)sgx;
        if (dbias != nullptr) dbias[c] = (float)sg;
    }
    float sc, sh;
    bn_scale_shift_of(weight, bias, mu_f, is_f, c, &sc, &sh);
    const T* px = x + (size_t)strip * len * V;
    T* pdx = dx + (size_t)strip * len * V;
    const uint8_t* pi = idx + (size_t)strip * Ho * Wo;
    const T* pg = dyp + (size_t)strip * Ho * Wo;
    const int j1 = min(len, ((int)blockIdx.y + 1) * plen);
#pragma unroll 2
    for (int j = blockIdx.y * plen + threadIdx.x; j < j1; j += blockDim.x) {
        float xv[V], gv[V];
        Pack<T>::load(px + (size_t)j * V, xv);
        const int r = j / w8, cg = j - r * w8;
        pool_gather_vec(pi, pg, r, cg, Ho, Wo, gv);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float g = Pack<T>::round(fmaf(xv[i], sc, sh)) > 0.0f ? gv[i] : 0.0f;
            xv[i] = fmaf(a, g, fmaf(b, xv[i], c0));
        }
        Pack<T>::store(pdx + (size_t)j * V, xv, stream);
    }
}

}  // namespace

extern "C" size_t aadg_bn_workspace_bytes(int C) { return C > 0 ? bn_ws(C).total * sizeof(float) : 0; }

// bytes of the activation bit mask of an [N, C, HW] tensor (one byte per 16-byte vector), 0 when the vector path does not
// apply (HW not a multiple of the vector length): the backward of a fused residual then reads the stored output instead
extern "C" size_t aadg_bn_mask_bytes(int N, int C, int HW, int dtype) {
    const int V = dtype == 0 ? 4 : 8;
    if (N <= 0 || C <= 0 || HW <= 0 || (dtype != 0 && dtype != 1) || (HW % V) != 0) return 0;
    return (size_t)N * C * (HW / V);
}

extern "C" int aadg_bn_forward(const void* x, const void* residual, void* y, void* act_mask, const float* weight, const float* bias,
                               float* running_mean, float* running_var, float momentum, float eps, int act, int training, int N,
                               int C, int HW, int dtype, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes,
                               long long y_image_stride, void* stream) {
    if (x == nullptr || y == nullptr || ws == nullptr || act < 0 || act > AADG_ACT_RELU6 || y_image_stride < 0) return AADG_E_BADARG;
    if (training && (save_mean == nullptr || save_invstd == nullptr)) return AADG_E_BADARG;
    if (!training && (running_mean == nullptr || running_var == nullptr)) return AADG_E_BADARG;
    if (C <= 0 || ws_bytes < aadg_bn_workspace_bytes(C)) return AADG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        return bn_forward<float>((const float*)x, (const float*)residual, (float*)y, (uint8_t*)act_mask, weight, bias, running_mean,
                                 running_var, momentum, eps, act, training, N, C, HW, save_mean, save_invstd, (float*)ws, y_image_stride, st);
    if (dtype == 1)
        return bn_forward<__hip_bfloat16>((const __hip_bfloat16*)x, (const __hip_bfloat16*)residual, (__hip_bfloat16*)y,
                                          (uint8_t*)act_mask, weight, bias, running_mean, running_var, momentum, eps, act, training,
                                          N, C, HW, save_mean, save_invstd, (float*)ws, y_image_stride, st);
    return AADG_E_BADARG;
}

extern "C" int aadg_bn_backward(const void* x, const void* y, const void* act_mask, const void* dy, const void* const* dy_extra,
                                int n_extra, const float* dy_plane_const, const float* weight, const float* bias, const float* save_mean,
                                const float* save_invstd, int act, void* dx, void* dres, float* dweight, float* dbias, int N, int C,
                                int HW, int dtype, void* ws, size_t ws_bytes, long long dy_image_stride, void* stream) {
    if (x == nullptr || dy == nullptr || dx == nullptr || save_mean == nullptr || save_invstd == nullptr || ws == nullptr ||
        act < 0 || act > AADG_ACT_RELU6 || dy_image_stride < 0)
        return AADG_E_BADARG;
    // a fused residual needs the forward's activation mask: the bit mask it wrote, or the stored output
    if (dres != nullptr && y == nullptr && act_mask == nullptr) return AADG_E_BADARG;
    if (n_extra < 0 || n_extra > BN_MAX_EXTRA || (n_extra > 0 && (dy_extra == nullptr || dres == nullptr)))
        return AADG_E_BADARG;                                     // summed gradients are materialised as dres
    if (dy_plane_const != nullptr && dres == nullptr) return AADG_E_BADARG;
    if (C <= 0 || ws_bytes < aadg_bn_workspace_bytes(C)) return AADG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        return bn_backward<float>((const float*)x, (const float*)y, (const uint8_t*)act_mask, (const float*)dy, dy_extra, n_extra,
                                  dy_plane_const, weight, bias, save_mean, save_invstd, act, (float*)dx, (float*)dres, dweight, dbias, N, C, HW,
                                  (float*)ws, dy_image_stride, st);
    if (dtype == 1)
        return bn_backward<__hip_bfloat16>((const __hip_bfloat16*)x, (const __hip_bfloat16*)y, (const uint8_t*)act_mask,
                                           (const __hip_bfloat16*)dy, dy_extra, n_extra, dy_plane_const, weight, bias, save_mean, save_invstd, act,
                                           (__hip_bfloat16*)dx, (__hip_bfloat16*)dres, dweight, dbias, N, C, HW, (float*)ws, dy_image_stride, st);
    return AADG_E_BADARG;
}

/* The backward of aadg_bn_forward_res_affine_f32 (float32): BOTH BatchNorm layers in the two passes of one -- x2 / weight2 / mean2 /
 * invstd2 describe the projection shortcut's BatchNorm (no activation), whose output gradient is this layer's masked gradient g: its
 * sums are taken beside this layer's (one more read of x2), its input gradient dx2 = a2 g + b2 x2 + c2 is written beside dx.  `dres`
 * (required) receives g -- nobody else needs it.  ws2: a second workspace of aadg_bn_workspace_bytes(C).  (ABI 10) */
extern "C" int aadg_bn_backward_res_bn_f32(const float* x, const void* act_mask, const float* dy, const void* const* dy_extra, int n_extra,
                                           const float* dy_plane_const, const float* weight, const float* bias, const float* save_mean,
                                           const float* save_invstd, int act, float* dx, float* dres, float* dweight, float* dbias,
                                           const float* x2, const float* weight2, const float* save_mean2, const float* save_invstd2,
                                           float* dx2, float* dweight2, float* dbias2, int N, int C, int HW, void* ws, size_t ws_bytes,
                                           void* ws2, size_t ws2_bytes, long long dy_image_stride, void* stream) {
    if (x == nullptr || act_mask == nullptr || dy == nullptr || dx == nullptr || dres == nullptr || save_mean == nullptr ||
        save_invstd == nullptr || x2 == nullptr || save_mean2 == nullptr || save_invstd2 == nullptr || dx2 == nullptr || ws == nullptr ||
        ws2 == nullptr || act <= 0 || act > AADG_ACT_RELU6 || dy_image_stride < 0)
        return AADG_E_BADARG;
    if (n_extra < 0 || n_extra > BN_MAX_EXTRA || (n_extra > 0 && dy_extra == nullptr)) return AADG_E_BADARG;
    if ((((uintptr_t)x2 | (uintptr_t)dx2) & 15u) != 0) return AADG_E_BADARG;
    if (C <= 0 || ws_bytes < aadg_bn_workspace_bytes(C) || ws2_bytes < aadg_bn_workspace_bytes(C)) return AADG_E_WORKSPACE;
    const BnDx2<float> d2 = {x2, dx2, nullptr, weight2, save_mean2, save_invstd2, dweight2, dbias2};
    return bn_backward<float>(x, nullptr, (const uint8_t*)act_mask, dy, dy_extra, n_extra, dy_plane_const, weight, bias, save_mean, save_invstd,
                              act, dx, dres, dweight, dbias, N, C, HW, (float*)ws, dy_image_stride, (hipStream_t)stream, 0, nullptr, nullptr,
                              &d2, (float*)ws2);
}

/* aadg_bn_backward_res_bn_f32 with synchronised statistics (ABI 11): phase 1 = both layers' local sums -> `sums` [4C] doubles (this
 * layer's (sum g, sum g x^) per channel, then the shortcut's), the masked gradient dres and the LOCAL dweight / dbias / dweight2 / dbias2;
 * the caller all-reduces `sums`; phase 2 = dx and dx2 from the totals and the forward's all-reduced element count `count`. */
extern "C" int aadg_bn_sync_backward_res_bn_f32(int phase, const float* x, const void* act_mask, const float* dy, const void* const* dy_extra,
                                                int n_extra, const float* dy_plane_const, const float* weight, const float* bias,
                                                const float* save_mean, const float* save_invstd, int act, float* dx, float* dres,
                                                float* dweight, float* dbias, const float* x2, const float* weight2, const float* save_mean2,
                                                const float* save_invstd2, float* dx2, float* dweight2, float* dbias2, int N, int C, int HW,
                                                double* sums, const double* count, void* ws, size_t ws_bytes, void* ws2, size_t ws2_bytes,
                                                long long dy_image_stride, void* stream) {
    if ((phase != 1 && phase != 2) || x == nullptr || act_mask == nullptr || dy == nullptr || dx == nullptr || dres == nullptr ||
        save_mean == nullptr || save_invstd == nullptr || x2 == nullptr || save_mean2 == nullptr || save_invstd2 == nullptr || dx2 == nullptr ||
        ws == nullptr || ws2 == nullptr || sums == nullptr || act <= 0 || act > AADG_ACT_RELU6 || dy_image_stride < 0 ||
        (((uintptr_t)sums) & 7u) != 0 || (phase == 2 && count == nullptr))
        return AADG_E_BADARG;
    if (n_extra < 0 || n_extra > BN_MAX_EXTRA || (n_extra > 0 && dy_extra == nullptr)) return AADG_E_BADARG;
    if ((((uintptr_t)x2 | (uintptr_t)dx2) & 15u) != 0) return AADG_E_BADARG;
    if (C <= 0 || ws_bytes < aadg_bn_workspace_bytes(C) || ws2_bytes < aadg_bn_workspace_bytes(C)) return AADG_E_WORKSPACE;
    const BnDx2<float> d2 = {x2, dx2, nullptr, weight2, save_mean2, save_invstd2, dweight2, dbias2};
    return bn_backward<float>(x, nullptr, (const uint8_t*)act_mask, dy, dy_extra, n_extra, dy_plane_const, weight, bias, save_mean, save_invstd,
                              act, dx, dres, dweight, dbias, N, C, HW, (float*)ws, dy_image_stride, (hipStream_t)stream, phase, sums, count,
                              &d2, (float*)ws2);
}

// ---- synchronised statistics over data-parallel ranks (the reference's single-GPU batch mixes all domains in every
// BatchNorm batch; sharded replicas reproduce that by summing the per-channel statistics over the ranks).  The library stays
// free of any communication: phase 1 leaves this rank's float64 sums in `sums`, the CALLER all-reduces them (RCCL through
// torch.distributed), phase 2 consumes the totals.  training mode only.
//   forward : sums [2C + 1] doubles = (sum x, sum x^2) per channel + this rank's element count N * HW
//   backward: sums [2C] doubles = (sum g, sum g * xhat) per channel; `count` = the forward's all-reduced element count
//             (device pointer, sums_fwd + 2C); phase 1 also writes the LOCAL dweight / dbias and the masked gradient dres
extern "C" int aadg_bn_sync_forward(int phase, const void* x, const void* residual, void* y, void* act_mask, const float* weight,
                                    const float* bias, float* running_mean, float* running_var, float momentum, float eps, int act,
                                    int N, int C, int HW, int dtype, float* save_mean, float* save_invstd, double* sums, void* ws,
                                    size_t ws_bytes, long long y_image_stride, void* stream) {
    if ((phase != 1 && phase != 2) || x == nullptr || sums == nullptr || ws == nullptr || act < 0 || act > AADG_ACT_RELU6 ||
        y_image_stride < 0 || (((uintptr_t)sums) & 7u) != 0)
        return AADG_E_BADARG;
    if (phase == 2 && (y == nullptr || save_mean == nullptr || save_invstd == nullptr)) return AADG_E_BADARG;
    if (C <= 0 || ws_bytes < aadg_bn_workspace_bytes(C)) return AADG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        return bn_forward<float>((const float*)x, (const float*)residual, (float*)y, (uint8_t*)act_mask, weight, bias, running_mean,
                                 running_var, momentum, eps, act, 1, N, C, HW, save_mean, save_invstd, (float*)ws, y_image_stride, st,
                                 phase, sums);
    if (dtype == 1)
        return bn_forward<__hip_bfloat16>((const __hip_bfloat16*)x, (const __hip_bfloat16*)residual, (__hip_bfloat16*)y,
                                          (uint8_t*)act_mask, weight, bias, running_mean, running_var, momentum, eps, act, 1, N, C, HW,
                                          save_mean, save_invstd, (float*)ws, y_image_stride, st, phase, sums);
    return AADG_E_BADARG;
}

/* aadg_bn_sync_forward(phase 2) of float32 tensors whose `residual` is the INPUT of another BatchNorm without activation (a bottleneck's
 * projection shortcut): res_scale / res_shift [C] (aadg_bn_finalize_f32 of that BatchNorm) are applied while the residual is read -- the
 * shortcut's normalised tensor is never written.  (ABI 10) */
extern "C" int aadg_bn_forward_res_affine_f32(const float* x, const float* residual, const float* res_scale, const float* res_shift, float* y,
                                              void* act_mask, const float* weight, const float* bias, float* running_mean, float* running_var,
                                              float momentum, float eps, int act, int N, int C, int HW, float* save_mean, float* save_invstd,
                                              double* sums, void* ws, size_t ws_bytes, void* stream) {
    if (x == nullptr || residual == nullptr || res_scale == nullptr || res_shift == nullptr || y == nullptr || sums == nullptr ||
        ws == nullptr || save_mean == nullptr || save_invstd == nullptr || act < 0 || act > AADG_ACT_RELU6 || (((uintptr_t)sums) & 7u) != 0)
        return AADG_E_BADARG;
    if (C <= 0 || ws_bytes < aadg_bn_workspace_bytes(C)) return AADG_E_WORKSPACE;
    return bn_forward<float>(x, residual, y, (uint8_t*)act_mask, weight, bias, running_mean, running_var, momentum, eps, act, 1, N, C, HW,
                             save_mean, save_invstd, (float*)ws, 0, (hipStream_t)stream, 2, sums, res_scale, res_shift);
}

extern "C" int aadg_bn_sync_backward(int phase, const void* x, const void* y, const void* act_mask, const void* dy,
                                     const void* const* dy_extra, int n_extra, const float* dy_plane_const, const float* weight,
                                     const float* bias, const float* save_mean, const float* save_invstd, int act, void* dx, void* dres,
                                     float* dweight, float* dbias, int N, int C, int HW, int dtype, double* sums, const double* count,
                                     void* ws, size_t ws_bytes, long long dy_image_stride, void* stream) {
    if ((phase != 1 && phase != 2) || x == nullptr || dy == nullptr || save_mean == nullptr || save_invstd == nullptr ||
        ws == nullptr || sums == nullptr || act < 0 || act > AADG_ACT_RELU6 || dy_image_stride < 0 || (((uintptr_t)sums) & 7u) != 0)
        return AADG_E_BADARG;
    if (phase == 2 && (dx == nullptr || count == nullptr)) return AADG_E_BADARG;
    if (dres != nullptr && y == nullptr && act_mask == nullptr) return AADG_E_BADARG;
    if (n_extra < 0 || n_extra > BN_MAX_EXTRA || (n_extra > 0 && (dy_extra == nullptr || dres == nullptr))) return AADG_E_BADARG;
    if (dy_plane_const != nullptr && dres == nullptr) return AADG_E_BADARG;
    if (C <= 0 || ws_bytes < aadg_bn_workspace_bytes(C)) return AADG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        return bn_backward<float>((const float*)x, (const float*)y, (const uint8_t*)act_mask, (const float*)dy, dy_extra, n_extra,
                                  dy_plane_const, weight, bias, save_mean, save_invstd, act, (float*)dx, (float*)dres, dweight, dbias, N, C, HW,
                                  (float*)ws, dy_image_stride, st, phase, sums, count);
    if (dtype == 1)
        return bn_backward<__hip_bfloat16>((const __hip_bfloat16*)x, (const __hip_bfloat16*)y, (const uint8_t*)act_mask,
                                           (const __hip_bfloat16*)dy, dy_extra, n_extra, dy_plane_const, weight, bias, save_mean,
                                           save_invstd, act, (__hip_bfloat16*)dx, (__hip_bfloat16*)dres, dweight, dbias, N, C, HW,
                                           (float*)ws, dy_image_stride, st, phase, sums, count);
    return AADG_E_BADARG;
}

// ---- statistics only (ABI 10): the BatchNorm whose normalise + ReLU pass the CONSUMING convolution applies on load ------------
namespace {
__global__ __launch_bounds__(256) void k_bn_finalize(const double* __restrict__ sums, int C, const float* __restrict__ weight,
                                                     const float* __restrict__ bias, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var, float momentum, float eps, float* __restrict__ save_mean,
                                                     float* __restrict__ save_invstd, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    // (k_bn_apply's finalisation, expression by expression: the backward re-derives the ReLU mask from these coefficients)
    const double count = sums[2 * (size_t)C];
    const double m = sums[2 * c] / count;
    double var = sums[2 * c + 1] / count - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    float sc, sh;
    bn_scale_shift_of(weight, bias, (float)m, is, c, &sc, &sh);
    scale[c] = sc; shift[c] = sh;
    save_mean[c] = (float)m;
    save_invstd[c] = is;
    if (running_mean != nullptr) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}
}  // namespace

/* Training-mode BatchNorm WITHOUT its elementwise pass: from the float64 totals of x (`sums` [2C + 1]: sum, sum of squares per channel,
 * element count -- aadg_conv3x3_nchw_f32x3_stats / aadg_conv1x1_nchw_f32x3_stats / aadg_bn_sync_forward(phase 1)) the saved mean /
 * invstd, the running statistics' update and scale / shift [C] of y = x * scale + shift -- the coefficients the consuming convolution
 * applies on load (aadg_conv1x1_nchw_f32x3_pre).  The backward is aadg_bn_backward with act = ReLU and no stored output (the mask is
 * re-derived from x). */
extern "C" int aadg_bn_finalize_f32(const double* sums, const float* weight, const float* bias, float* running_mean, float* running_var,
                                    float momentum, float eps, int C, float* save_mean, float* save_invstd, float* scale, float* shift,
                                    void* stream) {
    if (sums == nullptr || save_mean == nullptr || save_invstd == nullptr || scale == nullptr || shift == nullptr || C <= 0 ||
        (running_mean == nullptr) != (running_var == nullptr))
        return AADG_E_BADARG;
    hipLaunchKernelGGL(k_bn_finalize, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, C, weight, bias, running_mean, running_var,
                       momentum, eps, save_mean, save_invstd, scale, shift);
    AADG_LAUNCH_CHECK();
    return 0;
}

extern "C" int aadg_bn_relu_maxpool_supported(int H, int W, int dtype) {
    const int Ho = (H - 1) / 2 + 1, Wo = W / 2;
    return (dtype == 0 || dtype == 1) && H >= 2 && W >= 8 && (W % 8) == 0 && ((long long)Ho * Wo) % 1024 == 0 ? 1 : 0;
}

/* y [N, C, Ho, Wo] = max_pool2d(relu(batch_norm(x [N, C, H, W], training)), 3, 2, 1) and the pooling index (one byte per output,
 * as aadg_maxpool3x3s2_forward); save_mean / save_invstd / running statistics as aadg_bn_forward(training = 1).  Backward:
 * aadg_maxpool3x3s2_backward(index, dy) followed by aadg_bn_backward(x, ..., act = AADG_ACT_RELU). */
extern "C" int aadg_bn_relu_maxpool_forward(const void* x, void* y, void* index, const float* weight, const float* bias, float* running_mean,
                                            float* running_var, float momentum, float eps, int N, int C, int H, int W, int dtype,
                                            float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, void* stream) {
    if (x == nullptr || y == nullptr || index == nullptr || save_mean == nullptr || save_invstd == nullptr || ws == nullptr) return AADG_E_BADARG;
    if ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0 || (((uintptr_t)index) & 3u) != 0) return AADG_E_BADARG;
    if (N <= 0 || C <= 0 || ws_bytes < aadg_bn_workspace_bytes(C)) return AADG_E_WORKSPACE;
    if (!aadg_bn_relu_maxpool_supported(H, W, dtype)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)ws;
    const BnWs L = bn_ws(C);
    const int Ho = (H - 1) / 2 + 1, Wo = W / 2, HW = H * W;
    const long long quads = (long long)N * C * Ho * (Wo / 4);
    if (quads / 256 > 0x7FFFFFFFLL) return AADG_E_UNSUPPORTED;
    Shape s;
    if (dtype == 0) {
        if (!make_shape<float>(N, C, HW, x, nullptr, nullptr, nullptr, &s) || s.vec == 1) return AADG_E_UNSUPPORTED;
        hipLaunchKernelGGL((k_bn_reduce_fwd<float, 4>), BN_RED_GRID(s.split, C), dim3(s.threads), 0, st, (const float*)x, C, s.len, s.pc.per_strip,
                           s.pc.plen, s.pc.total, wsf + L.partial);
    } else {
        if (!make_shape<__hip_bfloat16>(N, C, HW, x, nullptr, nullptr, nullptr, &s) || s.vec == 1) return AADG_E_UNSUPPORTED;
        hipLaunchKernelGGL((k_bn_reduce_fwd<__hip_bfloat16, 8>), BN_RED_GRID(s.split, C), dim3(s.threads), 0, st, (const __hip_bfloat16*)x, C, s.len,
                           s.pc.per_strip, s.pc.plen, s.pc.total, wsf + L.partial);
    }
    AADG_LAUNCH_CHECK();
    const BnFin fin = {wsf + L.partial, s.split, (double)N * (double)HW, weight, bias, running_mean, running_var, momentum, eps, save_mean,
                       save_invstd};
    const dim3 grid((unsigned)(quads / 256));
    if (dtype == 0)
        hipLaunchKernelGGL(k_bn_relu_maxpool<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, (uint8_t*)index, fin, C, H, W, Ho, Wo);
    else
        hipLaunchKernelGGL(k_bn_relu_maxpool<__hip_bfloat16>, grid, dim3(256), 0, st, (const __hip_bfloat16*)x, (__hip_bfloat16*)y,
                           (uint8_t*)index, fin, C, H, W, Ho, Wo);
    AADG_LAUNCH_CHECK();
    return 0;
}

/* Backward of aadg_bn_relu_maxpool_forward (bfloat16 only): dx [N, C, H, W], dweight, dbias from x, the pooling index and the
 * pooled gradient dy [N, C, Ho, Wo] -- the gradient of the normalised map is rebuilt on the fly in both passes, never stored. */
namespace {
template <typename T>
int bn_pool_backward(const T* px, const uint8_t* index, const T* pg, const float* weight, const float* bias, const float* save_mean,
                     const float* save_invstd, T* dx, float* dweight, float* dbias, int N, int C, int H, int W, float* wsf, hipStream_t st) {
    const BnWs L = bn_ws(C);
    const int Ho = (H - 1) / 2 + 1, Wo = W / 2, HW = H * W;
    Shape s;
    if (!make_shape<T>(N, C, HW, px, dx, nullptr, nullptr, &s) || s.vec == 1) return AADG_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_bn_pool_reduce_bwd<T>, BN_RED_GRID(s.split, C), dim3(s.threads), 0, st, px, index, pg, save_mean, save_invstd,
                       weight, bias, C, H, W, Ho, Wo, s.len, s.pc.per_strip, s.pc.plen, s.pc.total, wsf + L.partial);
    AADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bn_pool_dx<T>, dim3(N * C, s.pc.per_strip), dim3(s.threads), 0, st, px, index, pg, dx,
                       (const float*)(wsf + L.partial), s.split, (double)N * (double)HW, weight, bias, save_mean, save_invstd, dweight,
                       dbias, C, H, W, Ho, Wo, s.len, s.pc.plen, bn_stream((size_t)N * C * H * W * sizeof(T)) ? 1 : 0);
    AADG_LAUNCH_CHECK();
    return 0;
}
}  // namespace

/* dtype 1 (bfloat16) and, since ABI 10, 0 (float32: W a multiple of 4, dy 8-byte aligned, index 2-byte aligned) */
extern "C" int aadg_bn_relu_maxpool_backward(const void* x, const void* index, const void* dy, const float* weight, const float* bias,
                                             const float* save_mean, const float* save_invstd, void* dx, float* dweight, float* dbias,
                                             int N, int C, int H, int W, int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (x == nullptr || index == nullptr || dy == nullptr || save_mean == nullptr || save_invstd == nullptr || dx == nullptr || ws == nullptr)
        return AADG_E_BADARG;
    if ((((uintptr_t)x | (uintptr_t)dx) & 15u) != 0 || (((uintptr_t)index) & 3u) != 0 || (((uintptr_t)dy) & 7u) != 0) return AADG_E_BADARG;
    if (N <= 0 || C <= 0 || ws_bytes < aadg_bn_workspace_bytes(C)) return AADG_E_WORKSPACE;
    if ((dtype != 0 && dtype != 1) || !aadg_bn_relu_maxpool_supported(H, W, dtype)) return AADG_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        if ((W % 4) != 0 || ((W / 2) % 2) != 0) return AADG_E_UNSUPPORTED;        // 4-column vectors, 2-byte / 8-byte aligned output pairs
        return bn_pool_backward<float>((const float*)x, (const uint8_t*)index, (const float*)dy, weight, bias, save_mean, save_invstd, (float*)dx,
                                       dweight, dbias, N, C, H, W, (float*)ws, st);
    }
    return bn_pool_backward<__hip_bfloat16>((const __hip_bfloat16*)x, (const uint8_t*)index, (const __hip_bfloat16*)dy, weight, bias, save_mean,
                                            save_invstd, (__hip_bfloat16*)dx, dweight, dbias, N, C, H, W, (float*)ws, st);
}

// Depthwise 3x3 convolution (stride 1, zero padding 1) + bias + exact GELU on TOKEN-layout activations [B, H, W, C] -- the middle
// of SegFormer's Mix-FFN (reference: models/mmseg/models/backbones/mix_transformer.py:19-46 `fc1 -> DWConv -> GELU -> fc2`, DWConv
// :149-159 reshapes the tokens to NCHW for nn.Conv2d(dim, dim, 3, 1, 1, groups=dim) and back).
//
// Why a kernel of its own: between the two Linear layers the hidden tensor ([B, N, 4C]: 400 MB in bfloat16 at stage 1 for 48 images)
// was touched by five passes -- a transposing copy to NCHW, the NCHW depthwise kernel, the bias add, GELU on the transposed view, and
// the copy that makes fc2's input contiguous again (and as many in the backward) -- 23 ms of strided ATen copies per step.  In the
// tokens' own layout the channel is the fastest dimension, so a lane owns 8 consecutive channels of one image column (16-byte
// accesses, coalesced across the lanes of a wave) and walks down the rows with a three-row register window; the left / right
// neighbour columns are the adjacent work items' pixels (cache hits).  One read and one write of the tensor per direction:
//   forward   out = GELU(dw(h) + bias)
//   backward  g = dout * GELU'(dw(h) + bias)        (z is recomputed, not stored)
//             dh = dw^T(g)                          (the same walk over g with mirrored taps)
//             dw[tap][c] = sum g * h(tap), db[c] = sum g        (float32 atomics into [9][C] / [C]: the order in which the
//             workgroups' partial sums arrive is not fixed, so dw / db are NOT bit-reproducible from run to run -- last-bit
//             differences, unlike the fixed-order reductions of csrc/layernorm.hip)
// HBM-bound by design, issue-heavy in practice (72 multiply-adds + 8 erf per 16 bytes).  C % 8 == 0.
#include "common.h"

namespace {

constexpr int DN_THREADS = 256;
constexpr int DN_ROWS = 16;          // rows per work item of the forward / input-gradient walks

struct Px8 { float v[8]; };

template <typename T> struct IO8;
template <> struct IO8<float> {
    static __device__ __forceinline__ Px8 load(const float* p) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        return Px8{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    }
    static __device__ __forceinline__ void store(float* p, const Px8& o) {
        *reinterpret_cast<float4*>(p) = make_float4(o.v[0], o.v[1], o.v[2], o.v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(o.v[4], o.v[5], o.v[6], o.v[7]);
    }
};
template <> struct IO8<uint16_t> {            // bfloat16 bits
    static __device__ __forceinline__ Px8 load(const uint16_t* p) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
        Px8 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o.v[2 * i] = __uint_as_float(w[i] << 16); o.v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u); }
        return o;
    }
    static __device__ __forceinline__ void store(uint16_t* p, const Px8& o) {
        *reinterpret_cast<uint4*>(p) = make_uint4(aadg_f2bf_pk(o.v[0], o.v[1]), aadg_f2bf_pk(o.v[2], o.v[3]), aadg_f2bf_pk(o.v[4], o.v[5]),
                                                  aadg_f2bf_pk(o.v[6], o.v[7]));
    }
};

__device__ __forceinline__ Px8 zero8() { return Px8{{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}}; }

// torch.nn.GELU() (approximate = 'none'): 0.5 z (1 + erf(z / sqrt 2)) and its derivative
__device__ __forceinline__ float gelu(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float z) {
    return 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
}

struct Row3 { Px8 l, m, r; };

template <typename T>
__device__ __forceinline__ Row3 load_row3(const T* __restrict__ img, int y, int x, int H, int W, int C, int c0) {
    Row3 o;
    o.l = o.m = o.r = zero8();
    if (y >= 0 && y < H) {                            // (uniform for the lanes of a wave that share the strip; rows outside = padding)
        const T* p = img + ((size_t)y * W + x) * C + c0;
        // neighbour columns at clamped addresses, zeroed afterwards at the image border: no per-lane conditional loads
        const Px8 l = IO8<T>::load(p - (x > 0 ? C : 0)), r = IO8<T>::load(p + (x < W - 1 ? C : 0));
        o.m = IO8<T>::load(p);
        if (x > 0) o.l = l;
        if (x < W - 1) o.r = r;
    }
    return o;
}

__device__ __forceinline__ void decode_item(long long item, int C8, int W, int NS, int& cg, int& x, int& strip, int& b) {
    cg = (int)(item % C8); item /= C8;
    x = (int)(item % W); item /= W;
    strip = (int)(item % NS);
    b = (int)(item / NS);
}

// MODE 0: out = GELU(conv + bias); MODE 1: out = dout * GELU'(conv + bias); MODE 2: out = conv with mirrored taps (no bias): the
// input gradient.  w9: [9][C] float32 (tap-major copy of the [C,1,3,3] weight).
template <typename T, int MODE>
__global__ __launch_bounds__(DN_THREADS) void k_dwn_walk(const T* __restrict__ in, const float* __restrict__ w9, const float* __restrict__ bias,
                                                         const T* __restrict__ dout, T* __restrict__ out, int B, int H, int W, int C,
                                                         long long n_items) {
    const int C8 = C >> 3, NS = (H + DN_ROWS - 1) / DN_ROWS;
    const long long item = (long long)blockIdx.x * DN_THREADS + threadIdx.x;
    if (item >= n_items) return;
    int cg, x, strip, b;
    decode_item(item, C8, W, NS, cg, x, strip, b);
    const int c0 = cg * 8;
    float wt[9][8], bs[8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const Px8 q = IO8<float>::load(w9 + (size_t)(MODE == 2 ? 8 - t : t) * C + c0);
#pragma unroll
        for (int i = 0; i < 8; ++i) wt[t][i] = q.v[i];
    }
    {
        const Px8 q = MODE == 2 ? zero8() : IO8<float>::load(bias + c0);
#pragma unroll
        for (int i = 0; i < 8; ++i) bs[i] = q.v[i];
    }
    const T* img = in + (size_t)b * H * W * C;
    const int y0 = strip * DN_ROWS, y1 = min(y0 + DN_ROWS, H);
    Row3 p = load_row3(img, y0 - 1, x, H, W, C, c0), c = load_row3(img, y0, x, H, W, C, c0);
    for (int y = y0; y < y1; ++y) {
        const Row3 n = load_row3(img, y + 1, x, H, W, C, c0);
        Px8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float z = bs[i];
            z = fmaf(wt[0][i], p.l.v[i], z); z = fmaf(wt[1][i], p.m.v[i], z); z = fmaf(wt[2][i], p.r.v[i], z);
            z = fmaf(wt[3][i], c.l.v[i], z); z = fmaf(wt[4][i], c.m.v[i], z); z = fmaf(wt[5][i], c.r.v[i], z);
            z = fmaf(wt[6][i], n.l.v[i], z); z = fmaf(wt[7][i], n.m.v[i], z); z = fmaf(wt[8][i], n.r.v[i], z);
            o.v[i] = z;
        }
        const size_t off = (((size_t)b * H + y) * W + x) * C + c0;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] = gelu(o.v[i]);
        } else if (MODE == 1) {
            const Px8 d = IO8<T>::load(dout + off);
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] = d.v[i] * gelu_grad(o.v[i]);
        }
        IO8<T>::store(out + off, o);
        p = c; c = n;
    }
}

// weight / bias gradient: dw9[tap][c] += sum g(y, x) * h(y + dy, x + dx), db[c] += sum g.  A workgroup = 8 channel groups (64
// channels: one 128-byte line per pixel) x 32 image columns x a strip of DW_ROWS rows; the 32 lanes that share a channel group fold
// their 80 partial sums by wave shuffles, the four waves through LDS, and the workgroup issues 80 x 8 float32 atomics.  (One work item
// per image column with per-thread atomics -- 15 M same-address atomics per call -- took 0.5-2.8 ms per call: 18 ms per SegFormer step.)
constexpr int DW_ROWS = 32;
template <typename T>
__global__ __launch_bounds__(DN_THREADS) void k_dwn_wgrad(const T* __restrict__ h, const T* __restrict__ g, float* __restrict__ dw9,
                                                          float* __restrict__ db, int B, int H, int W, int C) {
    const int C8 = C >> 3, NS = (H + DW_ROWS - 1) / DW_ROWS, XT = (W + 31) / 32;
    const int cgl = threadIdx.x & 7, xl = threadIdx.x >> 3;
    const int cg = blockIdx.x * 8 + cgl;
    int t = blockIdx.y;
    const int xt = t % XT; t /= XT;
    const int strip = t % NS, b = t / NS;
    const int x = xt * 32 + xl;
    const bool live = cg < C8 && x < W;
    const int c0 = min(cg, C8 - 1) * 8, xc = min(x, W - 1);
    float acc[9][8], ab[8];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[k][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ab[i] = 0.f;
    const T* img = h + (size_t)b * H * W * C;
    const int y0 = strip * DW_ROWS, y1 = min(y0 + DW_ROWS, H);
    Row3 p = load_row3(img, y0 - 1, xc, H, W, C, c0), c = load_row3(img, y0, xc, H, W, C, c0);
    for (int y = y0; y < y1; ++y) {
        const Row3 n = load_row3(img, y + 1, xc, H, W, C, c0);
        Px8 gv = IO8<T>::load(g + (((size_t)b * H + y) * W + xc) * C + c0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float q = live ? gv.v[i] : 0.f;
            acc[0][i] = fmaf(q, p.l.v[i], acc[0][i]); acc[1][i] = fmaf(q, p.m.v[i], acc[1][i]); acc[2][i] = fmaf(q, p.r.v[i], acc[2][i]);
            acc[3][i] = fmaf(q, c.l.v[i], acc[3][i]); acc[4][i] = fmaf(q, c.m.v[i], acc[4][i]); acc[5][i] = fmaf(q, c.r.v[i], acc[5][i]);
            acc[6][i] = fmaf(q, n.l.v[i], acc[6][i]); acc[7][i] = fmaf(q, n.m.v[i], acc[7][i]); acc[8][i] = fmaf(q, n.r.v[i], acc[8][i]);
            ab[i] += q;
        }
        p = c; c = n;
    }
    // lanes l, l + 8, ..., l + 56 of a wave share the channel group
    __shared__ float red[DN_THREADS / 64][80][8 + 1];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = k < 9 ? acc[k < 9 ? k : 0][i] : ab[i];
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            if (lane < 8) red[wv][k * 8 + i][lane] = v;
        }
    __syncthreads();
    for (int j = threadIdx.x; j < 80 * 8; j += DN_THREADS) {
        const int q = j >> 3, l = j & 7;                  // q = tap * 8 + channel-in-group (taps 0..8, 9 = bias), l = channel group
        const int cgo = blockIdx.x * 8 + l;
        if (cgo >= C8) continue;
        const float v = (red[0][q][l] + red[1][q][l]) + (red[2][q][l] + red[3][q][l]);
        const int k = q >> 3, i = q & 7;
        if (k < 9) atomicAdd(&dw9[(size_t)k * C + cgo * 8 + i], v);
        else atomicAdd(&db[cgo * 8 + i], v);
    }
}

template <typename T, int MODE>
int launch_walk(const void* in, const float* w9, const float* bias, const void* dout, void* out, int B, int H, int W, int C, hipStream_t st) {
    const long long n = (long long)B * ((H + DN_ROWS - 1) / DN_ROWS) * W * (C >> 3);
    hipLaunchKernelGGL((k_dwn_walk<T, MODE>), dim3((unsigned)((n + DN_THREADS - 1) / DN_THREADS)), dim3(DN_THREADS), 0, st,
                       reinterpret_cast<const T*>(in), w9, bias, reinterpret_cast<const T*>(dout), reinterpret_cast<T*>(out), B, H, W, C, n);
    AADG_LAUNCH_CHECK();
    return 0;
}

bool dwn_ok(int B, int H, int W, int C, int dtype) {
    return B > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && (dtype == 0 || dtype == 1) &&
           (long long)B * H * W * (C >> 3) < ((long long)1 << 31) * DN_THREADS &&
           (long long)((W + 31) / 32) * ((H + 31) / 32) * B <= 65535;            // grid.y of the weight-gradient kernel
}

}  // namespace

extern "C" int aadg_dwconv3x3_gelu_nhwc_supported(int B, int H, int W, int C, int dtype) { return dwn_ok(B, H, W, C, dtype) ? 1 : 0; }

// out[B,H,W,C] = GELU(depthwise3x3(h) + bias); w9 = the [C,1,3,3] weight as [9][C] float32
extern "C" int aadg_dwconv3x3_gelu_nhwc_forward(const void* h, const float* w9, const float* bias, void* out, int B, int H, int W, int C,
                                                int dtype, void* stream) {
    if (!h || !w9 || !bias || !out) return AADG_E_BADARG;
    if (!dwn_ok(B, H, W, C, dtype)) return AADG_E_UNSUPPORTED;
    if (((uintptr_t)h | (uintptr_t)out | (uintptr_t)w9 | (uintptr_t)bias) & 15) return AADG_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dtype == 0 ? launch_walk<float, 0>(h, w9, bias, nullptr, out, B, H, W, C, st)
                      : launch_walk<uint16_t, 0>(h, w9, bias, nullptr, out, B, H, W, C, st);
}

// g (scratch, same shape / dtype as h) = dout * GELU'(z); dh = depthwise3x3^T(g); dw9 [9][C] and db [C] float32 (overwritten)
extern "C" int aadg_dwconv3x3_gelu_nhwc_backward(const void* h, const float* w9, const float* bias, const void* dout, void* g, void* dh,
                                                 float* dw9, float* db, int B, int H, int W, int C, int dtype, void* stream) {
    if (!h || !w9 || !bias || !dout || !g || !dh || !dw9 || !db) return AADG_E_BADARG;
    if (!dwn_ok(B, H, W, C, dtype)) return AADG_E_UNSUPPORTED;
    if (((uintptr_t)h | (uintptr_t)dout | (uintptr_t)g | (uintptr_t)dh | (uintptr_t)w9 | (uintptr_t)bias) & 15) return AADG_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = dtype == 0 ? launch_walk<float, 1>(h, w9, bias, dout, g, B, H, W, C, st) : launch_walk<uint16_t, 1>(h, w9, bias, dout, g, B, H, W, C, st);
    if (rc) return rc;
    rc = dtype == 0 ? launch_walk<float, 2>(g, w9, bias, nullptr, dh, B, H, W, C, st) : launch_walk<uint16_t, 2>(g, w9, bias, nullptr, dh, B, H, W, C, st);
    if (rc) return rc;
    AADG_HIP_TRY(hipMemsetAsync(dw9, 0, (size_t)9 * C * sizeof(float), st));
    AADG_HIP_TRY(hipMemsetAsync(db, 0, (size_t)C * sizeof(float), st));
    const dim3 grid(((C >> 3) + 7) / 8, ((W + 31) / 32) * ((H + DW_ROWS - 1) / DW_ROWS) * B);
    if (dtype == 0)
        hipLaunchKernelGGL(k_dwn_wgrad<float>, grid, dim3(DN_THREADS), 0, st, reinterpret_cast<const float*>(h), reinterpret_cast<const float*>(g), dw9, db, B, H, W, C);
    else
        hipLaunchKernelGGL(k_dwn_wgrad<uint16_t>, grid, dim3(DN_THREADS), 0, st, reinterpret_cast<const uint16_t*>(h), reinterpret_cast<const uint16_t*>(g), dw9, db, B, H, W, C);
    AADG_LAUNCH_CHECK();
    return 0;
}

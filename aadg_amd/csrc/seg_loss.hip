// Per-policy BCE + samplewise Dice (+ the BCE gradient) in ONE pass over logits/labels
// (a21, a22; 8f item 2).
//
// Replaces, per inner iteration of the reference (search_dg.py:140-142,164-165):
//   seg_soft = sigmoid(seg_output)                                   1 read + 1 write of [N,K,H,W]
//   [BCELoss(seg_soft[j::M], mask_gt[j::M]) for j in range(M)]       2 reads (strided gathers)
//   M x { F1(stack([1-p, p]), gt.long())[1] per class }              2*M*K full-tensor passes (redundant: the
//                                                                    same value is recomputed for every j)
// with one streaming read of logits + labels (float4 per lane) and an optional write of d(loss)/d(logits),
// where loss = mean_j BCE_j, so that backward needs no further pass.
//
// Determinism: block partials go to the workspace and are combined in a fixed order by a second tiny
// kernel (no float atomics).
#include "common.h"

namespace {

constexpr int SL_THREADS = 256;
constexpr int SL_CHUNK = SL_THREADS * 4 * 8;  // elements per block: 8 float4 per thread

struct Partial {
    double bce;
    int tp, fp, fn, pad;
};

__device__ __forceinline__ void elem(float z, float y, float gscale, float& bce, int& tp, int& fp, int& fn, float* g) {
    // torch.sigmoid then nn.BCELoss on the ROUNDED probability (so fp32 saturation behaves as in the reference:
    // p == 1.0f -> log(1-p) clamps at -100).  Hardware exp/log/rcp (v_exp_f32, v_log_f32, v_rcp_f32; ~1 ulp)
    // keep the per-policy means well inside the 1e-4 parity tolerance and make the pass bandwidth bound.
    const float p = __frcp_rn(1.0f + __expf(-z));
    float lp = __logf(p), lq = __logf(1.0f - p);         // logs clamped at -100
    lp = fmaxf(lp, -100.0f);
    lq = fmaxf(lq, -100.0f);
    bce += (y - 1.0f) * lq - y * lp;
    const int pr = p > 0.5f, gt = ((long)y) != 0;        // argmax([1-p, p]) == 1  <=>  p > 0.5
    tp += pr & gt; fp += pr & (gt ^ 1); fn += (pr ^ 1) & gt;
    if (g) {
        // BCELoss backward (p - y) / max(p (1 - p), 1e-12) x sigmoid backward p (1 - p): the quotient is 1 unless the product
        // underflows the eps (|z| > 27) -- a select instead of a quarter-rate reciprocal
        const float pq = p * (1.0f - p);
        *g = gscale * (p - y) * (pq >= 1e-12f ? 1.0f : pq * 1e12f);
    }
}

// grid (chunks, N*K)
__global__ __launch_bounds__(SL_THREADS) void k_seg_partial(const float* __restrict__ logits, const float* __restrict__ labels,
                                                            int HW, float gscale, float* __restrict__ grad,
                                                            Partial* __restrict__ part, int stream) {
    const int plane = blockIdx.y;
    const size_t base = (size_t)plane * HW;
    const int c0 = blockIdx.x * SL_CHUNK;
    const int c1 = min(c0 + SL_CHUNK, HW);
    float bce = 0.f;
    int tp = 0, fp = 0, fn = 0;
    const bool vec = ((HW & 3) == 0) && ((((uintptr_t)logits | (uintptr_t)labels | (uintptr_t)grad) & 15) == 0);
    if (vec) {
        for (int i = c0 + threadIdx.x * 4; i < c1; i += SL_THREADS * 4) {
            const float4 z = aadg_load_stream(logits + base + i);       // logits and labels are read once
            const float4 y = aadg_load_stream(labels + base + i);
            float4 g;
            elem(z.x, y.x, gscale, bce, tp, fp, fn, grad ? &g.x : nullptr);
            elem(z.y, y.y, gscale, bce, tp, fp, fn, grad ? &g.y : nullptr);
            elem(z.z, y.z, gscale, bce, tp, fp, fn, grad ? &g.z : nullptr);
            elem(z.w, y.w, gscale, bce, tp, fp, fn, grad ? &g.w : nullptr);
            if (grad) aadg_store_out(grad + base + i, g, stream != 0);    // a large gradient is read by the backward only after ~1 GB of other traffic
        }
    } else {
        for (int i = c0 + threadIdx.x; i < c1; i += SL_THREADS) {
            float g;
            elem(logits[base + i], labels[base + i], gscale, bce, tp, fp, fn, grad ? &g : nullptr);
            if (grad) grad[base + i] = g;
        }
    }
    double b = wave_sum((double)bce);
    tp = wave_sum(tp); fp = wave_sum(fp); fn = wave_sum(fn);
    __shared__ double sb[4];
    __shared__ int st[4][3];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sb[wv] = b; st[wv][0] = tp; st[wv][1] = fp; st[wv][2] = fn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        Partial o;
        o.bce = (sb[0] + sb[1]) + (sb[2] + sb[3]);
        o.tp = st[0][0] + st[1][0] + st[2][0] + st[3][0];
        o.fp = st[0][1] + st[1][1] + st[2][1] + st[3][1];
        o.fn = st[0][2] + st[1][2] + st[2][2] + st[3][2];
        o.pad = 0;
        part[(size_t)plane * gridDim.x + blockIdx.x] = o;
    }
}

// grid M + K workgroups of 256 threads: workgroup t < M -> BCE of policy t; workgroup M + k -> Dice of class k.  Every thread sums a
// fixed subset of the partial records, the workgroup combines them in a fixed order (deterministic, no atomics).  (One wave per output
// walking its records one after the other took 17 us -- 8 % of the pass it finishes.)
__global__ __launch_bounds__(256) void k_seg_final(const Partial* __restrict__ part, int N, int K, int HW, int M,
                                                  int chunks, float* __restrict__ out_bce, float* __restrict__ out_dice) {
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ double red[4];
    double acc = 0.0;
    if (t < M) {
        // policy t owns the planes of rows t, t + M, ...: rows * K * chunks records
        const int rows = (N - t + M - 1) / M, per_row = K * chunks, total = rows * per_row;
#pragma unroll 4
        for (int i = tid; i < total; i += 256) {
            const int r = t + (i / per_row) * M;
            acc += part[(size_t)r * per_row + (i % per_row)].bce;
        }
        acc = wave_sum(acc);
        if (lane == 0) red[wv] = acc;
        __syncthreads();
        if (tid == 0) out_bce[t] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / ((double)rows * K * HW));
    } else {
        // class k: samplewise F1; 32 lanes walk the chunks of one sample, a wave holds two samples at a time
        const int k = t - M, half = lane >> 5, l = lane & 31;
        for (int r0 = (wv * 2 + half); r0 < N + 8; r0 += 8) {                     // uniform trip count (shuffles below)
            const bool live = r0 < N;
            const Partial* p = part + ((size_t)min(r0, N - 1) * K + k) * chunks;
            int tp = 0, fp = 0, fn = 0;
            if (live)
                for (int c = l; c < chunks; c += 32) { tp += p[c].tp; fp += p[c].fp; fn += p[c].fn; }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { tp += __shfl_xor(tp, o, 64); fp += __shfl_xor(fp, o, 64); fn += __shfl_xor(fn, o, 64); }
            const long den = 2l * tp + fp + fn;
            if (live && l == 0) acc += den ? (2.0 * (double)tp) / (double)den : 0.0;
        }
        acc = wave_sum(acc);
        if (lane == 0) red[wv] = acc;
        __syncthreads();
        if (tid == 0) out_dice[k] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / N);
    }
}

int chunks_of(int HW) { return (HW + SL_CHUNK - 1) / SL_CHUNK; }

}  // namespace

extern "C" size_t aadg_seg_loss_workspace_bytes(int N, int K, int HW) {
    if (N <= 0 || K <= 0 || HW <= 0) return 0;
    return aadg_align_up((size_t)N * K * chunks_of(HW) * sizeof(Partial), 256);
}

extern "C" int aadg_seg_bce_dice_scaled_f32(const float* logits, const float* labels, int N, int K, int HW, int M, float grad_scale,
                                            float* out_bce, float* out_dice, float* grad_logits, void* ws, size_t ws_bytes,
                                            void* stream) {
    if (!logits || !labels || !out_bce || !out_dice || !ws) return AADG_E_BADARG;
    if (N <= 0 || K <= 0 || HW <= 0 || M <= 0 || M > N || N % M) return AADG_E_BADARG;
    if (ws_bytes < aadg_seg_loss_workspace_bytes(N, K, HW)) return AADG_E_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int chunks = chunks_of(HW);
    Partial* part = reinterpret_cast<Partial*>(ws);
    // d (grad_scale * mean_j(BCE_j)) / dz: every element of policy j weighs grad_scale / (M * (N/M) * K * HW)
    const float gscale = (float)((double)grad_scale / ((double)N * K * HW));
    hipLaunchKernelGGL(k_seg_partial, dim3(chunks, N * K), dim3(SL_THREADS), 0, st, logits, labels, HW, gscale,
                       grad_logits, part, (size_t)N * K * HW * sizeof(float) > ((size_t)128 << 20) ? 1 : 0);
    AADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_seg_final, dim3(M + K), dim3(256), 0, st, part, N, K, HW, M, chunks, out_bce, out_dice);
    AADG_LAUNCH_CHECK();
    return 0;
}

extern "C" int aadg_seg_bce_dice_f32(const float* logits, const float* labels, int N, int K, int HW, int M,
                                     float* out_bce, float* out_dice, float* grad_logits, void* ws, size_t ws_bytes,
                                     void* stream) {
    return aadg_seg_bce_dice_scaled_f32(logits, labels, N, K, HW, M, 1.0f, out_bce, out_dice, grad_logits, ws, ws_bytes, stream);
}

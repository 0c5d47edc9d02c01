// Embedding prologue of the Sinkhorn reward (SURVEY (f)1): the EMA branch of the domain discriminator applied to the
// pooled encoder features under no_grad (models/discriminator.py:48-51, search_dg.py:133-135):
//
//     fe[n]  = LeakyReLU_slope(W1 x[n] + b1)      [E]      (mom_dis)
//     out[n] = W2 fe[n] + b2                      [D]      (mom_fc, logged as the discriminator loss)
//
// One workgroup per row: x[n] staged in LDS, one wavefront per output (lanes stride over the C_enc inputs: coalesced
// weight-row reads, wave-shuffle reduction).  Replaces two GEMM launches, two bias adds and the activation.
#include "common.h"

namespace {

constexpr int EMB_MAX_C = 4096, EMB_MAX_E = 256;

__global__ __launch_bounds__(256) void k_embed(const float* __restrict__ x, int ldx, const float* __restrict__ W1,
                                               const float* __restrict__ b1, const float* __restrict__ W2,
                                               const float* __restrict__ b2, int C, int E, int D, float slope,
                                               float* __restrict__ fe, float* __restrict__ out) {
    __shared__ float xs[EMB_MAX_C];
    __shared__ float fs[EMB_MAX_E];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int c = tid; c < C; c += 256) xs[c] = x[(size_t)n * ldx + c];
    __syncthreads();
    for (int e = wv; e < E; e += 4) {
        const float* w = W1 + (size_t)e * C;
        float s = 0.0f;
#pragma unroll 4
        for (int c = lane; c < C; c += 64) s = fmaf(w[c], xs[c], s);
        s = wave_sum(s);
        if (lane == 0) {
            s += b1[e];
            s = s > 0.0f ? s : s * slope;
            fs[e] = s;
            fe[(size_t)n * E + e] = s;
        }
    }
    if (out == nullptr) return;
    __syncthreads();
    for (int d = wv; d < D; d += 4) {
        float s = 0.0f;
        for (int e = lane; e < E; e += 64) s = fmaf(W2[(size_t)d * E + e], fs[e], s);
        s = wave_sum(s);
        if (lane == 0) out[(size_t)n * D + d] = s + b2[d];
    }
}

}  // namespace

extern "C" int aadg_embed_prologue_f32(const float* x, int ldx, int N, int C, const float* W1, const float* b1, int E,
                                       const float* W2, const float* b2, int D, float slope, float* fe, float* out,
                                       void* stream) {
    if (x == nullptr || W1 == nullptr || b1 == nullptr || fe == nullptr || N <= 0 || ldx < C) return AADG_E_BADARG;
    if (out != nullptr && (W2 == nullptr || b2 == nullptr || D <= 0)) return AADG_E_BADARG;
    if (C <= 0 || C > EMB_MAX_C || E <= 0 || E > EMB_MAX_E) return AADG_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_embed, dim3(N), dim3(256), 0, (hipStream_t)stream, x, ldx, W1, b1, W2, b2, C, E, D, slope, fe, out);
    AADG_LAUNCH_CHECK();
    return 0;
}

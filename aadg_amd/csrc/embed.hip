// Embedding prologue of the Sinkhorn reward (SURVEY (f)1): the EMA branch of the domain discriminator applied to the
// pooled encoder features under no_grad (models/discriminator.py:48-51, search_dg.py:133-135):
//
//     fe[n]  = LeakyReLU_slope(W1 x[n] + b1)      [E]      (mom_dis)
//     out[n] = W2 fe[n] + b2                      [D]      (mom_fc, logged as the discriminator loss)
//
// Replaces two GEMM launches, two bias adds and the activation; aadg_embed_prologue_norm_f32 also hands the Euclidean norm of
// every fe row to aadg_sinkhorn_rewards_norm_f32 (the cosine cost's denominators).  What SURVEY (f)1 sketched beyond that is
// deliberately NOT fused: the global average pool is shared with the ASPP image-pool branch (one reduction of the encoder output
// feeds both; a second one here would re-read 600 MB), and the rows stay un-normalised because geomloss derives its epsilon
// schedule from the diameter of the RAW point clouds -- unit rows would change the schedule, i.e. the reward.
#include "common.h"

namespace {

constexpr int EMB_MAX_C = 4096, EMB_MAX_E = 256;
constexpr int EMB_ROWS = 2;        // rows per workgroup: every weight row is read once per EMB_ROWS rows, 72 workgroups for N = 144
constexpr int EMB_OUTS = 4;        // outputs a wave accumulates at a time (independent weight-row loads in flight)

// One workgroup per EMB_ROWS rows: the rows staged in LDS, a wave owns the outputs e = 4 * (wave + 4 k) .. + 3 and strides its
// lanes over the C_enc inputs (coalesced weight-row reads, one shuffle reduction per (output, row)).  The workgroup holds the
// complete fe rows, so the logits and the Euclidean norm of each row (consumed by the Sinkhorn kernel's cosine cost) are
// finished here too.
__global__ __launch_bounds__(256) void k_embed(const float* __restrict__ x, int ldx, int N, const float* __restrict__ W1,
                                               const float* __restrict__ b1, const float* __restrict__ W2,
                                               const float* __restrict__ b2, int C, int E, int D, float slope,
                                               float* __restrict__ fe, float* __restrict__ out, float* __restrict__ nrm) {
    __shared__ float xs[EMB_ROWS][EMB_MAX_C];
    __shared__ float fs[EMB_ROWS][EMB_MAX_E];
    const int n0 = blockIdx.x * EMB_ROWS, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int rows = min(EMB_ROWS, N - n0);
#pragma unroll
    for (int r = 0; r < EMB_ROWS; ++r)
        for (int c = tid; c < C; c += 256) xs[r][c] = r < rows ? x[(size_t)(n0 + r) * ldx + c] : 0.0f;
    __syncthreads();
    for (int e0 = EMB_OUTS * wv; e0 < E; e0 += EMB_OUTS * 4) {
        float acc[EMB_OUTS][EMB_ROWS];
#pragma unroll
        for (int i = 0; i < EMB_OUTS; ++i)
#pragma unroll
            for (int r = 0; r < EMB_ROWS; ++r) acc[i][r] = 0.0f;
        const float* w[EMB_OUTS];
#pragma unroll
        for (int i = 0; i < EMB_OUTS; ++i) w[i] = W1 + (size_t)min(e0 + i, E - 1) * C;
#pragma unroll 2
        for (int c = lane; c < C; c += 64) {
            float xv[EMB_ROWS];
#pragma unroll
            for (int r = 0; r < EMB_ROWS; ++r) xv[r] = xs[r][c];
#pragma unroll
            for (int i = 0; i < EMB_OUTS; ++i) {
                const float wv_ = w[i][c];
#pragma unroll
                for (int r = 0; r < EMB_ROWS; ++r) acc[i][r] = fmaf(wv_, xv[r], acc[i][r]);
            }
        }
#pragma unroll
        for (int i = 0; i < EMB_OUTS; ++i)
#pragma unroll
            for (int r = 0; r < EMB_ROWS; ++r) {
                float s = wave_sum(acc[i][r]);
                if (lane == 0 && e0 + i < E && r < rows) {
                    s += b1[e0 + i];
                    s = s > 0.0f ? s : s * slope;
                    fs[r][e0 + i] = s;
                    fe[(size_t)(n0 + r) * E + e0 + i] = s;
                }
            }
    }
    __syncthreads();
    if (nrm != nullptr && wv < rows) {                       // wave r: |fe[n0 + r]|
        float ss = 0.0f;
        for (int e = lane; e < E; e += 64) ss = fmaf(fs[wv][e], fs[wv][e], ss);
        ss = wave_sum(ss);
        if (lane == 0) nrm[n0 + wv] = sqrtf(ss);
    }
    if (out == nullptr) return;
    for (int t = wv; t < rows * D; t += 4) {
        const int r = t / D, d = t - r * D;
        float s = 0.0f;
        for (int e = lane; e < E; e += 64) s = fmaf(W2[(size_t)d * E + e], fs[r][e], s);
        s = wave_sum(s);
        if (lane == 0) out[(size_t)(n0 + r) * D + d] = s + b2[d];
    }
}

}  // namespace

static int embed_launch(const float* x, int ldx, int N, int C, const float* W1, const float* b1, int E, const float* W2,
                        const float* b2, int D, float slope, float* fe, float* out, float* nrm, void* stream) {
    if (x == nullptr || W1 == nullptr || b1 == nullptr || fe == nullptr || N <= 0 || ldx < C) return AADG_E_BADARG;
    if (out != nullptr && (W2 == nullptr || b2 == nullptr || D <= 0)) return AADG_E_BADARG;
    if (C <= 0 || C > EMB_MAX_C || E <= 0 || E > EMB_MAX_E) return AADG_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_embed, dim3((N + EMB_ROWS - 1) / EMB_ROWS), dim3(256), 0, (hipStream_t)stream, x, ldx, N, W1, b1, W2, b2, C, E,
                       D, slope, fe, out, nrm);
    AADG_LAUNCH_CHECK();
    return 0;
}

extern "C" int aadg_embed_prologue_f32(const float* x, int ldx, int N, int C, const float* W1, const float* b1, int E,
                                       const float* W2, const float* b2, int D, float slope, float* fe, float* out,
                                       void* stream) {
    return embed_launch(x, ldx, N, C, W1, b1, E, W2, b2, D, slope, fe, out, nullptr, stream);
}

extern "C" int aadg_embed_prologue_norm_f32(const float* x, int ldx, int N, int C, const float* W1, const float* b1, int E,
                                            const float* W2, const float* b2, int D, float slope, float* fe, float* out,
                                            float* row_norm, void* stream) {
    if (row_norm == nullptr) return AADG_E_BADARG;
    return embed_launch(x, ldx, N, C, W1, b1, E, W2, b2, D, slope, fe, out, row_norm, stream);
}

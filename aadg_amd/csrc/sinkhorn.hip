// Fused Sinkhorn-divergence reward on gfx950 (a14-a16; kernel K1).
//
// Replaces geomloss.SamplesLoss("sinkhorn", cost="IntCst(1)-(X|Y)/(Norm2(X)*Norm2(Y))",
// backend="online") (search_dg.py:116) and the 18-call reward loop (search_dg.py:150-162): the
// reference spends ~44 KeOps log-sum-exp launches + one host sync per call; here ONE launch runs
// every (policy, domain-pair) problem, one 256-thread workgroup per problem, with
//   * both point clouds staged once in LDS (row stride E+1: conflict-free column and row walks),
//   * the three cosine-cost matrices C_xx, C_yy, C_xy built once in LDS (the cost does not depend
//     on eps, so the whole eps-scaling loop runs out of LDS -- no HBM traffic after the prologue),
//   * the eps schedule (float64, geomloss `epsilon_schedule`) computed in-kernel from the
//     in-kernel diameter, so there is no host round trip,
//   * log-sum-exp reductions over G-lane groups with wavefront shuffles.
// Algorithm = geomloss 0.2.4 sinkhorn_loop/sinkhorn_cost, debias=True, p=2.
#include "common.h"

namespace {

constexpr int EPS_CAP = 64;

struct Problem {
    int n, m;        // cloud sizes
};

template <bool LAW>
__global__ __launch_bounds__(256) void k_sinkhorn(const float* __restrict__ feat, const float* __restrict__ row_norm, int ld, int E,
                                                  const int* __restrict__ cloud_rows,
                                                  const int* __restrict__ cloud_off,
                                                  const int* __restrict__ prob_xy, int D, int B, int M,
                                                  int nmax, float blur, float scaling, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int p = blockIdx.x;
    const int ldx = E + 1;
    const int ldc = nmax | 1;

    // ---- carve LDS -------------------------------------------------------------------------
    double* eps_s = reinterpret_cast<double*>(smem_raw);                  // EPS_CAP doubles
    float* xs = reinterpret_cast<float*>(eps_s + EPS_CAP);                // nmax * ldx
    float* ys = xs + (size_t)nmax * ldx;                                  // nmax * ldx
    float* Cxx = ys + (size_t)nmax * ldx;                                 // nmax * ldc
    float* Cyy = Cxx + (size_t)nmax * ldc;
    float* Cxy = Cyy + (size_t)nmax * ldc;
    float* vec = Cxy + (size_t)nmax * ldc;                                // 10 * nmax
    float* a_x = vec, *b_y = vec + nmax, *a_y = vec + 2 * nmax, *b_x = vec + 3 * nmax;
    float* at_x = vec + 4 * nmax, *bt_y = vec + 5 * nmax, *at_y = vec + 6 * nmax, *bt_x = vec + 7 * nmax;
    float* nrm_x = vec + 8 * nmax, *nrm_y = vec + 9 * nmax;
    float* red = vec + 10 * nmax;                                          // 8 floats scratch
    int* s_nits = reinterpret_cast<int*>(red + 8);

    // ---- which rows ------------------------------------------------------------------------
    int n, m, cx = 0, cy = 0, j = 0, d1 = 0, d2 = 0;
    if (LAW) {
        // problem p = policy j, q-th domain pair in lexicographic (d1<d2) order
        const int P = D * (D - 1) / 2;
        j = p / P;
        int q = p - j * P;
        d1 = 0;
        while (q >= D - 1 - d1) { q -= D - 1 - d1; ++d1; }
        d2 = d1 + 1 + q;
        n = m = B;
    } else {
        cx = prob_xy[2 * p];
        cy = prob_xy[2 * p + 1];
        n = cloud_off[cx + 1] - cloud_off[cx];
        m = cloud_off[cy + 1] - cloud_off[cy];
    }
    auto row_of = [&](bool is_y, int r) -> int {
        if (LAW) return (r * D + (is_y ? d2 : d1)) * M + j;
        return cloud_rows[cloud_off[is_y ? cy : cx] + r];
    };

    // ---- stage clouds in LDS (coalesced along E) --------------------------------------------
    for (int r = tid >> 6; r < n + m; r += 4) {
        const bool is_y = r >= n;
        const int rr = is_y ? r - n : r;
        const int grow = row_of(is_y, rr);
        const float* src = feat + (size_t)grow * ld;
        float* dst = (is_y ? ys : xs) + (size_t)rr * ldx;
        if (row_norm != nullptr) {                           // norms from the producer (aadg_embed_prologue_norm_f32)
            for (int k = tid & 63; k < E; k += 64) dst[k] = src[k];
            if ((tid & 63) == 0) (is_y ? nrm_y : nrm_x)[rr] = row_norm[grow];
        } else {
            float ss = 0.f;
            for (int k = tid & 63; k < E; k += 64) { const float v = src[k]; dst[k] = v; ss = fmaf(v, v, ss); }
            ss = wave_sum(ss);
            if ((tid & 63) == 0) (is_y ? nrm_y : nrm_x)[rr] = sqrtf(ss);
        }
    }
    __syncthreads();

    // ---- diameter^2 = sum_k (max_k - min_k)^2 over x u y --------------------------------------
    {
        float part = 0.f;
        for (int k = tid; k < E; k += 256) {
            float lo = INFINITY, hi = -INFINITY;
            for (int r = 0; r < n; ++r) { const float v = xs[(size_t)r * ldx + k]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
            for (int r = 0; r < m; ++r) { const float v = ys[(size_t)r * ldx + k]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
            part += (hi - lo) * (hi - lo);
        }
        part = wave_sum(part);
        if ((tid & 63) == 0) red[tid >> 6] = part;
    }
    // ---- cost matrices -----------------------------------------------------------------------
    {
        const int nxx = n * n, nyy = m * m, nxy = n * m;
        for (int e = tid; e < nxx + nyy + nxy; e += 256) {
            const float *pa, *pb;
            float na, nb;
            float* dst;
            if (e < nxx) { const int i = e / n, jj = e - i * n; pa = xs + (size_t)i * ldx; pb = xs + (size_t)jj * ldx; na = nrm_x[i]; nb = nrm_x[jj]; dst = Cxx + i * ldc + jj; }
            else if (e < nxx + nyy) { const int t = e - nxx; const int i = t / m, jj = t - i * m; pa = ys + (size_t)i * ldx; pb = ys + (size_t)jj * ldx; na = nrm_y[i]; nb = nrm_y[jj]; dst = Cyy + i * ldc + jj; }
            else { const int t = e - nxx - nyy; const int i = t / m, jj = t - i * m; pa = xs + (size_t)i * ldx; pb = ys + (size_t)jj * ldx; na = nrm_x[i]; nb = nrm_y[jj]; dst = Cxy + i * ldc + jj; }
            float dot = 0.f;
            for (int k = 0; k < E; ++k) dot = fmaf(pa[k], pb[k], dot);
            *dst = 1.0f - dot / (na * nb);
        }
    }
    __syncthreads();
    // ---- eps schedule (geomloss epsilon_schedule, float64) -------------------------------------
    if (tid == 0) {
        const double diameter = (double)sqrtf(red[0] + red[1] + red[2] + red[3]);
        const double pw = 2.0;
        int c = 0;
        eps_s[c++] = diameter * diameter;
        const double start = pw * log(diameter), stop = pw * log((double)blur), step = pw * log((double)scaling);
        int len = (int)ceil((stop - start) / step);
        if (len < 0) len = 0;
        for (int i = 0; i < len && c < EPS_CAP - 1; ++i) eps_s[c++] = exp(start + (double)i * step);
        eps_s[c++] = (double)blur * (double)blur;
        *s_nits = c;
    }
    __syncthreads();
    const int nits = *s_nits;

    // ---- softmin machinery ---------------------------------------------------------------------
    int G = 8;
    const int big = n > m ? n : m;
    while (G < big && G < 64) G <<= 1;
    const int g = tid & (G - 1);
    const int slot = tid / G, nslots = 256 / G;
    const float a_log = logf(1.0f / (float)n), b_log = logf(1.0f / (float)m);

    // all four softmins of one sweep: dst <- -eps * LSE_j(logw + pot_j/eps - C_ij/eps)
    auto sweep = [&](double eps, bool use_pot, float* d_ax, float* d_by, float* d_ay, float* d_bx) {
        const float inv = (float)(1.0 / eps);
        const float feps = (float)eps;
        const int T = 2 * n + 2 * m;
        for (int o0 = 0; o0 < T; o0 += nslots) {
            const int o = o0 + slot;
            float mx = -INFINITY, s = 0.f;
            const float* Crow = nullptr; int cstride = 1, cnt = 0; const float* pot = nullptr; float lw = 0.f; float* dst = nullptr;
            if (o < n)              { Crow = Cxx + o * ldc;            cstride = 1;   cnt = n; pot = a_x; lw = a_log; dst = d_ax + o; }
            else if (o < 2 * n)     { const int i = o - n;     Crow = Cxy + i * ldc; cstride = 1;   cnt = m; pot = a_y; lw = b_log; dst = d_bx + i; }
            else if (o < 2 * n + m) { const int i = o - 2 * n; Crow = Cyy + i * ldc; cstride = 1;   cnt = m; pot = b_y; lw = b_log; dst = d_by + i; }
            else if (o < T)         { const int i = o - 2 * n - m; Crow = Cxy + i;   cstride = ldc; cnt = n; pot = b_x; lw = a_log; dst = d_ay + i; }
            for (int jj = g; jj < cnt; jj += G) {
                const float h = use_pot ? lw + pot[jj] / feps : lw;
                mx = fmaxf(mx, h - Crow[jj * cstride] * inv);
            }
            for (int off = G >> 1; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            for (int jj = g; jj < cnt; jj += G) {
                const float h = use_pot ? lw + pot[jj] / feps : lw;
                s += expf(h - Crow[jj * cstride] * inv - mx);
            }
            for (int off = G >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (g == 0 && dst) *dst = -feps * (mx + logf(s));
        }
    };

    // init at eps_s[0] (writes the potentials themselves), then the eps-scaling loop
    sweep(eps_s[0], false, at_x, bt_y, at_y, bt_x);
    __syncthreads();
    for (int i = tid; i < nmax; i += 256) { a_x[i] = at_x[i]; b_y[i] = bt_y[i]; a_y[i] = at_y[i]; b_x[i] = bt_x[i]; }
    __syncthreads();
    for (int it = 0; it < nits; ++it) {
        sweep(eps_s[it], true, at_x, bt_y, at_y, bt_x);
        __syncthreads();
        for (int i = tid; i < nmax; i += 256) {
            if (i < n) { a_x[i] = 0.5f * (a_x[i] + at_x[i]); b_x[i] = 0.5f * (b_x[i] + bt_x[i]); }
            if (i < m) { b_y[i] = 0.5f * (b_y[i] + bt_y[i]); a_y[i] = 0.5f * (a_y[i] + at_y[i]); }
        }
        __syncthreads();
    }
    // last extrapolation at the final eps (cross terms from the old values, simultaneously)
    sweep(eps_s[nits - 1], true, at_x, bt_y, at_y, bt_x);
    __syncthreads();
    // sinkhorn_cost: <alpha, b_x - a_x> + <beta, a_y - b_y>
    if (tid < 64) {
        float s1 = 0.f, s2 = 0.f;
        const float wa = 1.0f / (float)n, wb = 1.0f / (float)m;
        for (int i = tid; i < n; i += 64) s1 += wa * (bt_x[i] - at_x[i]);
        for (int i = tid; i < m; i += 64) s2 += wb * (at_y[i] - bt_y[i]);
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (tid == 0) out[p] = s1 + s2;
    }
}

// rewards[j] += ((d_0 + d_1) + d_2 ...) in pair order, one lane per policy
__global__ void k_rewards_accum(const float* dist, int M, int P, float* rewards) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    float acc = dist[(size_t)j * P];
    for (int q = 1; q < P; ++q) acc += dist[(size_t)j * P + q];
    rewards[j] += acc;
}

// (r - mean) / (std_unbiased + 1e-5); single wave, M <= 64 per pass (loops otherwise)
__global__ void k_normalize_rewards(const float* r, int M, float* out) {
    const int lane = threadIdx.x;
    float s = 0.f;
    for (int i = lane; i < M; i += 64) s += r[i];
    const float mean = wave_sum(s) / (float)M;
    float v = 0.f;
    for (int i = lane; i < M; i += 64) { const float d = r[i] - mean; v += d * d; }
    const float sd = sqrtf(wave_sum(v) / (float)(M - 1));
    for (int i = lane; i < M; i += 64) out[i] = (r[i] - mean) / (sd + 1e-5f);
}

size_t lds_bytes(int nmax, int E) {
    const size_t ldx = E + 1, ldc = nmax | 1;
    return EPS_CAP * sizeof(double) + sizeof(float) * (2 * (size_t)nmax * ldx + 3 * (size_t)nmax * ldc + 10 * (size_t)nmax + 8) + 16;
}

template <bool LAW>
int launch(const float* feat, int ld, int E, const int* cloud_rows, const int* cloud_off, const int* prob_xy, int D,
           int B, int M, int n_prob, int nmax, float blur, float scaling, float* out, hipStream_t st,
           const float* row_norm = nullptr) {
    const size_t lds = lds_bytes(nmax, E);
    if (lds > 160 * 1024) return AADG_E_UNSUPPORTED;
    if (lds > 48 * 1024)
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sinkhorn<LAW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_sinkhorn<LAW>, dim3(n_prob), dim3(256), lds, st, feat, row_norm, ld, E, cloud_rows, cloud_off, prob_xy, D,
                       B, M, nmax, blur, scaling, out);
    AADG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// large-cloud path (sinkhorn_big.hip)
size_t aadg_sinkhorn_big_workspace_bytes(int n_prob, int nmax, int E);
int aadg_sinkhorn_big_launch(const float* feat, int ld, int E, const int* cloud_rows, const int* cloud_off, const int* prob_xy,
                             int n_prob, int nmax, float blur, float scaling, float* out, void* ws, size_t ws_bytes,
                             hipStream_t st, int phases);

extern "C" size_t aadg_sinkhorn_workspace_bytes(int n_prob, int max_cloud, int E) {
    if (n_prob <= 0 || max_cloud <= 0 || E <= 0) return 0;
    const size_t small = aadg_align_up((size_t)n_prob * sizeof(float), 256);
    if (lds_bytes(max_cloud, E) <= 160 * 1024) return small;
    return aadg_align_up(aadg_sinkhorn_big_workspace_bytes(n_prob, max_cloud, E), 256);
}

extern "C" int aadg_sinkhorn_divergence_f32(const float* feat, int ld, int E, const int32_t* cloud_rows,
                                            const int32_t* cloud_off, const int32_t* prob_xy, int n_prob, int max_cloud,
                                            float blur, float scaling, float* out, void* ws, size_t ws_bytes,
                                            void* stream) {
    if (!feat || !cloud_rows || !cloud_off || !prob_xy || !out) return AADG_E_BADARG;
    if (E <= 0 || ld < E || n_prob <= 0 || max_cloud <= 0) return AADG_E_BADARG;
    if (!(blur > 0.f) || !(scaling > 0.f && scaling < 1.f)) return AADG_E_BADARG;
    if (lds_bytes(max_cloud, E) > 160 * 1024)           // clouds too large for the LDS-resident kernel
        return aadg_sinkhorn_big_launch(feat, ld, E, cloud_rows, cloud_off, prob_xy, n_prob, max_cloud, blur, scaling, out, ws,
                                        ws_bytes, reinterpret_cast<hipStream_t>(stream), 3);
    return launch<false>(feat, ld, E, cloud_rows, cloud_off, prob_xy, 0, 0, 0, n_prob, max_cloud, blur, scaling, out,
                         reinterpret_cast<hipStream_t>(stream));
}

/* Measurement entry (bench.py: roofline fractions of the two halves of the large-cloud path): as aadg_sinkhorn_divergence_f32 for clouds
 * beyond the LDS-resident kernel, running only `phases` -- bit 0: row normalisation + eps schedule + cost matrices into `ws`, bit 1: the
 * eps-scaling sweeps over the matrices already in `ws` + the result.  AADG_E_UNSUPPORTED for clouds the LDS-resident kernel takes. */
extern "C" int aadg_sinkhorn_divergence_phases_f32(const float* feat, int ld, int E, const int32_t* cloud_rows, const int32_t* cloud_off,
                                                   const int32_t* prob_xy, int n_prob, int max_cloud, float blur, float scaling,
                                                   float* out, void* ws, size_t ws_bytes, int phases, void* stream) {
    if (!feat || !cloud_rows || !cloud_off || !prob_xy || !out) return AADG_E_BADARG;
    if (E <= 0 || ld < E || n_prob <= 0 || max_cloud <= 0 || phases < 1 || phases > 3) return AADG_E_BADARG;
    if (!(blur > 0.f) || !(scaling > 0.f && scaling < 1.f)) return AADG_E_BADARG;
    if (lds_bytes(max_cloud, E) <= 160 * 1024) return AADG_E_UNSUPPORTED;
    return aadg_sinkhorn_big_launch(feat, ld, E, cloud_rows, cloud_off, prob_xy, n_prob, max_cloud, blur, scaling, out, ws, ws_bytes,
                                    reinterpret_cast<hipStream_t>(stream), phases);
}

static int sinkhorn_rewards(const float* fe, const float* row_norm, int D, int B, int M, int E, float blur, float scaling,
                            float* rewards_accum, void* ws, size_t ws_bytes, void* stream) {
    if (!fe || !rewards_accum || !ws) return AADG_E_BADARG;
    if (D < 2 || B <= 0 || M <= 0 || E <= 0) return AADG_E_BADARG;
    if (!(blur > 0.f) || !(scaling > 0.f && scaling < 1.f)) return AADG_E_BADARG;
    const int P = D * (D - 1) / 2;
    if (ws_bytes < aadg_align_up((size_t)M * P * sizeof(float), 256)) return AADG_E_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* dist = reinterpret_cast<float*>(ws);
    int rc = launch<true>(fe, E, E, nullptr, nullptr, nullptr, D, B, M, M * P, B, blur, scaling, dist, st, row_norm);
    if (rc) return rc;
    hipLaunchKernelGGL(k_rewards_accum, dim3((M + 63) / 64), dim3(64), 0, st, dist, M, P, rewards_accum);
    AADG_LAUNCH_CHECK();
    return 0;
}

extern "C" int aadg_sinkhorn_rewards_f32(const float* fe, int D, int B, int M, int E, float blur, float scaling,
                                         float* rewards_accum, void* ws, size_t ws_bytes, void* stream) {
    return sinkhorn_rewards(fe, nullptr, D, B, M, E, blur, scaling, rewards_accum, ws, ws_bytes, stream);
}

extern "C" int aadg_sinkhorn_rewards_norm_f32(const float* fe, const float* row_norm, int D, int B, int M, int E, float blur,
                                              float scaling, float* rewards_accum, void* ws, size_t ws_bytes, void* stream) {
    if (!row_norm) return AADG_E_BADARG;
    if ((long long)B > 0 && lds_bytes(B, E) > 160 * 1024) return AADG_E_UNSUPPORTED;      // the large-cloud path computes its own norms
    return sinkhorn_rewards(fe, row_norm, D, B, M, E, blur, scaling, rewards_accum, ws, ws_bytes, stream);
}

extern "C" int aadg_normalize_rewards_f32(const float* rewards, int M, float* out, void* stream) {
    if (!rewards || !out || M < 2) return AADG_E_BADARG;
    hipLaunchKernelGGL(k_normalize_rewards, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), rewards, M, out);
    AADG_LAUNCH_CHECK();
    return 0;
}

extern "C" int aadg_abi_version(void) { return AADG_ABI_VERSION; }

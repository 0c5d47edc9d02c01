// The policy controller as three kernels: the reference's Controller (models/controller.py:
// 9-145: embedding -> LSTMCell(32 -> 100) -> two tanh-squashed softmax heads, Q sub-policies x 2L decisions, state
// reset per sub-policy) and its PPO update (losses.py:117-157: 5 x {teacher-forced evaluate, clipped surrogate, backward,
// Adam}) are ~200 + ~5000 micro-launches in eager PyTorch and still ~1900 graph nodes when captured.  Here:
//
//   k_ctrl_transpose   W_ih^T, W_hh^T scratch copies (gate row = fastest index: coalesced forward reads) -- only for widths
//                      other than the module's 32 / 100
//   k_ctrl_rollout     grid M, one workgroup per policy row: its Q independent sequences run in lock-step; all
//                      activations of the 2L steps stay in LDS.  SAMPLE: inverse-CDF draw from caller-supplied uniforms,
//                      policies / sum log-prob / sum entropy / mean head probabilities.  UPDATE: teacher-forced
//                      forward, PPO ratio + clipped surrogate, full BPTT; per-(sequence, step) gate / logit / input
//                      gradients go to a scratch of R = M*Q*2L rows
//   k_ctrl_adam        one thread per parameter: gradient = sum over the R scratch rows (a [P x R] x [R x K] product,
//                      deterministic order), then the Adam update in place (torch.optim.Adam arithmetic) and the
//                      refreshed transposed copies for the next epoch
//
// float32 throughout; same arithmetic as the eager module up to summation order.
// Round 5: for the module's widths the calls run one workgroup per SEQUENCE instead (k_ctrl_sample_seq, k_ppo_rollout, k_ppo_grad_adam,
// further down); the three kernels above serve every other width.
//
// Round 2 (wall_clock64 stamps after every barrier, scripts/ubench/ctrl_phase_times.py): with one gate row per lane the gate
// phase took 10.5 us per step -- not load latency (weights in registers changed nothing) but 165 broadcast ds_read_b128 per wave
// and step through ONE LDS pipe shared by 8 waves.  For the module's widths (32 / 100, compile-time instantiation) a lane now
// owns 4 rows x one k-slice (forward) or 4 columns x one row slice (transposed products of the backward): 4x fewer LDS reads
// per FMA, weights register-resident for the whole rollout, the slices of a quad summed with two DPP adds.  Gate phase 10.5 ->
// 4.3 us, dh 4.9 -> 2.0, dx 2.5 -> 1.9; softmax on 16 lanes per sequence instead of one thread; k_ctrl_adam's conditional
// per-row loops (the head weights waited on 48 dependent L2 round trips) as unconditional batched loads: 49 -> 11.5 us.
// sample 80 -> 39 us, PPO update (5 epochs) 840 -> 320 us.
#include "common.h"

namespace {

constexpr int CT_THREADS = 512;
constexpr int CT_MAX_Q = 8;        // sequences per workgroup (sub-policies)
constexpr int CT_MAX_A = 16;       // actions per head
constexpr int CT_HP = 5, CT_XP = 10;   // row ranges of the transposed products in the backward (4H / CT_* % 4 == 0)

struct CtrlDims {
    int M, Q, S, E, H, NOPS, NMAGS;
    float cdiv;                    // C / T
};
struct CtrlParams {                // torch parameter order of the module
    float* emb;   // [NOPS + NMAGS, E]
    float* w_ih;  // [4H, E]
    float* w_hh;  // [4H, H]
    float* b_ih;  // [4H]
    float* b_hh;  // [4H]
    float* wop;   // [NOPS, H]
    float* bop;   // [NOPS]
    float* wmag;  // [NMAGS, H]
    float* bmag;  // [NMAGS]
};
struct CtrlPtrs9 { float* p[9]; };

// workspace layout (floats)
struct CtrlWs {
    size_t wt_ih, wt_hh;           // [E][4H], [H][4H]
    size_t dg, dl, dx, xin, hprev, hcur, tok;   // R rows each: 4H, CT_MAX_A, E, E, H, H, 1(int)
    size_t probs;                  // [M][2][CT_MAX_A] partial head-probability sums
    size_t counter;                // 1 int
    // k_ppo_rollout:
    size_t lpq;                    // [M * Q] 64-bit words {Adam step number of the epoch, sum log-prob of the sequence}; zero between epochs
    // k_ctrl_sample_seq:
    size_t sprobs;                 // [M * Q][2][CT_MAX_A] per-sequence head-probability sums
    size_t slp;                    // [M * Q][2] per-sequence sum log-prob, sum entropy
    size_t total;
};
__host__ __device__ inline int ctrl_n_params(const CtrlDims& d) {
    return (d.NOPS + d.NMAGS) * d.E + 4 * d.H * d.E + 4 * d.H * d.H + 8 * d.H + d.NOPS * d.H + d.NOPS + d.NMAGS * d.H + d.NMAGS;
}
__host__ __device__ inline CtrlWs ctrl_ws(const CtrlDims& d) {
    CtrlWs w;
    const size_t R = (size_t)d.M * d.Q * d.S, H4 = 4 * (size_t)d.H;
    size_t o = 0;
    w.wt_ih = o; o += (size_t)d.E * H4;
    w.wt_hh = o; o += (size_t)d.H * H4;
    w.dg = o; o += R * H4;
    w.dl = o; o += R * CT_MAX_A;
    w.dx = o; o += R * d.E;
    w.xin = o; o += R * d.E;
    w.hprev = o; o += R * d.H;
    w.hcur = o; o += R * d.H;
    w.tok = o; o += R;
    w.probs = o; o += (size_t)d.M * 2 * CT_MAX_A;
    w.counter = o; o += 4;
    o = (o + 3) / 4 * 4;
    w.lpq = o; o += (size_t)(2 * d.M * d.Q + 3) / 4 * 4;
    w.sprobs = o; o += (size_t)d.M * d.Q * 2 * CT_MAX_A;
    w.slp = o; o += (size_t)(2 * d.M * d.Q + 3) / 4 * 4;
    w.total = o;
    return w;
}

// LDS layout of k_ctrl_rollout (floats)
struct CtrlLds {
    size_t G, Cs, TC, Hs, X, P, TL, DL, DG, DH, DC, part, act, misc, Wh, Bh, Emb, total;
};
__host__ __device__ inline CtrlLds ctrl_lds(const CtrlDims& d) {
    CtrlLds l;
    const size_t Q = d.Q, S = d.S, H = d.H, E = d.E;
    size_t o = 0;
    l.G = o; o += Q * S * 4 * H;          // activated gates i, f, g, o
    l.Cs = o; o += Q * S * H;             // cell states
    l.TC = o; o += Q * S * H;             // tanh(c)
    l.Hs = o; o += Q * (S + 1) * H;       // hidden states, slot 0 = zeros
    l.X = o; o += Q * S * E;              // step inputs
    l.P = o; o += Q * S * CT_MAX_A;       // head probabilities
    l.TL = o; o += Q * S * CT_MAX_A;      // tanh(logits)
    l.DL = o; o += Q * CT_MAX_A;          // d logits of the current step
    l.DG = o; o += Q * 4 * H;             // d gate pre-activations of the current step
    l.DH = o; o += Q * H;
    l.DC = o; o += Q * H;
    l.part = o; o += (CT_HP * Q * H > CT_XP * Q * E ? CT_HP * Q * H : CT_XP * Q * E);   // partial transposed products
    l.act = o; o += Q * S;                // actions (int)
    l.misc = o; o += 8 + 2 * CT_MAX_Q + 8;
    l.Wh = o; o += (size_t)(d.NOPS + d.NMAGS) * H;      // head weights: op rows, then magnitude rows
    l.Bh = o; o += 2 * CT_MAX_A;                        // head biases
    l.Emb = o; o += (size_t)(d.NOPS + d.NMAGS) * E;     // embedding table
    l.total = o;
    return l;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ __launch_bounds__(256) void k_ctrl_transpose(CtrlParams P, CtrlDims d, float* ws) {
    const CtrlWs W = ctrl_ws(d);
    const int H4 = 4 * d.H, n_ih = H4 * d.E, n_hh = H4 * d.H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_ih + n_hh; i += gridDim.x * blockDim.x) {
        if (i < n_ih) {
            const int k = i / H4, j = i - k * H4;
            ws[W.wt_ih + i] = P.w_ih[(size_t)j * d.E + k];
        } else {
            const int i2 = i - n_ih, k = i2 / H4, j = i2 - k * H4;
            ws[W.wt_hh + i2] = P.w_hh[(size_t)j * d.H + k];
        }
    }
}

// SAMPLE = true: draw actions from `uniforms`; false: teacher-forced on `policies`, then the PPO backward.
// EC / HC > 0: embedding / hidden width known at compile time (the module's 32 / 100): every lane then keeps its weight column
// (EC + HC floats forward, 4 HC / CT_HP and 4 HC / CT_XP floats backward) in REGISTERS for the whole rollout -- the weights
// are read from L2 once per launch, all loads in flight together, instead of once per step in nine dependent chunks (the
// forward step was ~15 us of load latency for ~1 us of arithmetic).  EC = HC = 0: run-time widths, weights re-read per step.
template <bool SAMPLE, int EC, int HC>
__global__ __launch_bounds__(CT_THREADS) void k_ctrl_rollout(CtrlParams P, CtrlDims d, float* ws, const float* __restrict__ uniforms,
                                                             long long* __restrict__ policies, float* __restrict__ op_probs,
                                                             float* __restrict__ mag_probs, float* __restrict__ log_probs,
                                                             float* __restrict__ entropies, const float* __restrict__ old_log_probs,
                                                             const float* __restrict__ reward, float clip,
                                                             float* __restrict__ loss_terms) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    const CtrlWs W = ctrl_ws(d);
    const CtrlLds O = ctrl_lds(d);
    const int m = blockIdx.x, tid = threadIdx.x;
    constexpr bool FAST = EC > 0 && HC > 0;
    static_assert(!FAST || (EC % 4 == 0 && HC % 4 == 0 && 4 * HC <= CT_THREADS && CT_HP * HC <= CT_THREADS && CT_XP * EC <= CT_THREADS &&
                            (4 * HC) % CT_HP == 0 && (4 * HC / CT_HP) % 4 == 0 && (4 * HC) % CT_XP == 0 && (4 * HC / CT_XP) % 4 == 0),
                  "compile-time widths must fit the lane <-> weight-column mapping");
    const int Q = d.Q, S = d.S, H = FAST ? HC : d.H, E = FAST ? EC : d.E, H4 = 4 * H;
    float* G = L + O.G; float* Cs = L + O.Cs; float* TC = L + O.TC; float* Hs = L + O.Hs; float* X = L + O.X;
    float* Pp = L + O.P; float* TL = L + O.TL; float* DL = L + O.DL; float* DG = L + O.DG; float* DH = L + O.DH;
    float* DC = L + O.DC; float* part = L + O.part; int* act = reinterpret_cast<int*>(L + O.act); float* misc = L + O.misc;
    float* Wh = L + O.Wh; float* Bh = L + O.Bh; float* Emb = L + O.Emb;
    const float* wt_ih = ws + W.wt_ih;
    const float* wt_hh = ws + W.wt_hh;
    // the small tables live in LDS for the whole rollout
    for (int i = tid; i < d.NOPS * H; i += CT_THREADS) Wh[i] = P.wop[i];
    for (int i = tid; i < d.NMAGS * H; i += CT_THREADS) Wh[d.NOPS * H + i] = P.wmag[i];
    for (int i = tid; i < (d.NOPS + d.NMAGS) * E; i += CT_THREADS) Emb[i] = P.emb[i];
    if (tid < CT_MAX_A) { Bh[tid] = tid < d.NOPS ? P.bop[tid] : 0.0f; Bh[CT_MAX_A + tid] = tid < d.NMAGS ? P.bmag[tid] : 0.0f; }

    for (int i = tid; i < Q * (S + 1) * H; i += CT_THREADS) Hs[i] = 0.0f;
    for (int i = tid; i < Q * S * E; i += CT_THREADS) X[i] = 0.0f;
    if (!SAMPLE)
        for (int i = tid; i < Q * S; i += CT_THREADS) act[i] = (int)policies[(size_t)m * Q * S + i];
    if (tid < 8 + 2 * CT_MAX_Q) misc[tid] = 0.0f;
    __syncthreads();

    // ---------------------------------------------------------------------------------------------- forward
    float my_lp = 0.0f, my_ent = 0.0f;                     // lane 0 of each 16-lane group: its sequence's sum log-prob / entropy
    // FAST: lane (u = tid / 4, sl = tid % 4) keeps the weights of the FOUR gate rows of hidden unit u for k-slice sl of the inputs
    // (EC / 4 embedding inputs + HSL or HL hidden inputs).  One 16-byte LDS read of activations then feeds 16 FMAs instead of 4 (the
    // row-per-lane mapping was bound by the LDS broadcasts: 165 reads per wave and step, 10.5 us); the four k-slices of a quad are
    // summed with two DPP adds.
    constexpr int XSL = FAST ? EC / 4 : 4;                 // embedding inputs per slice
    constexpr int HSL = FAST ? (HC / 16) * 4 : 4;          // hidden inputs of slices 0..2
    constexpr int HL = FAST ? HC - 3 * HSL : 4;            // ... of slice 3 (the remainder); slices 0..2 carry zeros beyond HSL
    static_assert(!FAST || (EC % 16 == 0 && HL % 4 == 0 && HL >= HSL && HL - HSL <= HSL), "k-slices must be multiples of 4");
    float wxr[4][XSL], whr[4][HL];
    const int gu = tid >> 2, gsl = tid & 3;
    if constexpr (FAST) {
        if (tid < H4) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float* rx = P.w_ih + (size_t)(g * HC + gu) * EC + XSL * gsl;
                const float* rh = P.w_hh + (size_t)(g * HC + gu) * HC + HSL * gsl;
#pragma unroll
                for (int i = 0; i < XSL; i += 4) {
                    const float4 w = *reinterpret_cast<const float4*>(rx + i);
                    wxr[g][i] = w.x; wxr[g][i + 1] = w.y; wxr[g][i + 2] = w.z; wxr[g][i + 3] = w.w;
                }
#pragma unroll
                for (int i = 0; i < HL; i += 4) {
                    float4 w = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (i < HSL || gsl == 3) w = *reinterpret_cast<const float4*>(rh + i);
                    whr[g][i] = w.x; whr[g][i + 1] = w.y; whr[g][i + 2] = w.z; whr[g][i + 3] = w.w;
                }
            }
        }
    }
    for (int t = 0; t < S; ++t) {
        const bool op_step = (t & 1) == 0;
        const int NA = op_step ? d.NOPS : d.NMAGS;
        if constexpr (FAST) {
            if (tid < H4) {
                float acc[4][CT_MAX_Q];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int q = 0; q < CT_MAX_Q; ++q) acc[g][q] = 0.0f;
                if (t > 0) {
#pragma unroll
                    for (int i = 0; i < XSL; i += 4)
#pragma unroll
                        for (int q = 0; q < CT_MAX_Q; ++q)
                            if (q < Q) {
                                const float4 a = *reinterpret_cast<const float4*>(X + (q * S + t) * EC + XSL * gsl + i);
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    acc[g][q] = fmaf(wxr[g][i], a.x, acc[g][q]); acc[g][q] = fmaf(wxr[g][i + 1], a.y, acc[g][q]);
                                    acc[g][q] = fmaf(wxr[g][i + 2], a.z, acc[g][q]); acc[g][q] = fmaf(wxr[g][i + 3], a.w, acc[g][q]);
                                }
                            }
#pragma unroll
                    for (int i = 0; i < HL; i += 4)
#pragma unroll
                        for (int q = 0; q < CT_MAX_Q; ++q)
                            if (q < Q) {
                                // slices 0..2 read 4 inputs of the next slice in their last round: weights zero there
                                const float4 a = *reinterpret_cast<const float4*>(Hs + (q * (S + 1) + t) * HC + HSL * gsl + i);
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    acc[g][q] = fmaf(whr[g][i], a.x, acc[g][q]); acc[g][q] = fmaf(whr[g][i + 1], a.y, acc[g][q]);
                                    acc[g][q] = fmaf(whr[g][i + 2], a.z, acc[g][q]); acc[g][q] = fmaf(whr[g][i + 3], a.w, acc[g][q]);
                                }
                            }
                }
                // sum of the four k-slices (every lane of the quad ends with the total), then lane sl finishes gate sl:
                // bias, sigmoid / tanh, store
                const float b = P.b_ih[gsl * HC + gu] + P.b_hh[gsl * HC + gu];
#pragma unroll
                for (int q = 0; q < CT_MAX_Q; ++q)
                    if (q < Q) {
                        float mine = 0.0f;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float v = acc[g][q];
                            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // lane ^ 1
                            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // lane ^ 2
                            mine = gsl == g ? v : mine;
                        }
                        mine += b;
                        G[(q * S + t) * H4 + gsl * HC + gu] = gsl == 2 ? tanhf(mine) : sigmoidf_(mine);
                    }
            }
        } else if (tid < H4) {                             // gate row `tid` of all Q sequences
            float acc[CT_MAX_Q];
            const float b = P.b_ih[tid] + P.b_hh[tid];
#pragma unroll
            for (int q = 0; q < CT_MAX_Q; ++q) acc[q] = b;
            if (t > 0) {
                // 16 weights in flight per lane; activations are read as 16-byte LDS broadcasts (4 inputs per read):
                // the scalar version spent ~40 us per step here (wall_clock64), mostly waiting on 660 dependent ds_reads
                constexpr int KB = 16;
                const bool vec4 = ((E | H) & 3) == 0;
                for (int k0 = 0; k0 < E; k0 += KB) {
                    float w[KB];
#pragma unroll
                    for (int i = 0; i < KB; ++i) w[i] = k0 + i < E ? wt_ih[(size_t)(k0 + i) * H4 + tid] : 0.0f;
                    if (vec4) {
#pragma unroll
                        for (int i = 0; i < KB; i += 4)
#pragma unroll
                            for (int q = 0; q < CT_MAX_Q; ++q)
                                if (q < Q && k0 + i < E) {
                                    const float4 a = *reinterpret_cast<const float4*>(X + (q * S + t) * E + k0 + i);
                                    acc[q] = fmaf(w[i], a.x, acc[q]); acc[q] = fmaf(w[i + 1], a.y, acc[q]);
                                    acc[q] = fmaf(w[i + 2], a.z, acc[q]); acc[q] = fmaf(w[i + 3], a.w, acc[q]);
                                }
                    } else {
#pragma unroll
                        for (int i = 0; i < KB; ++i)
#pragma unroll
                            for (int q = 0; q < CT_MAX_Q; ++q)
                                if (q < Q && k0 + i < E) acc[q] = fmaf(w[i], X[(q * S + t) * E + k0 + i], acc[q]);
                    }
                }
                for (int k0 = 0; k0 < H; k0 += KB) {
                    float w[KB];
#pragma unroll
                    for (int i = 0; i < KB; ++i) w[i] = k0 + i < H ? wt_hh[(size_t)(k0 + i) * H4 + tid] : 0.0f;
                    if (vec4) {
#pragma unroll
                        for (int i = 0; i < KB; i += 4)
#pragma unroll
                            for (int q = 0; q < CT_MAX_Q; ++q)
                                if (q < Q && k0 + i < H) {
                                    const float4 a = *reinterpret_cast<const float4*>(Hs + (q * (S + 1) + t) * H + k0 + i);
                                    acc[q] = fmaf(w[i], a.x, acc[q]); acc[q] = fmaf(w[i + 1], a.y, acc[q]);
                                    acc[q] = fmaf(w[i + 2], a.z, acc[q]); acc[q] = fmaf(w[i + 3], a.w, acc[q]);
                                }
                    } else {
#pragma unroll
                        for (int i = 0; i < KB; ++i)
#pragma unroll
                            for (int q = 0; q < CT_MAX_Q; ++q)
                                if (q < Q && k0 + i < H) acc[q] = fmaf(w[i], Hs[(q * (S + 1) + t) * H + k0 + i], acc[q]);
                    }
                }
            }
            const bool is_g = tid >= 2 * H && tid < 3 * H;
#pragma unroll
            for (int q = 0; q < CT_MAX_Q; ++q)
                if (q < Q) G[(q * S + t) * H4 + tid] = is_g ? tanhf(acc[q]) : sigmoidf_(acc[q]);
        }
        __syncthreads();
        for (int i = tid; i < Q * H; i += CT_THREADS) {    // cell / hidden update
            const int q = i / H, u = i - q * H;
            const float* g4 = G + (q * S + t) * H4;
            const float c_prev = t > 0 ? Cs[(q * S + t - 1) * H + u] : 0.0f;
            const float c = g4[H + u] * c_prev + g4[u] * g4[2 * H + u];
            const float tc = tanhf(c);
            Cs[(q * S + t) * H + u] = c;
            TC[(q * S + t) * H + u] = tc;
            Hs[(q * (S + 1) + t + 1) * H + u] = g4[3 * H + u] * tc;
        }
        __syncthreads();
        {                                                  // head logits: 4 lanes per (sequence, action) output
            const int o = tid >> 2, part = tid & 3;
            const int q = o / NA, a = o - q * NA;
            const bool live = q < Q;
            float z = 0.0f;
            if (live) {
                const float* wrow = Wh + (size_t)((op_step ? 0 : d.NOPS) + a) * H;
                const float* h = Hs + (q * (S + 1) + t + 1) * H;
#pragma unroll 5
                for (int k = part; k < H; k += 4) z = fmaf(wrow[k], h[k], z);
            }
            z += __shfl_xor(z, 1, 64); z += __shfl_xor(z, 2, 64);
            if (live && part == 0) {
                z += Bh[(op_step ? 0 : CT_MAX_A) + a];
                const float tl = tanhf(z);
                TL[(q * S + t) * CT_MAX_A + a] = tl;
                Pp[(q * S + t) * CT_MAX_A + a] = d.cdiv * tl;      // squashed logit, turned into a probability below
            }
        }
        __syncthreads();
        if (tid < Q * CT_MAX_A) {                          // softmax, draw / gather, next input token: 16 lanes per sequence
            const int q = tid / CT_MAX_A, a = tid % CT_MAX_A;
            float* p = Pp + (q * S + t) * CT_MAX_A;
            const bool live = a < NA;
            const float v = live ? p[a] : -INFINITY;
            float mx = v;
#pragma unroll
            for (int o = CT_MAX_A / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, CT_MAX_A));
            const float z = v - mx;
            float sum = live ? expf(z) : 0.0f;
#pragma unroll
            for (int o = CT_MAX_A / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, CT_MAX_A);
            const float lp = z - logf(sum), pr = live ? expf(lp) : 0.0f;
            float ent = live ? -lp * pr : 0.0f;
#pragma unroll
            for (int o = CT_MAX_A / 2; o > 0; o >>= 1) ent += __shfl_xor(ent, o, CT_MAX_A);
            if (live) p[a] = pr;
            int a_sel = 0;
            if (SAMPLE) {
                // inverse CDF in action order (a sequential sum, as a host-side scan of the same probabilities would do it)
                const float u = uniforms[(size_t)m * Q * S + q * S + t];
                float cum = 0.0f;
                a_sel = NA - 1;
                bool found = false;
#pragma unroll
                for (int b = 0; b < CT_MAX_A; ++b) {
                    cum += __shfl(pr, b, CT_MAX_A);
                    if (!found && b < NA && u < cum) { a_sel = b; found = true; }
                }
                if (a == 0) {
                    act[q * S + t] = a_sel;
                    policies[(size_t)m * Q * S + q * S + t] = a_sel;
                }
            } else {
                a_sel = act[q * S + t];
            }
            const float lp_sel = __shfl(lp, a_sel, CT_MAX_A);
            my_lp += lp_sel;                               // per-sequence sums, combined in fixed order after the loop
            my_ent += ent;
            if (t + 1 < S) {
                const int tok = a_sel + (op_step ? 0 : d.NOPS);
                for (int k = a; k < E; k += CT_MAX_A) X[(q * S + t + 1) * E + k] = Emb[(size_t)tok * E + k];
            }
        }
        __syncthreads();
    }

    if (tid < Q * CT_MAX_A && tid % CT_MAX_A == 0) { misc[8 + tid / CT_MAX_A] = my_lp; misc[8 + CT_MAX_Q + tid / CT_MAX_A] = my_ent; }
    __syncthreads();
    if (tid == 0) {
        float a = 0.0f, b = 0.0f;
        for (int q = 0; q < Q; ++q) { a += misc[8 + q]; b += misc[8 + CT_MAX_Q + q]; }
        misc[1] = a; misc[2] = b;
    }
    __syncthreads();
    if (SAMPLE) {
        if (tid == 0) { log_probs[m] = misc[1]; entropies[m] = misc[2]; }
        // mean head probabilities over all M*Q*L rows: per-workgroup partial sums, last workgroup combines
        float* partial = ws + W.probs + (size_t)m * 2 * CT_MAX_A;
        if (tid < 2 * CT_MAX_A) {
            const int head = tid / CT_MAX_A, a = tid - head * CT_MAX_A;
            float s = 0.0f;
            for (int q = 0; q < Q; ++q)
                for (int t = head; t < S; t += 2) s += Pp[(q * S + t) * CT_MAX_A + a];
            partial[tid] = (a < (head == 0 ? d.NOPS : d.NMAGS)) ? s : 0.0f;
        }
        __threadfence();
        __syncthreads();
        int* counter = reinterpret_cast<int*>(ws + W.counter);
        __shared__ int last;
        if (tid == 0) last = atomicAdd(counter, 1) == d.M - 1;
        __syncthreads();
        if (last) {
            __threadfence();
            if (tid < 2 * CT_MAX_A) {
                const int head = tid / CT_MAX_A, a = tid - head * CT_MAX_A;
                float s = 0.0f;
                for (int mm = 0; mm < d.M; ++mm) s += ws[W.probs + (size_t)mm * 2 * CT_MAX_A + tid];
                const float inv = 1.0f / (float)(d.M * Q * (S / 2));
                if (head == 0 && a < d.NOPS) op_probs[a] = s * inv;
                if (head == 1 && a < d.NMAGS) mag_probs[a] = s * inv;
            }
            if (tid == 0) *counter = 0;
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------- PPO surrogate
    if (tid == 0) {
        const float lp = misc[1];
        const float ratio = expf(lp - old_log_probs[m]);
        const float clipped = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
        const float r = reward[m];
        const float a = ratio * r, b = clipped * r;
        loss_terms[m] = -fminf(a, b);                      // the caller averages over M
        // d(-min(a, b))/d lp: through a when a <= b (ties: both halves reach `ratio` because clamp is then the identity)
        const bool inside = ratio >= 1.0f - clip && ratio <= 1.0f + clip;
        float g = 0.0f;
        if (a < b || (a == b && inside)) g = -r * ratio;
        else if (a == b && !inside) g = -0.5f * r * ratio;
        misc[3] = g / (float)d.M;
    }
    for (int i = tid; i < Q * H; i += CT_THREADS) { DH[i] = 0.0f; DC[i] = 0.0f; }
    __syncthreads();
    const float gl = misc[3];

    // ---------------------------------------------------------------------------------------------- backward
    // FAST: the transposed products dh = W_hh^T dg and dx = W_ih^T dg with 4 output columns x JS gate rows per lane: lane
    // (jg, kg, jl) owns columns 4 kg .. 4 kg + 3 and gate rows [(4 jg + jl) JS, +JS); its JS x 4 weights stay in registers for all
    // steps, one 16-byte LDS read of dg feeds 16 FMAs, the four row slices of a quad are summed with DPP and lane jl stores column
    // 4 kg + jl of partial sum jg (CT_HP of them, combined below in fixed order)
    constexpr int JS = FAST ? 4 * HC / (4 * CT_HP) : 4;
    static_assert(!FAST || (JS % 4 == 0 && JS * 4 * CT_HP == 4 * HC && CT_HP * HC <= CT_THREADS && CT_HP <= CT_XP), "row slices of the transposed products");
    float wbh[JS][4], wbx[JS][4];
    const int bjl = tid & 3;
    const int hkg = (tid >> 2) % (FAST ? HC / 4 : 1), hjg = (tid >> 2) / (FAST ? HC / 4 : 1);      // dh lanes: tid < CT_HP * HC
    const int xkg = (tid >> 2) % (FAST ? EC / 4 : 1), xjg = (tid >> 2) / (FAST ? EC / 4 : 1);      // dx lanes: tid < CT_HP * EC
    if constexpr (FAST) {
        if (tid < CT_HP * HC) {
#pragma unroll
            for (int i = 0; i < JS; ++i) {
                const float4 w = *reinterpret_cast<const float4*>(P.w_hh + (size_t)((4 * hjg + bjl) * JS + i) * HC + 4 * hkg);
                wbh[i][0] = w.x; wbh[i][1] = w.y; wbh[i][2] = w.z; wbh[i][3] = w.w;
            }
        }
        if (tid < CT_HP * EC) {
#pragma unroll
            for (int i = 0; i < JS; ++i) {
                const float4 w = *reinterpret_cast<const float4*>(P.w_ih + (size_t)((4 * xjg + bjl) * JS + i) * EC + 4 * xkg);
                wbx[i][0] = w.x; wbx[i][1] = w.y; wbx[i][2] = w.z; wbx[i][3] = w.w;
            }
        }
    }
    for (int t = S - 1; t >= 0; --t) {
        const bool op_step = (t & 1) == 0;
        const int NA = op_step ? d.NOPS : d.NMAGS;
        if (tid < Q * CT_MAX_A) {                          // d logits
            const int q = tid / CT_MAX_A, a = tid - q * CT_MAX_A;
            const size_t r = ((size_t)m * Q + q) * S + t;
            float dl = 0.0f;
            if (a < NA) {
                const float p = Pp[(q * S + t) * CT_MAX_A + a], tl = TL[(q * S + t) * CT_MAX_A + a];
                const float dz = gl * ((a == act[q * S + t] ? 1.0f : 0.0f) - p);
                dl = dz * d.cdiv * (1.0f - tl * tl);
            }
            DL[q * CT_MAX_A + a] = dl;
            ws[W.dl + r * CT_MAX_A + a] = dl;
        }
        __syncthreads();
        for (int i = tid; i < Q * H; i += CT_THREADS) {    // through the head and the cell
            const int q = i / H, u = i - q * H;
            const size_t r = ((size_t)m * Q + q) * S + t;
            const float* wh = Wh + (size_t)(op_step ? 0 : d.NOPS) * H;
            float dh = DH[i];
            for (int a = 0; a < NA; ++a) dh = fmaf(wh[(size_t)a * H + u], DL[q * CT_MAX_A + a], dh);
            const float* g4 = G + (q * S + t) * H4;
            const float ig = g4[u], fg = g4[H + u], gg = g4[2 * H + u], og = g4[3 * H + u];
            const float tc = TC[(q * S + t) * H + u];
            const float c_prev = t > 0 ? Cs[(q * S + t - 1) * H + u] : 0.0f;
            const float d_o = dh * tc;
            const float dc = dh * og * (1.0f - tc * tc) + DC[i];
            DC[i] = dc * fg;
            const float dgi = dc * gg * ig * (1.0f - ig);
            const float dgf = dc * c_prev * fg * (1.0f - fg);
            const float dgg = dc * ig * (1.0f - gg * gg);
            const float dgo = d_o * og * (1.0f - og);
            DG[q * H4 + u] = dgi; DG[q * H4 + H + u] = dgf; DG[q * H4 + 2 * H + u] = dgg; DG[q * H4 + 3 * H + u] = dgo;
            float* dgr = ws + W.dg + r * H4;
            dgr[u] = dgi; dgr[H + u] = dgf; dgr[2 * H + u] = dgg; dgr[3 * H + u] = dgo;
            ws[W.hprev + r * H + u] = Hs[(q * (S + 1) + t) * H + u];
            ws[W.hcur + r * H + u] = Hs[(q * (S + 1) + t + 1) * H + u];
        }
        for (int i = tid; i < Q * E; i += CT_THREADS) {    // the step input (for dW_ih)
            const int q = i / E, k = i - q * E;
            const size_t r = ((size_t)m * Q + q) * S + t;
            ws[W.xin + r * E + k] = X[(q * S + t) * E + k];
        }
        if (tid < Q) {
            const size_t r = ((size_t)m * Q + tid) * S + t;
            // token whose embedding was this step's input (-1: the zero input of the first step)
            reinterpret_cast<int*>(ws + W.tok)[r] = t > 0 ? act[tid * S + t - 1] + (((t - 1) & 1) == 0 ? 0 : d.NOPS) : -1;
        }
        __syncthreads();
        if (t > 0) {
            // dh_{t-1} = W_hh^T dgates: CT_HP row ranges x H columns (reads of W_hh are contiguous over the column index)
            if constexpr (FAST) {
                if (tid < CT_HP * HC) {
                    float acc[4][CT_MAX_Q];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int q = 0; q < CT_MAX_Q; ++q) acc[c][q] = 0.0f;
#pragma unroll
                    for (int i = 0; i < JS; i += 4)
#pragma unroll
                        for (int q = 0; q < CT_MAX_Q; ++q)
                            if (q < Q) {
                                const float4 a = *reinterpret_cast<const float4*>(DG + q * H4 + (4 * hjg + bjl) * JS + i);
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    acc[c][q] = fmaf(wbh[i][c], a.x, acc[c][q]); acc[c][q] = fmaf(wbh[i + 1][c], a.y, acc[c][q]);
                                    acc[c][q] = fmaf(wbh[i + 2][c], a.z, acc[c][q]); acc[c][q] = fmaf(wbh[i + 3][c], a.w, acc[c][q]);
                                }
                            }
#pragma unroll
                    for (int q = 0; q < CT_MAX_Q; ++q)
                        if (q < Q) {
                            float mine = 0.0f;
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                float v = acc[c][q];
                                v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // lane ^ 1
                                v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // lane ^ 2
                                mine = bjl == c ? v : mine;
                            }
                            part[(hjg * Q + q) * H + 4 * hkg + bjl] = mine;
                        }
                }
            } else
            for (int i = tid; i < CT_HP * H; i += CT_THREADS) {
                const int pr = i / H, k = i - pr * H;
                const int j0 = pr * H4 / CT_HP, j1 = (pr + 1) * H4 / CT_HP;
                float acc[CT_MAX_Q];
#pragma unroll
                for (int q = 0; q < CT_MAX_Q; ++q) acc[q] = 0.0f;
                constexpr int JB = 16;
                for (int jb = j0; jb < j1; jb += JB) {
                    float w[JB];
#pragma unroll
                    for (int i = 0; i < JB; ++i) w[i] = jb + i < j1 ? P.w_hh[(size_t)(jb + i) * H + k] : 0.0f;
#pragma unroll
                    for (int i = 0; i < JB; i += 4)           // ranges start at multiples of 4: 16-byte LDS broadcasts
#pragma unroll
                        for (int q = 0; q < CT_MAX_Q; ++q)
                            if (q < Q && jb + i < j1) {
                                const float4 a = *reinterpret_cast<const float4*>(DG + q * H4 + jb + i);
                                acc[q] = fmaf(w[i], a.x, acc[q]); acc[q] = fmaf(w[i + 1], a.y, acc[q]);
                                acc[q] = fmaf(w[i + 2], a.z, acc[q]); acc[q] = fmaf(w[i + 3], a.w, acc[q]);
                            }
                }
#pragma unroll
                for (int q = 0; q < CT_MAX_Q; ++q)
                    if (q < Q) part[(pr * Q + q) * H + k] = acc[q];
            }
            __syncthreads();
            for (int i = tid; i < Q * H; i += CT_THREADS) {
                const int q = i / H, k = i - q * H;
                float s = 0.0f;
                for (int pr = 0; pr < CT_HP; ++pr) s += part[(pr * Q + q) * H + k];
                DH[i] = s;
            }
            __syncthreads();
            // dx_t = W_ih^T dgates -> gradient of the embedding row that fed this step: CT_XP row ranges x E columns
            if constexpr (FAST) {
                if (tid < CT_HP * EC) {
                    float acc[4][CT_MAX_Q];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int q = 0; q < CT_MAX_Q; ++q) acc[c][q] = 0.0f;
#pragma unroll
                    for (int i = 0; i < JS; i += 4)
#pragma unroll
                        for (int q = 0; q < CT_MAX_Q; ++q)
                            if (q < Q) {
                                const float4 a = *reinterpret_cast<const float4*>(DG + q * H4 + (4 * xjg + bjl) * JS + i);
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    acc[c][q] = fmaf(wbx[i][c], a.x, acc[c][q]); acc[c][q] = fmaf(wbx[i + 1][c], a.y, acc[c][q]);
                                    acc[c][q] = fmaf(wbx[i + 2][c], a.z, acc[c][q]); acc[c][q] = fmaf(wbx[i + 3][c], a.w, acc[c][q]);
                                }
                            }
#pragma unroll
                    for (int q = 0; q < CT_MAX_Q; ++q)
                        if (q < Q) {
                            float mine = 0.0f;
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                float v = acc[c][q];
                                v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // lane ^ 1
                                v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // lane ^ 2
                                mine = bjl == c ? v : mine;
                            }
                            part[(xjg * Q + q) * E + 4 * xkg + bjl] = mine;
                        }
                }
            } else
            for (int i = tid; i < CT_XP * E; i += CT_THREADS) {
                const int pr = i / E, k = i - pr * E;
                const int j0 = pr * H4 / CT_XP, j1 = (pr + 1) * H4 / CT_XP;
                float acc[CT_MAX_Q];
#pragma unroll
                for (int q = 0; q < CT_MAX_Q; ++q) acc[q] = 0.0f;
                constexpr int JB = 20;
                for (int jb = j0; jb < j1; jb += JB) {
                    float w[JB];
#pragma unroll
                    for (int i = 0; i < JB; ++i) w[i] = jb + i < j1 ? P.w_ih[(size_t)(jb + i) * E + k] : 0.0f;
#pragma unroll
                    for (int i = 0; i < JB; i += 4)           // ranges start at multiples of 4: 16-byte LDS broadcasts
#pragma unroll
                        for (int q = 0; q < CT_MAX_Q; ++q)
                            if (q < Q && jb + i < j1) {
                                const float4 a = *reinterpret_cast<const float4*>(DG + q * H4 + jb + i);
                                acc[q] = fmaf(w[i], a.x, acc[q]); acc[q] = fmaf(w[i + 1], a.y, acc[q]);
                                acc[q] = fmaf(w[i + 2], a.z, acc[q]); acc[q] = fmaf(w[i + 3], a.w, acc[q]);
                            }
                }
#pragma unroll
                for (int q = 0; q < CT_MAX_Q; ++q)
                    if (q < Q) part[(pr * Q + q) * E + k] = acc[q];
            }
            __syncthreads();
            for (int i = tid; i < Q * E; i += CT_THREADS) {
                const int q = i / E, k = i - q * E;
                const size_t r = ((size_t)m * Q + q) * S + t;
                float s = 0.0f;
                for (int pr = 0; pr < (FAST ? CT_HP : CT_XP); ++pr) s += part[(pr * Q + q) * E + k];
                ws[W.dx + r * E + k] = s;
            }
            __syncthreads();
        }
    }
}

// sum_r a[r * sa] * b[r * sb] (b == nullptr: sum_r a[r * sa]) with RB row pairs in flight; ascending r, one fma per row
__device__ __forceinline__ float dot_rows(const float* __restrict__ a, int sa, const float* __restrict__ b, int sb, int R) {
    constexpr int RB = 16;
    float g = 0.0f;
    for (int r0 = 0; r0 < R; r0 += RB) {
        float x[RB], y[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const bool in = r0 + i < R;
            x[i] = in ? a[(size_t)(r0 + i) * sa] : 0.0f;
            y[i] = b == nullptr ? 1.0f : (in ? b[(size_t)(r0 + i) * sb] : 0.0f);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) g = b == nullptr ? g + x[i] : fmaf(x[i], y[i], g);
    }
    return g;
}

// one thread per parameter element: gradient from the R scratch rows, then Adam in place
__global__ __launch_bounds__(256) void k_ctrl_adam(CtrlParams P, CtrlPtrs9 exp_avg, CtrlPtrs9 exp_avg_sq, CtrlDims d, float* ws,
                                                   float lr, float beta1, float beta2, float eps, float bc1, float bc2_sqrt) {
    const CtrlWs W = ctrl_ws(d);
    const int H4 = 4 * d.H, NT = d.NOPS + d.NMAGS, R = d.M * d.Q * d.S;
    const int sizes[9] = {NT * d.E, H4 * d.E, H4 * d.H, H4, H4, d.NOPS * d.H, d.NOPS, d.NMAGS * d.H, d.NMAGS};
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int which = 0;
    while (which < 9 && idx >= sizes[which]) { idx -= sizes[which]; ++which; }
    if (which >= 9) return;
    const float* dg = ws + W.dg;
    const float* dl = ws + W.dl;
    float g = 0.0f;
    float* param;
    switch (which) {
        case 0: {                                          // embedding[row][k]: rows whose input token was `row`
            const int row = idx / d.E, k = idx - row * d.E;
            const int* tok = reinterpret_cast<const int*>(ws + W.tok);
            constexpr int RB = 16;
            for (int r0 = 0; r0 < R; r0 += RB) {           // unconditional loads, the token test selects afterwards
                int tk[RB];
                float x[RB];
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int r = min(r0 + i, R - 1);
                    tk[i] = r0 + i < R ? tok[r] : -2;
                    x[i] = ws[W.dx + (size_t)r * d.E + k];
                }
#pragma unroll
                for (int i = 0; i < RB; ++i) g = tk[i] == row ? g + x[i] : g;
            }
            param = P.emb;
            break;
        }
        case 1: {                                          // W_ih[j][k]
            const int j = idx / d.E, k = idx - j * d.E;
            g = dot_rows(dg + j, H4, ws + W.xin + k, d.E, R);
            param = P.w_ih;
            break;
        }
        case 2: {                                          // W_hh[j][k]
            const int j = idx / d.H, k = idx - j * d.H;
            g = dot_rows(dg + j, H4, ws + W.hprev + k, d.H, R);
            param = P.w_hh;
            break;
        }
        case 3: case 4: {                                  // b_ih[j], b_hh[j]
            g = dot_rows(dg + idx, H4, nullptr, 0, R);
            param = which == 3 ? P.b_ih : P.b_hh;
            break;
        }
        case 5: case 7: {                                  // head weight[a][k]: steps of that head only (S is even: row parity = step parity)
            const int a = idx / d.H, k = idx - a * d.H, par = which == 5 ? 0 : 1;
            g = dot_rows(dl + (size_t)par * CT_MAX_A + a, 2 * CT_MAX_A, ws + W.hcur + (size_t)par * d.H + k, 2 * d.H, R / 2);
            param = which == 5 ? P.wop : P.wmag;
            break;
        }
        default: {                                         // head bias[a]
            const int par = which == 6 ? 0 : 1;
            g = dot_rows(dl + (size_t)par * CT_MAX_A + idx, 2 * CT_MAX_A, nullptr, 0, R / 2);
            param = which == 6 ? P.bop : P.bmag;
            break;
        }
    }
    // torch.optim.Adam (no weight decay, no amsgrad): m, v EMAs, step = lr / bc1, denom = sqrt(v) / sqrt(bc2) + eps
    float* mp = exp_avg.p[which] + idx;
    float* vp = exp_avg_sq.p[which] + idx;
    const float mnew = *mp + (g - *mp) * (1.0f - beta1);   // lerp, as torch does
    const float vnew = beta2 * *vp + (1.0f - beta2) * g * g;
    *mp = mnew;
    *vp = vnew;
    const float denom = sqrtf(vnew) / bc2_sqrt + eps;
    const float pnew = param[idx] - (lr / bc1) * (mnew / denom);
    param[idx] = pnew;
    if (which == 1) { const int j = idx / d.E, k = idx - j * d.E; ws[W.wt_ih + (size_t)k * H4 + j] = pnew; }
    if (which == 2) { const int j = idx / d.H, k = idx - j * d.H; ws[W.wt_hh + (size_t)k * H4 + j] = pnew; }
}

// ------------------------------------------------------------------------------------------------
// PPO epoch as two short kernels, one workgroup per SEQUENCE (round 5): k_ppo_rollout, k_ppo_grad_adam.
//   With the actions given, the M * Q sub-policies are independent sequences of S = 2L steps (the reference resets the LSTM
//   state per sub-policy, models/controller.py:82-84), the step inputs are known up front, and the heads / softmax of a step
//   do not feed the recurrence.  So instead of M workgroups that walk Q sequences in lock-step through S x (gates, cell,
//   head, softmax) with 4 barriers per step, and one thread per parameter summing R = M Q S scratch rows:
//   k_ppo_rollout (grid M Q):
//     * workgroup w owns sequence (m, q) = (w / Q, w % Q): the recurrence is S x (gate products + cell update fused in the
//       quad that holds the four gate sums: ONE LDS-only barrier per step), then the heads and soft-maxes of all S steps at once;
//       W_ih lives in LDS, the W_hh slices in registers;
//     * the policy's log-probability (sum over its Q sequences -> PPO ratio -> d loss / d log-prob) crosses workgroups as
//       tagged 64-bit words {Adam step number, value}: published with one atomic store, polled by the first Q lanes -- no
//       fence, no counter (the Q workgroups of a policy are co-resident: the grid is <= 128 workgroups, one per CU suffices);
//     * BPTT: d logits of all steps at once, S x (cell backward, W_hh^T dgates: two barriers), W_ih^T dgates of all steps at
//       once; the sequence's FACTORS (d gates, d logits, d inputs, inputs, hidden states: 11 KB) go to the scratch rows;
//   k_ppo_grad_adam (grid M Q): workgroup w owns a slice of the parameters (14 gate rows of W_ih / W_hh / the biases, one
//     head row, one embedding row), stages all sequences' factors in LDS with one batch of loads, sums over the rows in
//     ascending order and applies torch.optim.Adam's update to its slice.
//   Measured and dropped in the same round: ALL epochs in ONE persistent launch (grid barriers on arrival counters, the
//   gradient phase as a noinline function, then fence-free with sc1 atomics for everything shared): 333-395 us against the
//   324 us of the 10 launches above -- a device-scope acquire / release is L2 maintenance on every XCD (~8 us until the
//   first load behind it returns), without the fences every shared access goes to the coherence point, and the
//   ~6000-instruction kernel made the register allocator spill the weight slices inside the recurrence.
// Widths: the module's 32 / 100 only; S <= 8, M Q S <= 128; anything else takes the launches above.
// ------------------------------------------------------------------------------------------------
constexpr int PPO_MAX_S = 8;
constexpr int PPO_WIP = 4;             // padding of the LDS copy of W_ih (row pitch E + 4 floats)
struct PpoLds { size_t G, Cs, TC, Hs, X, P, TL, DL, DG, DX, partx, part, Wi, act, tok, misc, Wh, Bh, Emb, total; };
__host__ __device__ inline PpoLds ppo_lds(const CtrlDims& d) {
    PpoLds l;
    const size_t S = d.S, H = d.H, E = d.E;
    size_t o = 0;
    l.G = o; o += S * 4 * H;              // activated gates i, f, g, o
    l.Cs = o; o += S * H;                 // cell states
    l.TC = o; o += S * H;                 // tanh(c)
    l.Hs = o; o += (S + 1) * H;           // hidden states, slot 0 = zeros
    l.X = o; o += S * E;                  // step inputs (slot 0 = zeros)
    l.P = o; o += S * CT_MAX_A;           // head probabilities
    l.TL = o; o += S * CT_MAX_A;          // tanh(logits)
    l.DL = o; o += S * CT_MAX_A;          // d logits, all steps
    l.DG = o; o += S * 4 * H;             // d gate pre-activations, all steps
    l.DX = o; o += S * E;                 // d step inputs
    l.partx = o; o += S * CT_HP * E;      // partial W_ih^T dgates
    l.part = o; o += CT_HP * H;           // partial W_hh^T dgates of the current step
    l.Wi = o; o += 4 * H * (E + PPO_WIP); // W_ih (for W_ih^T dgates: registers are for the recurrence's weights)
    l.act = o; o += S;                    // actions (int)
    l.tok = o; o += S;                    // input token of each step (int, -1: zero input)
    l.misc = o; o += 8 + PPO_MAX_S;
    o = (o + 3) / 4 * 4;
    l.Wh = o; o += (size_t)(d.NOPS + d.NMAGS) * H;
    l.Bh = o; o += 2 * CT_MAX_A;
    l.Emb = o; o += (size_t)(d.NOPS + d.NMAGS) * E;
    l.total = o;
    return l;
}

// gradient + Adam phase: gate rows per pass, thread ranges of the four roles, 16-byte vectors / words staged per thread
constexpr int PPO_JP = 14;
constexpr int PPO_GA_GRID = 100;       // workgroups of k_ppo_grad_adam: 4 gate rows each
constexpr int PPO_T_WIH = 352, PPO_T_HEAD = 464, PPO_T_EMB = 496;
constexpr int PPO_GA_HS = 8, PPO_GA_XI = 2, PPO_GA_DG = 4;
constexpr int PPO_GA_UR = 16;           // rows per round of the sums: the per-row tables are padded to a multiple of it
struct PpoGaLds { size_t HS, XI, DXg, DGs, DLs, HIT, ROW, IDN, PT, total; };
__host__ __device__ inline PpoGaLds ppo_ga_lds(const CtrlDims& d) {
    PpoGaLds l;
    const size_t R = (size_t)d.M * d.Q * d.S, RS = (size_t)d.M * d.Q * (d.S + 1);
    size_t o = 0;
    l.HS = o; o += RS * d.H;              // hidden states of every sequence [M Q][S + 1][H]
    l.XI = o; o += R * d.E;               // step inputs
    l.DXg = o; o += R * d.E;              // d step inputs
    const size_t Rp = (R + PPO_GA_UR - 1) / PPO_GA_UR * PPO_GA_UR;
    l.DGs = o; o += Rp * 16;              // the pass's columns of d gates (pitch 16), zero rows behind R
    l.DLs = o; o += Rp;                   // the pass's column of d logits (zero at the other head's steps)
    l.HIT = o; o += Rp;                   // 1 where the row's input token is the pass's embedding row
    l.ROW = o; o += Rp;                   // float offset of h_{t-1} in HS: (r + r / S) * H
    l.IDN = o; o += Rp;                   // float offset of row r in XI / DXg: r * E
    o = (o + 3) / 4 * 4;
    l.PT = o; o += 2 * 28;               // 27 pointers: parameters, exp_avg, exp_avg_sq
    l.total = o;
    return l;
}
// the staging registers of the gradient phase cover these many rows
inline bool ppo_ga_fits(const CtrlDims& d) {
    const int R = d.M * d.Q * d.S, RS = d.M * d.Q * (d.S + 1);
    const int Rp = (R + PPO_GA_UR - 1) / PPO_GA_UR * PPO_GA_UR;
    return RS * (d.H / 4) <= PPO_GA_HS * CT_THREADS && R * (d.E / 4) <= PPO_GA_XI * CT_THREADS && Rp * 16 <= PPO_GA_DG * CT_THREADS && Rp <= CT_THREADS;
}


template <int CTRL>
__device__ __forceinline__ float quad_lane(float v) {          // lane CTRL of the caller's quad
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL * 0x55, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum(float v) {           // every lane of the quad ends with the quad's sum
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // lane ^ 1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // lane ^ 2
    return v;
}

// workgroup barrier that orders LDS accesses only: global loads issued before it stay in flight (a __syncthreads() waits for
// vmcnt(0): every weight prefetch would be exposed at the next barrier)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// k_ppo_grad_adam: the gradient + Adam half of a PPO epoch, grid M * Q.
// Workgroup w owns the gate rows [w JW, (w + 1) JW) of W_ih / W_hh / both biases and the head / embedding rows
// [w AW, (w + 1) AW): it stages every sequence's hidden states, inputs and d inputs in LDS once (one batch of loads), per
// pass its <= 14 columns of d gates and one column of d logits, sums over the rows r in ascending order (the order of
// k_ctrl_adam) and applies torch.optim.Adam's update to its parameters.
template <int EC, int HC>
__global__ __launch_bounds__(CT_THREADS) void k_ppo_grad_adam(CtrlParams P, CtrlPtrs9 exp_avg, CtrlPtrs9 exp_avg_sq, CtrlDims d, float* ws,
                                                              float bc1, float bc2s, float lr, float beta1, float beta2, float eps) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    constexpr int H4 = 4 * HC, A = CT_MAX_A;
    const CtrlWs W = ctrl_ws(d);
    const PpoGaLds Og = ppo_ga_lds(d);
    const int w = blockIdx.x, tid = threadIdx.x;
    const int S = d.S, nseq = d.M * d.Q, NT = d.NOPS + d.NMAGS;
    // parameter / moment pointers as a table (role-dependent index below): 0..8 parameters, 9..17 exp_avg, 18..26 exp_avg_sq
    float** PT = reinterpret_cast<float**>(L + Og.PT);
    if (tid == 0) {
        PT[0] = P.emb; PT[1] = P.w_ih; PT[2] = P.w_hh; PT[3] = P.b_ih; PT[4] = P.b_hh; PT[5] = P.wop; PT[6] = P.bop; PT[7] = P.wmag; PT[8] = P.bmag;
        for (int i = 0; i < 9; ++i) { PT[9 + i] = exp_avg.p[i]; PT[18 + i] = exp_avg_sq.p[i]; }
    }
    // the rollout's {step, log-prob} slots: zero again before the next rollout (every tag >= 1 is then fresh, whatever step0 a caller passes)
    if (w == 0 && tid < nseq) reinterpret_cast<unsigned long long*>(ws + W.lpq)[tid] = 0ull;
    __syncthreads();
    {
            const int R = nseq * S, RS = nseq * (S + 1);
            float* HSg = L + Og.HS; float* XI = L + Og.XI; float* DXg = L + Og.DXg; float* DGs = L + Og.DGs; float* DLs = L + Og.DLs;
            float* HIT = L + Og.HIT; int* ROW = reinterpret_cast<int*>(L + Og.ROW); int* IDN = reinterpret_cast<int*>(L + Og.IDN);
            const int Rp = (R + PPO_GA_UR - 1) / PPO_GA_UR * PPO_GA_UR;
            const int NG = gridDim.x;                                       // the parameter slices are per workgroup of THIS grid
            const int JW = (H4 + NG - 1) / NG, AW = (NT + NG - 1) / NG;
            // role of this thread (the same in every pass): 0 W_hh rows, 1 W_ih rows + biases, 2 head row + its bias, 3 embedding row
            const int role = tid < PPO_JP * (HC / 4) ? 0 : (tid >= PPO_T_WIH && tid < PPO_T_WIH + PPO_JP * (EC / 4)) ? 1 :
                             (tid >= PPO_T_HEAD && tid < PPO_T_HEAD + HC / 4 + 1) ? 2 : (tid >= PPO_T_EMB && tid < PPO_T_EMB + EC / 4) ? 3 : 4;
            const int l = role == 0 ? tid : role == 1 ? tid - PPO_T_WIH : role == 2 ? tid - PPO_T_HEAD : tid - PPO_T_EMB;
            const int jj = role == 0 ? l / (HC / 4) : l / (EC / 4);
            const int k4 = role == 0 ? l - jj * (HC / 4) : role == 1 ? l - jj * (EC / 4) : l;
            int tk = -2;                                                    // input token of row `tid` (loaded in the first pass)
            // ONE pass: the grid (PPO_GA_GRID workgroups) is sized so that a workgroup owns <= PPO_JP gate rows and <= 1 head / embedding row
            {
                constexpr int ps = 0;
                constexpr bool FIRST = true;
                const int j0 = w * JW + ps * PPO_JP;                            // first gate row of the pass
                const int jn = max(0, min(min(H4, (w + 1) * JW) - j0, PPO_JP)); // its rows
                const int arow = w * AW + ps;                                   // head / embedding row of the pass
                const bool has_a = ps < AW && arow < NT;
                const int apar = arow < d.NOPS ? 0 : 1, acol = arow - (apar ? d.NOPS : 0);
                // this thread's parameters (4 consecutive ones, or 1): parameter / moment pointers of the pass
                float* pp = nullptr; float* mp = nullptr; float* vp = nullptr;
                bool vec = true, live = false;
                if (role == 0 && jj < jn) {
                    const size_t o = (size_t)(j0 + jj) * HC + 4 * k4;
                    pp = PT[2] + o; mp = PT[11] + o; vp = PT[20] + o; live = true;
                } else if (role == 1 && jj < jn) {
                    const size_t o = (size_t)(j0 + jj) * EC + 4 * k4;
                    pp = PT[1] + o; mp = PT[10] + o; vp = PT[19] + o; live = true;
                } else if (role == 2 && has_a) {
                    live = true;
                    if (k4 < HC / 4) {
                        const size_t o = (size_t)acol * HC + 4 * k4;
                        const int wi = apar ? 7 : 5;
                        pp = PT[wi] + o; mp = PT[9 + wi] + o; vp = PT[18 + wi] + o;
                    } else {
                        vec = false;
                        const int wi = apar ? 8 : 6;
                        pp = PT[wi] + acol; mp = PT[9 + wi] + acol; vp = PT[18 + wi] + acol;
                    }
                } else if (role == 3 && has_a) {
                    const size_t o = (size_t)arow * EC + 4 * k4;
                    pp = PT[0] + o; mp = PT[9] + o; vp = PT[18] + o; live = true;
                }
                // bias lanes of role 1: k4 == 0 -> b_ih[j], k4 == 1 -> b_hh[j]
                const bool blane = role == 1 && jj < jn && k4 < 2;
                float* bp = nullptr; float* bm = nullptr; float* bv = nullptr;
                if (blane) {
                    bp = PT[3 + k4] + j0 + jj; bm = PT[12 + k4] + j0 + jj; bv = PT[21 + k4] + j0 + jj;
                }
                // ---- ONE batch of loads: (first pass) every sequence's hidden states / inputs / d inputs / tokens; the pass's columns of
                // d gates and d logits; the parameters and moments this thread will update
                float4 rh[PPO_GA_HS], rx[PPO_GA_XI], rd[PPO_GA_XI];
                const int nhs = RS * (HC / 4), nxi = R * (EC / 4);
                if constexpr (FIRST) {
                    const float4* ghs = reinterpret_cast<const float4*>(ws + W.hprev);
                    const float4* gxi = reinterpret_cast<const float4*>(ws + W.xin);
                    const float4* gdx = reinterpret_cast<const float4*>(ws + W.dx);
#pragma unroll
                    for (int i = 0; i < PPO_GA_HS; ++i) rh[i] = ghs[min(tid + i * CT_THREADS, nhs - 1)];
#pragma unroll
                    for (int i = 0; i < PPO_GA_XI; ++i) {
                        rx[i] = gxi[min(tid + i * CT_THREADS, nxi - 1)];
                        rd[i] = gdx[min(tid + i * CT_THREADS, nxi - 1)];
                    }
                    tk = reinterpret_cast<const int*>(ws + W.tok)[min(tid, R - 1)];
                }
                float rg[PPO_GA_DG];
#pragma unroll
                for (int i = 0; i < PPO_GA_DG; ++i) {
                    const int e = tid + i * CT_THREADS, r = min(e >> 4, R - 1), cj = e & 15;
                    rg[i] = ws[W.dg + (size_t)r * H4 + min(j0 + cj, H4 - 1)];
                }
                const float rl = ws[W.dl + (size_t)min(tid, R - 1) * A + (has_a ? acol : 0)];
                float pv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, mv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, vv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                float bpv = 0.0f, bmv = 0.0f, bvv = 0.0f;
                if (live && vec) {
                    const float4 p0 = *reinterpret_cast<const float4*>(pp), m0 = *reinterpret_cast<const float4*>(mp), v0 = *reinterpret_cast<const float4*>(vp);
                    pv[0] = p0.x; pv[1] = p0.y; pv[2] = p0.z; pv[3] = p0.w; mv[0] = m0.x; mv[1] = m0.y; mv[2] = m0.z; mv[3] = m0.w;
                    vv[0] = v0.x; vv[1] = v0.y; vv[2] = v0.z; vv[3] = v0.w;
                } else if (live) {
                    pv[0] = *pp; mv[0] = *mp; vv[0] = *vp;
                }
                if (blane) { bpv = *bp; bmv = *bm; bvv = *bv; }
                if constexpr (FIRST) {
#pragma unroll
                    for (int i = 0; i < PPO_GA_HS; ++i)
                        if (tid + i * CT_THREADS < nhs) reinterpret_cast<float4*>(HSg)[tid + i * CT_THREADS] = rh[i];
#pragma unroll
                    for (int i = 0; i < PPO_GA_XI; ++i)
                        if (tid + i * CT_THREADS < nxi) {
                            reinterpret_cast<float4*>(XI)[tid + i * CT_THREADS] = rx[i];
                            reinterpret_cast<float4*>(DXg)[tid + i * CT_THREADS] = rd[i];
                        }
                    if (tid < Rp) {
                        const int r = min(tid, R - 1);
                        ROW[tid] = (r + r / S) * HC;
                        IDN[tid] = r * EC;
                    }
                } else {
                    lds_barrier();                                              // the previous pass has read its columns
                }
#pragma unroll
                for (int i = 0; i < PPO_GA_DG; ++i) {
                    const int e = tid + i * CT_THREADS;
                    if (e < Rp * 16) DGs[e] = e < R * 16 ? rg[i] : 0.0f;
                }
                if (tid < Rp) {
                    DLs[tid] = tid < R && (tid & 1) == apar ? rl : 0.0f;    // S is even: row parity = step parity = head
                    HIT[tid] = tid < R && has_a && tk == arow ? 1.0f : 0.0f;
                }
                lds_barrier();
                // ---- the sums over the rows r = (sequence, step), ascending; UR rows' operands in flight at a time.  ONE loop for the four
                // roles (a wave that holds several roles would otherwise walk several latency-bound loops one after the other):
                //   sum_r g[r] * y[r][4 k4 ..]   with   g = the row's d gate (roles 0, 1) / its d logit, zero at the other head's steps
                //   (role 2) / 1 where the row's input token is this embedding row, else 0 (role 3) -- adding 0 * y changes nothing --
                //   and y = h_{t-1} / x_t / h_t / d x_t
                // No load of the loop is conditional (a padded row has g = 0) and no address needs a multiply (the tables hold float offsets):
                // a round is 16 table reads, then 32 operand reads, in flight together.
                constexpr int UR = PPO_GA_UR;
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                float bsum = 0.0f;
                if (live) {
                    const float* gcol = role <= 1 ? DGs + jj : role == 2 ? DLs : HIT;
                    const int gsh = role <= 1 ? 4 : 0;
                    const int* rowtab = (role == 0 || role == 2) ? ROW : IDN;      // hidden-state rows: [sequence][S + 1] slots
                    const float* ycol = role == 0 ? HSg + 4 * k4 : role == 1 ? XI + 4 * k4 : role == 2 ? HSg + HC + 4 * min(k4, HC / 4 - 1) : DXg + 4 * k4;
                    for (int r0 = 0; r0 < Rp; r0 += UR) {
                        int yo[UR];
                        float g[UR];
                        float4 y[UR];
#pragma unroll
                        for (int i = 0; i < UR; ++i) yo[i] = rowtab[r0 + i];
#pragma unroll
                        for (int i = 0; i < UR; ++i) {
                            g[i] = gcol[(r0 + i) << gsh];
                            y[i] = *reinterpret_cast<const float4*>(ycol + yo[i]);
                        }
#pragma unroll
                        for (int i = 0; i < UR; ++i) {
                            acc[0] = fmaf(g[i], y[i].x, acc[0]); acc[1] = fmaf(g[i], y[i].y, acc[1]);
                            acc[2] = fmaf(g[i], y[i].z, acc[2]); acc[3] = fmaf(g[i], y[i].w, acc[3]);
                            bsum += g[i];
                        }
                    }
                }
                // ---- torch.optim.Adam (no weight decay, no amsgrad): m, v EMAs (lerp, as torch does), step = lr / bc1,
                // denom = sqrt(v) / sqrt(bc2) + eps
                if (live) {
                    if (!vec) acc[0] = bsum;
                    float pn[4], mn[4], vn[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        mn[c] = mv[c] + (acc[c] - mv[c]) * (1.0f - beta1);
                        vn[c] = beta2 * vv[c] + (1.0f - beta2) * acc[c] * acc[c];
                        const float denom = sqrtf(vn[c]) / bc2s + eps;
                        pn[c] = pv[c] - (lr / bc1) * (mn[c] / denom);
                    }
                    if (vec) {
                        *reinterpret_cast<float4*>(mp) = make_float4(mn[0], mn[1], mn[2], mn[3]);
                        *reinterpret_cast<float4*>(vp) = make_float4(vn[0], vn[1], vn[2], vn[3]);
                        *reinterpret_cast<float4*>(pp) = make_float4(pn[0], pn[1], pn[2], pn[3]);
                    } else {
                        *mp = mn[0]; *vp = vn[0]; *pp = pn[0];
                    }
                }
                if (blane) {
                    const float mnew = bmv + (bsum - bmv) * (1.0f - beta1);
                    const float vnew = beta2 * bvv + (1.0f - beta2) * bsum * bsum;
                    *bm = mnew; *bv = vnew;
                    *bp = bpv - (lr / bc1) * (mnew / (sqrtf(vnew) / bc2s + eps));
                }
            }
    }
}

template <int EC, int HC>
__global__ __launch_bounds__(CT_THREADS) void k_ppo_rollout(CtrlParams P, CtrlDims d, float* ws, const long long* __restrict__ policies,
                                                            const float* __restrict__ old_log_probs, const float* __restrict__ reward,
                                                            float clip, unsigned tag, float* __restrict__ loss_terms /* this epoch's [M] */) {
    static_assert(EC % 16 == 0 && HC % 4 == 0 && 4 * HC <= CT_THREADS && CT_HP * HC <= CT_THREADS && 3 * CT_HP * EC <= CT_THREADS &&
                  (4 * HC) % CT_HP == 0 && (4 * HC / CT_HP) % 4 == 0, "widths must fit the lane <-> weight-slice mappings");
    extern __shared__ __attribute__((aligned(16))) float L[];
    const CtrlWs W = ctrl_ws(d);
    const PpoLds O = ppo_lds(d);
    const int w = blockIdx.x;
    const int Q = d.Q, S = d.S;
    constexpr int H4 = 4 * HC, A = CT_MAX_A;
    const int NT = d.NOPS + d.NMAGS;
    const int m = w / Q;
    float* G = L + O.G; float* Cs = L + O.Cs; float* TC = L + O.TC; float* Hs = L + O.Hs; float* X = L + O.X;
    float* Pp = L + O.P; float* TL = L + O.TL; float* DL = L + O.DL; float* DG = L + O.DG; float* DX = L + O.DX;
    float* partx = L + O.partx; float* part = L + O.part; float* Wi = L + O.Wi; int* act = reinterpret_cast<int*>(L + O.act);
    int* tok = reinterpret_cast<int*>(L + O.tok); float* misc = L + O.misc;
    float* Wh = L + O.Wh; float* Bh = L + O.Bh; float* Emb = L + O.Emb;
    unsigned long long* lpq = reinterpret_cast<unsigned long long*>(ws + W.lpq);
    constexpr int XSL = EC / 4, HSL = (HC / 16) * 4, HL = HC - 3 * HSL;
    static_assert(HL % 4 == 0 && HL >= HSL && HL - HSL <= HSL, "k-slices must be multiples of 4");
    constexpr int JS = 4 * HC / (4 * CT_HP);
    constexpr int XT = CT_HP * EC;                                               // dx lanes per step group

    {
        const int tid = threadIdx.x;
        {
            const float* pw_ih = P.w_ih;
            const float* pw_hh = P.w_hh;
            const uint32_t otid = (uint32_t)tid;                    // 32-bit element offsets from the uniform bases
            const uint32_t ogu = otid >> 2, osl = otid & 3u;
            // lane <-> weight-slice mappings (those of k_ctrl_rollout's compile-time-width instantiation)
            const int gu = tid >> 2, gsl = tid & 3;
            const int hkg = (tid >> 2) % (HC / 4), hjg = (tid >> 2) / (HC / 4);          // dh lanes: tid < CT_HP * HC
            const int xgrp = tid / XT, xl = tid - xgrp * XT;
            const int xkg = (xl >> 2) % (EC / 4), xjg = (xl >> 2) / (EC / 4);
            // ------------------------------------------------------------------------------------------ tables, weights
            // every global load of the forward in ONE batch (behind the kernel boundary they all miss the caches: ~3 us per
            // dependent round trip): head weights and embedding rows as one 16-byte vector per thread, then the weight slices
            float4 r_op = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r_mag = r_op, r_emb = r_op;
            float r_bop = 0.0f, r_bmag = 0.0f;
            if (tid < d.NOPS * (HC / 4)) r_op = reinterpret_cast<const float4*>(P.wop)[tid];
            if (tid < d.NMAGS * (HC / 4)) r_mag = reinterpret_cast<const float4*>(P.wmag)[tid];
            if (tid < NT * (EC / 4)) r_emb = reinterpret_cast<const float4*>(P.emb)[tid];
            if (tid < d.NOPS) r_bop = P.bop[tid];
            if (tid < d.NMAGS) r_bmag = P.bmag[tid];
            constexpr int WIV = H4 * EC / 4, WIN = (WIV + CT_THREADS - 1) / CT_THREADS;     // W_ih as 16-byte vectors
            float4 r_wi[WIN];
#pragma unroll
            for (int i = 0; i < WIN; ++i) r_wi[i] = reinterpret_cast<const float4*>(pw_ih)[min(otid + (uint32_t)(i * CT_THREADS), (uint32_t)(WIV - 1))];
            float whr[4][HL];                                   // (the input-to-hidden weights are read from the LDS copy of W_ih)
            float bias = 0.0f;
            if (tid < H4) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float* rh = pw_hh + ((uint32_t)(g * HC) + ogu) * (uint32_t)HC + (uint32_t)HSL * osl;
#pragma unroll
                    for (int i = 0; i < HL; i += 4) {
                        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        if (i < HSL || gsl == 3) v = *reinterpret_cast<const float4*>(rh + i);
                        whr[g][i] = v.x; whr[g][i + 1] = v.y; whr[g][i + 2] = v.z; whr[g][i + 3] = v.w;
                    }
                }
                bias = P.b_ih[gsl * HC + gu] + P.b_hh[gsl * HC + gu];
            }
            if (tid < S) {                                      // the sequence's actions and input tokens
                const int a = (int)policies[(size_t)w * S + tid];
                act[tid] = a;
                if (tid + 1 < S) tok[tid + 1] = a + ((tid & 1) == 0 ? 0 : d.NOPS);
                if (tid == 0) tok[0] = -1;
            }
            if (tid < d.NOPS * (HC / 4)) reinterpret_cast<float4*>(Wh)[tid] = r_op;
            if (tid < d.NMAGS * (HC / 4)) reinterpret_cast<float4*>(Wh + d.NOPS * HC)[tid] = r_mag;
            if (tid < NT * (EC / 4)) reinterpret_cast<float4*>(Emb)[tid] = r_emb;
            if (tid < A) { Bh[tid] = r_bop; Bh[A + tid] = r_bmag; }
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                const int v = tid + i * CT_THREADS;
                if (v < WIV) *reinterpret_cast<float4*>(Wi + (v / (EC / 4)) * (EC + PPO_WIP) + 4 * (v % (EC / 4))) = r_wi[i];
            }
            for (int i = tid; i < HC; i += CT_THREADS) Hs[i] = 0.0f;
            lds_barrier();                                      // Emb, act, tok
            for (int i = tid; i < S * EC; i += CT_THREADS) {    // teacher forcing: every step's input is known
                const int t = i / EC, k = i - t * EC;
                X[i] = t > 0 ? Emb[(size_t)tok[t] * EC + k] : 0.0f;
            }
            lds_barrier();

            // ------------------------------------------------------------------------------------------ forward recurrence
            float c_prev = 0.0f;
            for (int t = 0; t < S; ++t) {
                if (tid < H4) {
                    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (t > 0) {
#pragma unroll
                        for (int i = 0; i < XSL; i += 4) {
                            const float4 a = *reinterpret_cast<const float4*>(X + t * EC + XSL * gsl + i);
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const float4 wv = *reinterpret_cast<const float4*>(Wi + (g * HC + gu) * (EC + PPO_WIP) + XSL * gsl + i);
                                acc[g] = fmaf(wv.x, a.x, acc[g]); acc[g] = fmaf(wv.y, a.y, acc[g]);
                                acc[g] = fmaf(wv.z, a.z, acc[g]); acc[g] = fmaf(wv.w, a.w, acc[g]);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < HL; i += 4) {
                            // slices 0..2 read 4 inputs of the next slice in their last round: weights zero there
                            const float4 a = *reinterpret_cast<const float4*>(Hs + t * HC + HSL * gsl + i);
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                acc[g] = fmaf(whr[g][i], a.x, acc[g]); acc[g] = fmaf(whr[g][i + 1], a.y, acc[g]);
                                acc[g] = fmaf(whr[g][i + 2], a.z, acc[g]); acc[g] = fmaf(whr[g][i + 3], a.w, acc[g]);
                            }
                        }
                    }
                    // sum of the four k-slices; lane sl activates gate sl, then the quad exchanges the four gates and every
                    // lane carries the cell state of hidden unit gu
                    float mine = 0.0f;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float v = quad_sum(acc[g]);
                        mine = gsl == g ? v : mine;
                    }
                    mine += bias;
                    const float av = gsl == 2 ? tanhf(mine) : sigmoidf_(mine);
                    G[t * H4 + gsl * HC + gu] = av;
                    const float ig = quad_lane<0>(av), fg = quad_lane<1>(av), gg = quad_lane<2>(av), og = quad_lane<3>(av);
                    const float c = fg * c_prev + ig * gg;
                    const float tc = tanhf(c);
                    c_prev = c;
                    if (gsl == 0) {
                        Cs[t * HC + gu] = c;
                        TC[t * HC + gu] = tc;
                        Hs[(t + 1) * HC + gu] = og * tc;
                    }
                }
                lds_barrier();
            }
            // the backward's weight slices: in flight while the heads run
            float wbh[JS][4];
            const uint32_t ohrow = (4u * ((otid >> 2) / (uint32_t)(HC / 4)) + osl) * (uint32_t)JS, ohcol = 4u * ((otid >> 2) % (uint32_t)(HC / 4));
            if (tid < CT_HP * HC) {
#pragma unroll
                for (int i = 0; i < JS; ++i) {
                    const float4 v = *reinterpret_cast<const float4*>(pw_hh + (ohrow + (uint32_t)i) * (uint32_t)HC + ohcol);
                    wbh[i][0] = v.x; wbh[i][1] = v.y; wbh[i][2] = v.z; wbh[i][3] = v.w;
                }
            }
            // ------------------------------------------------------------------------------------------ heads, all steps
            {                                                   // 4 lanes per (step, action) logit
                const int o = tid >> 2, prt = tid & 3;
                const int t = o / A, a = o - t * A;
                const bool op_step = (t & 1) == 0;
                const bool live = t < S && a < (op_step ? d.NOPS : d.NMAGS);
                float z = 0.0f;
                if (live) {
                    const float* wrow = Wh + (size_t)((op_step ? 0 : d.NOPS) + a) * HC;
                    const float* h = Hs + (t + 1) * HC;
#pragma unroll 5
                    for (int k = prt; k < HC; k += 4) z = fmaf(wrow[k], h[k], z);
                }
                z += __shfl_xor(z, 1, 64); z += __shfl_xor(z, 2, 64);
                if (live && prt == 0) {
                    z += Bh[(op_step ? 0 : A) + a];
                    const float tl = tanhf(z);
                    TL[t * A + a] = tl;
                    Pp[t * A + a] = d.cdiv * tl;
                }
            }
            lds_barrier();
            if (tid < S * A) {                                  // soft-max of every step: 16 lanes per step
                const int t = tid / A, a = tid - t * A;
                const int NA = (t & 1) == 0 ? d.NOPS : d.NMAGS;
                const bool live = a < NA;
                const float v = live ? Pp[t * A + a] : -INFINITY;
                float mx = v;
#pragma unroll
                for (int o = A / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, A));
                const float z = v - mx;
                float sum = live ? expf(z) : 0.0f;
#pragma unroll
                for (int o = A / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, A);
                const float lp = z - logf(sum);
                if (live) Pp[t * A + a] = expf(lp);
                const float lp_sel = __shfl(lp, act[t], A);
                if (a == 0) misc[8 + t] = lp_sel;
            }
            lds_barrier();
            // the policy's log-probability = sum over its Q sequences: every workgroup publishes {epoch tag, value} as ONE 64-bit word
            // and the first Q lanes poll the policy's slots -- the payload travels inside the atomic, no fence, no counter
            if (tid == 0) {
                float s = 0.0f;
                for (int t = 0; t < S; ++t) s += misc[8 + t];
                const unsigned long long word = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(s);
                __hip_atomic_store(lpq + w, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid < Q) {
                unsigned long long word;
                do {
                    word = __hip_atomic_load(lpq + m * Q + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while ((unsigned)(word >> 32) != tag);
                misc[8 + tid] = __uint_as_float((unsigned)word);
            }
            lds_barrier();
            if (tid == 0) {
                float lp = 0.0f;
                for (int q = 0; q < Q; ++q) lp += misc[8 + q];
                const float ratio = expf(lp - old_log_probs[m]);
                const float clipped = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
                const float r = reward[m];
                const float a = ratio * r, b = clipped * r;
                if (w == m * Q) loss_terms[m] = -fminf(a, b);      // the caller averages over M
                // d(-min(a, b))/d lp: through a when a <= b (ties: both halves reach `ratio` because clamp is then the identity)
                const bool inside = ratio >= 1.0f - clip && ratio <= 1.0f + clip;
                float g = 0.0f;
                if (a < b || (a == b && inside)) g = -r * ratio;
                else if (a == b && !inside) g = -0.5f * r * ratio;
                misc[3] = g / (float)d.M;
            }
            lds_barrier();
            const float gl = misc[3];

            // ------------------------------------------------------------------------------------------ backward
            if (tid < S * A) {                                  // d logits of every step
                const int t = tid / A, a = tid - t * A;
                const int NA = (t & 1) == 0 ? d.NOPS : d.NMAGS;
                float dl = 0.0f;
                if (a < NA) {
                    const float p = Pp[t * A + a], tl = TL[t * A + a];
                    const float dz = gl * ((a == act[t] ? 1.0f : 0.0f) - p);
                    dl = dz * d.cdiv * (1.0f - tl * tl);
                }
                DL[t * A + a] = dl;
            }
            lds_barrier();
            float dc_carry = 0.0f;                              // thread u < HC: d loss / d c_t carried to step t - 1
            for (int t = S - 1; t >= 0; --t) {
                const bool op_step = (t & 1) == 0;
                const int NA = op_step ? d.NOPS : d.NMAGS;
                if (tid < HC) {                                 // through the head and the cell
                    const int u = tid;
                    float dh = 0.0f;
                    if (t < S - 1)
                        for (int pr = 0; pr < CT_HP; ++pr) dh += part[pr * HC + u];
                    const float* wh = Wh + (size_t)(op_step ? 0 : d.NOPS) * HC;
                    for (int a = 0; a < NA; ++a) dh = fmaf(wh[(size_t)a * HC + u], DL[t * A + a], dh);
                    const float* g4 = G + t * H4;
                    const float ig = g4[u], fg = g4[HC + u], gg = g4[2 * HC + u], og = g4[3 * HC + u];
                    const float tc = TC[t * HC + u];
                    const float cp = t > 0 ? Cs[(t - 1) * HC + u] : 0.0f;
                    const float d_o = dh * tc;
                    const float dc = dh * og * (1.0f - tc * tc) + dc_carry;
                    dc_carry = dc * fg;
                    float* dg = DG + t * H4;
                    dg[u] = dc * gg * ig * (1.0f - ig);
                    dg[HC + u] = dc * cp * fg * (1.0f - fg);
                    dg[2 * HC + u] = dc * ig * (1.0f - gg * gg);
                    dg[3 * HC + u] = d_o * og * (1.0f - og);
                }
                lds_barrier();
                if (t > 0) {
                    // dh_{t-1} = W_hh^T dgates: lane (jg, kg, jl) owns columns 4 kg .. 4 kg + 3 x gate rows [(4 jg + jl) JS, + JS)
                    if (tid < CT_HP * HC) {
                        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int i = 0; i < JS; i += 4) {
                            const float4 a = *reinterpret_cast<const float4*>(DG + t * H4 + (4 * hjg + gsl) * JS + i);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                acc[c] = fmaf(wbh[i][c], a.x, acc[c]); acc[c] = fmaf(wbh[i + 1][c], a.y, acc[c]);
                                acc[c] = fmaf(wbh[i + 2][c], a.z, acc[c]); acc[c] = fmaf(wbh[i + 3][c], a.w, acc[c]);
                            }
                        }
                        float mine = 0.0f;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float v = quad_sum(acc[c]);
                            mine = gsl == c ? v : mine;
                        }
                        part[hjg * HC + 4 * hkg + gsl] = mine;
                    }
                    lds_barrier();
                }
            }
            // dx_t = W_ih^T dgates of every step t >= 1 (gradient of the embedding row that fed the step): three steps at a time
            if (tid < 3 * XT) {
                for (int t = 1 + xgrp; t < S; t += 3) {
                    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int i = 0; i < JS; i += 4) {
                        const float4 a = *reinterpret_cast<const float4*>(DG + t * H4 + (4 * xjg + gsl) * JS + i);
                        float wv[4][4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float4 v = *reinterpret_cast<const float4*>(Wi + ((4 * xjg + gsl) * JS + i + r) * (EC + PPO_WIP) + 4 * xkg);
                            wv[r][0] = v.x; wv[r][1] = v.y; wv[r][2] = v.z; wv[r][3] = v.w;
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            acc[c] = fmaf(wv[0][c], a.x, acc[c]); acc[c] = fmaf(wv[1][c], a.y, acc[c]);
                            acc[c] = fmaf(wv[2][c], a.z, acc[c]); acc[c] = fmaf(wv[3][c], a.w, acc[c]);
                        }
                    }
                    float mine = 0.0f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float v = quad_sum(acc[c]);
                        mine = gsl == c ? v : mine;
                    }
                    partx[(t * CT_HP + xjg) * EC + 4 * xkg + gsl] = mine;
                }
            }
            lds_barrier();
            for (int i = tid; i < S * EC; i += CT_THREADS) {
                const int t = i / EC, k = i - t * EC;
                float s = 0.0f;
                if (t > 0)
                    for (int pr = 0; pr < CT_HP; ++pr) s += partx[(t * CT_HP + pr) * EC + k];
                DX[i] = s;
            }
            lds_barrier();

            // ------------------------------------------------------------------------------------------ the sequence's factors
            // rows r = w S + t of the scratch arrays: d gates, d logits, d inputs, inputs, tokens; hidden states as [w][S + 1][H]
            // (slot t = h_{t-1}, slot t + 1 = h_t): 11 KB per sequence instead of its 225 KB of parameter gradients
            {
                const size_t r0 = (size_t)w * S;
                float4* gdg = reinterpret_cast<float4*>(ws + W.dg + r0 * H4);
                for (int i = tid; i < S * (H4 / 4); i += CT_THREADS) gdg[i] = reinterpret_cast<const float4*>(DG)[i];
                float4* ghs = reinterpret_cast<float4*>(ws + W.hprev + (size_t)w * (S + 1) * HC);
                for (int i = tid; i < (S + 1) * (HC / 4); i += CT_THREADS) ghs[i] = reinterpret_cast<const float4*>(Hs)[i];
                if (tid < S * (EC / 4)) {
                    reinterpret_cast<float4*>(ws + W.xin + r0 * EC)[tid] = reinterpret_cast<const float4*>(X)[tid];
                    reinterpret_cast<float4*>(ws + W.dx + r0 * EC)[tid] = reinterpret_cast<const float4*>(DX)[tid];
                }
                if (tid < S * (A / 4)) reinterpret_cast<float4*>(ws + W.dl + r0 * A)[tid] = reinterpret_cast<const float4*>(DL)[tid];
                if (tid < S) reinterpret_cast<int*>(ws + W.tok)[r0 + tid] = tok[tid];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_ctrl_sample_seq (round 5): Controller.sample with one workgroup per SEQUENCE, as k_ppo_rollout's forward half -- the Q
// sub-policies of a policy row are independent sequences (state reset per sub-policy, models/controller.py:82-84), each draws
// from its own uniforms.  Per step: gate products + cell update (one barrier), the step's head logits (4 lanes per action),
// soft-max + inverse-CDF draw + next input on 16 lanes.  The per-policy sums (log-prob, entropy) and the mean head
// probabilities cross workgroups as per-sequence partials that the last workgroup to arrive combines in ascending order.
// 43 -> 17 us for M = 6, Q = 5, S = 4.  Widths 32 / 100 only; anything else takes k_ctrl_rollout<true>.
// ------------------------------------------------------------------------------------------------
template <int EC, int HC>
__global__ __launch_bounds__(CT_THREADS) void k_ctrl_sample_seq(CtrlParams P, CtrlDims d, float* ws, const float* __restrict__ uniforms,
                                                                long long* __restrict__ policies, float* __restrict__ op_probs,
                                                                float* __restrict__ mag_probs, float* __restrict__ log_probs,
                                                                float* __restrict__ entropies) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    const CtrlWs W = ctrl_ws(d);
    const PpoLds O = ppo_lds(d);
    const int w = blockIdx.x, tid = threadIdx.x;
    const int Q = d.Q, S = d.S, nseq = d.M * d.Q;
    constexpr int H4 = 4 * HC, A = CT_MAX_A;
    const int NT = d.NOPS + d.NMAGS;
    float* Hs = L + O.Hs; float* X = L + O.X; float* Pp = L + O.P; float* Wi = L + O.Wi; float* misc = L + O.misc;
    float* Wh = L + O.Wh; float* Bh = L + O.Bh; float* Emb = L + O.Emb;
    constexpr int XSL = EC / 4, HSL = (HC / 16) * 4, HL = HC - 3 * HSL;
    const int gu = tid >> 2, gsl = tid & 3;
    // ---- every global load in one batch: head weights, embedding rows, W_ih (-> LDS), this lane's W_hh slices and bias
    float4 r_op = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r_mag = r_op, r_emb = r_op;
    float r_bop = 0.0f, r_bmag = 0.0f;
    if (tid < d.NOPS * (HC / 4)) r_op = reinterpret_cast<const float4*>(P.wop)[tid];
    if (tid < d.NMAGS * (HC / 4)) r_mag = reinterpret_cast<const float4*>(P.wmag)[tid];
    if (tid < NT * (EC / 4)) r_emb = reinterpret_cast<const float4*>(P.emb)[tid];
    if (tid < d.NOPS) r_bop = P.bop[tid];
    if (tid < d.NMAGS) r_bmag = P.bmag[tid];
    constexpr int WIV = H4 * EC / 4, WIN = (WIV + CT_THREADS - 1) / CT_THREADS;
    float4 r_wi[WIN];
#pragma unroll
    for (int i = 0; i < WIN; ++i) r_wi[i] = reinterpret_cast<const float4*>(P.w_ih)[min(tid + i * CT_THREADS, WIV - 1)];
    float whr[4][HL];
    float bias = 0.0f;
    if (tid < H4) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* rh = P.w_hh + (size_t)(g * HC + gu) * HC + HSL * gsl;
#pragma unroll
            for (int i = 0; i < HL; i += 4) {
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (i < HSL || gsl == 3) v = *reinterpret_cast<const float4*>(rh + i);
                whr[g][i] = v.x; whr[g][i + 1] = v.y; whr[g][i + 2] = v.z; whr[g][i + 3] = v.w;
            }
        }
        bias = P.b_ih[gsl * HC + gu] + P.b_hh[gsl * HC + gu];
    }
    if (tid < d.NOPS * (HC / 4)) reinterpret_cast<float4*>(Wh)[tid] = r_op;
    if (tid < d.NMAGS * (HC / 4)) reinterpret_cast<float4*>(Wh + d.NOPS * HC)[tid] = r_mag;
    if (tid < NT * (EC / 4)) reinterpret_cast<float4*>(Emb)[tid] = r_emb;
    if (tid < A) { Bh[tid] = r_bop; Bh[A + tid] = r_bmag; }
#pragma unroll
    for (int i = 0; i < WIN; ++i) {
        const int v = tid + i * CT_THREADS;
        if (v < WIV) *reinterpret_cast<float4*>(Wi + (v / (EC / 4)) * (EC + PPO_WIP) + 4 * (v % (EC / 4))) = r_wi[i];
    }
    for (int i = tid; i < HC; i += CT_THREADS) Hs[i] = 0.0f;
    for (int i = tid; i < EC; i += CT_THREADS) X[i] = 0.0f;
    lds_barrier();

    float c_prev = 0.0f, my_lp = 0.0f, my_ent = 0.0f;
    for (int t = 0; t < S; ++t) {
        const bool op_step = (t & 1) == 0;
        const int NA = op_step ? d.NOPS : d.NMAGS;
        if (tid < H4) {                                     // gate products + cell update, as k_ppo_rollout
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (t > 0) {
#pragma unroll
                for (int i = 0; i < XSL; i += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(X + t * EC + XSL * gsl + i);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 wv = *reinterpret_cast<const float4*>(Wi + (g * HC + gu) * (EC + PPO_WIP) + XSL * gsl + i);
                        acc[g] = fmaf(wv.x, a.x, acc[g]); acc[g] = fmaf(wv.y, a.y, acc[g]);
                        acc[g] = fmaf(wv.z, a.z, acc[g]); acc[g] = fmaf(wv.w, a.w, acc[g]);
                    }
                }
#pragma unroll
                for (int i = 0; i < HL; i += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(Hs + t * HC + HSL * gsl + i);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[g] = fmaf(whr[g][i], a.x, acc[g]); acc[g] = fmaf(whr[g][i + 1], a.y, acc[g]);
                        acc[g] = fmaf(whr[g][i + 2], a.z, acc[g]); acc[g] = fmaf(whr[g][i + 3], a.w, acc[g]);
                    }
                }
            }
            float mine = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v = quad_sum(acc[g]);
                mine = gsl == g ? v : mine;
            }
            mine += bias;
            const float av = gsl == 2 ? tanhf(mine) : sigmoidf_(mine);
            const float ig = quad_lane<0>(av), fg = quad_lane<1>(av), gg = quad_lane<2>(av), og = quad_lane<3>(av);
            const float c = fg * c_prev + ig * gg;
            c_prev = c;
            if (gsl == 0) Hs[(t + 1) * HC + gu] = og * tanhf(c);
        }
        lds_barrier();
        {                                                   // head logits of the step: 4 lanes per action
            const int a = tid >> 2, prt = tid & 3;
            const bool live = a < NA;
            float z = 0.0f;
            if (live) {
                const float* wrow = Wh + (size_t)((op_step ? 0 : d.NOPS) + a) * HC;
                const float* h = Hs + (t + 1) * HC;
#pragma unroll 5
                for (int k = prt; k < HC; k += 4) z = fmaf(wrow[k], h[k], z);
            }
            z += __shfl_xor(z, 1, 64); z += __shfl_xor(z, 2, 64);
            if (live && prt == 0) Pp[t * A + a] = d.cdiv * tanhf(z + Bh[(op_step ? 0 : A) + a]);
        }
        lds_barrier();
        if (tid < A) {                                      // soft-max, draw, next input token: 16 lanes
            const int a = tid;
            float* p = Pp + t * A;
            const bool live = a < NA;
            const float v = live ? p[a] : -INFINITY;
            float mx = v;
#pragma unroll
            for (int o = A / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, A));
            const float z = v - mx;
            float sum = live ? expf(z) : 0.0f;
#pragma unroll
            for (int o = A / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, A);
            const float lp = z - logf(sum), pr = live ? expf(lp) : 0.0f;
            float ent = live ? -lp * pr : 0.0f;
#pragma unroll
            for (int o = A / 2; o > 0; o >>= 1) ent += __shfl_xor(ent, o, A);
            if (live) p[a] = pr;
            // inverse CDF in action order (a sequential sum, as a host-side scan of the same probabilities would do it)
            const float u = uniforms[(size_t)w * S + t];
            float cum = 0.0f;
            int a_sel = NA - 1;
            bool found = false;
#pragma unroll
            for (int b = 0; b < A; ++b) {                   // (lanes 0..15 of wave 0: lane b's value through a scalar register)
                cum += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(pr), b));
                if (!found && b < NA && u < cum) { a_sel = b; found = true; }
            }
            if (a == 0) policies[(size_t)w * S + t] = a_sel;
            my_lp += __shfl(lp, a_sel, A);
            my_ent += ent;
            if (t + 1 < S) {
                const int tok = a_sel + (op_step ? 0 : d.NOPS);
                for (int k = a; k < EC; k += A) X[(t + 1) * EC + k] = Emb[(size_t)tok * EC + k];
            }
        }
        lds_barrier();
    }
    // ---- per-sequence partials; the last workgroup to arrive combines them in ascending order.  No device-scope fence (cache
    // maintenance on every XCD, several us): the partials are written and read with relaxed device-scope atomics (sc1: served at the
    // device's coherence point); every storing wave waits for its stores' acknowledgements (an explicit s_waitcnt vmcnt(0), as k_seg_loss
    // does: a workgroup barrier alone does not promise it -- ADVICE r5) before the barrier behind which the arrival is counted.
    auto st_dev = [](float* p, float v) { __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto ld_dev = [](const float* p) {
        return __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned*>(const_cast<float*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    float* sp = ws + W.sprobs + (size_t)w * 2 * A;
    if (tid < 2 * A) {
        const int head = tid / A, a = tid - head * A;
        float s = 0.0f;
        for (int t = head; t < S; t += 2) s += Pp[t * A + a];
        st_dev(sp + tid, (a < (head == 0 ? d.NOPS : d.NMAGS)) ? s : 0.0f);
    }
    if (tid == 0) { st_dev(ws + W.slp + 2 * w, my_lp); st_dev(ws + W.slp + 2 * w + 1, my_ent); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* counter = reinterpret_cast<int*>(ws + W.counter);
    __shared__ int last;
    if (tid == 0) last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nseq - 1;
    __syncthreads();
    if (last) {
        if (tid < 2 * A) {
            const int head = tid / A, a = tid - head * A;
            float s = 0.0f;
            for (int i = 0; i < nseq; ++i) s += ld_dev(ws + W.sprobs + (size_t)i * 2 * A + tid);
            const float inv = 1.0f / (float)(nseq * (S / 2));
            if (head == 0 && a < d.NOPS) op_probs[a] = s * inv;
            if (head == 1 && a < d.NMAGS) mag_probs[a] = s * inv;
        }
        if (tid >= 64 && tid < 64 + d.M) {
            const int mm = tid - 64;
            float a = 0.0f, b = 0.0f;
            for (int q = 0; q < Q; ++q) { a += ld_dev(ws + W.slp + 2 * (mm * Q + q)); b += ld_dev(ws + W.slp + 2 * (mm * Q + q) + 1); }
            log_probs[mm] = a; entropies[mm] = b;
        }
        if (tid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

inline bool ctrl_ok(const CtrlDims& d) {
    return d.M > 0 && d.Q > 0 && d.Q <= CT_MAX_Q && d.S > 0 && (d.S % 2) == 0 && d.E > 0 && d.H > 0 && 4 * d.H <= CT_THREADS &&
           d.NOPS > 0 && d.NOPS <= CT_MAX_A && d.NMAGS > 0 && d.NMAGS <= CT_MAX_A && d.Q * CT_MAX_A * 4 <= CT_THREADS &&
           (4 * d.H / CT_HP) % 4 == 0 && (4 * d.H) % CT_HP == 0 && (4 * d.H / CT_XP) % 4 == 0 && (4 * d.H) % CT_XP == 0 &&
           ctrl_lds(d).total * sizeof(float) <= 160 * 1024 - 256;
}
// the widths of the reference's Controller (models/controller.py:10): the register-resident-weights instantiation
constexpr int CT_E = 32, CT_H = 100;
inline bool ctrl_fast(const CtrlDims& d) { return d.E == CT_E && d.H == CT_H; }
inline CtrlParams as_params(void* const* p) {
    CtrlParams P;
    P.emb = (float*)p[0]; P.w_ih = (float*)p[1]; P.w_hh = (float*)p[2]; P.b_ih = (float*)p[3]; P.b_hh = (float*)p[4];
    P.wop = (float*)p[5]; P.bop = (float*)p[6]; P.wmag = (float*)p[7]; P.bmag = (float*)p[8];
    return P;
}
template <bool SAMPLE, int EC, int HC>
int set_lds(size_t bytes) {
    static size_t current = 0;
    if (bytes > current) {
        AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctrl_rollout<SAMPLE, EC, HC>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        current = bytes;
    }
    return 0;
}

}  // namespace

extern "C" int aadg_controller_supported(int M, int Q, int S, int E, int H, int n_ops, int n_mags) {
    CtrlDims d = {M, Q, S, E, H, n_ops, n_mags, 1.0f};
    return ctrl_ok(d) ? 1 : 0;
}

extern "C" size_t aadg_controller_workspace_bytes(int M, int Q, int S, int E, int H, int n_ops, int n_mags) {
    CtrlDims d = {M, Q, S, E, H, n_ops, n_mags, 1.0f};
    return ctrl_ok(d) ? ctrl_ws(d).total * sizeof(float) : 0;
}

extern "C" int aadg_controller_sample_f32(void* const* params, int M, int Q, int S, int E, int H, int n_ops, int n_mags, float c_over_t,
                                          const float* uniforms, long long* policies, float* op_probs, float* mag_probs,
                                          float* log_probs, float* entropies, void* ws, size_t ws_bytes, void* stream) {
    CtrlDims d = {M, Q, S, E, H, n_ops, n_mags, c_over_t};
    if (params == nullptr || uniforms == nullptr || policies == nullptr || op_probs == nullptr || mag_probs == nullptr ||
        log_probs == nullptr || entropies == nullptr || ws == nullptr)
        return AADG_E_BADARG;
    if (!ctrl_ok(d)) return AADG_E_UNSUPPORTED;
    if (ws_bytes < ctrl_ws(d).total * sizeof(float)) return AADG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const CtrlParams P = as_params(params);
    const size_t lds = ctrl_lds(d).total * sizeof(float);
    const bool fast = ctrl_fast(d);
    if (fast && M * Q <= 128 && M <= CT_THREADS - 64 && ppo_lds(d).total * sizeof(float) <= 150 * 1024) {      // one workgroup per sequence
        const size_t lds_s = ppo_lds(d).total * sizeof(float);
        static size_t set_s = 0;
        if (lds_s > set_s) {
            AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctrl_sample_seq<CT_E, CT_H>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
            set_s = lds_s;
        }
        hipLaunchKernelGGL((k_ctrl_sample_seq<CT_E, CT_H>), dim3(M * Q), dim3(CT_THREADS), lds_s, st, P, d, (float*)ws, uniforms, policies, op_probs,
                           mag_probs, log_probs, entropies);
        AADG_LAUNCH_CHECK();
        return 0;
    }
    const int rc = fast ? set_lds<true, CT_E, CT_H>(lds) : set_lds<true, 0, 0>(lds);
    if (rc) return rc;
    if (!fast) {                                           // the compile-time-width kernel reads the parameters as they lie
        hipLaunchKernelGGL(k_ctrl_transpose, dim3(64), dim3(256), 0, st, P, d, (float*)ws);
        AADG_LAUNCH_CHECK();
    }
    auto* kern = fast ? &k_ctrl_rollout<true, CT_E, CT_H> : &k_ctrl_rollout<true, 0, 0>;
    hipLaunchKernelGGL(kern, dim3(M), dim3(CT_THREADS), lds, st, P, d, (float*)ws, uniforms, policies, op_probs,
                       mag_probs, log_probs, entropies, (const float*)nullptr, (const float*)nullptr, 0.0f, (float*)nullptr);
    AADG_LAUNCH_CHECK();
    return 0;
}

extern "C" int aadg_controller_ppo_update_f32(void* const* params, void* const* exp_avg, void* const* exp_avg_sq, int M, int Q, int S,
                                              int E, int H, int n_ops, int n_mags, float c_over_t, const long long* policies,
                                              const float* old_log_probs, const float* reward, float clip, int n_updates,
                                              int step0, float lr, float beta1, float beta2, float eps, float* loss_terms,
                                              void* ws, size_t ws_bytes, void* stream) {
    CtrlDims d = {M, Q, S, E, H, n_ops, n_mags, c_over_t};
    if (params == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr || policies == nullptr || old_log_probs == nullptr ||
        reward == nullptr || loss_terms == nullptr || ws == nullptr || n_updates <= 0 || step0 < 0)
        return AADG_E_BADARG;
    if (!ctrl_ok(d)) return AADG_E_UNSUPPORTED;
    if (ws_bytes < ctrl_ws(d).total * sizeof(float)) return AADG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const CtrlParams P = as_params(params);
    CtrlPtrs9 m1, m2;
    for (int i = 0; i < 9; ++i) { m1.p[i] = (float*)exp_avg[i]; m2.p[i] = (float*)exp_avg_sq[i]; }
    const size_t lds = ctrl_lds(d).total * sizeof(float);
    const bool fast = ctrl_fast(d);
    const int rc = fast ? set_lds<false, CT_E, CT_H>(lds) : set_lds<false, 0, 0>(lds);
    if (rc) return rc;
    auto* kern = fast ? &k_ctrl_rollout<false, CT_E, CT_H> : &k_ctrl_rollout<false, 0, 0>;
    const int n_params = ctrl_n_params(d);
    {
        // the module's widths, short sequences: two short kernels per epoch, one workgroup per sequence
        const int grid = M * Q;
        const size_t lds_r = ppo_lds(d).total * sizeof(float), lds_g = ppo_ga_lds(d).total * sizeof(float);
        if (fast && S <= PPO_MAX_S && grid <= 128 && lds_r <= 150 * 1024 && lds_g <= 150 * 1024 && ppo_ga_fits(d) &&
            (4 * H + PPO_GA_GRID - 1) / PPO_GA_GRID <= PPO_JP && n_ops + n_mags <= PPO_GA_GRID) {
            static size_t set_r = 0, set_g = 0;
            if (lds_r > set_r) {
                AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ppo_rollout<CT_E, CT_H>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
                set_r = lds_r;
            }
            if (lds_g > set_g) {
                AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ppo_grad_adam<CT_E, CT_H>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g));
                set_g = lds_g;
            }
            for (int it = 0; it < n_updates; ++it) {
                const double step = (double)(step0 + it + 1);
                const float bc1 = (float)(1.0 - pow((double)beta1, step));
                const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
                hipLaunchKernelGGL((k_ppo_rollout<CT_E, CT_H>), dim3(grid), dim3(CT_THREADS), lds_r, st, P, d, (float*)ws, policies, old_log_probs,
                                   reward, clip, (unsigned)(step0 + it + 1), loss_terms + (size_t)it * M);
                AADG_LAUNCH_CHECK();
                hipLaunchKernelGGL((k_ppo_grad_adam<CT_E, CT_H>), dim3(PPO_GA_GRID), dim3(CT_THREADS), lds_g, st, P, m1, m2, d, (float*)ws, bc1, bc2_sqrt,
                                   lr, beta1, beta2, eps);
                AADG_LAUNCH_CHECK();
            }
            return 0;
        }
    }
    if (!fast) {
        hipLaunchKernelGGL(k_ctrl_transpose, dim3(64), dim3(256), 0, st, P, d, (float*)ws);
        AADG_LAUNCH_CHECK();
    }
    for (int it = 0; it < n_updates; ++it) {
        hipLaunchKernelGGL(kern, dim3(M), dim3(CT_THREADS), lds, st, P, d, (float*)ws, (const float*)nullptr,
                           const_cast<long long*>(policies), (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                           old_log_probs, reward, clip, loss_terms + (size_t)it * M);
        AADG_LAUNCH_CHECK();
        const double step = (double)(step0 + it + 1);
        const float bc1 = (float)(1.0 - pow((double)beta1, step));
        const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
        hipLaunchKernelGGL(k_ctrl_adam, dim3((n_params + 255) / 256), dim3(256), 0, st, P, m1, m2, d, (float*)ws, lr, beta1, beta2, eps,
                           bc1, bc2_sqrt);
        AADG_LAUNCH_CHECK();
    }
    return 0;
}

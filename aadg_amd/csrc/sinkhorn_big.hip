// Sinkhorn divergence for point clouds that do not fit the LDS-resident kernel of sinkhorn.hip (more than ~100
// points per cloud; the reference's own size is 8).  Same algorithm (geomloss 0.2.4 sinkhorn_loop, debiased,
// cosine cost, p = 2); the difference is where the cost matrices live:
//
//   k_big_prep    rows normalised to unit length (so C = 1 - <x^, y^>), per-chunk column min / max
//   k_big_schedule diameter of x u y, eps schedule (float64)
//   k_big_cost    the cost matrices C_xx, C_yy, C_xy as dense 64x64-tiled contractions on the matrix cores:
//                 v_mfma_f32_32x32x2_f32 -- exact f32 (an fmaf chain), operands staged in LDS with an odd row stride; this
//                 is the one place on the path where a dense feature GEMM is the bottleneck (N = 4096, E = 128: 3 x 4.3
//                 GFLOP), so it is the one place MFMA is used.  C_yx = C_xy^T is not a fourth contraction (round 5): the
//                 C_xy workgroups write their tile a second time, transposed through LDS (coalesced rows of C_yx)
//   k_big_sweep   one eps-scaling step: every row of the four matrices is an independent log-sum-exp,
//                 one wavefront per row, online (max, sum) in registers, wave-shuffle merge; potentials are
//                 double-buffered so the symmetrised update needs no second pass.  HBM-bound: 4 * N^2 * 4 bytes per
//                 step.  Round 5: the per-COLUMN term h_j = log w_j + pot_j / eps (a true division per matrix element
//                 before) is written once per step by the wave that produces pot_j, pre-scaled by log2(e) so that an
//                 element costs one FMA + one v_exp_f32; eight streaming float4 loads per lane are in flight before the
//                 first is consumed, and four elements share one running-maximum update (5 exps per 4 elements,
//                 branch-free).
//   k_big_final   <alpha, b_x - a_x> + <beta, a_y - b_y>
//
// The number of eps steps depends on the data (diameter), which lives on the device: the host enqueues a fixed
// number of sweep launches (MAX_ITS + 2) and the surplus ones exit immediately -- no host synchronisation.
#include "common.h"

namespace {

constexpr int BIG_EPS_CAP = 32;     // eps entries: 2 + ceil(log2(diameter / blur)) <= 32  <=>  diameter/blur < 2^30
constexpr int BIG_PREP_CHUNKS = 32;        // row chunks of the prep pass (one workgroup each)
constexpr int BIG_MAX_ITS = BIG_EPS_CAP;   // sweeps enqueued: init + up to BIG_EPS_CAP eps steps + final extrapolation

struct BigLayout {                  // per-problem float offsets into the workspace
    size_t xn, yn, cxx, cyy, cxy, cyx, pot, hh, til, meta, scr, total;
};
__host__ __device__ inline BigLayout big_layout(int nmax, int E) {
    BigLayout l;
    size_t o = 0;
    const size_t mat = (size_t)nmax * nmax;
    l.xn = o; o += (size_t)nmax * E;
    l.yn = o; o += (size_t)nmax * E;
    l.cxx = o; o += mat;
    l.cyy = o; o += mat;
    l.cxy = o; o += mat;
    l.cyx = o; o += mat;
    l.pot = o; o += (size_t)2 * 4 * nmax;     // [buffer][a_x, b_y, a_y, b_x][nmax]
    l.hh = o; o += (size_t)2 * 4 * nmax;      // [buffer][..][nmax]: (log w + pot / eps_next) * log2(e), what the NEXT step's softmins add per column
    l.til = o; o += (size_t)4 * nmax;
    l.meta = o; o += 2 * BIG_EPS_CAP + 8;      // eps_s as doubles (2 floats each) + nits
    l.scr = o; o += (size_t)BIG_PREP_CHUNKS * 2 * 512;   // per-chunk column min / max (E <= 512)
    l.total = (o + 63) / 64 * 64;
    return l;
}

struct BigProb {
    const int* rows_x; const int* rows_y; int n, m;
};
__device__ __forceinline__ BigProb big_problem(const int* cloud_rows, const int* cloud_off, const int* prob_xy, int p) {
    const int cx = prob_xy[2 * p], cy = prob_xy[2 * p + 1];
    BigProb r;
    r.rows_x = cloud_rows + cloud_off[cx]; r.n = cloud_off[cx + 1] - cloud_off[cx];
    r.rows_y = cloud_rows + cloud_off[cy]; r.m = cloud_off[cy + 1] - cloud_off[cy];
    return r;
}

// ---- prep: grid (BIG_PREP_CHUNKS, n_prob), 256 threads: wave <-> rows of this chunk, lanes <-> feature columns ------
constexpr int PREP_KPL = 8;   // E <= 64 * PREP_KPL = 512

__global__ __launch_bounds__(256) void k_big_prep(const float* __restrict__ feat, int ld, int E, const int* cloud_rows,
                                                  const int* cloud_off, const int* prob_xy, int nmax, float* ws) {
    const int p = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const BigProb pr = big_problem(cloud_rows, cloud_off, prob_xy, p);
    const BigLayout L = big_layout(nmax, E);
    float* base = ws + (size_t)p * L.total;
    // one pass over the rows: unit-length copy (so that C = 1 - <x^, y^>) and per-column min / max of the raw values
    float lo[PREP_KPL], hi[PREP_KPL];
#pragma unroll
    for (int q = 0; q < PREP_KPL; ++q) { lo[q] = INFINITY; hi[q] = -INFINITY; }
    const int total = pr.n + pr.m, per = (total + BIG_PREP_CHUNKS - 1) / BIG_PREP_CHUNKS;
    const int r1 = min(total, (chunk + 1) * per);
    for (int r = chunk * per + wv; r < r1; r += 4) {
        const bool isy = r >= pr.n;
        const int rr = isy ? r - pr.n : r;
        const float* src = feat + (size_t)(isy ? pr.rows_y[rr] : pr.rows_x[rr]) * ld;
        float v[PREP_KPL];
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < PREP_KPL; ++q) {
            const int k = lane + 64 * q;
            v[q] = k < E ? src[k] : 0.0f;
            ss = fmaf(v[q], v[q], ss);
            if (k < E) { lo[q] = fminf(lo[q], v[q]); hi[q] = fmaxf(hi[q], v[q]); }
        }
        ss = wave_sum(ss);
        const float inv = 1.0f / sqrtf(ss);
        float* dst = base + (isy ? L.yn : L.xn) + (size_t)rr * E;
#pragma unroll
        for (int q = 0; q < PREP_KPL; ++q) {
            const int k = lane + 64 * q;
            if (k < E) dst[k] = v[q] * inv;
        }
    }
    __shared__ float s_lo[4][64 * PREP_KPL], s_hi[4][64 * PREP_KPL];
#pragma unroll
    for (int q = 0; q < PREP_KPL; ++q) { s_lo[wv][lane + 64 * q] = lo[q]; s_hi[wv][lane + 64 * q] = hi[q]; }
    __syncthreads();
    float* scr = base + L.scr + (size_t)chunk * 2 * 512;
    for (int k = tid; k < E; k += 256) {
        scr[k] = fminf(fminf(s_lo[0][k], s_lo[1][k]), fminf(s_lo[2][k], s_lo[3][k]));
        scr[512 + k] = fmaxf(fmaxf(s_hi[0][k], s_hi[1][k]), fmaxf(s_hi[2][k], s_hi[3][k]));
    }
}

// diameter of x u y and the eps schedule: grid n_prob, 256 threads
__global__ __launch_bounds__(256) void k_big_schedule(int nmax, int E, float blur, float scaling, float* ws) {
    const int p = blockIdx.x, tid = threadIdx.x;
    const BigLayout L = big_layout(nmax, E);
    float* base = ws + (size_t)p * L.total;
    __shared__ float red[4];
    float part = 0.f;
    for (int k = tid; k < E; k += 256) {
        float a = INFINITY, b = -INFINITY;
        for (int c = 0; c < BIG_PREP_CHUNKS; ++c) {
            a = fminf(a, base[L.scr + (size_t)c * 1024 + k]);
            b = fmaxf(b, base[L.scr + (size_t)c * 1024 + 512 + k]);
        }
        part += (b - a) * (b - a);
    }
    part = wave_sum(part);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        const double diameter = (double)sqrtf(red[0] + red[1] + red[2] + red[3]);
        double* eps_s = reinterpret_cast<double*>(base + L.meta);
        int c = 0;
        eps_s[c++] = diameter * diameter;
        const double start = 2.0 * log(diameter), stop = 2.0 * log((double)blur), step = 2.0 * log((double)scaling);
        int len = (int)ceil((stop - start) / step);
        if (len < 0) len = 0;
        for (int i = 0; i < len && c < BIG_EPS_CAP - 1; ++i) eps_s[c++] = exp(start + (double)i * step);
        eps_s[c++] = (double)blur * (double)blur;
        reinterpret_cast<int*>(base + L.meta + 2 * BIG_EPS_CAP)[0] = c;
    }
}

// ---- cost matrices on the matrix cores: grid (ceil(nmax/64), ceil(nmax/64), n_prob*4), 256 threads ---------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CT = 64;          // block tile (4 waves, 32x32 each)
constexpr int CK = 64;          // K chunk staged in LDS (2 x 64 x 65 floats = 33 KB)

__global__ __launch_bounds__(256) void k_big_cost(const int* cloud_off, const int* prob_xy, int nmax, int E, float* ws) {
    const int p = blockIdx.z / 3, which = blockIdx.z - 3 * p;         // 0 xx, 1 yy, 2 xy (+ yx = its transpose)
    const BigLayout L = big_layout(nmax, E);
    float* base = ws + (size_t)p * L.total;
    const int cx = prob_xy[2 * p], cy = prob_xy[2 * p + 1];
    const int n = cloud_off[cx + 1] - cloud_off[cx], m = cloud_off[cy + 1] - cloud_off[cy];
    const float* Am = base + ((which == 0 || which == 2) ? L.xn : L.yn);
    const float* Bm = base + (which == 0 ? L.xn : L.yn);
    const int rows = (which == 0 || which == 2) ? n : m, cols = which == 0 ? n : m;
    float* C = base + (which == 0 ? L.cxx : which == 1 ? L.cyy : L.cxy);
    const int i0 = blockIdx.y * CT, j0 = blockIdx.x * CT;
    if (i0 >= rows || j0 >= cols) return;
    __shared__ float As[CT][CK + 1];
    __shared__ float Bs[CT][CK + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wi = (wv >> 1) * 32, wj = (wv & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int k0 = 0; k0 < E; k0 += CK) {
        const int kc = min(CK, E - k0);
        for (int idx = tid; idx < CT * CK; idx += 256) {
            const int r = idx / CK, k = idx - r * CK;
            As[r][k] = (i0 + r < rows && k < kc) ? Am[(size_t)(i0 + r) * E + k0 + k] : 0.0f;
            Bs[r][k] = (j0 + r < cols && k < kc) ? Bm[(size_t)(j0 + r) * E + k0 + k] : 0.0f;
        }
        __syncthreads();
        const int kk = (kc + 1) & ~1;                                  // zero-padded to even
        const float* ap = &As[wi + (lane & 31)][lane >> 5];
        const float* bp = &Bs[wj + (lane & 31)][lane >> 5];
        for (int k = 0; k < kk; k += 2)                               // A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k], acc, 0, 0, 0);
        __syncthreads();
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = i0 + wi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = j0 + wj + (lane & 31);
        if (row < rows && col < cols) C[(size_t)row * nmax + col] = 1.0f - acc[r];
    }
    if (which == 2) {
        // C_yx[j][i] = C_xy[i][j] (the same fmaf chain: bit-identical to a contraction of its own).  Through LDS, so that 32 lanes
        // write 128 contiguous bytes of a C_yx row: the wave's 32 x 32 tile goes to its slice of the (now free) operand buffers.
        float* T = (wv < 2 ? &As[0][0] : &Bs[0][0]) + (wv & 1) * 32 * 33;
#pragma unroll
        for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 33 + (lane & 31)] = 1.0f - acc[r];
        __builtin_amdgcn_wave_barrier();
        float* Ct = base + L.cyx;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int c = 2 * t + (lane >> 5), rr = lane & 31;            // element (row rr, column c) of the tile -> C_yx[j0 + wj + c][i0 + wi + rr]
            const int row = i0 + wi + rr, col = j0 + wj + c;
            if (row < rows && col < cols) Ct[(size_t)col * nmax + row] = T[rr * 33 + c];
        }
    }
}

// ---- the same three matrices through the bfloat16 matrix cores at float32 precision ("f32x3", csrc/common.h: aadg_split4): every
// operand value is split into bfloat16 halves x = hi + lo and every product formed as hi*hi + hi*lo + lo*hi with float32 accumulation:
// a third of the bfloat16 MFMA rate = 5.3x the float32 MFMA rate (gfx950 has no tf32).  The unit-length rows keep |C| error <= 2^-17
// (measured against the float32 kernel above: tests/test_gpu_sinkhorn.py).  Both operands are K-contiguous rows: a loaded float4 becomes
// one 16-byte LDS chunk [hi0..3][lo0..3], a fragment (8 K-values) is two chunks.  128 x 128 tile (4 waves of 64 x 64), K chunks of 64.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
constexpr int CX = 128, CXK = 64, CX_PITCH = 2 * CXK + 8;     // tile, K chunk, LDS row pitch (bfloat16 elements: 272 bytes)

__global__ __launch_bounds__(256) void k_big_cost_x3(const int* cloud_off, const int* prob_xy, int nmax, int E, float* ws) {
    const int p = blockIdx.z / 3, which = blockIdx.z - 3 * p;         // 0 xx, 1 yy, 2 xy (+ yx = its transpose)
    const BigLayout L = big_layout(nmax, E);
    float* base = ws + (size_t)p * L.total;
    const int cx = prob_xy[2 * p], cy = prob_xy[2 * p + 1];
    const int n = cloud_off[cx + 1] - cloud_off[cx], m = cloud_off[cy + 1] - cloud_off[cy];
    const float* Am = base + ((which == 0 || which == 2) ? L.xn : L.yn);
    const float* Bm = base + (which == 0 ? L.xn : L.yn);
    const int rows = (which == 0 || which == 2) ? n : m, cols = which == 0 ? n : m;
    float* C = base + (which == 0 ? L.cxx : which == 1 ? L.cyy : L.cxy);
    const int i0 = blockIdx.y * CX, j0 = blockIdx.x * CX;
    if (i0 >= rows || j0 >= cols) return;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds_x3[];
    uint16_t* As = lds_x3;                           // [128][CX_PITCH]
    uint16_t* Bs = lds_x3 + CX * CX_PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wi = (wv >> 1) * 64, wj = (wv & 1) * 64, g = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    for (int k0 = 0; k0 < E; k0 += CXK) {
        // 128 rows x CXK / 4 chunks of 4 K-values per operand (E % 4 == 0: the caller checks)
        __syncthreads();
        constexpr int NCH = CXK / 4, PER = CX * NCH / 256;           // chunks of 4 K-values per row; chunks per thread and operand
#pragma unroll
        for (int i = 0; i < 2 * PER; ++i) {
            const int id = tid + 256 * (i % PER), r = id / NCH, c = id % NCH, k = k0 + 4 * c;
            const bool isb = i >= PER;
            const int gr = (isb ? j0 : i0) + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < (isb ? cols : rows) && k < E) v = *reinterpret_cast<const float4*>((isb ? Bm : Am) + (size_t)gr * E + k);
            uint2 hi, lo;
            aadg_split4(v, hi, lo);
            *reinterpret_cast<uint4*>((isb ? Bs : As) + r * CX_PITCH + 8 * c) = make_uint4(hi.x, hi.y, lo.x, lo.y);
        }
        __syncthreads();
        const int kc = min(CXK, E - k0);
#pragma unroll
        for (int ks = 0; ks < CXK / 16; ++ks) {
            if (16 * ks >= kc) break;
            bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint16_t* pa = As + (wi + 32 * t + (lane & 31)) * CX_PITCH + 32 * ks + 16 * g;
                const uint4 q0 = *reinterpret_cast<const uint4*>(pa), q1 = *reinterpret_cast<const uint4*>(pa + 8);
                ah[t] = __builtin_bit_cast(bf16x8_t, make_uint4(q0.x, q0.y, q1.x, q1.y));
                al[t] = __builtin_bit_cast(bf16x8_t, make_uint4(q0.z, q0.w, q1.z, q1.w));
                const uint16_t* pb = Bs + (wj + 32 * t + (lane & 31)) * CX_PITCH + 32 * ks + 16 * g;
                const uint4 r0 = *reinterpret_cast<const uint4*>(pb), r1 = *reinterpret_cast<const uint4*>(pb + 8);
                bh[t] = __builtin_bit_cast(bf16x8_t, make_uint4(r0.x, r0.y, r1.x, r1.y));
                bl[t] = __builtin_bit_cast(bf16x8_t, make_uint4(r0.z, r0.w, r1.z, r1.w));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                }
        }
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wi + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * g, col = j0 + wj + 32 * b + (lane & 31);
                if (row < rows && col < cols) C[(size_t)row * nmax + col] = 1.0f - acc[a][b][r];
            }
    if (which == 2) {
        // C_yx = C_xy^T, transposed through LDS (the operand buffers are free): 32 lanes write 128 contiguous bytes of a C_yx row
        __syncthreads();
        float* T = reinterpret_cast<float*>(lds_x3) + wv * 32 * 33;
        float* Ct = base + L.cyx;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * g) * 33 + (lane & 31)] = 1.0f - acc[a][b][r];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int c = 2 * t + g, rr = lane & 31;
                    const int row = i0 + wi + 32 * a + rr, col = j0 + wj + 32 * b + c;
                    if (row < rows && col < cols) Ct[(size_t)col * nmax + row] = T[rr * 33 + c];
                }
                __builtin_amdgcn_wave_barrier();
            }
    }
}

// ---- one sweep: grid (4 * ceil(nmax/4), n_prob), 256 threads: wave <-> one row of one of the four softmins ---------
// step 0: initialisation at eps_s[0] (no potentials);  1..nits: eps-scaling with symmetrised update;
// nits+1: final extrapolation (no averaging, written to the tilde buffers);  beyond: nothing.
__global__ __launch_bounds__(256) void k_big_sweep(const int* cloud_off, const int* prob_xy, int nmax, int E, int step, float* ws) {
    const int p = blockIdx.y;
    const BigLayout L = big_layout(nmax, E);
    float* base = ws + (size_t)p * L.total;
    const int nits = reinterpret_cast<const int*>(base + L.meta + 2 * BIG_EPS_CAP)[0];
    if (step > nits + 1) return;
    const int cx = prob_xy[2 * p], cy = prob_xy[2 * p + 1];
    const int n = cloud_off[cx + 1] - cloud_off[cx], m = cloud_off[cy + 1] - cloud_off[cy];
    const int groups = (nmax + 3) / 4;
    const int kind = blockIdx.x / groups;                              // 0 a_x (C_xx), 1 b_y (C_yy), 2 a_y (C_yx), 3 b_x (C_xy)
    const int row = (blockIdx.x - kind * groups) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int rows = (kind == 0 || kind == 3) ? n : m, cols = (kind == 0 || kind == 2) ? n : m;
    if (row >= rows) return;
    const double* eps_s = reinterpret_cast<const double*>(base + L.meta);
    const bool init = step == 0, fin = step == nits + 1;
    const int it = init ? 0 : (fin ? nits - 1 : step - 1);
    const double eps = eps_s[it];
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    const float feps = (float)eps, inv2 = (float)(1.0 / eps) * LOG2E;
    // source potential of each softmin: a_x <- a_x, b_y <- b_y, a_y <- b_x, b_x <- a_y   (index into [a_x,b_y,a_y,b_x])
    const int src_kind = kind == 2 ? 3 : (kind == 3 ? 2 : kind);
    const int rd = init ? 0 : (fin ? (nits & 1) : ((step - 1) & 1));
    const float* hsrc = base + L.hh + ((size_t)rd * 4 + src_kind) * nmax;        // (log w_j + pot_j / eps) * log2(e), written by the previous step
    const float* own = base + L.pot + ((size_t)rd * 4 + kind) * nmax;
    const float h0 = logf(1.0f / (float)cols) * LOG2E;                              // the initialisation has no potentials
    const float* Crow = base + (kind == 0 ? L.cxx : kind == 1 ? L.cyy : kind == 2 ? L.cyx : L.cxy) + (size_t)row * nmax;
    // base-2 online log-sum-exp: running maximum mx, s = sum 2^(v - mx)
    float mx = -INFINITY, s = 0.f;
    const bool vec = (nmax & 3) == 0;
    if (vec) {
        constexpr int U = 8;                                   // float4 loads per lane in flight (a 4096-column row = 2 rounds)
        for (int j0 = lane * 4; j0 < cols; j0 += 256 * U) {
            float4 c4[U], h4[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 256 * u;
                // (a row of the padded matrix is nmax floats: a float4 that starts below `cols` never leaves the row)
                c4[u] = j < cols ? aadg_load_stream(Crow + j) : make_float4(0.f, 0.f, 0.f, 0.f);     // each element is read once per sweep
                h4[u] = (j < cols && !init) ? *reinterpret_cast<const float4*>(hsrc + j) : make_float4(h0, h0, h0, h0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 256 * u;
                if (j >= cols) continue;
                float v0 = fmaf(-c4[u].x, inv2, h4[u].x), v1 = fmaf(-c4[u].y, inv2, h4[u].y);
                float v2 = fmaf(-c4[u].z, inv2, h4[u].z), v3 = fmaf(-c4[u].w, inv2, h4[u].w);
                if (j + 1 >= cols) v1 = -INFINITY;
                if (j + 2 >= cols) v2 = -INFINITY;
                if (j + 3 >= cols) v3 = -INFINITY;
                const float mn = fmaxf(fmaxf(mx, fmaxf(v0, v1)), fmaxf(v2, v3));
                s = s * exp2f(mx - mn) + ((exp2f(v0 - mn) + exp2f(v1 - mn)) + (exp2f(v2 - mn) + exp2f(v3 - mn)));
                mx = mn;
            }
        }
    } else {
        for (int j = lane; j < cols; j += 64) {
            const float v = fmaf(-Crow[j], inv2, init ? h0 : hsrc[j]);
            const float mn = fmaxf(mx, v);
            s = s * exp2f(mx - mn) + exp2f(v - mn);
            mx = mn;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(mx, o, 64), s2 = __shfl_xor(s, o, 64);
        const float mn = fmaxf(mx, m2);
        const float e1 = mx == -INFINITY ? 0.f : exp2f(mx - mn), e2 = m2 == -INFINITY ? 0.f : exp2f(m2 - mn);
        s = s * e1 + s2 * e2;
        mx = mn;
    }
    if (lane == 0) {
        const float val = -feps * ((mx + log2f(s)) * LN2);
        if (fin) {
            base[L.til + (size_t)kind * nmax + row] = val;
        } else {
            // this step's output and, once per element instead of once per matrix entry, the per-column term of the step that will read
            // it: h = log w + pot / eps_next (the oracle's expression, a true division), scaled by log2(e)
            const int wr = init ? 0 : (((step - 1) & 1) ^ 1);
            const float pot = init ? val : 0.5f * (own[row] + val);
            const float feps_next = (float)eps_s[min(step, nits - 1)];
            base[L.pot + ((size_t)wr * 4 + kind) * nmax + row] = pot;
            base[L.hh + ((size_t)wr * 4 + kind) * nmax + row] = (logf(1.0f / (float)rows) + pot / feps_next) * LOG2E;
        }
    }
}

// ---- cost: grid n_prob, 256 threads ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_big_final(const int* cloud_off, const int* prob_xy, int nmax, int E, const float* ws,
                                                   float* out) {
    const int p = blockIdx.x, tid = threadIdx.x;
    const BigLayout L = big_layout(nmax, E);
    const float* base = ws + (size_t)p * L.total;
    const int cx = prob_xy[2 * p], cy = prob_xy[2 * p + 1];
    const int n = cloud_off[cx + 1] - cloud_off[cx], m = cloud_off[cy + 1] - cloud_off[cy];
    const float* a_x = base + L.til, *b_y = a_x + nmax, *a_y = b_y + nmax, *b_x = a_y + nmax;
    float s1 = 0.f, s2 = 0.f;
    const float wa = 1.0f / (float)n, wb = 1.0f / (float)m;
    for (int i = tid; i < n; i += 256) s1 += wa * (b_x[i] - a_x[i]);
    for (int i = tid; i < m; i += 256) s2 += wb * (a_y[i] - b_y[i]);
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    __shared__ float r1[4], r2[4];
    if ((tid & 63) == 0) { r1[tid >> 6] = s1; r2[tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) out[p] = ((r1[0] + r1[1]) + (r1[2] + r1[3])) + ((r2[0] + r2[1]) + (r2[2] + r2[3]));
}

}  // namespace

size_t aadg_sinkhorn_big_workspace_bytes(int n_prob, int nmax, int E) {
    return (size_t)n_prob * big_layout(nmax, E).total * sizeof(float);
}

int aadg_sinkhorn_big_launch(const float* feat, int ld, int E, const int* cloud_rows, const int* cloud_off, const int* prob_xy,
                             int n_prob, int nmax, float blur, float scaling, float* out, void* ws, size_t ws_bytes,
                             hipStream_t st, int phases) {
    // phases (measurement: bench.py times the two halves apart): bit 0 = prepare + schedule + cost matrices, bit 1 = sweeps + result
    if (!ws || ws_bytes < aadg_sinkhorn_big_workspace_bytes(n_prob, nmax, E)) return AADG_E_WORKSPACE;
    if (E > 64 * PREP_KPL) return AADG_E_UNSUPPORTED;
    float* w = reinterpret_cast<float*>(ws);
    if (phases & 1) {
    hipLaunchKernelGGL(k_big_prep, dim3(BIG_PREP_CHUNKS, n_prob), dim3(256), 0, st, feat, ld, E, cloud_rows, cloud_off, prob_xy,
                       nmax, w);
    AADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_big_schedule, dim3(n_prob), dim3(256), 0, st, nmax, E, blur, scaling, w);
    AADG_LAUNCH_CHECK();
    if ((E & 3) == 0) {
        // the float32-precision build on the bfloat16 matrix cores (three products per pair of split operands)
        const int tiles = (nmax + CX - 1) / CX;
        const size_t lds = (size_t)2 * CX * CX_PITCH * sizeof(uint16_t);
        static bool attr_set = false;
        if (!attr_set) {
            AADG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_big_cost_x3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        hipLaunchKernelGGL(k_big_cost_x3, dim3(tiles, tiles, n_prob * 3), dim3(256), lds, st, cloud_off, prob_xy, nmax, E, w);
    } else {
        const int tiles = (nmax + CT - 1) / CT;
        hipLaunchKernelGGL(k_big_cost, dim3(tiles, tiles, n_prob * 3), dim3(256), 0, st, cloud_off, prob_xy, nmax, E, w);
    }
    AADG_LAUNCH_CHECK();
    }
    if (!(phases & 2)) return 0;
    const dim3 gs(4 * ((nmax + 3) / 4), n_prob);
    for (int step = 0; step <= BIG_MAX_ITS + 1; ++step) {
        hipLaunchKernelGGL(k_big_sweep, gs, dim3(256), 0, st, cloud_off, prob_xy, nmax, E, step, w);
        AADG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_big_final, dim3(n_prob), dim3(256), 0, st, cloud_off, prob_xy, nmax, E, w, out);
    AADG_LAUNCH_CHECK();
    return 0;
}

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
// Stage kernels (this file, "v1" data flow):
//   k_hist   per-channel 256-bin histograms + sum of L   (AutoContrast / Equalize / Contrast)
//   k_lut    builds the 3x256 byte LUT of every LUT-class op (7 of the 10 ops)
//   k_apply  one op: LUT | Color | Sharpness (3x3 SMOOTH + blend) | Cutout, u8 -> u8
//   k_tables Pillow BILINEAR coefficient tables + NEAREST index tables for the crop window
//   k_final  horizontal+vertical fixed-point resample, pad, crop, normalise, HWC u8 -> CHW f32,
//            mask -> multilabel planes
#include "common.h"

namespace {

constexpr int KMAX = 8;           // max taps per output pixel (scale factor >= 1/3)
constexpr int HIST_STRIDE = 772;  // 768 bins + u64 L-sum + pad (u32 words)
constexpr int TAB_STRIDE = 2 * KMAX + 4;  // ints per crop position: xmin,xk[KMAX],ymin,yk[KMAX],xnn,ynn
constexpr int PRECISION_BITS = 22;

struct UnitRef {
    const aadg_unit* units;
    aadg_unit single;
    int use_single;
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

__device__ __forceinline__ uint32_t rgb2l(uint32_t r, uint32_t g, uint32_t b) {
    return (19595u * r + 38470u * g + 7471u * b + 0x8000u) >> 16;
}

// Image.blend(degenerate, image, alpha) for one byte; float32, unfused (Blend.c)
__device__ __forceinline__ uint32_t blend_px(int deg, int img, float alpha, bool interp) {
    float t = __fadd_rn((float)deg, __fmul_rn(alpha, (float)(img - deg)));
    if (interp) return (uint32_t)(int)t;
    if (t <= 0.0f) return 0u;
    if (t >= 255.0f) return 255u;
    return (uint32_t)(int)t;
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
__global__ __launch_bounds__(256) void k_hist(Bufs bufs, UnitRef ur, int stage, int npix, uint32_t* hist) {
    const int u = blockIdx.y;
    const aadg_unit& un = pick(ur, u);
    if (un.n_ops <= stage || !op_needs_stats(un.op[stage])) return;
    const uint8_t* in = stage_input(bufs, un, u, stage);
    __shared__ uint32_t sh[4][768];
    const int tid = threadIdx.x, wv = tid >> 6;
    for (int i = tid; i < 4 * 768; i += 256) (&sh[0][0])[i] = 0;
    __syncthreads();
    uint32_t* h = sh[wv];
    unsigned long long lsum = 0;
    const bool vec = ((npix & 3) == 0) && ((((uintptr_t)in) & 3) == 0);
    if (vec) {
        const int ngroups = npix >> 2;
        const uint32_t* p32 = reinterpret_cast<const uint32_t*>(in);
        for (int g = blockIdx.x * 256 + tid; g < ngroups; g += gridDim.x * 256) {
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
        for (int p = blockIdx.x * 256 + tid; p < npix; p += gridDim.x * 256) {
            uint32_t r = in[3 * (size_t)p], g = in[3 * (size_t)p + 1], b = in[3 * (size_t)p + 2];
            atomicAdd(&h[r], 1u); atomicAdd(&h[256 + g], 1u); atomicAdd(&h[512 + b], 1u);
            lsum += rgb2l(r, g, b);
        }
    }
    lsum = wave_sum(lsum);
    __syncthreads();
    uint32_t* gh = hist + (size_t)u * HIST_STRIDE;
    for (int i = tid; i < 768; i += 256) {
        uint32_t v = sh[0][i] + sh[1][i] + sh[2][i] + sh[3][i];
        if (v) atomicAdd(&gh[i], v);
    }
    if ((tid & 63) == 0 && lsum) atomicAdd(reinterpret_cast<unsigned long long*>(gh + 768), lsum);
}

// ------------------------------------------------------------------------------------------------
// k_lut: grid N, 256 threads; thread i owns LUT entry i of each channel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lut(UnitRef ur, int stage, int npix, const uint32_t* hist, uint8_t* lut) {
    const int u = blockIdx.x;
    const aadg_unit& un = pick(ur, u);
    if (un.n_ops <= stage) return;
    const int op = un.op[stage];
    if (!op_is_lut(op)) return;
    const int i = threadIdx.x;
    uint8_t* L = lut + (size_t)u * 768;
    const uint32_t* gh = hist + (size_t)u * HIST_STRIDE;
    __shared__ uint32_t scan[256];
    __shared__ int s_lo, s_hi, s_nnz;
    __shared__ unsigned long long s_sum;
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
    } else {  // AUTOCONTRAST / EQUALIZE
        for (int c = 0; c < 3; ++c) {
            const uint32_t hv = gh[256 * c + i];
            __syncthreads();
            if (i == 0) { s_lo = 256; s_hi = -1; s_nnz = 0; s_sum = 0; }
            scan[i] = hv;
            __syncthreads();
            if (hv) { atomicMin(&s_lo, i); atomicMax(&s_hi, i); atomicAdd(&s_nnz, 1); atomicAdd(&s_sum, (unsigned long long)hv); }
            // inclusive Hillis-Steele scan of the 256 bins
            for (int o = 1; o < 256; o <<= 1) {
                __syncthreads();
                uint32_t t = i >= o ? scan[i - o] : 0u;
                __syncthreads();
                scan[i] += t;
            }
            __syncthreads();
            uint8_t v = (uint8_t)i;
            if (op == AADG_OP_AUTOCONTRAST) {
                const int lo = s_lo, hi = s_hi;
                if (hi > lo) {
                    const double scale = 255.0 / (double)(hi - lo);
                    const double offset = (double)(-lo) * scale;
                    int ix = (int)((double)i * scale + offset);
                    ix = ix < 0 ? 0 : (ix > 255 ? 255 : ix);
                    v = (uint8_t)ix;
                }
            } else {
                if (s_nnz > 1) {
                    const unsigned long long last = gh[256 * c + s_hi];
                    const unsigned long long step = (s_sum - last) / 255ull;
                    if (step) {
                        const unsigned long long n = step / 2 + (unsigned long long)(scan[i] - hv);  // exclusive prefix
                        const unsigned long long q = n / step;
                        v = (uint8_t)(q > 255ull ? 255ull : q);
                    }
                }
            }
            L[256 * c + i] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_apply: grid (chunks, N), 256 threads; one op, u8 HWC -> u8 HWC.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_apply(Bufs bufs, UnitRef ur, int stage, int H, int W, const uint8_t* lut,
                                               uint8_t* out_override) {
    const int u = blockIdx.y;
    const aadg_unit& un = pick(ur, u);
    if (un.n_ops <= stage) return;
    const int op = un.op[stage];
    const uint8_t* in = stage_input(bufs, un, u, stage);
    uint8_t* out = out_override ? out_override : ((stage & 1) ? bufs.buf1 : bufs.buf0) + (size_t)u * bufs.img_bytes;
    const int tid = threadIdx.x;
    const int npix = H * W;
    __shared__ uint8_t sl[768];
    if (op_is_lut(op)) {
        const uint8_t* L = lut + (size_t)u * 768;
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

__global__ __launch_bounds__(256) void k_tables(UnitRef ur, int Hs, int Ws, int crop, int* tab) {
    const int u = blockIdx.x;
    const aadg_unit& un = pick(ur, u);
    int* base = tab + (size_t)u * crop * TAB_STRIDE;
    int* xmin = base;
    int* xk = xmin + crop;
    int* ymin = xk + (size_t)crop * KMAX;
    int* yk = ymin + crop;
    int* xnn = yk + (size_t)crop * KMAX;
    int* ynn = xnn + crop;
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
    // NEAREST tables: ImagingScaleAffine accumulates xo += a0 in double, sequentially
    if (threadIdx.x == 0 || threadIdx.x == 64) {
        const bool isx = threadIdx.x == 0;
        const int outSize = isx ? w : h, inSize = isx ? Ws : Hs, off = isx ? ox : oy;
        int* nn = isx ? xnn : ynn;
        const double a0 = (double)inSize / (double)outSize;
        double xo = 0.0 + a0 * 0.5;
        int lim = off + crop;
        if (lim > outSize) lim = outSize;
        for (int o = 0; o < crop; ++o) {
            const int s = o + off;
            if (s < 0 || s >= outSize) nn[o] = -1;
        }
        for (int s = 0; s < lim; ++s) {
            if (s >= off) {
                int xin = xo < 0.0 ? -1 : (int)xo;
                if (xin >= inSize) xin = -1;
                nn[s - off] = xin;
            }
            xo += a0;
        }
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
                                               int dataset, const int* tab, float* out_img, float* out_lbl) {
    const int u = blockIdx.z;
    const aadg_unit& un = pick(ur, u);
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
                *reinterpret_cast<float4*>(oi + c * plane + off) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
            *reinterpret_cast<float4*>(ol + off) = make_float4(l0[0], l0[1], l0[2], l0[3]);
            if (K == 2) *reinterpret_cast<float4*>(ol + plane + off) = make_float4(l1[0], l1[1], l1[2], l1[3]);
        } else {
            for (int i = 0; i < nvalid; ++i) {
                for (int c = 0; c < 3; ++c) oi[c * plane + off + i] = o[c][i];
                ol[off + i] = l0[i];
                if (K == 2) ol[plane + off + i] = l1[i];
            }
        }
    }
}

struct WsLayout {
    size_t hist, lut, tab, buf0, buf1, total;
};
WsLayout ws_layout(int N, int Hs, int Ws, int crop) {
    WsLayout l;
    size_t o = 0;
    l.hist = o; o = aadg_align_up(o + (size_t)N * HIST_STRIDE * 4, 256);
    l.lut = o;  o = aadg_align_up(o + (size_t)N * 768, 256);
    l.tab = o;  o = aadg_align_up(o + (size_t)N * crop * TAB_STRIDE * 4, 256);
    const size_t img = (size_t)Hs * Ws * 3;
    l.buf0 = o; o = aadg_align_up(o + (size_t)N * img, 256);
    l.buf1 = o; o = aadg_align_up(o + (size_t)N * img, 256);
    l.total = o;
    return l;
}

int chunks_for(int npix) {
    int c = (npix / 4 + 256 * 8 - 1) / (256 * 8);  // ~8 groups of 4 pixels per thread
    return c < 1 ? 1 : (c > 1024 ? 1024 : c);
}

// runs stages [0, max_ops) for N units
int run_stages(const Bufs& bufs, const UnitRef& ur, int N, int Hs, int Ws, int max_ops, uint8_t* ws8, const WsLayout& L,
               uint8_t* out_override, hipStream_t st) {
    const int npix = Hs * Ws;
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws8 + L.hist);
    uint8_t* lut = ws8 + L.lut;
    const dim3 g(chunks_for(npix), N);
    for (int k = 0; k < max_ops; ++k) {
        AADG_HIP_TRY(hipMemsetAsync(hist, 0, (size_t)N * HIST_STRIDE * 4, st));
        hipLaunchKernelGGL(k_hist, g, dim3(256), 0, st, bufs, ur, k, npix, hist);
        AADG_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_lut, dim3(N), dim3(256), 0, st, ur, k, npix, hist, lut);
        AADG_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_apply, g, dim3(256), 0, st, bufs, ur, k, Hs, Ws, lut, out_override);
        AADG_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

extern "C" size_t aadg_aug_u8_workspace_bytes(int N, int Hs, int Ws, int crop) {
    if (N <= 0 || Hs <= 0 || Ws <= 0 || crop < 0) return 0;
    return ws_layout(N, Hs, Ws, crop).total;
}

extern "C" int aadg_aug_u8_forward_ex(const uint8_t* pool, const uint8_t* masks, int P, int Hs, int Ws,
                                      const aadg_unit* units, int N, int max_ops, int crop, int dataset,
                                      float* out_img, float* out_lbl, void* ws, size_t ws_bytes, void* stream,
                                      void* ev_before_final, void* ev_after_final);

extern "C" int aadg_aug_u8_forward(const uint8_t* pool, const uint8_t* masks, int P, int Hs, int Ws,
                                   const aadg_unit* units, int N, int max_ops, int crop, int dataset,
                                   float* out_img, float* out_lbl, void* ws, size_t ws_bytes, void* stream) {
    return aadg_aug_u8_forward_ex(pool, masks, P, Hs, Ws, units, N, max_ops, crop, dataset, out_img, out_lbl, ws,
                                  ws_bytes, stream, nullptr, nullptr);
}

extern "C" int aadg_aug_u8_forward_ex(const uint8_t* pool, const uint8_t* masks, int P, int Hs, int Ws,
                                      const aadg_unit* units, int N, int max_ops, int crop, int dataset,
                                      float* out_img, float* out_lbl, void* ws, size_t ws_bytes, void* stream,
                                      void* ev_before_final, void* ev_after_final) {
    if (!pool || !masks || !units || !out_img || !out_lbl || !ws) return AADG_E_BADARG;
    if (P <= 0 || Hs <= 0 || Ws <= 0 || N <= 0 || crop <= 0) return AADG_E_BADARG;
    if (max_ops < 0 || max_ops > AADG_MAX_OPS) return AADG_E_BADARG;
    if (dataset != AADG_DATASET_OPTIC && dataset != AADG_DATASET_VESSEL) return AADG_E_BADARG;
    if ((size_t)N * (size_t)crop > (size_t)1 << 28) return AADG_E_BADARG;
    const WsLayout L = ws_layout(N, Hs, Ws, crop);
    if (ws_bytes < L.total) return AADG_E_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    uint8_t* ws8 = reinterpret_cast<uint8_t*>(ws);
    // pool images and workspace images are both densely packed (Hs*Ws*3 bytes each)
    Bufs bufs{pool, ws8 + L.buf0, ws8 + L.buf1, (size_t)Hs * Ws * 3};
    UnitRef ur;
    ur.units = units;
    ur.use_single = 0;
    ::memset(&ur.single, 0, sizeof(ur.single));
    int rc = 0;
    rc = run_stages(bufs, ur, N, Hs, Ws, max_ops, ws8, L, nullptr, st);
    if (rc) return rc;
    int* tab = reinterpret_cast<int*>(ws8 + L.tab);
    hipLaunchKernelGGL(k_tables, dim3(N), dim3(256), 0, st, ur, Hs, Ws, crop, tab);
    AADG_LAUNCH_CHECK();
    const dim3 g((crop + 255) / 256, (crop + FIN_ROWS - 1) / FIN_ROWS, N);
    if (ev_before_final) AADG_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_before_final), st));
    hipLaunchKernelGGL(k_final, g, dim3(256), 0, st, bufs, masks, ur, Hs, Ws, crop, dataset, tab, out_img, out_lbl);
    AADG_LAUNCH_CHECK();
    if (ev_after_final) AADG_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_after_final), st));
    return 0;
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
    ::memset(&ur.single, 0, sizeof(ur.single));
    ur.single.src = 0;
    ur.single.n_ops = 1;
    ur.single.op[0] = op;
    ur.single.iarg[0] = iarg;
    ur.single.farg[0] = farg;
    for (int i = 0; i < 4; ++i) ur.single.rect[0][i] = rect_host ? rect_host[i] : (i < 2 ? 0 : -1);
    ur.single.scaled_w = W;
    ur.single.scaled_h = H;
    return run_stages(bufs, ur, 1, H, W, 1, ws8, L, out, st);
}

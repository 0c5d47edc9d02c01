// Float tensor ops on gfx950 (a8-a13).
//
// Semantics = the reference's data/functional.py (batched [B,3,H,W] float32 images in [0,1], magnitude a
// scalar or one value per sample, output clamped to [0,1] by tensor_function, data/functional.py:49-73):
//   pointwise   invert :158, solarize :165, posterize :172 (floor(255x)/255 -- the shift pair is a no-op),
//               gray :183, saturate :210, brightness :217, sample_pairing :233, hue :224, hflip/vflip :138-147
//   statistics  contrast :189 (mean of gray(255x) per sample), auto_contrast :196 (per-(b,c) min/max),
//               equalize :241 (per-(b,c) 256-bin torch.histc + CDF LUT)
//   stencil     sharpness :265 / gaussian_blur3x3 :274 through _blur :98 (reflect pad 1, depthwise 3x3)
//   geometric   shear_x/y :110-121, translate_x/y :124-135, rotate :150 -- the arithmetic lives in kornia
//               (absent, unpinned): restated as inverse affine map about the image centre + bilinear
//               sampling on pixel centres with zero padding ("parity unpinned").
//
// All kernels are HBM-bound: one read and one write of the image (24*H*W bytes per image), a second read for the
// statistics ops.  Round 3 ("Float tensor ops"):
//   * every op is its own template instantiation (no per-pixel switch), a lane owns 4 consecutive pixels of all three
//     planes (16-byte accesses), inputs that are read once are loaded and all outputs stored with streaming
//     (non-temporal) accesses -- the batch is larger than the Infinity Cache;
//   * statistics ops are two launches instead of four: the reduction of the per-block partial records (and the
//     equalize LUT) happens in the prologue of every apply workgroup, and the apply pass walks the samples in the
//     opposite order to the statistics pass, so that it starts on the part of the batch the cache still holds;
//   * the 3x3 stencil streams down row strips with a three-row register window (neighbour columns by wave shuffles,
//     no LDS, no barriers);
//   * translate_x / translate_y / shear_x read their taps with one 16-byte load per source row at a 4-byte-aligned address (taps of
//     weight zero are not loaded); rotate and shear_y stage the footprint of a 32 x 32 tile in LDS, plane after plane, the next plane in
//     flight; coordinates and weights serve the three planes; their workgroups are numbered image-major per XCD (one L2 per image);
// Shapes the vector paths cannot take (W % 4 != 0, W < 8, unaligned pointers) go through the `_any` kernels (one pixel per lane),
// which share the per-pixel arithmetic with the vector kernels (tests/test_gpu_functional.py compares them bit for bit).
#include "common.h"

namespace {

constexpr int TO_THREADS = 256;
constexpr int STAT_CHUNKS_MAX = 64;      // partial records per sample: one wave reduces them

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ float gray_of(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.110f * b; }
// _blend_image(img1, img2, alpha) = clamp(img2 + alpha * (img1 - img2))
__device__ __forceinline__ float blendf(float img1, float img2, float alpha) { return clamp01(img2 + alpha * (img1 - img2)); }

__device__ __forceinline__ void rgb2hsv(float r, float g, float b, float& h, float& s, float& v) {
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    const float d = mx - mn;
    v = mx;
    s = mx > 0.0f ? d / mx : 0.0f;
    float hh = 0.0f;
    if (d > 0.0f) {
        if (mx == r) hh = (g - b) / d;
        else if (mx == g) hh = 2.0f + (b - r) / d;
        else hh = 4.0f + (r - g) / d;
        hh /= 6.0f;
        hh -= floorf(hh);
    }
    h = hh;
}
__device__ __forceinline__ void hsv2rgb(float h, float s, float v, float& r, float& g, float& b) {
    const float h6 = h * 6.0f;
    const float fi = floorf(h6);
    const float f = h6 - fi;
    const int i = ((int)fi) % 6;
    const float p = v * (1.0f - s), q = v * (1.0f - f * s), t = v * (1.0f - (1.0f - f) * s);
    switch (i) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

// per-block partial record of the statistics pass
struct Part {
    float mn[3], mx[3];      // auto_contrast / equalize: min / max of clamp(x)*255 per channel
    double gsum;             // contrast: sum of gray(255x)
    unsigned int pad[2];
};

// what an apply workgroup derives from the partial records of its sample
struct SampleStats {
    float mn[3], mx[3];
    float mean;              // floor(mean(gray(255x)) + 0.5) / 255
};

// ---- one pixel of a pointwise op (FOP is a compile-time constant) ----------------------------------------------------
template <int FOP>
__device__ __forceinline__ void point_px(float& R, float& G, float& Bv, float m, const SampleStats& st, const float* lut /* LDS [3][256] */,
                                         float R2, float G2, float B2) {
    if (FOP == AADG_FOP_INVERT) { R = 1.0f - R; G = 1.0f - G; Bv = 1.0f - Bv; }
    else if (FOP == AADG_FOP_SOLARIZE) { R = R < m ? R : 1.0f - R; G = G < m ? G : 1.0f - G; Bv = Bv < m ? Bv : 1.0f - Bv; }
    else if (FOP == AADG_FOP_POSTERIZE) {
        R = (float)(long long)(R * 255.0f) / 255.0f; G = (float)(long long)(G * 255.0f) / 255.0f;
        Bv = (float)(long long)(Bv * 255.0f) / 255.0f;
    } else if (FOP == AADG_FOP_GRAY) { const float y = gray_of(R, G, Bv); R = G = Bv = y; }
    else if (FOP == AADG_FOP_SATURATE) {
        const float y = gray_of(R, G, Bv), a = 1.0f - m;
        R = blendf(R, y, a); G = blendf(G, y, a); Bv = blendf(Bv, y, a);
    } else if (FOP == AADG_FOP_BRIGHTNESS) {
        const float a = 1.0f - m;
        R = blendf(R, 0.0f, a); G = blendf(G, 0.0f, a); Bv = blendf(Bv, 0.0f, a);
    } else if (FOP == AADG_FOP_CONTRAST) {
        const float a = 1.0f - m;
        R = blendf(R, st.mean, a); G = blendf(G, st.mean, a); Bv = blendf(Bv, st.mean, a);
    } else if (FOP == AADG_FOP_SAMPLE_PAIRING) {
        R = (1.0f - m) * R + m * R2; G = (1.0f - m) * G + m * G2; Bv = (1.0f - m) * Bv + m * B2;
    } else if (FOP == AADG_FOP_AUTO_CONTRAST) {
        float* ch[3] = {&R, &G, &Bv};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = clamp01(*ch[c]) * 255.0f;
            const float scale = 255.0f / (st.mx[c] - st.mn[c] + 0.1f);
            *ch[c] = floorf(((float)(long long)v - st.mn[c]) * scale) / 255.0f;
        }
    } else if (FOP == AADG_FOP_HUE) {
        float h, s, v;
        rgb2hsv(R, G, Bv, h, s, v);
        h = h + m;
        h = h - floorf(h);          // python % 1
        hsv2rgb(h, s, v, R, G, Bv);
    }
    (void)lut;
}

// equalize: output = lut.view(-1)[shifted.long()], shifted = clamp(x)*255 + 256*plane evaluated in float32 as the reference does (for
// large plane indices the sum is rounded to the float32 grid before it is truncated); clamp(x)*255 <= 255, so the index stays inside
// the plane's 256 entries: `lut` = the three tables of sample b in LDS
__device__ __forceinline__ float equalize_px(float x, int plane, int nplanes, const float* lut_lds, int b) {
    const float shifted = clamp01(x) * 255.0f + 256.0f * (float)plane;
    int idx = (int)shifted;
    idx = idx < nplanes * 256 ? idx : nplanes * 256 - 1;
    return lut_lds[idx - 768 * b];
}

// ---- prologue of an apply workgroup: reduce the sample's partial records / build its equalize LUT --------------------
template <int FOP>
__device__ __forceinline__ void sample_prologue(const Part* __restrict__ part, int chunks, int HW, const unsigned int* __restrict__ hist,
                                                int b, int nplanes, SampleStats& st, float* lut_lds /* [3][256] */) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ SampleStats sst;
    if (FOP == AADG_FOP_CONTRAST || FOP == AADG_FOP_AUTO_CONTRAST) {
        if (wv == 0) {
            float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            double gs = 0.0;
            if (lane < chunks) {
                const Part p = part[(size_t)b * chunks + lane];
#pragma unroll
                for (int c = 0; c < 3; ++c) { mn[c] = p.mn[c]; mx[c] = p.mx[c]; }
                gs = p.gsum;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) { mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]); }
            gs = wave_sum(gs);          // a fixed butterfly: every workgroup of the sample gets the same bits
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { sst.mn[c] = mn[c]; sst.mx[c] = mx[c]; }
                sst.mean = floorf((float)(gs / (double)HW) + 0.5f) / 255.0f;
            }
        }
        __syncthreads();
        st = sst;
    }
    if (FOP == AADG_FOP_EQUALIZE) {
        // LUT per plane: cdf = cumsum(h); step = floor((cdf[-1]-h[-1])/255); lut[k] = floor((cdf_exclusive[k] + floor(step/2)) / (step + 0.1)) / 255
        // wave c <-> channel c, lane l <-> bins 4l .. 4l+3; counts are integers < 2^24, so the float32 sums are exact in any order
        if (wv < 3) {
            const uint4 h4 = *reinterpret_cast<const uint4*>(hist + (size_t)(b * 3 + wv) * 256 + 4 * lane);
            const float h[4] = {(float)h4.x, (float)h4.y, (float)h4.z, (float)h4.w};
            const float loc = (h[0] + h[1]) + (h[2] + h[3]);
            float incl = loc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            const float total = __shfl(incl, 63, 64), last = __shfl(h[3], 63, 64);
            const float step = floorf((total - last) / 255.0f);
            float excl = incl - loc;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lut_lds[wv * 256 + 4 * lane + k] = floorf((excl + floorf(step / 2.0f)) / (step + 0.1f)) / 255.0f;
                excl += h[k];
            }
        }
        __syncthreads();
    }
    (void)nplanes;
}

// sample_pairing reads every image twice: as itself and as the partner of the sample that perm maps onto it.  Walking the samples along
// the cycles of perm (b, perm[b], perm[perm[b]], ...) makes the image fetched as a partner at one step the "self" image of the next, so the second
// read finds it in the Infinity Cache instead of HBM (the batch is larger than the cache: in index order half of the second reads miss).
// One workgroup; perm staged in LDS; every sample appears exactly once whatever perm holds (out-of-range entries end a chain).
constexpr int PAIR_ORDER_MAX = 8192;
__global__ __launch_bounds__(256) void k_fop_pair_order(const int* __restrict__ perm, int B, int* __restrict__ order) {
    __shared__ int sp[PAIR_ORDER_MAX];
    __shared__ unsigned char seen[PAIR_ORDER_MAX];
    for (int i = threadIdx.x; i < B; i += 256) { sp[i] = perm[i]; seen[i] = 0; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int n = 0;
    for (int start = 0; start < B; ++start) {
        int b = start;
        while (b >= 0 && b < B && !seen[b]) {
            seen[b] = 1;
            order[n++] = b;
            b = sp[b];
        }
    }
}

// ---- pointwise / LUT apply kernel: grid (chunks, B); a lane owns 4 consecutive pixels x 3 planes, UNR groups in flight ------------
template <int FOP, bool STREAM>
__global__ __launch_bounds__(TO_THREADS) void k_fop_point(const float* __restrict__ in, float* __restrict__ out,
                                                          const float* __restrict__ mag, int mag_n, const int* __restrict__ perm,
                                                          const Part* __restrict__ part, int chunks, const unsigned int* __restrict__ hist,
                                                          int HW, const int* __restrict__ order) {
    constexpr bool STAT = FOP == AADG_FOP_CONTRAST || FOP == AADG_FOP_AUTO_CONTRAST || FOP == AADG_FOP_EQUALIZE;
    // statistics ops: opposite sample order to the statistics pass (its last samples are the ones the caches still hold);
    // sample_pairing: along the cycles of perm (k_fop_pair_order)
    const int b = STAT ? (int)gridDim.y - 1 - (int)blockIdx.y
                       : (FOP == AADG_FOP_SAMPLE_PAIRING && order != nullptr) ? order[blockIdx.y] : (int)blockIdx.y;
    const int nplanes = gridDim.y * 3;
    const size_t base = (size_t)b * 3 * HW;
    const float m = mag ? mag[mag_n == 1 ? 0 : b] : 0.0f;
    const float* pin = in + base;
    const float* pin2 = FOP == AADG_FOP_SAMPLE_PAIRING ? in + (size_t)perm[b] * 3 * HW : pin;
    float* po = out + base;
    __shared__ float lut_lds[FOP == AADG_FOP_EQUALIZE ? 768 : 4];
    SampleStats st = {};
    if (STAT) sample_prologue<FOP>(part, chunks, HW, hist, b, nplanes, st, lut_lds);
    const int ngroups = HW >> 2;
    constexpr int UNR = 2;
    for (int g0 = blockIdx.x * (TO_THREADS * UNR) + threadIdx.x; g0 < ngroups; g0 += gridDim.x * (TO_THREADS * UNR)) {
        float4 a[UNR], c[UNR], d[UNR], a2[UNR], c2[UNR], d2[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int g = g0 + u * TO_THREADS;
            if (g < ngroups) {
                const int i0 = g << 2;
                // STAT ops re-read what the statistics pass has just read: ordinary loads (cache hits wanted); everything else is read once
                if (STREAM && !STAT && FOP != AADG_FOP_SAMPLE_PAIRING) { a[u] = aadg_load_stream(pin + i0); c[u] = aadg_load_stream(pin + HW + i0); d[u] = aadg_load_stream(pin + 2 * HW + i0); }
                else {
                    a[u] = *reinterpret_cast<const float4*>(pin + i0); c[u] = *reinterpret_cast<const float4*>(pin + HW + i0);
                    d[u] = *reinterpret_cast<const float4*>(pin + 2 * HW + i0);
                }
                if (FOP == AADG_FOP_SAMPLE_PAIRING) {
                    a2[u] = *reinterpret_cast<const float4*>(pin2 + i0); c2[u] = *reinterpret_cast<const float4*>(pin2 + HW + i0);
                    d2[u] = *reinterpret_cast<const float4*>(pin2 + 2 * HW + i0);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int g = g0 + u * TO_THREADS;
            if (g >= ngroups) break;
            const int i0 = g << 2;
            float r[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, gg[4] = {c[u].x, c[u].y, c[u].z, c[u].w}, bl[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
            float r2[4] = {0, 0, 0, 0}, g2[4] = {0, 0, 0, 0}, b2[4] = {0, 0, 0, 0};
            if (FOP == AADG_FOP_SAMPLE_PAIRING) {
                r2[0] = a2[u].x; r2[1] = a2[u].y; r2[2] = a2[u].z; r2[3] = a2[u].w;
                g2[0] = c2[u].x; g2[1] = c2[u].y; g2[2] = c2[u].z; g2[3] = c2[u].w;
                b2[0] = d2[u].x; b2[1] = d2[u].y; b2[2] = d2[u].z; b2[3] = d2[u].w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (FOP == AADG_FOP_EQUALIZE) {
                    r[k] = equalize_px(r[k], b * 3 + 0, nplanes, lut_lds, b);
                    gg[k] = equalize_px(gg[k], b * 3 + 1, nplanes, lut_lds, b);
                    bl[k] = equalize_px(bl[k], b * 3 + 2, nplanes, lut_lds, b);
                } else {
                    point_px<FOP>(r[k], gg[k], bl[k], m, st, lut_lds, r2[k], g2[k], b2[k]);
                }
                r[k] = clamp01(r[k]); gg[k] = clamp01(gg[k]); bl[k] = clamp01(bl[k]);
            }
            aadg_store_out(po + i0, make_float4(r[0], r[1], r[2], r[3]), STREAM);
            aadg_store_out(po + HW + i0, make_float4(gg[0], gg[1], gg[2], gg[3]), STREAM);
            aadg_store_out(po + 2 * HW + i0, make_float4(bl[0], bl[1], bl[2], bl[3]), STREAM);
        }
    }
}

// scalar variant of the same ops for HW % 4 != 0 or unaligned tensors: one pixel per lane
template <int FOP>
__global__ __launch_bounds__(TO_THREADS) void k_fop_point_any(const float* __restrict__ in, float* __restrict__ out,
                                                              const float* __restrict__ mag, int mag_n, const int* __restrict__ perm,
                                                              const Part* __restrict__ part, int chunks, const unsigned int* __restrict__ hist,
                                                              int HW) {
    constexpr bool STAT = FOP == AADG_FOP_CONTRAST || FOP == AADG_FOP_AUTO_CONTRAST || FOP == AADG_FOP_EQUALIZE;
    const int b = blockIdx.y, nplanes = gridDim.y * 3;
    const size_t base = (size_t)b * 3 * HW;
    const float m = mag ? mag[mag_n == 1 ? 0 : b] : 0.0f;
    const float* pin = in + base;
    const float* pin2 = FOP == AADG_FOP_SAMPLE_PAIRING ? in + (size_t)perm[b] * 3 * HW : pin;
    float* po = out + base;
    __shared__ float lut_lds[FOP == AADG_FOP_EQUALIZE ? 768 : 4];
    SampleStats st = {};
    if (STAT) sample_prologue<FOP>(part, chunks, HW, hist, b, nplanes, st, lut_lds);
    for (int i = blockIdx.x * TO_THREADS + threadIdx.x; i < HW; i += gridDim.x * TO_THREADS) {
        float R = pin[i], G = pin[HW + i], Bv = pin[2 * HW + i];
        if (FOP == AADG_FOP_EQUALIZE) {
            R = equalize_px(R, b * 3 + 0, nplanes, lut_lds, b);
            G = equalize_px(G, b * 3 + 1, nplanes, lut_lds, b);
            Bv = equalize_px(Bv, b * 3 + 2, nplanes, lut_lds, b);
        } else {
            float R2 = 0.f, G2 = 0.f, B2 = 0.f;
            if (FOP == AADG_FOP_SAMPLE_PAIRING) { R2 = pin2[i]; G2 = pin2[HW + i]; B2 = pin2[2 * HW + i]; }
            point_px<FOP>(R, G, Bv, m, st, lut_lds, R2, G2, B2);
        }
        po[i] = clamp01(R); po[HW + i] = clamp01(G); po[2 * HW + i] = clamp01(Bv);
    }
}

// ---- statistics pass: grid (chunks <= 64, B); one partial record per workgroup ----------------------------------------------------
// equalize: torch.histc(bins=256*BC, min=0, max=256*BC-1) on v + 256*(b*3+c): per plane this is
// bin = int(shifted * nb / (nb - 1)) - 256*(b*3+c), evaluated in float32 like ATen; a value can land in the neighbouring plane's
// first bin, torch counts it there
template <int FOP, bool VEC>
__global__ __launch_bounds__(TO_THREADS) void k_fop_stats(const float* __restrict__ in, int HW, Part* __restrict__ part,
                                                          unsigned int* __restrict__ hist /*[B*3][256]*/) {
    const int b = blockIdx.y, tid = threadIdx.x;
    const float* pin = in + (size_t)b * 3 * HW;
    constexpr bool HIST = FOP == AADG_FOP_EQUALIZE;
    constexpr int COPIES = 4;        // interleaved sub-histograms: lanes of a wave that hit one bin spread over 4 banks
    __shared__ unsigned int sh[HIST ? 3 * 256 * COPIES : 1];
    if (HIST) {
        for (int i = tid; i < 3 * 256 * COPIES; i += TO_THREADS) sh[i] = 0;
        __syncthreads();
    }
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    double gs = 0.0;
    const float nb = (float)(gridDim.y * 3 * 256);
    const int copy = tid & (COPIES - 1);
    auto pixel = [&](float r, float g, float bl) {
        if (FOP == AADG_FOP_CONTRAST) {
            gs += (double)gray_of(r * 255.0f, g * 255.0f, bl * 255.0f);
        } else {
            const float v[3] = {clamp01(r) * 255.0f, clamp01(g) * 255.0f, clamp01(bl) * 255.0f};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                mn[c] = fminf(mn[c], v[c]); mx[c] = fmaxf(mx[c], v[c]);
                if (HIST) {
                    const int plane = b * 3 + c;
                    const float shifted = v[c] + 256.0f * (float)plane;
                    int bin = (int)(shifted * nb / (nb - 1.0f));
                    if (bin >= (int)nb) bin = (int)nb - 1;
                    bin -= 256 * plane;
                    if (bin >= 0 && bin < 256) atomicAdd(&sh[(c * 256 + bin) * COPIES + copy], 1u);
                    else atomicAdd(&hist[(size_t)(plane + (bin < 0 ? -1 : 1)) * 256 + (bin < 0 ? bin + 256 : bin - 256)], 1u);
                }
            }
        }
    };
    if (VEC) {
        const int ngroups = HW >> 2;
        constexpr int UNR = 2;
        for (int g0 = blockIdx.x * (TO_THREADS * UNR) + tid; g0 < ngroups; g0 += gridDim.x * (TO_THREADS * UNR)) {
            float4 a[UNR], c[UNR], d[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int g = g0 + u * TO_THREADS;
                if (g < ngroups) {
                    a[u] = *reinterpret_cast<const float4*>(pin + 4 * g); c[u] = *reinterpret_cast<const float4*>(pin + HW + 4 * g);
                    d[u] = *reinterpret_cast<const float4*>(pin + 2 * HW + 4 * g);
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (g0 + u * TO_THREADS >= ngroups) break;
                pixel(a[u].x, c[u].x, d[u].x); pixel(a[u].y, c[u].y, d[u].y);
                pixel(a[u].z, c[u].z, d[u].z); pixel(a[u].w, c[u].w, d[u].w);
            }
        }
    } else {
        for (int i = blockIdx.x * TO_THREADS + tid; i < HW; i += gridDim.x * TO_THREADS) pixel(pin[i], pin[HW + i], pin[2 * HW + i]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]); }
    gs = wave_sum(gs);
    __shared__ float smn[4][3], smx[4][3];
    __shared__ double sgs[4];
    const int wv = tid >> 6;
    if ((tid & 63) == 0) {
        for (int c = 0; c < 3; ++c) { smn[wv][c] = mn[c]; smx[wv][c] = mx[c]; }
        sgs[wv] = gs;
    }
    __syncthreads();
    if (tid == 0) {
        Part p;
        for (int c = 0; c < 3; ++c) {
            p.mn[c] = fminf(fminf(smn[0][c], smn[1][c]), fminf(smn[2][c], smn[3][c]));
            p.mx[c] = fmaxf(fmaxf(smx[0][c], smx[1][c]), fmaxf(smx[2][c], smx[3][c]));
        }
        p.gsum = (sgs[0] + sgs[1]) + (sgs[2] + sgs[3]);
        p.pad[0] = p.pad[1] = 0;
        part[(size_t)b * gridDim.x + blockIdx.x] = p;
    }
    if (HIST)
        for (int i = tid; i < 768; i += TO_THREADS) {
            const uint4 s4 = *reinterpret_cast<const uint4*>(&sh[i * COPIES]);
            const unsigned int s = (s4.x + s4.y) + (s4.z + s4.w);
            if (s) atomicAdd(&hist[(size_t)(b * 3 + i / 256) * 256 + (i & 255)], s);
        }
}

// ---- 3x3 depthwise stencil with reflect padding --------------------------------------------------------------------------------
// F.pad(..., 'reflect'): -1 -> 1, n -> n-2
__device__ __forceinline__ int reflect1(int i, int n) {
    i = i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i);
    return min(max(i, 0), n - 1);
}
// the 3x3 kernel of the op: the caller's, the reference default of sharpness (data/kernels.py:9-13), or the Gaussian of sigma := mean(mag)^2
// as variance (data/kernels.py:16-25); uniform per launch -- every lane evaluates it (nine expf)
template <int FOP>
__device__ __forceinline__ void stencil_kernel(const float* __restrict__ kern, const float* __restrict__ mag, int mag_n, float* k) {
    if (kern) {
#pragma unroll
        for (int j = 0; j < 9; ++j) k[j] = kern[j];
    } else if (FOP == AADG_FOP_SHARPNESS) {
#pragma unroll
        for (int j = 0; j < 9; ++j) k[j] = (j == 4 ? 5.0f : 1.0f) / 13.0f;
    } else {
        float sm = 0.f;
        for (int i = 0; i < mag_n; ++i) sm += mag[i];
        sm /= (float)mag_n;
        const float var = sm * sm;
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float dy = (float)(j / 3 - 1), dx = (float)(j % 3 - 1);
            k[j] = expf(-(dx * dx + dy * dy) / (2.0f * var));
            tot += k[j];
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) k[j] = k[j] / tot;
    }
}

// one output of the 3x3 window rp / rc / rn (three consecutive values each, centre = rc[1]): row-major accumulation with fused
// multiply-adds (both kernels below use this function, so they agree bit for bit); sharpness blends with the centre pixel
template <int FOP>
__device__ __forceinline__ float stencil_px(const float* rp, const float* rc, const float* rn, const float* k, float m) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) acc = fmaf(rp[j], k[j], acc);
#pragma unroll
    for (int j = 0; j < 3; ++j) acc = fmaf(rc[j], k[3 + j], acc);
#pragma unroll
    for (int j = 0; j < 3; ++j) acc = fmaf(rn[j], k[6 + j], acc);
    return FOP == AADG_FOP_SHARPNESS ? blendf(rc[1], acc, 1.0f - m) : acc;
}

// register-window variant: a wave owns a strip of 64 * NPX columns (NPX = 4 or 8 consecutive pixels per lane) x ST_ROWS rows of one
// plane and walks down it; the three source rows of an output row live in registers, the neighbour columns of a lane's run come from
// the adjacent lanes (wave shuffles) and the strip's two outer columns from the lane's own pixels when the strip ends at the image
// border (reflect padding: column -1 = column 1, column W = column W-2) -- a 512-wide image is one strip and needs no extra load --
// or from one 4-byte load by the first / last lane otherwise.  No LDS, no barriers.  grid (strips, ceil(row strips / 4), planes).
template <int NPX>
struct RowWin { float v[NPX + 2]; };        // v[0] = left neighbour, v[1..NPX] = the lane's pixels, v[NPX+1] = right neighbour

template <int NPX, bool STREAM>
__device__ __forceinline__ RowWin<NPX> stencil_load_row(const float* __restrict__ pin, int y, int H, int W, int x, int lane, int last_lane,
                                                        bool active, bool left_is_border, bool right_is_border) {
    RowWin<NPX> o;
    const float* rowp = pin + (size_t)reflect1(y, H) * W;
#pragma unroll
    for (int q = 0; q < NPX / 4; ++q) {
        float4 v = make_float4(0, 0, 0, 0);
        if (active) v = STREAM ? aadg_load_stream(rowp + x + 4 * q) : *reinterpret_cast<const float4*>(rowp + x + 4 * q);
        o.v[1 + 4 * q] = v.x; o.v[2 + 4 * q] = v.y; o.v[3 + 4 * q] = v.z; o.v[4 + 4 * q] = v.w;
    }
    float el = o.v[2], er = o.v[NPX - 1];         // reflect at the image border: the lane's own second / second-to-last pixel
    if (!left_is_border && lane == 0) el = rowp[x - 1];                 // uniform branches: interior strip edges only (W > 64 * NPX)
    if (!right_is_border && lane == last_lane) er = rowp[x + NPX];
    const float l = __shfl_up(o.v[NPX], 1, 64), r = __shfl_down(o.v[1], 1, 64);
    o.v[0] = lane == 0 ? el : l;
    o.v[NPX + 1] = lane == last_lane ? er : r;
    return o;
}

// 1-D grids in group-major order per XCD.  Consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2; with the plain
// (x, y, image) order neighbouring workgroups of one image -- whose source footprints overlap (halo rows, rotated boxes, row pairs) -- sit on
// eight different L2s and every one of them fetches the overlap again (rotate: 2.4 x the source bytes from the fabric).  Here workgroup id L
// works for group 8 * (L / 8 / per_group) + L % 8 (group = image or plane), so a group's workgroups share one L2; the last groups % 8
// groups keep the plain order.
__device__ __forceinline__ void xcd_group_major(int L, int per_group, int groups, int& g, int& t) {
    if (L < per_group * (groups & ~7)) {
        const int j = L >> 3;
        g = (j / per_group) * 8 + (L & 7);
        t = j % per_group;
    } else {
        g = L / per_group;
        t = L % per_group;
    }
}

template <int FOP, int NPX, int ROWS, bool STREAM>
__global__ __launch_bounds__(TO_THREADS) void k_fop_stencil_rows(const float* __restrict__ in, float* __restrict__ out,
                                                                 const float* __restrict__ mag, int mag_n,
                                                                 const float* __restrict__ kern, int H, int W) {
    const int plane = blockIdx.z, b = plane / 3;
    const float* pin = in + (size_t)plane * H * W;
    float* po = out + (size_t)plane * H * W;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x0 = blockIdx.x * (64 * NPX), x = x0 + NPX * lane;
    const int cols = min(64 * NPX, W - x0);            // a multiple of NPX, >= NPX
    const int last_lane = cols / NPX - 1;
    const bool active = lane <= last_lane;
    const bool lb = x0 == 0, rb = x0 + cols == W;
    const int ys = (blockIdx.y * 4 + wv) * ROWS;
    if (ys >= H) return;
    const int ye = min(ys + ROWS, H);
    float k[9];
    stencil_kernel<FOP>(kern, mag, mag_n, k);
    const float m = mag ? mag[mag_n == 1 ? 0 : b] : 0.0f;
    // two rows ahead: the load of row y + 2 is in flight while row y is computed (a wave has nothing else to hide the latency with)
    RowWin<NPX> p = stencil_load_row<NPX, false>(pin, ys - 1, H, W, x, lane, last_lane, active, lb, rb);   // halo row: the strip above streams it
    RowWin<NPX> c = stencil_load_row<NPX, STREAM>(pin, ys, H, W, x, lane, last_lane, active, lb, rb);
    RowWin<NPX> n = stencil_load_row<NPX, STREAM>(pin, ys + 1, H, W, x, lane, last_lane, active, lb, rb);
    for (int y = ys; y < ye; ++y) {
        RowWin<NPX> nn = n;
        if (y + 1 < ye) nn = stencil_load_row<NPX, STREAM>(pin, y + 2, H, W, x, lane, last_lane, active, lb, rb);
        float o[NPX];
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            o[i] = clamp01(stencil_px<FOP>(&p.v[i], &c.v[i], &n.v[i], k, m));
        }
        if (active) {
#pragma unroll
            for (int q = 0; q < NPX / 4; ++q)
                aadg_store_out(po + (size_t)y * W + x + 4 * q, make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]), STREAM);
        }
        p = c; c = n; n = nn;
    }
}

// LDS-tiled variant for any shape: grid (tiles_x, tiles_y, B*3)
constexpr int ST_W = 64, ST_H = 16;
template <int FOP>
__global__ __launch_bounds__(TO_THREADS) void k_fop_stencil_any(const float* __restrict__ in, float* __restrict__ out,
                                                                const float* __restrict__ mag, int mag_n,
                                                                const float* __restrict__ kern, int H, int W) {
    const int plane = blockIdx.z, b = plane / 3;
    const float* pin = in + (size_t)plane * H * W;
    float* po = out + (size_t)plane * H * W;
    __shared__ float tile[ST_H + 2][ST_W + 2 + 1];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * ST_W, y0 = blockIdx.y * ST_H;
    float k[9];
    stencil_kernel<FOP>(kern, mag, mag_n, k);
    for (int i = tid; i < (ST_H + 2) * (ST_W + 2); i += TO_THREADS) {
        const int ty = i / (ST_W + 2), tx = i - ty * (ST_W + 2);
        tile[ty][tx] = pin[(size_t)reflect1(y0 + ty - 1, H) * W + reflect1(x0 + tx - 1, W)];
    }
    __syncthreads();
    const float m = mag ? mag[mag_n == 1 ? 0 : b] : 0.0f;
    for (int i = tid; i < ST_H * ST_W; i += TO_THREADS) {
        const int ty = i / ST_W, tx = i - ty * ST_W;
        const int y = y0 + ty, x = x0 + tx;
        if (y >= H || x >= W) continue;
        po[(size_t)y * W + x] = clamp01(stencil_px<FOP>(&tile[ty][tx], &tile[ty + 1][tx], &tile[ty + 2][tx], k, m));
    }
}

// ---- affine warps -----------------------------------------------------------------------------------------------------------------
// forward (src -> dst) matrix A and offset t about the centre; the kernels apply the inverse
struct Affine { float i00, i01, i10, i11, tx, ty, cx, cy; };
template <int FOP>
__device__ __forceinline__ Affine affine_of(float m, int H, int W) {
    Affine A;
    A.tx = 0.f; A.ty = 0.f;
    A.cx = 0.5f * (float)(W - 1); A.cy = 0.5f * (float)(H - 1);
    if (FOP == AADG_FOP_ROTATE) {
        const float rad = m * 0.017453292519943295f;   // degrees, counter-clockwise (OpenCV convention)
        const float cs = cosf(rad), sn = sinf(rad);
        const float a00 = cs, a01 = sn, a10 = -sn, a11 = cs;
        const float det = a00 * a11 - a01 * a10;
        A.i00 = a11 / det; A.i01 = -a01 / det; A.i10 = -a10 / det; A.i11 = a00 / det;
    } else {
        // unit determinant: the inverse of [[1, a01], [a10, 1]] with one off-diagonal entry is [[1, -a01], [-a10, 1]] exactly
        // (the general formula divides by det = 1 - 0 = 1: the same bits, four divisions less per thread)
        A.i00 = 1.f; A.i11 = 1.f;
        A.i01 = FOP == AADG_FOP_SHEAR_X ? -m : -0.f;
        A.i10 = FOP == AADG_FOP_SHEAR_Y ? -m : -0.f;
        if (FOP == AADG_FOP_TRANSLATE_X) A.tx = m * (float)W;
        if (FOP == AADG_FOP_TRANSLATE_Y) A.ty = m * (float)H;
    }
    return A;
}

// rotate / shear_y: 32 x 32 output tile per workgroup, lane <-> 4 consecutive columns of one row.  The source footprint of the tile (bounding box
// of its four corners + margin: at most 56 x 50 pixels at any angle) is staged in LDS with 16-byte row loads, positions outside the
// image as zeros -- so a tap needs neither a validity test nor a clamp, and coordinates and weights are computed once for the three
// planes.  (4-byte gathers through the texture path deliver 0.16-0.28 of the HBM roofline for this access pattern, whatever the lane
// mapping.)  The planes go through ONE 11 KB buffer one after the other, the next plane's rows in flight (registers) while the
// current one is interpolated: with all three planes staged at once (34 KB) a CU held 4 workgroups and 74 % of the wave cycles were
// spent waiting; with one plane per workgroup the per-thread set-up (sine, cosine, four divisions, coordinates) tripled and the vector
// ALUs saturated.  grid (ceil(W/32), ceil(H/32), B).  Requires W % 4 == 0.
constexpr int ROT_T = 32, ROT_G = 14, ROT_PITCH = 4 * ROT_G + 1, ROT_ROWS = 50, ROT_PASSES = (ROT_ROWS + 15) / 16;
template <int FOP, bool STREAM>
__global__ __launch_bounds__(TO_THREADS) void k_fop_affine_tile(const float* __restrict__ in, float* __restrict__ out,
                                                                const float* __restrict__ mag, int mag_n, int H, int W, int B) {
    // 1-D grid, image-major per XCD (xcd_group_major): the tiles of one image share one L2
    const int gxn = (W + ROT_T - 1) / ROT_T, tiles = gxn * ((H + ROT_T - 1) / ROT_T);
    int b, t;
    xcd_group_major(blockIdx.x, tiles, B, b, t);
    const int HW = H * W, tid = threadIdx.x;
    const float* pin = in + (size_t)b * 3 * HW;
    float* po = out + (size_t)b * 3 * HW;
    const int tx0 = (t % gxn) * ROT_T, ty0 = (t / gxn) * ROT_T;
    const int x = tx0 + 4 * (tid & 7), y = ty0 + (tid >> 3);
    const float m = mag ? mag[mag_n == 1 ? 0 : b] : 0.0f;
    const Affine A = affine_of<FOP>(m, H, W);
    __shared__ float tile[ROT_PITCH * ROT_ROWS];
    // footprint: the map is affine, so the extremes over the tile are attained at its corners
    float sxmin = INFINITY, symin = INFINITY, sxmax = -INFINITY, symax = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dx = (float)(tx0 + (k & 1) * (ROT_T - 1)) - A.cx - A.tx, dy = (float)(ty0 + (k >> 1) * (ROT_T - 1)) - A.cy - A.ty;
        const float sx = A.i00 * dx + A.i01 * dy + A.cx, sy = A.i10 * dx + A.i11 * dy + A.cy;
        sxmin = fminf(sxmin, sx); sxmax = fmaxf(sxmax, sx); symin = fminf(symin, sy); symax = fmaxf(symax, sy);
    }
    // one pixel of margin for the rounding of interior coordinates; the box is NOT clipped to the image (outside = zeros)
    const float big = 1.0e6f;
    const int bx0 = ((int)floorf(fminf(fmaxf(sxmin, -big), big)) - 1) & ~3;
    const int by0 = (int)floorf(fminf(fmaxf(symin, -big), big)) - 1;
    // a rotation always fits; a shear beyond |m| ~ 0.45 does not: those workgroups gather their taps from global memory (uniform branch)
    // columns / rows of the box this tile really touches (a small angle needs 34 x 34 of the 56 x 50)
    const int need_x = (int)floorf(fminf(fmaxf(sxmax, -big), big)) + 3 - bx0, need_y = (int)floorf(fminf(fmaxf(symax, -big), big)) + 3 - by0;
    const bool fits = FOP == AADG_FOP_ROTATE || (need_x <= 4 * ROT_G && need_y <= ROT_ROWS);
    if (!fits) {
        if (x >= W || y >= H) return;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* pc = pin + c * HW;
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dx = (float)(x + i) - A.cx - A.tx, dy = (float)y - A.cy - A.ty;
                const float sx = A.i00 * dx + A.i01 * dy + A.cx, sy = A.i10 * dx + A.i11 * dy + A.cy;
                const float fx = floorf(fminf(fmaxf(sx, -big), big)), fy = floorf(fminf(fmaxf(sy, -big), big));
                const int x0 = (int)fx, y0 = (int)fy;
                const float wx = sx - fx, wy = sy - fy;
                auto at = [&](int yy, int xx) -> float { return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? pc[yy * W + xx] : 0.0f; };
                const float v = (1.f - wy) * ((1.f - wx) * at(y0, x0) + wx * at(y0, x0 + 1)) + wy * ((1.f - wx) * at(y0 + 1, x0) + wx * at(y0 + 1, x0 + 1));
                o[i] = clamp01(v);
            }
            aadg_store_out(po + c * HW + y * W + x, make_float4(o[0], o[1], o[2], o[3]), STREAM);
        }
        return;
    }
    // staging: 16 column groups x 16 rows per pass; W % 4 == 0 and gx % 4 == 0, so a group is inside or outside the image as a whole
    const int g = tid & 15, r0 = tid >> 4;
    const int gx = bx0 + 4 * g;
    const bool col_in = g < ROT_G && 4 * g < need_x && gx >= 0 && gx < W;
    int goff[ROT_PASSES];
    bool gin[ROT_PASSES];
#pragma unroll
    for (int p = 0; p < ROT_PASSES; ++p) {
        const int r = r0 + 16 * p, gy = by0 + r;
        gin[p] = col_in && r < ROT_ROWS && r < need_y && gy >= 0 && gy < H;
        goff[p] = gin[p] ? gy * W + gx : 0;
    }
    float4 stage[ROT_PASSES];
    auto fetch = [&](int c) {
#pragma unroll
        for (int p = 0; p < ROT_PASSES; ++p) {
            // clamped address, value replaced afterwards: no conditional load
            const float4 v = *reinterpret_cast<const float4*>(pin + c * HW + goff[p]);
            stage[p] = gin[p] ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0);
    // coordinates and weights of the lane's four pixels (shared by the planes)
    int o0[4];
    float wx[4], wy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float dx = (float)(x + i) - A.cx - A.tx, dy = (float)y - A.cy - A.ty;
        const float sx = A.i00 * dx + A.i01 * dy + A.cx, sy = A.i10 * dx + A.i11 * dy + A.cy;
        const float fx = floorf(fminf(fmaxf(sx, -big), big)), fy = floorf(fminf(fmaxf(sy, -big), big));
        wx[i] = sx - fx; wy[i] = sy - fy;
        // inside the staged box by construction (margin above); the clamp only guards the LDS against a NaN magnitude
        const int rx = min(max((int)fx - bx0, 0), ROT_PITCH - 2), ry = min(max((int)fy - by0, 0), ROT_ROWS - 2);
        o0[i] = ry * ROT_PITCH + rx;
    }
    const bool live = x < W && y < H;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (g < ROT_G) {
#pragma unroll
            for (int p = 0; p < ROT_PASSES; ++p) {
                const int r = r0 + 16 * p;
                if (r < ROT_ROWS) {
                    float* d = &tile[r * ROT_PITCH + 4 * g];
                    d[0] = stage[p].x; d[1] = stage[p].y; d[2] = stage[p].z; d[3] = stage[p].w;
                }
            }
        }
        __syncthreads();
        if (c < 2) fetch(c + 1);                     // in flight during the interpolation below
        if (live) {
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = tile[o0[i]], bq = tile[o0[i] + 1], cq = tile[o0[i] + ROT_PITCH], d = tile[o0[i] + ROT_PITCH + 1];
                const float v = (1.f - wy[i]) * ((1.f - wx[i]) * a + wx[i] * bq) + wy[i] * ((1.f - wx[i]) * cq + wx[i] * d);
                o[i] = clamp01(v);
            }
            aadg_store_out(po + c * HW + y * W + x, make_float4(o[0], o[1], o[2], o[3]), STREAM);
        }
        if (c < 2) __syncthreads();                  // everyone has read plane c before the buffer is overwritten
    }
}

// Row-preserving maps (translate_x / translate_y / shear_x) without LDS: a lane owns 4 consecutive output pixels of one row, the
// source row pair (y0, y0 + 1) is the same for the whole lane and consecutive outputs read consecutive source columns: per plane one
// 16-byte load at a 4-byte-aligned address per source row + one 4-byte load.  Taps whose weight is exactly zero are not loaded
// (0 * tap adds nothing for finite taps): translate_x / shear_x keep the row (sy = y exactly, wy = 0), translate_y keeps the column
// (wx = 0).  Every pixel still evaluates its own coordinates with the generic formula; a lane whose pixels do not have that structure
// (float rounding at an integer boundary, the image border) takes the per-tap path, a lane that maps outside the image writes zeros.
// grid (ceil(W/256), ceil(H/4), B).  Requires W % 4 == 0.
typedef float aadg_f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float4 load4_unaligned(const float* p) {
    const aadg_f32x4_u q = *reinterpret_cast<const aadg_f32x4_u*>(p);
    return make_float4(q.x, q.y, q.z, q.w);
}

template <int FOP, bool STREAM>
__global__ __launch_bounds__(TO_THREADS) void k_fop_warp_rows(const float* __restrict__ in, float* __restrict__ out,
                                                              const float* __restrict__ mag, int mag_n, int H, int W) {
    const int b = blockIdx.z, HW = H * W;
    const float* pin = in + (size_t)b * 3 * HW;
    float* po = out + (size_t)b * 3 * HW;
    static_assert(FOP == AADG_FOP_TRANSLATE_X || FOP == AADG_FOP_TRANSLATE_Y || FOP == AADG_FOP_SHEAR_X, "row-preserving maps only");
    constexpr bool ROW_ONLY = FOP == AADG_FOP_TRANSLATE_X || FOP == AADG_FOP_SHEAR_X, COL_ONLY = FOP == AADG_FOP_TRANSLATE_Y;
    const int x = blockIdx.x * 256 + 4 * (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float m = mag ? mag[mag_n == 1 ? 0 : b] : 0.0f;
    const Affine A = affine_of<FOP>(m, H, W);
    int x0[4], y0[4];
    float wx[4], wy[4];
    const float big = 1.0e6f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float dx = (float)(x + i) - A.cx - A.tx, dy = (float)y - A.cy - A.ty;
        const float sx = A.i00 * dx + A.i01 * dy + A.cx, sy = A.i10 * dx + A.i11 * dy + A.cy;
        const float fx = floorf(fminf(fmaxf(sx, -big), big)), fy = floorf(fminf(fmaxf(sy, -big), big));
        x0[i] = (int)fx; y0[i] = (int)fy;
        wx[i] = sx - fx; wy[i] = sy - fy;
    }
    bool fast;
    {
        fast = x0[1] == x0[0] + 1 && x0[2] == x0[0] + 2 && x0[3] == x0[0] + 3 && y0[1] == y0[0] && y0[2] == y0[0] && y0[3] == y0[0] &&
               x0[0] >= 0 && x0[0] + 4 < W && y0[0] >= 0 && y0[0] + 1 < H;      // (one column / row more than a zero-weight lane needs)
        if (ROW_ONLY) fast = fast && wy[0] == 0.f && wy[1] == 0.f && wy[2] == 0.f && wy[3] == 0.f;
        if (COL_ONLY) fast = fast && wx[0] == 0.f && wx[1] == 0.f && wx[2] == 0.f && wx[3] == 0.f;
    }
    // every tap of every pixel of the lane outside the image: zeros without a load (the region a translation / shear moves in)
    bool outside = true;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        outside = outside && (y0[i] + 1 < 0 || y0[i] >= H || x0[i] + 1 < 0 || x0[i] >= W);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* pc = pin + c * HW;
        float a[4], bq[4], cq[4], d[4];
        if (fast) {
            {
                // taps whose weight is exactly zero are not loaded (0 * tap adds nothing for finite taps): translate_x / shear_x keep the
                // row (sy = y exactly, wy = 0), translate_y keeps the column (wx = 0) -- part of `fast`, so no per-lane branches here
                const float* r0 = pc + y0[0] * W + x0[0];
                const float4 u0 = load4_unaligned(r0);
                float4 u1 = make_float4(0.f, 0.f, 0.f, 0.f);
                float e0 = 0.f, e1 = 0.f;
                if (!ROW_ONLY) u1 = load4_unaligned(r0 + W);
                if (!COL_ONLY) e0 = r0[4];
                if (!COL_ONLY && !ROW_ONLY) e1 = r0[W + 4];
                a[0] = u0.x; a[1] = u0.y; a[2] = u0.z; a[3] = u0.w;
                bq[0] = u0.y; bq[1] = u0.z; bq[2] = u0.w; bq[3] = e0;
                cq[0] = u1.x; cq[1] = u1.y; cq[2] = u1.z; cq[3] = u1.w;
                d[0] = u1.y; d[1] = u1.z; d[2] = u1.w; d[3] = e1;
            }
        } else if (outside) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = bq[i] = cq[i] = d[i] = 0.0f;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                auto at = [&](int yy, int xx) -> float { return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? pc[yy * W + xx] : 0.0f; };
                a[i] = at(y0[i], x0[i]); bq[i] = at(y0[i], x0[i] + 1); cq[i] = at(y0[i] + 1, x0[i]); d[i] = at(y0[i] + 1, x0[i] + 1);
            }
        }
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = (1.f - wy[i]) * ((1.f - wx[i]) * a[i] + wx[i] * bq[i]) + wy[i] * ((1.f - wx[i]) * cq[i] + wx[i] * d[i]);
            o[i] = clamp01(v);
        }
        aadg_store_out(po + c * HW + y * W + x, make_float4(o[0], o[1], o[2], o[3]), STREAM);
    }
}

// flips: 16 bytes per lane and plane row; grid (chunks, B*3)
template <int FOP, bool STREAM>
__global__ __launch_bounds__(TO_THREADS) void k_fop_flip(const float* __restrict__ in, float* __restrict__ out, int H, int W) {
    const int plane = blockIdx.y, q = W >> 2, n = H * q;
    const float* pin = in + (size_t)plane * H * W;
    float* po = out + (size_t)plane * H * W;
    for (int g = blockIdx.x * TO_THREADS + threadIdx.x; g < n; g += gridDim.x * TO_THREADS) {
        const int y = g / q, xg = g - y * q;
        float4 v;
        if (FOP == AADG_FOP_HFLIP) {
            const float* s = pin + (size_t)y * W + (W - 4 - 4 * xg);
            const float4 t = STREAM ? aadg_load_stream(s) : *reinterpret_cast<const float4*>(s);
            v = make_float4(t.w, t.z, t.y, t.x);
        } else {
            const float* s = pin + (size_t)(H - 1 - y) * W + 4 * xg;
            v = STREAM ? aadg_load_stream(s) : *reinterpret_cast<const float4*>(s);
        }
        v.x = clamp01(v.x); v.y = clamp01(v.y); v.z = clamp01(v.z); v.w = clamp01(v.w);
        aadg_store_out(po + (size_t)y * W + 4 * xg, v, STREAM);
    }
}

// one pixel per lane, any shape: grid (chunks, B*3)
template <int FOP>
__global__ __launch_bounds__(TO_THREADS) void k_fop_warp_any(const float* __restrict__ in, float* __restrict__ out,
                                                             const float* __restrict__ mag, int mag_n, int H, int W) {
    const int plane = blockIdx.y, b = plane / 3;
    const float* pin = in + (size_t)plane * H * W;
    float* po = out + (size_t)plane * H * W;
    const float m = mag ? mag[mag_n == 1 ? 0 : b] : 0.0f;
    const Affine A = affine_of<FOP>(m, H, W);
    const int HW = H * W;
    for (int i = blockIdx.x * TO_THREADS + threadIdx.x; i < HW; i += gridDim.x * TO_THREADS) {
        const int y = i / W, x = i - y * W;
        float v;
        if (FOP == AADG_FOP_HFLIP) v = pin[(size_t)y * W + (W - 1 - x)];
        else if (FOP == AADG_FOP_VFLIP) v = pin[(size_t)(H - 1 - y) * W + x];
        else {
            const float dx = (float)x - A.cx - A.tx, dy = (float)y - A.cy - A.ty;
            const float sx = A.i00 * dx + A.i01 * dy + A.cx, sy = A.i10 * dx + A.i11 * dy + A.cy;
            const float fx = floorf(sx), fy = floorf(sy);
            const int x0 = (int)fx, y0 = (int)fy;
            const float wx = sx - fx, wy = sy - fy;
            auto at = [&](int yy, int xx) -> float {
                return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? pin[(size_t)yy * W + xx] : 0.0f;
            };
            v = (1.f - wy) * ((1.f - wx) * at(y0, x0) + wx * at(y0, x0 + 1)) + wy * ((1.f - wx) * at(y0 + 1, x0) + wx * at(y0 + 1, x0 + 1));
        }
        po[i] = clamp01(v);
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------------
int stat_chunks(int HW) {
    int c = (HW / 4 + TO_THREADS * 4 - 1) / (TO_THREADS * 4);
    return c < 1 ? 1 : (c > STAT_CHUNKS_MAX ? STAT_CHUNKS_MAX : c);
}
int point_chunks(int HW) {                 // 2 groups of 4 pixels per lane and iteration, ~2 iterations per lane
    int c = (HW / 4 + TO_THREADS * 4 - 1) / (TO_THREADS * 4);
    return c < 1 ? 1 : (c > 512 ? 512 : c);
}
struct FopWs { size_t part, hist, total; };
FopWs fop_ws(int B, int HW) {
    FopWs w;
    size_t o = 0;
    w.part = o; o = aadg_align_up(o + (size_t)B * STAT_CHUNKS_MAX * sizeof(Part), 256);
    w.hist = o; o = aadg_align_up(o + (size_t)(B * 3 + 1) * 256 * 4, 256);
    w.total = o;
    (void)HW;
    return w;
}

constexpr size_t FOP_STREAM_BYTES = (size_t)128 << 20;     // batches beyond this cannot stay cached until their consumer runs

template <int FOP>
int launch_point(bool vec, bool stream_io, const float* in, float* out, const float* mag, int mag_n, const int32_t* perm, const Part* part,
                 int chunks, const unsigned int* hist, int B, int HW, hipStream_t st, int* order = nullptr) {
    const dim3 g(point_chunks(HW), B);
    if (order != nullptr) {
        hipLaunchKernelGGL(k_fop_pair_order, dim3(1), dim3(256), 0, st, perm, B, order);
        AADG_LAUNCH_CHECK();
    }
    if (!vec) hipLaunchKernelGGL(k_fop_point_any<FOP>, g, dim3(TO_THREADS), 0, st, in, out, mag, mag_n, perm, part, chunks, hist, HW);
    else if (stream_io) hipLaunchKernelGGL((k_fop_point<FOP, true>), g, dim3(TO_THREADS), 0, st, in, out, mag, mag_n, perm, part, chunks, hist, HW, order);
    else hipLaunchKernelGGL((k_fop_point<FOP, false>), g, dim3(TO_THREADS), 0, st, in, out, mag, mag_n, perm, part, chunks, hist, HW, order);
    AADG_LAUNCH_CHECK();
    return 0;
}
template <int FOP>
int launch_stats(bool vec, const float* in, int B, int HW, int chunks, Part* part, unsigned int* hist, hipStream_t st) {
    if (vec) hipLaunchKernelGGL((k_fop_stats<FOP, true>), dim3(chunks, B), dim3(TO_THREADS), 0, st, in, HW, part, hist);
    else hipLaunchKernelGGL((k_fop_stats<FOP, false>), dim3(chunks, B), dim3(TO_THREADS), 0, st, in, HW, part, hist);
    AADG_LAUNCH_CHECK();
    return 0;
}
template <int FOP>
int launch_stencil(bool vec, bool stream_io, const float* in, float* out, const float* mag, int mag_n, const float* kern, int B, int H, int W,
                   hipStream_t st) {
    if (vec) {
        // measured at [144,3,512,512] (4 / 8 pixels per lane x 8 / 16 / 32-row strips): 4 x 8 rows 0.164 ms, 4 x 16 0.169, 4 x 32 0.177; 8 pixels per
        // lane 0.22 ms -- a lane's two 16-byte accesses per row then leave every wave-level access half-coalesced (32-byte lane stride)
        const dim3 g((W + 255) / 256, ((H + 7) / 8 + 3) / 4, B * 3);
        if (stream_io) hipLaunchKernelGGL((k_fop_stencil_rows<FOP, 4, 8, true>), g, dim3(TO_THREADS), 0, st, in, out, mag, mag_n, kern, H, W);
        else hipLaunchKernelGGL((k_fop_stencil_rows<FOP, 4, 8, false>), g, dim3(TO_THREADS), 0, st, in, out, mag, mag_n, kern, H, W);
    } else {
        const dim3 g((W + ST_W - 1) / ST_W, (H + ST_H - 1) / ST_H, B * 3);
        hipLaunchKernelGGL(k_fop_stencil_any<FOP>, g, dim3(TO_THREADS), 0, st, in, out, mag, mag_n, kern, H, W);
    }
    AADG_LAUNCH_CHECK();
    return 0;
}
template <int FOP>
int launch_warp(bool vec, bool stream_io, const float* in, float* out, const float* mag, int mag_n, int B, int H, int W, hipStream_t st) {
    if (!vec) {
        hipLaunchKernelGGL(k_fop_warp_any<FOP>, dim3(point_chunks(H * W) * 2, B * 3), dim3(TO_THREADS), 0, st, in, out, mag, mag_n, H, W);
    } else if constexpr (FOP == AADG_FOP_HFLIP || FOP == AADG_FOP_VFLIP) {
        const dim3 g(point_chunks(H * W), B * 3);
        if (stream_io) hipLaunchKernelGGL((k_fop_flip<FOP, true>), g, dim3(TO_THREADS), 0, st, in, out, H, W);
        else hipLaunchKernelGGL((k_fop_flip<FOP, false>), g, dim3(TO_THREADS), 0, st, in, out, H, W);
    } else if constexpr (FOP == AADG_FOP_ROTATE || FOP == AADG_FOP_SHEAR_Y) {
        const dim3 g((unsigned)(((W + ROT_T - 1) / ROT_T) * ((H + ROT_T - 1) / ROT_T)) * (unsigned)B);
        if (stream_io) hipLaunchKernelGGL((k_fop_affine_tile<FOP, true>), g, dim3(TO_THREADS), 0, st, in, out, mag, mag_n, H, W, B);
        else hipLaunchKernelGGL((k_fop_affine_tile<FOP, false>), g, dim3(TO_THREADS), 0, st, in, out, mag, mag_n, H, W, B);
    } else {
        const dim3 g((W + 255) / 256, (H + 3) / 4, B);
        if (stream_io) hipLaunchKernelGGL((k_fop_warp_rows<FOP, true>), g, dim3(TO_THREADS), 0, st, in, out, mag, mag_n, H, W);
        else hipLaunchKernelGGL((k_fop_warp_rows<FOP, false>), g, dim3(TO_THREADS), 0, st, in, out, mag, mag_n, H, W);
    }
    AADG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" size_t aadg_fop_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return fop_ws(B, H * W).total;
}

extern "C" int aadg_fop_f32(int fop, const float* in, float* out, const float* mag, int mag_n, const float* kernel3x3,
                            const int32_t* perm, int B, int C, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !out || fop < 0 || fop >= AADG_FOP_COUNT) return AADG_E_BADARG;
    if (B <= 0 || C != 3 || H <= 0 || W <= 0) return AADG_E_BADARG;
    if (mag && mag_n != 1 && mag_n != B) return AADG_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int HW = H * W;
    const bool needs_mag = !(fop == AADG_FOP_INVERT || fop == AADG_FOP_GRAY || fop == AADG_FOP_AUTO_CONTRAST ||
                             fop == AADG_FOP_EQUALIZE || fop == AADG_FOP_HFLIP || fop == AADG_FOP_VFLIP);
    if (needs_mag && !mag && !(fop == AADG_FOP_GAUSSIAN_BLUR3X3 && kernel3x3)) return AADG_E_BADARG;
    if (fop == AADG_FOP_SAMPLE_PAIRING && !perm) return AADG_E_BADARG;
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const bool vec_px = aligned && (HW & 3) == 0;                  // 4 consecutive pixels of a plane per lane
    const bool vec_row = aligned && (W & 3) == 0 && W >= 8;        // ... that lie in one image row
    // streaming accesses when the batch cannot stay in the 256 MiB Infinity Cache until whoever reads the output runs
    const bool stream_io = (size_t)B * 3 * HW * sizeof(float) > FOP_STREAM_BYTES;
#define FOP_CASE(F, CALL) case F: return CALL
    if (fop == AADG_FOP_SHARPNESS || fop == AADG_FOP_GAUSSIAN_BLUR3X3) {
        if (in == out || H < 2 || W < 2) return AADG_E_BADARG;
        if (fop == AADG_FOP_SHARPNESS) return launch_stencil<AADG_FOP_SHARPNESS>(vec_row, stream_io, in, out, mag, mag_n, kernel3x3, B, H, W, st);
        return launch_stencil<AADG_FOP_GAUSSIAN_BLUR3X3>(vec_row, stream_io, in, out, mag, mag_n, kernel3x3, B, H, W, st);
    }
    if (fop >= AADG_FOP_SHEAR_X) {
        if (in == out) return AADG_E_BADARG;
        switch (fop) {
            FOP_CASE(AADG_FOP_SHEAR_X, launch_warp<AADG_FOP_SHEAR_X>(vec_row, stream_io, in, out, mag, mag_n, B, H, W, st));
            FOP_CASE(AADG_FOP_SHEAR_Y, launch_warp<AADG_FOP_SHEAR_Y>(vec_row, stream_io, in, out, mag, mag_n, B, H, W, st));
            FOP_CASE(AADG_FOP_TRANSLATE_X, launch_warp<AADG_FOP_TRANSLATE_X>(vec_row, stream_io, in, out, mag, mag_n, B, H, W, st));
            FOP_CASE(AADG_FOP_TRANSLATE_Y, launch_warp<AADG_FOP_TRANSLATE_Y>(vec_row, stream_io, in, out, mag, mag_n, B, H, W, st));
            FOP_CASE(AADG_FOP_ROTATE, launch_warp<AADG_FOP_ROTATE>(vec_row, stream_io, in, out, mag, mag_n, B, H, W, st));
            FOP_CASE(AADG_FOP_HFLIP, launch_warp<AADG_FOP_HFLIP>(vec_row, stream_io, in, out, mag, mag_n, B, H, W, st));
            default: return launch_warp<AADG_FOP_VFLIP>(vec_row, stream_io, in, out, mag, mag_n, B, H, W, st);
        }
    }
    const Part* part = nullptr;
    const unsigned int* hist = nullptr;
    int chunks = 0;
    if (fop == AADG_FOP_CONTRAST || fop == AADG_FOP_AUTO_CONTRAST || fop == AADG_FOP_EQUALIZE) {
        if (!ws) return AADG_E_BADARG;
        const FopWs w = fop_ws(B, HW);
        if (ws_bytes < w.total) return AADG_E_WORKSPACE;
        uint8_t* ws8 = reinterpret_cast<uint8_t*>(ws);
        Part* p = reinterpret_cast<Part*>(ws8 + w.part);
        unsigned int* h = reinterpret_cast<unsigned int*>(ws8 + w.hist);
        chunks = stat_chunks(HW);
        int rc;
        if (fop == AADG_FOP_EQUALIZE) {
            AADG_HIP_TRY(hipMemsetAsync(h, 0, (size_t)(B * 3 + 1) * 256 * 4, st));
            rc = launch_stats<AADG_FOP_EQUALIZE>(vec_px, in, B, HW, chunks, p, h, st);
        } else if (fop == AADG_FOP_CONTRAST) rc = launch_stats<AADG_FOP_CONTRAST>(vec_px, in, B, HW, chunks, p, h, st);
        else rc = launch_stats<AADG_FOP_AUTO_CONTRAST>(vec_px, in, B, HW, chunks, p, h, st);
        if (rc) return rc;
        part = p;
        hist = h;
    }
    if (fop == AADG_FOP_SAMPLE_PAIRING && in == out) return AADG_E_BADARG;
    switch (fop) {
        FOP_CASE(AADG_FOP_INVERT, launch_point<AADG_FOP_INVERT>(vec_px, stream_io, in, out, mag, mag_n, perm, part, chunks, hist, B, HW, st));
        FOP_CASE(AADG_FOP_SOLARIZE, launch_point<AADG_FOP_SOLARIZE>(vec_px, stream_io, in, out, mag, mag_n, perm, part, chunks, hist, B, HW, st));
        FOP_CASE(AADG_FOP_POSTERIZE, launch_point<AADG_FOP_POSTERIZE>(vec_px, stream_io, in, out, mag, mag_n, perm, part, chunks, hist, B, HW, st));
        FOP_CASE(AADG_FOP_GRAY, launch_point<AADG_FOP_GRAY>(vec_px, stream_io, in, out, mag, mag_n, perm, part, chunks, hist, B, HW, st));
        FOP_CASE(AADG_FOP_CONTRAST, launch_point<AADG_FOP_CONTRAST>(vec_px, stream_io, in, out, mag, mag_n, perm, part, chunks, hist, B, HW, st));
        FOP_CASE(AADG_FOP_AUTO_CONTRAST, launch_point<AADG_FOP_AUTO_CONTRAST>(vec_px, stream_io, in, out, mag, mag_n, perm, part, chunks, hist, B, HW, st));
        FOP_CASE(AADG_FOP_SATURATE, launch_point<AADG_FOP_SATURATE>(vec_px, stream_io, in, out, mag, mag_n, perm, part, chunks, hist, B, HW, st));
        FOP_CASE(AADG_FOP_BRIGHTNESS, launch_point<AADG_FOP_BRIGHTNESS>(vec_px, stream_io, in, out, mag, mag_n, perm, part, chunks, hist, B, HW, st));
        FOP_CASE(AADG_FOP_HUE, launch_point<AADG_FOP_HUE>(vec_px, stream_io, in, out, mag, mag_n, perm, part, chunks, hist, B, HW, st));
        case AADG_FOP_SAMPLE_PAIRING: {
            // with a workspace (>= 4 B bytes) the samples are walked along the cycles of perm; without one, in index order
            int* order = (vec_px && stream_io && ws != nullptr && ws_bytes >= (size_t)B * 4 && B <= PAIR_ORDER_MAX) ? reinterpret_cast<int*>(ws) : nullptr;
            return launch_point<AADG_FOP_SAMPLE_PAIRING>(vec_px, stream_io, in, out, mag, mag_n, perm, part, chunks, hist, B, HW, st, order);
        }
        default: return launch_point<AADG_FOP_EQUALIZE>(vec_px, stream_io, in, out, mag, mag_n, perm, part, chunks, hist, B, HW, st);
    }
#undef FOP_CASE
}

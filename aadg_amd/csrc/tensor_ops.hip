// Float tensor ops on gfx950 (a8-a13; kernels K2-K5).
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
// All kernels are HBM-bound: one read and one write of the image (24*H*W bytes per image), a second read
// for the statistics ops.  A thread owns 4 consecutive pixels of all three planes (3 float4 loads, 3 float4
// stores); statistics are reduced per block (wave shuffles / LDS histograms) into the workspace and
// finished by a one-block-per-(b,c) kernel.
#include "common.h"

namespace {

constexpr int TO_THREADS = 256;

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

struct Stats {           // per (b,c) workspace record
    float mn, mx;        // auto_contrast: min / max of clamp(x)*255
    float mean;          // contrast: floor(mean(gray(255x)) + 0.5) / 255 (stored per b in channel-0 record)
    float pad;
};

// ---- pointwise / LUT kernel: grid (chunks, B) ------------------------------------------------------------------
__global__ __launch_bounds__(TO_THREADS) void k_fop_point(int fop, const float* __restrict__ in, float* __restrict__ out,
                                                          const float* __restrict__ mag, int mag_n, const int* __restrict__ perm,
                                                          const Stats* __restrict__ stats, const float* __restrict__ lut,
                                                          int H, int W) {
    const int b = blockIdx.y;
    const int HW = H * W;
    const size_t base = (size_t)b * 3 * HW;
    const float m = mag ? mag[mag_n == 1 ? 0 : b] : 0.0f;
    const float* pin = in + base;
    const float* pin2 = perm ? in + (size_t)perm[b] * 3 * HW : pin;
    float* po = out + base;
    const Stats* st = stats ? stats + (size_t)b * 3 : nullptr;
    const int nplanes = gridDim.y * 3;
    const bool vec = (HW & 3) == 0;
    const int step = vec ? 4 : 1;
    for (int i0 = (blockIdx.x * TO_THREADS + threadIdx.x) * step; i0 < HW; i0 += gridDim.x * TO_THREADS * step) {
        float r[4], g[4], bl[4];
        if (vec) {
            const float4 a = *reinterpret_cast<const float4*>(pin + i0);
            const float4 c = *reinterpret_cast<const float4*>(pin + HW + i0);
            const float4 d = *reinterpret_cast<const float4*>(pin + 2 * HW + i0);
            r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
            g[0] = c.x; g[1] = c.y; g[2] = c.z; g[3] = c.w;
            bl[0] = d.x; bl[1] = d.y; bl[2] = d.z; bl[3] = d.w;
        } else {
            r[0] = pin[i0]; g[0] = pin[HW + i0]; bl[0] = pin[2 * HW + i0];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= step) break;
            float R = r[k], G = g[k], Bv = bl[k];
            switch (fop) {
                case AADG_FOP_INVERT: R = 1.0f - R; G = 1.0f - G; Bv = 1.0f - Bv; break;
                case AADG_FOP_SOLARIZE:
                    R = R < m ? R : 1.0f - R; G = G < m ? G : 1.0f - G; Bv = Bv < m ? Bv : 1.0f - Bv; break;
                case AADG_FOP_POSTERIZE:
                    R = (float)(long long)(R * 255.0f) / 255.0f; G = (float)(long long)(G * 255.0f) / 255.0f;
                    Bv = (float)(long long)(Bv * 255.0f) / 255.0f; break;
                case AADG_FOP_GRAY: { const float y = gray_of(R, G, Bv); R = G = Bv = y; break; }
                case AADG_FOP_SATURATE: {
                    const float y = gray_of(R, G, Bv), a = 1.0f - m;
                    R = blendf(R, y, a); G = blendf(G, y, a); Bv = blendf(Bv, y, a); break;
                }
                case AADG_FOP_BRIGHTNESS: {
                    const float a = 1.0f - m;
                    R = blendf(R, 0.0f, a); G = blendf(G, 0.0f, a); Bv = blendf(Bv, 0.0f, a); break;
                }
                case AADG_FOP_CONTRAST: {
                    const float mean = st[0].mean, a = 1.0f - m;
                    R = blendf(R, mean, a); G = blendf(G, mean, a); Bv = blendf(Bv, mean, a); break;
                }
                case AADG_FOP_SAMPLE_PAIRING: {
                    const int i = i0 + k;
                    R = (1.0f - m) * R + m * pin2[i]; G = (1.0f - m) * G + m * pin2[HW + i];
                    Bv = (1.0f - m) * Bv + m * pin2[2 * HW + i]; break;
                }
                case AADG_FOP_AUTO_CONTRAST: {
                    float* ch[3] = {&R, &G, &Bv};
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v = clamp01(*ch[c]) * 255.0f;
                        const float scale = 255.0f / (st[c].mx - st[c].mn + 0.1f);
                        *ch[c] = floorf(((float)(long long)v - st[c].mn) * scale) / 255.0f;
                    }
                    break;
                }
                case AADG_FOP_EQUALIZE: {
                    float* ch[3] = {&R, &G, &Bv};
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        // output = lut.view(-1)[shifted.long()], shifted = clamp(x)*255 + 256*plane in float32
                        const float shifted = clamp01(*ch[c]) * 255.0f + 256.0f * (float)(b * 3 + c);
                        int idx = (int)shifted;
                        idx = idx < nplanes * 256 ? idx : nplanes * 256 - 1;
                        *ch[c] = lut[idx];
                    }
                    break;
                }
                case AADG_FOP_HUE: {
                    float h, s, v;
                    rgb2hsv(R, G, Bv, h, s, v);
                    h = h + m;
                    h = h - floorf(h);          // python % 1
                    hsv2rgb(h, s, v, R, G, Bv);
                    break;
                }
                default: break;
            }
            r[k] = clamp01(R); g[k] = clamp01(G); bl[k] = clamp01(Bv);
        }
        if (vec) {
            *reinterpret_cast<float4*>(po + i0) = make_float4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<float4*>(po + HW + i0) = make_float4(g[0], g[1], g[2], g[3]);
            *reinterpret_cast<float4*>(po + 2 * HW + i0) = make_float4(bl[0], bl[1], bl[2], bl[3]);
        } else {
            po[i0] = r[0]; po[HW + i0] = g[0]; po[2 * HW + i0] = bl[0];
        }
    }
}

// ---- statistics pass 1: grid (chunks, B); partial records in ws --------------------------------------------------
struct Part {
    float mn[3], mx[3];
    double gsum;
    unsigned int pad[2];
};

__global__ __launch_bounds__(TO_THREADS) void k_fop_stats(int fop, const float* __restrict__ in, int H, int W, Part* __restrict__ part,
                                                          unsigned int* __restrict__ hist /*[B*3][256]*/) {
    const int b = blockIdx.y, HW = H * W, tid = threadIdx.x;
    const float* pin = in + (size_t)b * 3 * HW;
    __shared__ unsigned int sh[3 * 256];
    const bool do_hist = fop == AADG_FOP_EQUALIZE;
    if (do_hist) {
        for (int i = tid; i < 768; i += TO_THREADS) sh[i] = 0;
        __syncthreads();
    }
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    double gs = 0.0;
    for (int i = blockIdx.x * TO_THREADS + tid; i < HW; i += gridDim.x * TO_THREADS) {
        const float r = pin[i], g = pin[HW + i], bl = pin[2 * HW + i];
        if (fop == AADG_FOP_CONTRAST) {
            gs += (double)gray_of(r * 255.0f, g * 255.0f, bl * 255.0f);
        } else {
            const float v[3] = {clamp01(r) * 255.0f, clamp01(g) * 255.0f, clamp01(bl) * 255.0f};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                mn[c] = fminf(mn[c], v[c]); mx[c] = fmaxf(mx[c], v[c]);
                if (do_hist) {
                    // torch.histc(bins=256*BC, min=0, max=256*BC-1) on v + 256*(b*3+c): per-plane this is
                    // bin = int(shifted * nb / (nb - 1)) - 256*(b*3+c), evaluated in float32 like ATen
                    const int plane = b * 3 + c;
                    const float nb = (float)(gridDim.y * 3 * 256);
                    const float shifted = v[c] + 256.0f * (float)plane;
                    int bin = (int)(shifted * nb / (nb - 1.0f));
                    if (bin >= (int)nb) bin = (int)nb - 1;
                    bin -= 256 * plane;
                    // a value can land in the neighbouring plane's first bin; torch counts it there
                    if (bin >= 0 && bin < 256) atomicAdd(&sh[c * 256 + bin], 1u);
                    else atomicAdd(&hist[(size_t)(plane + (bin < 0 ? -1 : 1)) * 256 + (bin < 0 ? bin + 256 : bin - 256)], 1u);
                }
            }
        }
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
    if (do_hist)
        for (int i = tid; i < 768; i += TO_THREADS)
            if (sh[i]) atomicAdd(&hist[(size_t)(b * 3 + i / 256) * 256 + (i & 255)], sh[i]);
}

// ---- statistics pass 2: grid B, 256 threads ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fop_stats_final(int fop, const Part* __restrict__ part, int chunks, int HW,
                                                         const unsigned int* __restrict__ hist, Stats* __restrict__ stats,
                                                         float* __restrict__ lut) {
    const int b = blockIdx.x, i = threadIdx.x;
    if (i == 0) {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        double gs = 0.0;
        for (int k = 0; k < chunks; ++k) {
            const Part& p = part[(size_t)b * chunks + k];
            for (int c = 0; c < 3; ++c) { mn[c] = fminf(mn[c], p.mn[c]); mx[c] = fmaxf(mx[c], p.mx[c]); }
            gs += p.gsum;
        }
        for (int c = 0; c < 3; ++c) {
            stats[b * 3 + c].mn = mn[c];
            stats[b * 3 + c].mx = mx[c];
            stats[b * 3 + c].mean = floorf((float)(gs / (double)HW) + 0.5f) / 255.0f;
            stats[b * 3 + c].pad = 0.f;
        }
    }
    if (fop != AADG_FOP_EQUALIZE) return;
    // equalize LUT per plane: cdf = cumsum(h); step = floor((cdf[-1]-h[-1])/255);
    // lut[k] = floor((cdf_exclusive[k] + floor(step/2)) / (step + 0.1)) / 255
    __shared__ float scan[256];
    for (int c = 0; c < 3; ++c) {
        const float hv = (float)hist[(size_t)(b * 3 + c) * 256 + i];
        __syncthreads();
        scan[i] = hv;
        for (int o = 1; o < 256; o <<= 1) {
            __syncthreads();
            const float t = i >= o ? scan[i - o] : 0.0f;
            __syncthreads();
            scan[i] += t;
        }
        __syncthreads();
        const float total = scan[255], last = (float)hist[(size_t)(b * 3 + c) * 256 + 255];
        const float step = floorf((total - last) / 255.0f);
        const float excl = scan[i] - hv;
        lut[(size_t)(b * 3 + c) * 256 + i] = floorf((excl + floorf(step / 2.0f)) / (step + 0.1f)) / 255.0f;
    }
}

// ---- 3x3 depthwise stencil with reflect padding: grid (tiles_x, tiles_y, B*3) ---------------------------------------
constexpr int ST_W = 64, ST_H = 16;
__global__ __launch_bounds__(TO_THREADS) void k_fop_stencil(int fop, const float* __restrict__ in, float* __restrict__ out,
                                                            const float* __restrict__ mag, int mag_n,
                                                            const float* __restrict__ kern, int H, int W) {
    const int plane = blockIdx.z, b = plane / 3;
    const float* pin = in + (size_t)plane * H * W;
    float* po = out + (size_t)plane * H * W;
    __shared__ float tile[ST_H + 2][ST_W + 2 + 1];
    __shared__ float k[9];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * ST_W, y0 = blockIdx.y * ST_H;
    if (tid < 9) {
        if (kern) k[tid] = kern[tid];
        else if (fop == AADG_FOP_SHARPNESS) k[tid] = (tid == 4 ? 5.0f : 1.0f) / 13.0f;
    }
    if (!kern && fop == AADG_FOP_GAUSSIAN_BLUR3X3 && tid == 0) {
        // kernels._gaussian: sigma := mean(mag)^2 is used as the variance
        float sm = 0.f;
        for (int i = 0; i < mag_n; ++i) sm += mag[i];
        sm /= (float)mag_n;
        const float var = sm * sm;
        float w[9], tot = 0.f;
        for (int j = 0; j < 9; ++j) {
            const float dy = (float)(j / 3 - 1), dx = (float)(j % 3 - 1);
            w[j] = expf(-(dx * dx + dy * dy) / (2.0f * var));
            tot += w[j];
        }
        for (int j = 0; j < 9; ++j) k[j] = w[j] / tot;
    }
    for (int i = tid; i < (ST_H + 2) * (ST_W + 2); i += TO_THREADS) {
        const int ty = i / (ST_W + 2), tx = i - ty * (ST_W + 2);
        int y = y0 + ty - 1, x = x0 + tx - 1;
        // F.pad(..., 'reflect'): -1 -> 1, H -> H-2
        y = y < 0 ? -y : (y >= H ? 2 * H - 2 - y : y);
        x = x < 0 ? -x : (x >= W ? 2 * W - 2 - x : x);
        y = min(max(y, 0), H - 1); x = min(max(x, 0), W - 1);
        tile[ty][tx] = pin[(size_t)y * W + x];
    }
    __syncthreads();
    const float m = mag ? mag[mag_n == 1 ? 0 : b] : 0.0f;
    for (int i = tid; i < ST_H * ST_W; i += TO_THREADS) {
        const int ty = i / ST_W, tx = i - ty * ST_W;
        const int y = y0 + ty, x = x0 + tx;
        if (y >= H || x >= W) continue;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) acc += tile[ty + j / 3][tx + j % 3] * k[j];
        const float c = tile[ty + 1][tx + 1];
        const float v = fop == AADG_FOP_SHARPNESS ? blendf(c, acc, 1.0f - m) : acc;
        po[(size_t)y * W + x] = clamp01(v);
    }
}

// ---- affine warp / flips: grid (chunks, B*3) -------------------------------------------------------------------------
__global__ __launch_bounds__(TO_THREADS) void k_fop_warp(int fop, const float* __restrict__ in, float* __restrict__ out,
                                                         const float* __restrict__ mag, int mag_n, int H, int W) {
    const int plane = blockIdx.y, b = plane / 3;
    const float* pin = in + (size_t)plane * H * W;
    float* po = out + (size_t)plane * H * W;
    const float m = mag ? mag[mag_n == 1 ? 0 : b] : 0.0f;
    const float cx = 0.5f * (float)(W - 1), cy = 0.5f * (float)(H - 1);
    // forward (src -> dst) matrix A and offset t about the centre; the kernel applies the inverse
    float a00 = 1.f, a01 = 0.f, a10 = 0.f, a11 = 1.f, tx = 0.f, ty = 0.f;
    switch (fop) {
        case AADG_FOP_SHEAR_X: a01 = m; break;
        case AADG_FOP_SHEAR_Y: a10 = m; break;
        case AADG_FOP_TRANSLATE_X: tx = m * (float)W; break;
        case AADG_FOP_TRANSLATE_Y: ty = m * (float)H; break;
        case AADG_FOP_ROTATE: {
            const float rad = m * 0.017453292519943295f;   // degrees, counter-clockwise (OpenCV convention)
            const float cs = cosf(rad), sn = sinf(rad);
            a00 = cs; a01 = sn; a10 = -sn; a11 = cs;
            break;
        }
        default: break;
    }
    const float det = a00 * a11 - a01 * a10;
    const float i00 = a11 / det, i01 = -a01 / det, i10 = -a10 / det, i11 = a00 / det;
    const int HW = H * W;
    for (int i = blockIdx.x * TO_THREADS + threadIdx.x; i < HW; i += gridDim.x * TO_THREADS) {
        const int y = i / W, x = i - y * W;
        float v;
        if (fop == AADG_FOP_HFLIP) v = pin[(size_t)y * W + (W - 1 - x)];
        else if (fop == AADG_FOP_VFLIP) v = pin[(size_t)(H - 1 - y) * W + x];
        else {
            const float dx = (float)x - cx - tx, dy = (float)y - cy - ty;
            const float sx = i00 * dx + i01 * dy + cx, sy = i10 * dx + i11 * dy + cy;
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

int chunks(int HW) {
    int c = (HW / 4 + TO_THREADS * 4 - 1) / (TO_THREADS * 4);
    return c < 1 ? 1 : (c > 512 ? 512 : c);
}
struct FopWs { size_t part, hist, stats, lut, total; };
FopWs fop_ws(int B, int HW) {
    FopWs w;
    size_t o = 0;
    w.part = o; o = aadg_align_up(o + (size_t)B * chunks(HW) * sizeof(Part), 256);
    w.hist = o; o = aadg_align_up(o + (size_t)B * 3 * 256 * 4, 256);
    w.stats = o; o = aadg_align_up(o + (size_t)B * 3 * sizeof(Stats), 256);
    w.lut = o; o = aadg_align_up(o + (size_t)B * 3 * 256 * 4, 256);
    w.total = o;
    return w;
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
    if (fop == AADG_FOP_SHARPNESS || fop == AADG_FOP_GAUSSIAN_BLUR3X3) {
        if (in == out || H < 2 || W < 2) return AADG_E_BADARG;
        const dim3 g((W + ST_W - 1) / ST_W, (H + ST_H - 1) / ST_H, B * 3);
        hipLaunchKernelGGL(k_fop_stencil, g, dim3(TO_THREADS), 0, st, fop, in, out, mag, mag_n, kernel3x3, H, W);
        AADG_LAUNCH_CHECK();
        return 0;
    }
    if (fop >= AADG_FOP_SHEAR_X) {
        if (in == out) return AADG_E_BADARG;
        hipLaunchKernelGGL(k_fop_warp, dim3(chunks(HW) * 2, B * 3), dim3(TO_THREADS), 0, st, fop, in, out, mag, mag_n, H, W);
        AADG_LAUNCH_CHECK();
        return 0;
    }
    const Stats* stats = nullptr;
    const float* lut = nullptr;
    if (fop == AADG_FOP_CONTRAST || fop == AADG_FOP_AUTO_CONTRAST || fop == AADG_FOP_EQUALIZE) {
        if (!ws) return AADG_E_BADARG;
        const FopWs w = fop_ws(B, HW);
        if (ws_bytes < w.total) return AADG_E_WORKSPACE;
        uint8_t* ws8 = reinterpret_cast<uint8_t*>(ws);
        Part* part = reinterpret_cast<Part*>(ws8 + w.part);
        unsigned int* hist = reinterpret_cast<unsigned int*>(ws8 + w.hist);
        Stats* s = reinterpret_cast<Stats*>(ws8 + w.stats);
        float* l = reinterpret_cast<float*>(ws8 + w.lut);
        if (fop == AADG_FOP_EQUALIZE) AADG_HIP_TRY(hipMemsetAsync(hist, 0, (size_t)B * 3 * 256 * 4, st));
        const int ch = chunks(HW);
        hipLaunchKernelGGL(k_fop_stats, dim3(ch, B), dim3(TO_THREADS), 0, st, fop, in, H, W, part, hist);
        AADG_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_fop_stats_final, dim3(B), dim3(256), 0, st, fop, part, ch, HW, hist, s, l);
        AADG_LAUNCH_CHECK();
        stats = s;
        lut = l;
    }
    if (fop == AADG_FOP_SAMPLE_PAIRING && in == out) return AADG_E_BADARG;
    hipLaunchKernelGGL(k_fop_point, dim3(chunks(HW), B), dim3(TO_THREADS), 0, st, fop, in, out, mag, mag_n, perm, stats, lut, H, W);
    AADG_LAUNCH_CHECK();
    return 0;
}

/*
 * aadg_oracle.c -- CPU restatement of the AADG policy-search hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (aadg_amd/) never links or calls it.
 *
 * It restates, in plain scalar C, what the reference's live augmentation path computes through
 * Pillow's C core, and what its reward loop computes through geomloss/pykeops:
 *
 *   reference call site                     | what is restated here
 *   ----------------------------------------+----------------------------------------------------
 *   data/basic.py:70-120,137-167,231-260    | the 10 selectable uint8 ops (orc_op_u8)
 *   data/transform.py:97-135 (scale, crop)  | Image.resize BILINEAR / NEAREST + pad + crop
 *   data/transform.py:149-172,244-249       | Normalize_dg (u8/127.5-1, mask -> multilabel)
 *   data/transform.py:217-236               | ToTensor (HWC -> CHW float32)
 *   search_dg.py:116,150-162,214            | debiased Sinkhorn divergence, cosine cost; rewards
 *   search_dg.py:140-142, losses.py:21-25   | per-policy BCE on sigmoid
 *   search_dg.py:112,164-165 (torchmetrics) | samplewise Dice (F1 of the foreground class)
 *
 * Parity pins: the uint8 ops, resize and normalise stages are checked bit-for-bit against golden
 * fixtures produced by importing the reference in the build container (tests/golden/make_golden.py)
 * and, when Pillow is importable, against live Pillow.  The Sinkhorn part has NO importable
 * reference (geomloss 0.2.4 / pykeops 1.5 are absent from /root/reference and from the image):
 * "parity unpinned" for that function -- it follows the published geomloss 0.2.4 algorithm
 * (sinkhorn_divergence.py: scaling_parameters, epsilon_schedule, sinkhorn_loop, sinkhorn_cost;
 * sinkhorn_samples.py: softmin_online) and is pinned by analytic known answers in tests/.
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -o liboracle.so aadg_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_OPS 4

/* op ids = index in augment_list(), data/basic.py:231-243 */
enum { OP_AUTOCONTRAST = 0, OP_INVERT, OP_EQUALIZE, OP_SOLARIZE, OP_POSTERIZE, OP_CONTRAST,
       OP_COLOR, OP_BRIGHTNESS, OP_SHARPNESS, OP_CUTOUT, OP_COUNT };

/* ------------------------------------------------------------------------------------------- */
/* uint8 helpers                                                                               */
/* ------------------------------------------------------------------------------------------- */

/* Pillow RGB->L (ImageEnhance.Color / Contrast go through image.convert("L")) */
static inline uint8_t rgb2l(uint8_t r, uint8_t g, uint8_t b) {
    return (uint8_t)((19595u * r + 38470u * g + 7471u * b + 0x8000u) >> 16);
}

/* Image.blend(degenerate, img, alpha): C float arithmetic, see ImagingBlend */
static inline uint8_t blend_px(uint8_t deg, uint8_t img, float alpha, int interp) {
    float t = (float)((int)deg + alpha * (float)((int)img - (int)deg));
    if (interp) return (uint8_t)t;
    if (t <= 0.0f) return 0;
    if (t >= 255.0f) return 255;
    return (uint8_t)t;
}

static void hist3(const uint8_t* im, int npix, uint32_t* h /*768*/) {
    memset(h, 0, 768 * sizeof(uint32_t));
    for (int i = 0; i < npix; ++i) {
        h[im[3 * i]]++;
        h[256 + im[3 * i + 1]]++;
        h[512 + im[3 * i + 2]]++;
    }
}

static void apply_lut3(const uint8_t* in, uint8_t* out, int npix, const uint8_t* lut /*768*/) {
    for (int i = 0; i < npix; ++i) {
        out[3 * i] = lut[in[3 * i]];
        out[3 * i + 1] = lut[256 + in[3 * i + 1]];
        out[3 * i + 2] = lut[512 + in[3 * i + 2]];
    }
}

/* ImageOps.autocontrast(cutoff=0) */
static void lut_autocontrast(const uint32_t* h, uint8_t* lut) {
    for (int c = 0; c < 3; ++c) {
        const uint32_t* hc = h + 256 * c;
        int lo = 0, hi = 255;
        while (lo < 256 && !hc[lo]) lo++;
        while (hi >= 0 && !hc[hi]) hi--;
        if (hi <= lo) {
            for (int i = 0; i < 256; ++i) lut[256 * c + i] = (uint8_t)i;
        } else {
            double scale = 255.0 / (hi - lo);
            double offset = -lo * scale;
            for (int i = 0; i < 256; ++i) {
                int ix = (int)(i * scale + offset);
                if (ix < 0) ix = 0; else if (ix > 255) ix = 255;
                lut[256 * c + i] = (uint8_t)ix;
            }
        }
    }
}

/* ImageOps.equalize */
static void lut_equalize(const uint32_t* h, uint8_t* lut) {
    for (int c = 0; c < 3; ++c) {
        const uint32_t* hc = h + 256 * c;
        int nnz = 0; uint64_t sum = 0; uint32_t last = 0;
        for (int i = 0; i < 256; ++i) if (hc[i]) { nnz++; sum += hc[i]; last = hc[i]; }
        uint64_t step = nnz <= 1 ? 0 : (sum - last) / 255;
        if (!step) {
            for (int i = 0; i < 256; ++i) lut[256 * c + i] = (uint8_t)i;
        } else {
            uint64_t n = step / 2;
            for (int i = 0; i < 256; ++i) {
                uint64_t v = n / step;
                lut[256 * c + i] = (uint8_t)(v > 255 ? 255 : v);
                n += hc[i];
            }
        }
    }
}

/* One selectable op of data/basic.py on an HWC uint8 RGB image.
 *   iarg: Solarize -> ceil(threshold) (i < v  <=>  i < ceil(v));  Posterize -> bits (int(v))
 *   farg: Contrast/Color/Brightness/Sharpness -> blend factor as C float
 *   rect: Cutout -> inclusive, already clipped (x0,y0,x1,y1); empty when x1<x0 or y1<y0
 */
void orc_op_u8(const uint8_t* in, uint8_t* out, int H, int W, int op, int iarg, float farg,
               const int* rect) {
    const int npix = H * W;
    uint8_t lut[768];
    uint32_t h[768];
    switch (op) {
    case OP_AUTOCONTRAST:
        hist3(in, npix, h); lut_autocontrast(h, lut); apply_lut3(in, out, npix, lut); break;
    case OP_EQUALIZE:
        hist3(in, npix, h); lut_equalize(h, lut); apply_lut3(in, out, npix, lut); break;
    case OP_INVERT:
        for (int i = 0; i < 3 * npix; ++i) out[i] = (uint8_t)(255 - in[i]);
        break;
    case OP_SOLARIZE:
        for (int i = 0; i < 3 * npix; ++i) out[i] = (int)in[i] < iarg ? in[i] : (uint8_t)(255 - in[i]);
        break;
    case OP_POSTERIZE: {
        uint8_t m = (uint8_t)~((1u << (8 - iarg)) - 1u);
        for (int i = 0; i < 3 * npix; ++i) out[i] = in[i] & m;
        break;
    }
    case OP_CONTRAST: case OP_COLOR: case OP_BRIGHTNESS: case OP_SHARPNESS: {
        float alpha = farg;
        if (alpha == 1.0f) { memcpy(out, in, (size_t)3 * npix); break; }
        int interp = (alpha >= 0.0f && alpha <= 1.0f);
        if (op == OP_CONTRAST) {
            /* mean of the L image: ImageStat sum/count in double, then int(mean + 0.5) */
            double sum = 0.0;
            for (int i = 0; i < npix; ++i) sum += rgb2l(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
            uint8_t m = (uint8_t)(int)(sum / npix + 0.5);
            for (int i = 0; i < 3 * npix; ++i) out[i] = alpha == 0.0f ? m : blend_px(m, in[i], alpha, interp);
        } else if (op == OP_COLOR) {
            for (int i = 0; i < npix; ++i) {
                uint8_t l = rgb2l(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
                for (int c = 0; c < 3; ++c)
                    out[3 * i + c] = alpha == 0.0f ? l : blend_px(l, in[3 * i + c], alpha, interp);
            }
        } else if (op == OP_BRIGHTNESS) {
            for (int i = 0; i < 3 * npix; ++i) out[i] = alpha == 0.0f ? 0 : blend_px(0, in[i], alpha, interp);
        } else {
            /* ImageFilter.SMOOTH: (1,1,1;1,5,1;1,1,1)/13, +0.5, truncate; 1-px border copied */
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x)
                    for (int c = 0; c < 3; ++c) {
                        uint8_t d;
                        if (y == 0 || x == 0 || y == H - 1 || x == W - 1) {
                            d = in[(y * W + x) * 3 + c];
                        } else {
                            int s = 4 * in[(y * W + x) * 3 + c];
                            for (int dy = -1; dy <= 1; ++dy)
                                for (int dx = -1; dx <= 1; ++dx)
                                    s += in[((y + dy) * W + x + dx) * 3 + c];
                            d = (uint8_t)((s + 6) / 13);
                        }
                        out[(y * W + x) * 3 + c] =
                            alpha == 0.0f ? d : blend_px(d, in[(y * W + x) * 3 + c], alpha, interp);
                    }
        }
        break;
    }
    case OP_CUTOUT:
        if (out != in) memcpy(out, in, (size_t)3 * npix);
        if (rect && rect[2] >= rect[0] && rect[3] >= rect[1])
            for (int y = rect[1]; y <= rect[3]; ++y)
                for (int x = rect[0]; x <= rect[2]; ++x)
                    out[(y * W + x) * 3] = out[(y * W + x) * 3 + 1] = out[(y * W + x) * 3 + 2] = 127;
        break;
    default:
        if (out != in) memcpy(out, in, (size_t)3 * npix);
    }
}

/* ------------------------------------------------------------------------------------------- */
/* Image.resize (Pillow Resample.c semantics)                                                  */
/* ------------------------------------------------------------------------------------------- */

#define PRECISION_BITS (32 - 8 - 2)

static inline double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}

/* precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (support 1.0), box = full */
static int precompute_coeffs(int inSize, int outSize, int** boundsp, int32_t** kkp) {
    double scale, filterscale, support;
    filterscale = scale = (double)inSize / outSize;
    if (filterscale < 1.0) filterscale = 1.0;
    support = 1.0 * filterscale;
    int ksize = (int)ceil(support) * 2 + 1;
    int* bounds = (int*)malloc(sizeof(int) * 2 * outSize);
    int32_t* kk = (int32_t*)malloc(sizeof(int32_t) * (size_t)outSize * ksize);
    double* k = (double*)malloc(sizeof(double) * ksize);
    for (int xx = 0; xx < outSize; ++xx) {
        double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > inSize) xmax = inSize;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; ++x) {
            double w = bilinear_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; ++x) k[x] = 0.0;
        for (x = 0; x < ksize; ++x) {
            if (k[x] < 0) kk[xx * ksize + x] = (int32_t)(-0.5 + k[x] * (1 << PRECISION_BITS));
            else kk[xx * ksize + x] = (int32_t)(0.5 + k[x] * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    free(k);
    *boundsp = bounds;
    *kkp = kk;
    return ksize;
}

static inline uint8_t clip8(int32_t v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/* RGB HWC uint8 BILINEAR resize (two passes, uint8 intermediate, horizontal first) */
void orc_resize_bilinear_u8(const uint8_t* in, int H, int W, uint8_t* out, int h, int w) {
    const uint8_t* cur = in;
    uint8_t* tmp = NULL;
    if (w != W) {
        int* b; int32_t* kk;
        int ks = precompute_coeffs(W, w, &b, &kk);
        tmp = (uint8_t*)malloc((size_t)H * w * 3);
        for (int y = 0; y < H; ++y)
            for (int xx = 0; xx < w; ++xx) {
                int xmin = b[2 * xx], n = b[2 * xx + 1];
                const int32_t* k = kk + xx * ks;
                for (int c = 0; c < 3; ++c) {
                    int32_t ss = 1 << (PRECISION_BITS - 1);
                    for (int x = 0; x < n; ++x) ss += in[(y * W + x + xmin) * 3 + c] * k[x];
                    tmp[(y * w + xx) * 3 + c] = clip8(ss);
                }
            }
        free(b); free(kk);
        cur = tmp;
    }
    if (h != H) {
        int* b; int32_t* kk;
        int ks = precompute_coeffs(H, h, &b, &kk);
        for (int yy = 0; yy < h; ++yy) {
            int ymin = b[2 * yy], n = b[2 * yy + 1];
            const int32_t* k = kk + yy * ks;
            for (int x = 0; x < w * 3; ++x) {
                int32_t ss = 1 << (PRECISION_BITS - 1);
                for (int y = 0; y < n; ++y) ss += cur[((size_t)(y + ymin) * w) * 3 + x] * k[y];
                out[(size_t)yy * w * 3 + x] = clip8(ss);
            }
        }
        free(b); free(kk);
    } else {
        memcpy(out, cur, (size_t)h * w * 3);
    }
    free(tmp);
}

/* nearest index table, ImagingScaleAffine: xo = a0*0.5, index = (int)xo, xo += a0 (accumulated) */
void orc_nearest_table(int inSize, int outSize, int* tab) {
    double a0 = (double)inSize / outSize;
    double xo = 0.0 + a0 * 0.5;
    for (int x = 0; x < outSize; ++x) {
        int xin = xo < 0.0 ? -1 : (int)xo;
        if (xin >= inSize) xin = -1; /* not written by Pillow (fill=0); cannot happen for full box */
        tab[x] = xin;
        xo += a0;
    }
}

/* L uint8 NEAREST resize */
void orc_resize_nearest_u8(const uint8_t* in, int H, int W, uint8_t* out, int h, int w) {
    int* xt = (int*)malloc(sizeof(int) * w);
    int* yt = (int*)malloc(sizeof(int) * h);
    orc_nearest_table(W, w, xt);
    orc_nearest_table(H, h, yt);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            out[(size_t)y * w + x] = (yt[y] < 0 || xt[x] < 0) ? 0 : in[(size_t)yt[y] * W + xt[x]];
    free(xt); free(yt);
}

/* ------------------------------------------------------------------------------------------- */
/* One (sample, policy) unit of the live pipeline:                                             */
/*   Policy ops (data/policy.py:23-28) -> DGRandomScaleCrop (data/transform.py:104-131)        */
/*   -> Normalize_dg (:149-172) -> ToTensor (:217-236).                                         */
/* All random draws are made by the caller and passed in explicitly.                           */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t src;                    /* source image index */
    int32_t n_ops;
    int32_t op[ORC_MAX_OPS];
    int32_t iarg[ORC_MAX_OPS];
    float farg[ORC_MAX_OPS];
    int32_t rect[ORC_MAX_OPS][4];
    int32_t scaled_w, scaled_h;     /* == source size when the p=0.8 scale branch is not taken */
    int32_t pad, crop_x, crop_y;    /* RandomCrop: border (fill 0) then crop offset */
} orc_unit;

/* dataset_kind: 0 = optic (K=2 multilabel), 1 = vessel (K=1) */
void orc_aug_units(const uint8_t* src, const uint8_t* masks, int S, int Hs, int Ws,
                   const orc_unit* units, int N, int crop, int dataset_kind,
                   float* out_img, float* out_lbl) {
    (void)S;
    const int K = dataset_kind == 0 ? 2 : 1;
    const size_t npx = (size_t)Hs * Ws;
    uint8_t* a = (uint8_t*)malloc(npx * 3);
    uint8_t* b = (uint8_t*)malloc(npx * 3);
    float lutf[256];
    for (int i = 0; i < 256; ++i) { float f = (float)i; f /= 127.5f; f -= 1.0f; lutf[i] = f; }
    for (int u = 0; u < N; ++u) {
        const orc_unit* p = &units[u];
        const uint8_t* cur = src + (size_t)p->src * npx * 3;
        const uint8_t* msk = masks + (size_t)p->src * npx;
        for (int k = 0; k < p->n_ops; ++k) {
            uint8_t* dst = (cur == a) ? b : a;
            orc_op_u8(cur, dst, Hs, Ws, p->op[k], p->iarg[k], p->farg[k], p->rect[k]);
            cur = dst;
        }
        const int w = p->scaled_w, h = p->scaled_h;
        uint8_t* rs = (uint8_t*)malloc((size_t)w * h * 3);
        uint8_t* rm = (uint8_t*)malloc((size_t)w * h);
        if (w == Ws && h == Hs) {
            memcpy(rs, cur, npx * 3);
            memcpy(rm, msk, npx);
        } else {
            orc_resize_bilinear_u8(cur, Hs, Ws, rs, h, w);
            orc_resize_nearest_u8(msk, Hs, Ws, rm, h, w);
        }
        float* oi = out_img + (size_t)u * 3 * crop * crop;
        float* ol = out_lbl + (size_t)u * K * crop * crop;
        for (int y = 0; y < crop; ++y)
            for (int x = 0; x < crop; ++x) {
                int sx = x + p->crop_x - p->pad, sy = y + p->crop_y - p->pad;
                uint8_t r = 0, g = 0, bl = 0, m = 0;
                if (sx >= 0 && sx < w && sy >= 0 && sy < h) {
                    r = rs[((size_t)sy * w + sx) * 3];
                    g = rs[((size_t)sy * w + sx) * 3 + 1];
                    bl = rs[((size_t)sy * w + sx) * 3 + 2];
                    m = rm[(size_t)sy * w + sx];
                }
                size_t o = (size_t)y * crop + x, pl = (size_t)crop * crop;
                oi[o] = lutf[r]; oi[pl + o] = lutf[g]; oi[2 * pl + o] = lutf[bl];
                if (dataset_kind == 0) {
                    /* >200 bg [0,0]; 51..200 disc ring [0,1]; <=50 cup [1,1]  (transform.py:155-165,244-249) */
                    ol[o] = m <= 50 ? 1.0f : 0.0f;
                    ol[pl + o] = m <= 200 ? 1.0f : 0.0f;
                } else {
                    ol[o] = m != 0 ? 1.0f : 0.0f;
                }
            }
        free(rs); free(rm);
    }
    free(a); free(b);
}

/* ------------------------------------------------------------------------------------------- */
/* Debiased Sinkhorn divergence, geomloss 0.2.4 semantics ("parity unpinned", see header).      */
/*   SamplesLoss("sinkhorn", p=2, blur=.05, scaling=.5, debias=True, backend="online",          */
/*               cost="IntCst(1) - (X|Y)/(Norm2(X)*Norm2(Y))")      search_dg.py:116            */
/* ------------------------------------------------------------------------------------------- */

static float cos_cost(const float* x, const float* y, int E) {
    float xy = 0.f, xx = 0.f, yy = 0.f;
    for (int k = 0; k < E; ++k) { xy += x[k] * y[k]; xx += x[k] * x[k]; yy += y[k] * y[k]; }
    return 1.0f - xy / (sqrtf(xx) * sqrtf(yy));
}

/* softmin(eps, C, h)_i = -eps * LSE_j( h_j - C_ij * fp32(1/eps) ); C is [n][m] row-major */
static void softmin(double eps, const float* C, int n, int m, int ldc, int transposed,
                    const float* h, float* out) {
    const float inv = (float)(1.0 / eps);
    const float feps = (float)eps;
    for (int i = 0; i < n; ++i) {
        float mx = -INFINITY;
        for (int j = 0; j < m; ++j) {
            float c = transposed ? C[j * ldc + i] : C[i * ldc + j];
            float v = h[j] - c * inv;
            if (v > mx) mx = v;
        }
        float s = 0.f;
        for (int j = 0; j < m; ++j) {
            float c = transposed ? C[j * ldc + i] : C[i * ldc + j];
            s += expf(h[j] - c * inv - mx);
        }
        out[i] = -feps * (mx + logf(s));
    }
}

/* eps schedule: [d^p] + [exp(e) for e in arange(p ln d, p ln blur, p ln scaling)] + [blur^p] */
int orc_epsilon_schedule(double diameter, double blur, double scaling, double p, double* eps_s, int cap) {
    int n = 0;
    eps_s[n++] = pow(diameter, p);
    double start = p * log(diameter), stop = p * log(blur), step = p * log(scaling);
    int len = (int)ceil((stop - start) / step);
    if (len < 0) len = 0;
    for (int i = 0; i < len && n < cap - 1; ++i) eps_s[n++] = exp(start + i * step);
    eps_s[n++] = pow(blur, p);
    return n;
}

float orc_sinkhorn_divergence(const float* x, int n, const float* y, int m, int E,
                              double blur, double scaling) {
    /* diameter: ||max(x u y) - min(x u y)||_2 in fp32 */
    float d2 = 0.f;
    for (int k = 0; k < E; ++k) {
        float lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < n; ++i) { float v = x[i * E + k]; if (v < lo) lo = v; if (v > hi) hi = v; }
        for (int i = 0; i < m; ++i) { float v = y[i * E + k]; if (v < lo) lo = v; if (v > hi) hi = v; }
        d2 += (hi - lo) * (hi - lo);
    }
    double diameter = (double)sqrtf(d2);
    double eps_s[128];
    int nits = orc_epsilon_schedule(diameter, blur, scaling, 2.0, eps_s, 128);

    float* Cxx = (float*)malloc(sizeof(float) * n * n);
    float* Cyy = (float*)malloc(sizeof(float) * m * m);
    float* Cxy = (float*)malloc(sizeof(float) * n * m);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Cxx[i * n + j] = cos_cost(x + i * E, x + j * E, E);
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Cyy[i * m + j] = cos_cost(y + i * E, y + j * E, E);
    for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) Cxy[i * m + j] = cos_cost(x + i * E, y + j * E, E);

    int mx = n > m ? n : m;
    float* buf = (float*)malloc(sizeof(float) * mx * 12);
    float *a_log = buf, *b_log = buf + mx, *a_x = buf + 2 * mx, *b_y = buf + 3 * mx, *a_y = buf + 4 * mx,
          *b_x = buf + 5 * mx, *at_x = buf + 6 * mx, *bt_y = buf + 7 * mx, *at_y = buf + 8 * mx,
          *bt_x = buf + 9 * mx, *hx = buf + 10 * mx, *hy = buf + 11 * mx;
    for (int i = 0; i < n; ++i) a_log[i] = logf(1.0f / n);
    for (int j = 0; j < m; ++j) b_log[j] = logf(1.0f / m);

    double eps = eps_s[0];
    softmin(eps, Cxx, n, n, n, 0, a_log, a_x);
    softmin(eps, Cyy, m, m, m, 0, b_log, b_y);
    softmin(eps, Cxy, m, n, m, 1, a_log, a_y); /* C_yx = C_xy^T */
    softmin(eps, Cxy, n, m, m, 0, b_log, b_x);
    for (int it = 0; it < nits; ++it) {
        eps = eps_s[it];
        const float feps = (float)eps;
        for (int i = 0; i < n; ++i) hx[i] = a_log[i] + a_x[i] / feps;
        softmin(eps, Cxx, n, n, n, 0, hx, at_x);
        for (int j = 0; j < m; ++j) hy[j] = b_log[j] + b_y[j] / feps;
        softmin(eps, Cyy, m, m, m, 0, hy, bt_y);
        for (int i = 0; i < n; ++i) hx[i] = a_log[i] + b_x[i] / feps;
        softmin(eps, Cxy, m, n, m, 1, hx, at_y);
        for (int j = 0; j < m; ++j) hy[j] = b_log[j] + a_y[j] / feps;
        softmin(eps, Cxy, n, m, m, 0, hy, bt_x);
        for (int i = 0; i < n; ++i) { a_x[i] = 0.5f * (a_x[i] + at_x[i]); b_x[i] = 0.5f * (b_x[i] + bt_x[i]); }
        for (int j = 0; j < m; ++j) { b_y[j] = 0.5f * (b_y[j] + bt_y[j]); a_y[j] = 0.5f * (a_y[j] + at_y[j]); }
    }
    /* last extrapolation at the final eps; cross terms from the OLD values simultaneously */
    {
        const float feps = (float)eps;
        for (int i = 0; i < n; ++i) hx[i] = a_log[i] + a_x[i] / feps;
        softmin(eps, Cxx, n, n, n, 0, hx, at_x);
        for (int j = 0; j < m; ++j) hy[j] = b_log[j] + b_y[j] / feps;
        softmin(eps, Cyy, m, m, m, 0, hy, bt_y);
        for (int i = 0; i < n; ++i) hx[i] = a_log[i] + b_x[i] / feps;
        softmin(eps, Cxy, m, n, m, 1, hx, at_y);
        for (int j = 0; j < m; ++j) hy[j] = b_log[j] + a_y[j] / feps;
        softmin(eps, Cxy, n, m, m, 0, hy, bt_x);
    }
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < n; ++i) s1 += (1.0f / n) * (bt_x[i] - at_x[i]);
    for (int j = 0; j < m; ++j) s2 += (1.0f / m) * (at_y[j] - bt_y[j]);
    free(Cxx); free(Cyy); free(Cxy); free(buf);
    return s1 + s2;
}

/* float64 master of the same algorithm (used to bound fp32 error in tests) */
static void softmin64(double eps, const double* C, int n, int m, int ldc, int tr, const double* h, double* out) {
    for (int i = 0; i < n; ++i) {
        double mx = -INFINITY;
        for (int j = 0; j < m; ++j) { double v = h[j] - (tr ? C[j * ldc + i] : C[i * ldc + j]) / eps; if (v > mx) mx = v; }
        double s = 0;
        for (int j = 0; j < m; ++j) s += exp(h[j] - (tr ? C[j * ldc + i] : C[i * ldc + j]) / eps - mx);
        out[i] = -eps * (mx + log(s));
    }
}
static double cos_cost64(const float* x, const float* y, int E) {
    double xy = 0, xx = 0, yy = 0;
    for (int k = 0; k < E; ++k) { xy += (double)x[k] * y[k]; xx += (double)x[k] * x[k]; yy += (double)y[k] * y[k]; }
    return 1.0 - xy / (sqrt(xx) * sqrt(yy));
}
double orc_sinkhorn_divergence_f64(const float* x, int n, const float* y, int m, int E, double blur, double scaling) {
    float d2 = 0.f;
    for (int k = 0; k < E; ++k) {
        float lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < n; ++i) { float v = x[i * E + k]; if (v < lo) lo = v; if (v > hi) hi = v; }
        for (int i = 0; i < m; ++i) { float v = y[i * E + k]; if (v < lo) lo = v; if (v > hi) hi = v; }
        d2 += (hi - lo) * (hi - lo);
    }
    double eps_s[128];
    int nits = orc_epsilon_schedule((double)sqrtf(d2), blur, scaling, 2.0, eps_s, 128);
    double* Cxx = (double*)malloc(sizeof(double) * n * n);
    double* Cyy = (double*)malloc(sizeof(double) * m * m);
    double* Cxy = (double*)malloc(sizeof(double) * n * m);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Cxx[i * n + j] = cos_cost64(x + i * E, x + j * E, E);
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Cyy[i * m + j] = cos_cost64(y + i * E, y + j * E, E);
    for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) Cxy[i * m + j] = cos_cost64(x + i * E, y + j * E, E);
    int mx = n > m ? n : m;
    double* buf = (double*)malloc(sizeof(double) * mx * 12);
    double *a_log = buf, *b_log = buf + mx, *a_x = buf + 2 * mx, *b_y = buf + 3 * mx, *a_y = buf + 4 * mx,
           *b_x = buf + 5 * mx, *at_x = buf + 6 * mx, *bt_y = buf + 7 * mx, *at_y = buf + 8 * mx,
           *bt_x = buf + 9 * mx, *hx = buf + 10 * mx, *hy = buf + 11 * mx;
    for (int i = 0; i < n; ++i) a_log[i] = -log((double)n);
    for (int j = 0; j < m; ++j) b_log[j] = -log((double)m);
    double eps = eps_s[0];
    softmin64(eps, Cxx, n, n, n, 0, a_log, a_x);
    softmin64(eps, Cyy, m, m, m, 0, b_log, b_y);
    softmin64(eps, Cxy, m, n, m, 1, a_log, a_y);
    softmin64(eps, Cxy, n, m, m, 0, b_log, b_x);
    for (int it = 0; it <= nits; ++it) {
        int last = it == nits;
        if (!last) eps = eps_s[it];
        for (int i = 0; i < n; ++i) hx[i] = a_log[i] + a_x[i] / eps;
        softmin64(eps, Cxx, n, n, n, 0, hx, at_x);
        for (int j = 0; j < m; ++j) hy[j] = b_log[j] + b_y[j] / eps;
        softmin64(eps, Cyy, m, m, m, 0, hy, bt_y);
        for (int i = 0; i < n; ++i) hx[i] = a_log[i] + b_x[i] / eps;
        softmin64(eps, Cxy, m, n, m, 1, hx, at_y);
        for (int j = 0; j < m; ++j) hy[j] = b_log[j] + a_y[j] / eps;
        softmin64(eps, Cxy, n, m, m, 0, hy, bt_x);
        if (!last) {
            for (int i = 0; i < n; ++i) { a_x[i] = 0.5 * (a_x[i] + at_x[i]); b_x[i] = 0.5 * (b_x[i] + bt_x[i]); }
            for (int j = 0; j < m; ++j) { b_y[j] = 0.5 * (b_y[j] + bt_y[j]); a_y[j] = 0.5 * (a_y[j] + at_y[j]); }
        }
    }
    double s = 0;
    for (int i = 0; i < n; ++i) s += (bt_x[i] - at_x[i]) / n;
    for (int j = 0; j < m; ++j) s += (at_y[j] - bt_y[j]) / m;
    free(Cxx); free(Cyy); free(Cxy); free(buf);
    return s;
}

/* reward loop, search_dg.py:150-162: rows are fe[(b*D + d)*M + j]; rewards[j] += sum over domain pairs */
void orc_sinkhorn_rewards(const float* fe, int D, int B, int M, int E, double blur, double scaling,
                          float* rewards /*[M], accumulated*/) {
    float* clouds = (float*)malloc(sizeof(float) * (size_t)D * B * E);
    for (int j = 0; j < M; ++j) {
        for (int d = 0; d < D; ++d)
            for (int b = 0; b < B; ++b)
                memcpy(clouds + ((size_t)d * B + b) * E, fe + ((size_t)(b * D + d) * M + j) * E, sizeof(float) * E);
        float acc = 0.f;
        int first = 1;
        /* reference order for D=3: dist_12 + dist_13 + dist_23 */
        for (int d1 = 0; d1 < D; ++d1)
            for (int d2 = d1 + 1; d2 < D; ++d2) {
                float s = orc_sinkhorn_divergence(clouds + (size_t)d1 * B * E, B, clouds + (size_t)d2 * B * E, B, E, blur, scaling);
                acc = first ? s : acc + s;
                first = 0;
            }
        rewards[j] += acc;
    }
    free(clouds);
}

/* reward normalisation, search_dg.py:214 (unbiased std) */
void orc_normalize_rewards(const float* r, int M, float* out) {
    float mean = 0.f;
    for (int j = 0; j < M; ++j) mean += r[j];
    mean /= M;
    float var = 0.f;
    for (int j = 0; j < M; ++j) var += (r[j] - mean) * (r[j] - mean);
    float sd = sqrtf(var / (M - 1));
    for (int j = 0; j < M; ++j) out[j] = (r[j] - mean) / (sd + 1e-5f);
}

/* ------------------------------------------------------------------------------------------- */
/* per-policy BCE (search_dg.py:140-142) and samplewise Dice (torchmetrics F1, :164-165)        */
/* ------------------------------------------------------------------------------------------- */

/* logits/labels [N,K,HW]; out_bce[M] = mean over rows j::M of BCE(sigmoid(z), y) (log clamped at -100) */
void orc_policy_bce(const float* logits, const float* labels, int N, int K, int HW, int M, double* out_bce) {
    for (int j = 0; j < M; ++j) {
        double acc = 0; size_t cnt = 0;
        for (int r = j; r < N; r += M)
            for (size_t i = 0; i < (size_t)K * HW; ++i) {
                double z = logits[(size_t)r * K * HW + i], y = labels[(size_t)r * K * HW + i];
                double p = 1.0 / (1.0 + exp(-z));
                double lp = log(p), lq = log(1.0 - p);
                if (lp < -100) lp = -100;
                if (lq < -100) lq = -100;
                acc += -(y * lp + (1 - y) * lq);
                cnt++;
            }
        out_bce[j] = acc / (double)cnt;
    }
}

/* dice[k] = mean over samples of 2TP/(2TP+FP+FN) with pred = sigmoid(z) > 0.5 (0 when denominator 0) */
void orc_dice(const float* logits, const float* labels, int N, int K, int HW, double* out_dice /*[K]*/) {
    for (int k = 0; k < K; ++k) {
        double acc = 0;
        for (int r = 0; r < N; ++r) {
            long tp = 0, fp = 0, fn = 0;
            const float* z = logits + ((size_t)r * K + k) * HW;
            const float* y = labels + ((size_t)r * K + k) * HW;
            for (int i = 0; i < HW; ++i) {
                float pf = 1.0f / (1.0f + expf(-z[i]));
                int pr = pf > 0.5f, gt = (long)y[i] != 0; /* argmax([1-p,p]) picks class 1 iff p > 1-p */
                tp += pr & gt; fp += pr & !gt; fn += !pr & gt;
            }
            long den = 2 * tp + fp + fn;
            acc += den ? (2.0 * tp) / den : 0.0;
        }
        out_dice[k] = acc / N;
    }
}

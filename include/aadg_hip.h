/*
 * aadg_hip.h -- C ABI of libaadg_hip.so, the MI355X (gfx950) implementation of the AADG
 * policy-search hot path.
 *
 * Conventions (SURVEY.md 8b "Ownership / errors / threading"):
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - the caller owns and allocates every buffer, including the workspace (`ws`, size from the
 *     matching *_workspace_bytes query); the library keeps no state, allocates nothing, and is
 *     re-entrant;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and the call returns without
 *     synchronising;
 *   - return value: 0 = ok, <0 = bad argument (AADG_E_*), >0 = hipError_t of a failed launch.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository root).
 */
#ifndef AADG_HIP_H
#define AADG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AADG_ABI_VERSION 12
#define AADG_MAX_OPS 4

#define AADG_E_BADARG (-1)
#define AADG_E_WORKSPACE (-2)
#define AADG_E_UNSUPPORTED (-3)

/* op ids = position in augment_list(), data/basic.py:231-243 */
enum aadg_op {
    AADG_OP_AUTOCONTRAST = 0,
    AADG_OP_INVERT = 1,
    AADG_OP_EQUALIZE = 2,
    AADG_OP_SOLARIZE = 3,
    AADG_OP_POSTERIZE = 4,
    AADG_OP_CONTRAST = 5,
    AADG_OP_COLOR = 6,
    AADG_OP_BRIGHTNESS = 7,
    AADG_OP_SHARPNESS = 8,
    AADG_OP_CUTOUT = 9,
    AADG_OP_COUNT = 10
};

/* dataset kinds of Normalize_dg, data/transform.py:149-172 */
enum aadg_dataset { AADG_DATASET_OPTIC = 0 /* K=2 multilabel */, AADG_DATASET_VESSEL = 1 /* K=1 */ };

/*
 * One (sample, policy) unit of the live augmentation path: every random draw the reference makes
 * in Policy.__call__ (data/policy.py:15-30), Cutout (data/basic.py:146-155) and
 * DGRandomScaleCrop / RandomCrop (data/transform.py:27-55,104-131) is made on the host and
 * recorded here; the kernels only apply.  140 bytes, no padding.
 */
typedef struct aadg_unit {
    int32_t src;                     /* index of the source image in the pool */
    int32_t n_ops;                   /* 0..AADG_MAX_OPS (CONTROLLER.L) */
    int32_t op[AADG_MAX_OPS];        /* enum aadg_op */
    int32_t iarg[AADG_MAX_OPS];      /* Solarize: ceil(threshold); Posterize: int(bits) */
    float farg[AADG_MAX_OPS];        /* Contrast/Color/Brightness/Sharpness: factor as C float */
    int32_t rect[AADG_MAX_OPS][4];   /* Cutout: inclusive clipped (x0,y0,x1,y1); empty if x1<x0 */
    int32_t scaled_w, scaled_h;      /* Image.resize target; == source size when not scaled */
    int32_t pad, crop_x, crop_y;     /* RandomCrop border (fill 0) and crop offset */
} aadg_unit;

int aadg_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Live uint8 augmentation path, replaces the DataLoader-worker chain
 *   DGMultiPolicy (data/policy.py:45-61) -> DGRandomScaleCrop (data/transform.py:97-135)
 *   -> Normalize_dg (:138-186) -> ToTensor (:208-236) -> train_dg_collate_fn (:323-340).
 *
 *   pool      uint8 [P, Hs, Ws, 3]  RGB source images (HWC)
 *   masks     uint8 [P, Hs, Ws]     label images (mode L)
 *   units     aadg_unit[N]
 *   out_img   float [N, 3, crop, crop]   = u8/127.5 - 1
 *   out_lbl   float [N, K, crop, crop]   K = 2 (optic multilabel) or 1 (vessel)
 * ------------------------------------------------------------------------------------------- */
size_t aadg_aug_u8_workspace_bytes(int N, int Hs, int Ws, int crop);
/* max_ops = largest n_ops over the units (number of op stages to run, <= AADG_MAX_OPS).
 * Every unit must satisfy 3*scaled_w >= Ws and 3*scaled_h >= Hs (8 filter taps). */
int aadg_aug_u8_forward(const uint8_t* pool, const uint8_t* masks, int P, int Hs, int Ws,
                        const aadg_unit* units, int N, int max_ops, int crop, int dataset,
                        float* out_img, float* out_lbl, void* ws, size_t ws_bytes, void* stream);
/* Same, with launch hints from a caller that holds the unit records on the host, and two optional
 * hipEvent_t (may be NULL) recorded on `stream` immediately before and after the dominant kernel(s) (the fused
 * resample/normalise/store kernel), so a caller can time exactly that part.
 *   classes_hint    bit 0: some unit takes the fused up-scaling tile kernel; bit 1: some unit takes the staged
 *                   flow (scale < 1/2, > 2 Sharpness ops, or sizes not multiples of 4); bit 2: some unit takes
 *                   the fused generic (down-scaling by <= 2x) tile kernel; 0 = unknown.
 *   stats_mask_hint bit k: some unit's k-th op needs image statistics (AutoContrast/Equalize/Contrast);
 *                   -1 = unknown. */
int aadg_aug_u8_forward_ex(const uint8_t* pool, const uint8_t* masks, int P, int Hs, int Ws,
                           const aadg_unit* units, int N, int max_ops, int crop, int dataset,
                           float* out_img, float* out_lbl, void* ws, size_t ws_bytes, void* stream,
                           int classes_hint, int stats_mask_hint, void* ev_before_final, void* ev_after_final);

/* Same, with the caller's work lists (a HOST struct of DEVICE index arrays; NULL = aadg_aug_u8_forward_ex).  The kernels of the call
 * then launch one workgroup per tile of a unit THEY process instead of one per tile of EVERY unit (a workgroup that finds the wrong
 * class returns at once but still occupies a slot with its LDS for about a microsecond).
 *   order       int32 [N]: unit indices grouped by tile class -- first the n_plain up-scaling units (both scaled sizes >= the
 *               source's) that chain no Sharpness stencil, then the n_sharp up-scaling units that do, then the n_generic units that
 *               shrink an axis by at most 2x -- among them FIRST the n_generic_wonly ones that shrink the width only and chain no
 *               stencil (ABI 9: one-pass tile), LAST the n_generic_sharp ones that chain a Sharpness stencil (ABI 5: their
 *               horizontal pass is a launch of its own, with the stencil's ping-pong buffer) --; the remaining (staged) units follow
 *               in any order.
 *   stat_units  per op slot k: the n_stat[k] units whose k-th op needs image statistics (AutoContrast / Equalize / Contrast) --
 *               the work list of the histogram kernels of that stage.  stat_units[0] == NULL: no statistics lists.
 *   pool_hist   uint32 [P][AADG_HIST_STRIDE] from aadg_pool_histograms_u8 (or NULL): the policy ops run on the RAW source image
 *               (data/policy.py:17-23 precedes DGRandomScaleCrop), so the statistics of an op in slot 0 -- and, pushed through
 *               the earlier byte maps, of a later slot -- are those of the POOL image, the same for every unit and every batch
 *               that draws it.  With the pool resident for the whole run they are computed once instead of per unit and call.
 *               The caller owns the cache: recompute after writing to the pool.
 * A unit listed under the wrong class is skipped (its outputs are not written): the lists must follow unit_flow() of
 * csrc/aug_u8.hip (aadg_amd/_lib/aug.py: launch_hints builds them from the host copy of the unit records). */
#define AADG_HIST_STRIDE 772    /* uint32 words per image: 3 x 256 bins, uint64 sum of L (ImageStat mean of convert('L')), pad */
typedef struct aadg_aug_lists {
    const int32_t* order;
    int32_t n_plain, n_sharp, n_generic;
    const int32_t* stat_units[AADG_MAX_OPS];
    int32_t n_stat[AADG_MAX_OPS];
    const uint32_t* pool_hist;
    /* With pool_hist: the n_late units whose op in a slot k >= 1 needs a pixel pass for its statistics ("late" units: Contrast behind
     * anything, AutoContrast / Equalize behind Color / Cutout / Sharpness).  The call then builds every other byte map, together with
     * the tables, in its first launch and re-does only the late units' maps behind the histogram passes.
     * late_units == NULL: stage by stage for all units. */
    const int32_t* late_units;
    int32_t n_late;
    int32_t n_generic_sharp;   /* ABI 5: how many of the n_generic units (the last ones in `order`) chain a Sharpness stencil */
    /* ABI 7: how many of the n_stat[k] units of stat_units[k] -- the FIRST ones -- have a Sharpness stencil among ops [0, k): the
     * statistics pass gives each of their tiles a workgroup of its own (the image after k ops is rebuilt in LDS) and streams the
     * others.  A wrong split costs time, not correctness (the kernel chooses the data flow from the unit record). */
    int32_t n_stat_stencil[AADG_MAX_OPS];
    /* ABI 7: list slots per chunk of the two-pass flow of the down-scaling units (the horizontally resampled rows of one chunk share one
     * slice of the workspace).  0 = the library's choice (as many as keep a chunk's intermediate within 128 MB); a smaller positive
     * value is honoured (tests of the chunk boundaries), a larger one is clamped to the library's. */
    int32_t gen_chunk;
    /* ABI 9: how many of the n_generic units -- the FIRST ones in `order`, none of them a Sharpness unit -- shrink the WIDTH only (scaled
     * height >= source height, source width >= 8): they run through the one-pass tile k_fused3w instead of the two passes. */
    int32_t n_generic_wonly;
    /* ABI 12: how many of the n_plain / n_sharp units -- the FIRST ones of their class in `order` -- wait for no statistics pass (the late
     * units close their class).  With the cached-statistics call the library runs the late units' chain (histogram passes, byte maps) and
     * their tiles on a helper stream beside the tile kernel of these early units and joins before it returns to the caller's stream (not
     * while ev_before / ev_after time the tile kernel).  0 / 0 (a zero-initialised struct, or lists whose classes are not ordered that way):
     * the call stays on the caller's stream.  The helper stream and its events are one per process: forking calls must be stream-ordered
     * with one another (they already are: they share the caller's workspace). */
    int32_t n_plain_early, n_sharp_early;
} aadg_aug_lists;
/* per-image histograms of a source pool [P, Hs, Ws, 3] (what PIL's Image.histogram() / ImageStat.Stat(convert('L')).mean read:
 * data/basic.py AutoContrast / Equalize / Contrast via ImageOps / ImageEnhance) */
int aadg_pool_histograms_u8(const uint8_t* pool, int P, int Hs, int Ws, uint32_t* hist, void* stream);
int aadg_aug_u8_forward_ex2(const uint8_t* pool, const uint8_t* masks, int P, int Hs, int Ws,
                            const aadg_unit* units, int N, int max_ops, int crop, int dataset,
                            float* out_img, float* out_lbl, void* ws, size_t ws_bytes, void* stream,
                            int classes_hint, int stats_mask_hint, void* ev_before_final, void* ev_after_final,
                            const aadg_aug_lists* lists);

/* (ABI 6) Host-side planning of one aadg_aug_u8_forward_ex2 call: validates the unit records (AADG_E_BADARG: source index, op count / ids,
 * scale factor below 1/3, Cutout box not clipped to the image, Posterize bits) and fills the work lists `aadg_aug_lists` carries -- no GPU
 * work, host pointers (the caller copies the lists to the device next to the records).
 *   order [N]: unit indices by tile class;  stat_units [AADG_MAX_OPS][N]: per op slot the units that need a statistics pass;
 *   late_units [N];  summary [11 + 2 * AADG_MAX_OPS] = n_plain, n_sharp, n_generic, n_generic_sharp, n_late, classes_hint, stats_mask_hint,
 *   max_ops, n_stat[0 .. AADG_MAX_OPS), n_stat_stencil[0 .. AADG_MAX_OPS) (ABI 7; stat_units[k] lists the stencil units first),
 *   n_generic_wonly (ABI 9), n_plain_early, n_sharp_early (ABI 12: summary [11 + 2 * AADG_MAX_OPS]; inside the plain and the Sharpness class of
 *   `order` the late units come last).  (The reference does this work implicitly, op by op, in PIL: data/policy.py:45-61.) */
int aadg_aug_u8_plan(const aadg_unit* units, int N, int P, int Hs, int Ws, int crop, int32_t* order, int32_t* stat_units,
                     int32_t* late_units, int32_t* summary);

/* (ABI 11) Host-side planner, part 2 (no GPU work): every draw one training batch of the standard pipeline makes from PYTHON's `random`
 * generator, in the reference's order -- per (item, domain): per policy the CutMix-queue draw and the sub-policy draw (data/policy.py:17-23),
 * DGRandomScaleCrop for the original and each of the M augmented images (data/transform.py:104-131, RandomCrop :38-53), the soft domain code
 * (:260-274) -- on a copy of the interpreter's MT19937 state: mt_state [625] = random.getstate()[1], advanced in place (the caller puts it
 * back with random.setstate).  queue_lens [M] in / out: the CutMix queue lengths (capped at 10).  sub [S M] int64: sub-policy per (sample,
 * policy); geo [(S + S M) 5] int32: (scaled_w, scaled_h, pad, crop_x, crop_y) per row (rows [0, S): the un-augmented images; S + s M + j: the
 * augmented ones); codes [S n_code] float64.  S = n_items * D.  AADG_E_BADARG: bad sizes, or an empty crop range (python raises ValueError). */
int aadg_draw_python_stream(uint32_t* mt_state, int n_items, int D, int M, const int32_t* nsub, int32_t* queue_lens, double scale_lo,
                            double scale_hi, int crop_h, int crop_w, int crop_pad, int n_code, int W0, int H0, int64_t* sub, int32_t* geo,
                            double* codes);

/* One registry op on one image, replaces fn(img, mask, v) of augment_list()
 * (data/basic.py:70-120,137-167) for uint8 HWC tensors.  ws >= aadg_aug_u8_workspace_bytes(1,H,W,0);
 * in != out. */
int aadg_op_u8(const uint8_t* in, uint8_t* out, int H, int W, int op, int iarg, float farg,
               const int32_t* rect_host, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sinkhorn reward, replaces geomloss.SamplesLoss("sinkhorn", cost=cosine, backend="online")
 * (search_dg.py:116) and the reward loop (search_dg.py:150-162).
 *
 * Problems are described by index tables into one feature matrix `feat` [rows, E] (row stride
 * `ld` floats): cloud c = rows cloud_rows[cloud_off[c] .. cloud_off[c+1]); problem p compares
 * clouds prob_xy[2p] and prob_xy[2p+1].  out[p] = S_eps(x, y), debiased, p=2.
 * ------------------------------------------------------------------------------------------- */
/* clouds up to ~100 points run out of LDS (no workspace beyond n_prob floats); larger clouds keep their cost
 * matrices in the workspace (4 * max_cloud^2 floats per problem) and build them on the matrix cores. */
size_t aadg_sinkhorn_workspace_bytes(int n_prob, int max_cloud, int E);
int aadg_sinkhorn_divergence_f32(const float* feat, int ld, int E, const int32_t* cloud_rows,
                                 const int32_t* cloud_off, const int32_t* prob_xy, int n_prob,
                                 int max_cloud, float blur, float scaling, float* out, void* ws,
                                 size_t ws_bytes, void* stream);
/* ABI 8, measurement: the large-cloud path (clouds beyond the LDS-resident kernel) in two halves -- phases bit 0 = row normalisation +
 * eps schedule + cost matrices into ws, bit 1 = the sweeps over the matrices in ws + the result -- so that bench.py can price the cost
 * build against the matrix-core peak and the sweeps against HBM.  -3 for clouds the LDS-resident kernel takes. */
int aadg_sinkhorn_divergence_phases_f32(const float* feat, int ld, int E, const int32_t* cloud_rows, const int32_t* cloud_off,
                                        const int32_t* prob_xy, int n_prob, int max_cloud, float blur, float scaling, float* out,
                                        void* ws, size_t ws_bytes, int phases, void* stream);
/* fe [D*B*M, E], row (b*D + d)*M + j  (train_dg_collate_fn order);  rewards[j] += sum over the
 * D(D-1)/2 domain pairs, added in the reference's order (d1<d2 lexicographic). */
int aadg_sinkhorn_rewards_f32(const float* fe, int D, int B, int M, int E, float blur,
                              float scaling, float* rewards_accum, void* ws, size_t ws_bytes,
                              void* stream);
/* the same with the Euclidean norm of every row supplied by the producer of fe (aadg_embed_prologue_norm_f32): the kernel skips
 * its own norm reduction; row_norm [D*B*M].  LDS-resident clouds only (AADG_E_UNSUPPORTED otherwise). */
int aadg_sinkhorn_rewards_norm_f32(const float* fe, const float* row_norm, int D, int B, int M, int E, float blur,
                                   float scaling, float* rewards_accum, void* ws, size_t ws_bytes, void* stream);
/* (r - mean) / (std_unbiased + 1e-5), search_dg.py:214 */
int aadg_normalize_rewards_f32(const float* rewards, int M, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-policy BCE + samplewise Dice in one pass over logits/labels, replaces
 *   [BCELoss(sigmoid(z)[j::M], y[j::M]) for j in range(M)]        (search_dg.py:140-142)
 *   torchmetrics F1(num_classes=2, average=None, mdmc_average='samplewise')[1]  (:164-165)
 * logits/labels float [N, K, HW]; out_bce[M]; out_dice[K]; optional grad_logits (d mean_j BCE_j / dz).
 * ------------------------------------------------------------------------------------------- */
/* ABI 8: ONE launch.  `ws` holds integer accumulators and an arrival counter: it must be zero-filled before the first call that uses it
 * (and not be shared by calls that may overlap on different streams); every call leaves it zero-filled for the next one. */
size_t aadg_seg_loss_workspace_bytes(int N, int K, int HW);
int aadg_seg_bce_dice_f32(const float* logits, const float* labels, int N, int K, int HW, int M,
                          float* out_bce, float* out_dice, float* grad_logits, void* ws,
                          size_t ws_bytes, void* stream);
/* (ABI 7) the same with the gradient of grad_scale * mean_j BCE_j: the factor a caller would otherwise apply to the loss before
 * calling backward (count-weighted mean of a row-sharded batch: aadg_amd/distributed.py RowPlan.loss_weight) -- autograd's
 * `grad * d loss` is then not needed: the gradient is handed to logits.backward() as written by the kernel (one pass over the
 * [N,K,HW] tensor less per step).  out_bce / out_dice are NOT scaled. */
int aadg_seg_bce_dice_scaled_f32(const float* logits, const float* labels, int N, int K, int HW, int M, float grad_scale,
                                 float* out_bce, float* out_dice, float* grad_logits, void* ws,
                                 size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Float tensor ops, data/functional.py (batched [B,3,H,W] float in [0,1], mag scalar or [B]).
 * `mag` is a device pointer to mag_n (1 or B) floats, or NULL for ops without magnitude.
 * Output is clamped to [0,1] as tensor_function does (data/functional.py:49-73).
 * ------------------------------------------------------------------------------------------- */
enum aadg_fop {
    AADG_FOP_INVERT = 0, AADG_FOP_SOLARIZE, AADG_FOP_POSTERIZE, AADG_FOP_GRAY, AADG_FOP_CONTRAST,
    AADG_FOP_AUTO_CONTRAST, AADG_FOP_SATURATE, AADG_FOP_BRIGHTNESS, AADG_FOP_HUE,
    AADG_FOP_SAMPLE_PAIRING, AADG_FOP_EQUALIZE, AADG_FOP_SHARPNESS, AADG_FOP_GAUSSIAN_BLUR3X3,
    AADG_FOP_SHEAR_X, AADG_FOP_SHEAR_Y, AADG_FOP_TRANSLATE_X, AADG_FOP_TRANSLATE_Y,
    AADG_FOP_ROTATE, AADG_FOP_HFLIP, AADG_FOP_VFLIP, AADG_FOP_COUNT
};
size_t aadg_fop_workspace_bytes(int B, int H, int W);
/* kernel3x3: 9 floats (device) for SHARPNESS / GAUSSIAN_BLUR3X3, NULL = reference default;
 * perm: B int32 (device) for SAMPLE_PAIRING. */
int aadg_fop_f32(int fop, const float* in, float* out, const float* mag, int mag_n,
                 const float* kernel3x3, const int32_t* perm, int B, int C, int H, int W, void* ws,
                 size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Depthwise 3x3 (stride 1, zero padding 1) + bias + exact GELU on token-layout activations [B, H, W, C] (C fastest): the middle of
 * SegFormer's Mix-FFN (mix_transformer.py:19-46,149-159) without the NCHW round trip.  dtype 0 float32, 1 bfloat16; C % 8 == 0.
 * w9 = the [C,1,3,3] weight as [9][C] float32 (tap-major), bias float32 [C].
 * backward: g = scratch of h's shape / dtype; dh = gradient w.r.t. h; dw9 [9][C], db [C] float32 (overwritten; float32 atomics).
 * ------------------------------------------------------------------------------------------- */
int aadg_dwconv3x3_gelu_nhwc_supported(int B, int H, int W, int C, int dtype);
int aadg_dwconv3x3_gelu_nhwc_forward(const void* h, const float* w9, const float* bias, void* out, int B, int H, int W, int C, int dtype,
                                     void* stream);
int aadg_dwconv3x3_gelu_nhwc_backward(const void* h, const float* w9, const float* bias, const void* dout, void* g, void* dh, float* dw9,
                                      float* db, int B, int H, int W, int C, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Residual add + LayerNorm over the last dimension (the pre-norm blocks of the SegFormer backbone, BASELINE configs[4];
 * mix_transformer.py:96-117): s = x + rscale[row / rows_per_sample] * r (r == NULL: s = x, nothing written to s_out; rscale == NULL: 1),
 * y = LayerNorm(s) * gamma + beta.  x, r, s_out, y, dy, ds_extra, dx, dr: [R, C] in `dtype` (0 float32, 1 bfloat16); gamma, beta,
 * mean, rstd, dgamma, dbeta float32.  C % 8 == 0, C <= 512.  Backward: dx = d/ds (LayerNorm gradient + ds_extra), dr = rscale * dx.
 * ------------------------------------------------------------------------------------------- */
int aadg_layernorm_supported(int R, int C, int dtype);
size_t aadg_layernorm_workspace_bytes(int R, int C);
int aadg_layernorm_forward(const void* x, const void* r, const float* rscale, int rows_per_sample, const float* gamma,
                           const float* beta, float eps, void* s_out, void* y, float* mean, float* rstd, int R, int C, int dtype,
                           void* stream);
int aadg_layernorm_backward(const void* s, const void* dy, const void* ds_extra, const float* gamma, const float* mean,
                            const float* rstd, const float* rscale, int rows_per_sample, void* dx, void* dr, float* dgamma,
                            float* dbeta, void* ws, size_t ws_bytes, int R, int C, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Bilinear up-sampling, align_corners = True, NCHW planes (the x4 up-samplings between the augmentation output
 * and the BCE/Dice kernel in DeepLabV3+; same arithmetic as torch.nn.functional.interpolate / ATen
 * upsample_bilinear2d).  in [planes, h, w] -> out [planes, H, W]; dtype 0 = float32, 1 = bfloat16.
 * ------------------------------------------------------------------------------------------- */
int aadg_upsample_bilinear2d(const void* in, void* out, int planes, int h, int w, int H, int W, int dtype, void* stream);
/* out as a channel slice of a wider tensor (e.g. a concatenation buffer): plane (n, c) at n * out_image_stride + c * H * W */
int aadg_upsample_bilinear2d_strided(const void* in, void* out, int N, int C, int h, int w, int H, int W,
                                     long long out_image_stride, int dtype, void* stream);
/* gradient w.r.t. the input: dy [planes, H, W] -> dx [planes, h, w] (gathered, deterministic; _supported: the dy
 * rectangle feeding an 8 x 32 input tile must fit 48 KiB of LDS, i.e. up-sampling factors up to ~6) */
int aadg_upsample_bilinear2d_backward_supported(int h, int w, int H, int W);
size_t aadg_upsample_bilinear2d_backward_workspace_bytes(int h, int w);
/* dy as a channel slice of a wider tensor: plane (n, c) starts at n * dy_image_stride + c * H * W elements */
int aadg_upsample_bilinear2d_backward_strided(const void* dy, void* dx, int N, int C, int h, int w, int H, int W,
                                              long long dy_image_stride, int dtype, void* ws, size_t ws_bytes,
                                              void* stream);
int aadg_upsample_bilinear2d_backward(const void* dy, void* dx, int planes, int h, int w, int H, int W, int dtype,
                                      void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d (+ activation, + residual add) over NCHW planes: the normalisation layers of the segmentation
 * backbone that consumes the augmentation batch (reference: smp DeepLabV3+ built at models/__init__.py:17-23; every
 * `nn.BatchNorm2d` followed by `ReLU`/`ReLU6`, and the `relu(bn3(conv3(x)) + identity)` tail of a bottleneck).
 * Same arithmetic as torch.nn.functional.batch_norm: biased variance for normalisation, unbiased for running_var,
 * running <- (1 - momentum) * running + momentum * batch.  dtype 0 = float32, 1 = bfloat16 (x, residual, y, dy, dx);
 * weight / bias / statistics are float32 [C].  act: AADG_ACT_*.  residual (nullable): y = act(bn(x) + residual).
 * training = 0 normalises with running_mean / running_var and writes no statistics.
 * Backward: dx, dweight, dbias (nullable) and, if dres != NULL, dres = dy * act'(.) = gradient of the residual
 * branch (then y, the stored forward output, or the forward's act_mask must be given: the activation mask is taken from it).  dy_extra (host array of
 * n_extra <= 6 device pointers, only with dres): further gradients of the same output -- a block output feeds the next
 * block's first convolution AND its residual branch, the encoder output feeds five ASPP branches -- summed on the fly
 * instead of by separate elementwise passes.
 * ------------------------------------------------------------------------------------------- */
enum { AADG_ACT_NONE = 0, AADG_ACT_RELU = 1, AADG_ACT_RELU6 = 2 };
size_t aadg_bn_workspace_bytes(int C);
/* bytes of the optional activation bit mask (one byte per 16-byte vector; 0 = not available for this shape): a forward with
 * a fused residual may write it (act_mask != NULL) so that the backward reads 1 byte instead of a 16-byte vector of y */
size_t aadg_bn_mask_bytes(int N, int C, int HW, int dtype);
/* y_image_stride / dy_image_stride (elements, 0 = dense): plane (n, c) of y / dy starts at n * stride + c * HW -- the output may be
 * written straight into a channel slice of a concatenation buffer, and the gradient read from a slice of that buffer's gradient */
int aadg_bn_forward(const void* x, const void* residual, void* y, void* act_mask, const float* weight, const float* bias,
                    float* running_mean, float* running_var, float momentum, float eps, int act, int training,
                    int N, int C, int HW, int dtype, float* save_mean, float* save_invstd, void* ws,
                    size_t ws_bytes, long long y_image_stride, void* stream);
/* dy_plane_const (optional, needs dres): [N*C] float32, a further gradient that is constant over each plane -- what a global
 * average pool of the output sends back -- added to dy without materialising it */
int aadg_bn_backward(const void* x, const void* y, const void* act_mask, const void* dy, const void* const* dy_extra,
                     int n_extra, const float* dy_plane_const, const float* weight, const float* bias, const float* save_mean,
                     const float* save_invstd, int act, void* dx, void* dres, float* dweight, float* dbias, int N,
                     int C, int HW, int dtype, void* ws, size_t ws_bytes, long long dy_image_stride, void* stream);

/* Synchronised statistics over data-parallel ranks (replaces torch.nn.SyncBatchNorm around the same kernels; the reference wraps
 * its model in DDP at models/__init__.py:39 and its single-GPU batch mixes all source domains in every BatchNorm batch, which
 * domain-sharded replicas only reproduce with all-reduced statistics -- SURVEY.md 8e).  The library does no communication:
 *   phase 1  local float64 sums -> `sums`; the CALLER all-reduces `sums` over the ranks (SUM);   phase 2  consumes the totals.
 * forward : sums = [2C + 1] doubles: (sum x, sum x^2) per channel, then this rank's element count N * HW.  Phase 2 writes y, the
 *           saved / running statistics (from the global totals, identical on every rank).
 * backward: sums = [2C] doubles: (sum g, sum g * xhat); count = device pointer to the forward's all-reduced element count
 *           (its sums + 2C).  Phase 1 writes the masked gradient `dres` (when given) and the LOCAL dweight / dbias (the
 *           parameter gradients are averaged over the ranks like every other gradient); phase 2 writes dx.  Pass the same
 *           tensors in both phases.  Training mode only; other arguments as aadg_bn_forward / aadg_bn_backward. */
int aadg_bn_sync_forward(int phase, const void* x, const void* residual, void* y, void* act_mask, const float* weight,
                         const float* bias, float* running_mean, float* running_var, float momentum, float eps, int act,
                         int N, int C, int HW, int dtype, float* save_mean, float* save_invstd, double* sums, void* ws,
                         size_t ws_bytes, long long y_image_stride, void* stream);
int aadg_bn_sync_backward(int phase, const void* x, const void* y, const void* act_mask, const void* dy,
                          const void* const* dy_extra, int n_extra, const float* dy_plane_const, const float* weight,
                          const float* bias, const float* save_mean, const float* save_invstd, int act, void* dx, void* dres,
                          float* dweight, float* dbias, int N, int C, int HW, int dtype, double* sums, const double* count,
                          void* ws, size_t ws_bytes, long long dy_image_stride, void* stream);

/* Training BatchNorm + ReLU + MaxPool2d(3, 2, 1) in one pass (the ResNet stem after its convolution): the normalised map is
 * never written.  y [N, C, Ho, Wo] and the pooling index (one byte per output, as aadg_maxpool3x3s2_forward); statistics as
 * aadg_bn_forward(training = 1).  Backward = aadg_maxpool3x3s2_backward(index, dy) then aadg_bn_backward(x, ..., AADG_ACT_RELU).
 * W a multiple of 8, Ho * Wo a multiple of 1024. */
int aadg_bn_relu_maxpool_supported(int H, int W, int dtype);
int aadg_bn_relu_maxpool_forward(const void* x, void* y, void* index, const float* weight, const float* bias, float* running_mean,
                                 float* running_var, float momentum, float eps, int N, int C, int H, int W, int dtype,
                                 float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, void* stream);
/* its backward in two passes over x (bfloat16 only): the gradient of the normalised map is rebuilt from the pooled gradient dy
 * [N, C, Ho, Wo] and the index inside both BatchNorm backward passes instead of being stored and read back */
int aadg_bn_relu_maxpool_backward(const void* x, const void* index, const void* dy, const float* weight, const float* bias,
                                  const float* save_mean, const float* save_invstd, void* dx, float* dweight, float* dbias, int N,
                                  int C, int H, int W, int dtype, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Depthwise 3x3 convolution, stride 1, padding = dilation, no bias, NCHW planes (the atrous separable convolutions
 * of the DeepLabV3+ head: smp's SeparableConv2d / ASPPSeparableConv built at models/__init__.py:17-23).
 * x, y: [N, C, H, W] float32 (dtype 0) or bfloat16 (dtype 1); weight: float32 [C, 3, 3].
 * flip = 1 applies the kernel rotated by 180 degrees: dx = aadg_dwconv3x3(dy, weight, flip = 1).
 * aadg_dwconv3x3_wgrad: dweight[c, a, b] = sum_{n, i, j} dy[n, c, i, j] * x[n, c, i + (a-1)d, j + (b-1)d].
 * Supported: W <= 256, W a multiple of the 16-byte vector (4 / 8 elements), (rows + halo) * W * 4 <= 64 KiB.
 * ------------------------------------------------------------------------------------------- */
int aadg_dwconv3x3_supported(int H, int W, int dilation, int dtype);
size_t aadg_dwconv3x3_workspace_bytes(int C);
int aadg_dwconv3x3(const void* x, const float* weight, void* y, int N, int C, int H, int W, int dilation, int flip,
                   int dtype, void* stream);
int aadg_dwconv3x3_wgrad(const void* x, const void* dy, float* dweight, int N, int C, int H, int W, int dilation,
                         int dtype, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 1x1 / stride-1 convolution on NCHW bfloat16 tensors as a per-image matrix-core GEMM (csrc/conv1x1_fwd.hip):
 * out [N, M, HW] = a [M, K] x in [N, K, HW], float32 accumulation.  Forward of torch.nn.Conv2d(K, M, 1, bias=False): a = weight;
 * its input gradient: a = weight^T (contiguous [Ci, Co]), in = dY.  K and HW multiples of 8.
 * ------------------------------------------------------------------------------------------- */
int aadg_conv1x1_nchw_supported(int M, int K, int HW);
int aadg_conv1x1_nchw_bf16(const void* a, const void* in, void* out, int N, int M, int K, int HW, void* stream);
/* ABI 8 -- "f32x3": the same contraction on float32 NCHW tensors at float32 precision (the reference runs its backbone in float32:
 * search_dg.py:123-206).  gfx950 has no tf32 and its float32 MFMA runs at 1/16 of the bfloat16 rate, so every operand is split into
 * bfloat16 halves x = hi + lo (hi = bf16(x), lo = bf16(x - hi); ~2^-17 relative) and every product is formed as hi*hi + hi*lo + lo*hi
 * on the bfloat16 matrix cores with float32 accumulation: three MFMAs per product = 5.3x the float32-MFMA rate.  a_hi / a_lo [M, K]
 * bfloat16: the halves of the float32 operand a (aadg_weight_layouts_split_bf16); in / out float32. */
int aadg_conv1x1_nchw_f32x3(const void* a_hi, const void* a_lo, const float* in, float* out, int N, int M, int K, int HW, void* stream);
/* ... with the BatchNorm statistics of `out` from the convolution's epilogue: bn_sums [2 M + 1] doubles = (sum, sum of squares) per output
 * channel + the element count N * HW -- the buffer aadg_bn_sync_forward(phase 2, ...) normalises with (the statistics pass over the
 * float32 output is not run); bn_sums == NULL: plain convolution */
int aadg_conv1x1_nchw_f32x3_stats(const void* a_hi, const void* a_lo, const float* in, float* out, int N, int M, int K, int HW,
                                  double* bn_sums, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The ResNet stem convolution, forward: y [N, 64, H/2, W/2] = conv2d(x [N, 3, H, W], weight [64, 3, 7, 7], stride 2, padding 3),
 * bfloat16 activations, float32 master weights (rounded to bfloat16 in the kernel, float32 accumulation) -- what
 * torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False) computes under bfloat16 autocast (smp's ResNet encoder `conv1`).  MFMA implicit
 * GEMM straight from NCHW (csrc/stem_conv.hip).  H even, W a multiple of 16.  `ws`: aadg_stem_conv7x7_workspace_bytes().
 * ------------------------------------------------------------------------------------------- */
int aadg_stem_conv7x7_supported(int H, int W);
size_t aadg_stem_conv7x7_workspace_bytes(void);
/* x_dtype: 0 = x is float32 (rounded to bfloat16 on load: no separate cast of the batch), 1 = bfloat16 */
int aadg_stem_conv7x7_bf16(const void* x, int x_dtype, const float* weight, void* y, int N, int H, int W, void* ws, size_t ws_bytes,
                           void* stream);
/* dweight [64,3,7,7] (float32, overwritten) from x [N,3,H,W] and dy [N,64,H/2,W/2] (bfloat16): MFMA GEMM over the output pixels */
int aadg_stem_conv7x7_wgrad_bf16(const void* x, int x_dtype, const void* dy, float* dweight, int N, int H, int W, void* stream);
/* ABI 8 -- f32x3 (see aadg_conv1x1_nchw_f32x3): the stem convolution and its weight gradient on float32 tensors at float32 precision.
 * aadg_stem_conv7x7_workspace_bytes() doubled with ABI 8 (hi + lo weight fragments). */
int aadg_stem_conv7x7_f32x3(const float* x, const float* weight, float* y, int N, int H, int W, void* ws, size_t ws_bytes, void* stream);
int aadg_stem_conv7x7_wgrad_f32x3(const float* x, const float* dy, float* dweight, int N, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stride-2 pixel sub-sampling of NCHW planes: y[p][i][j] = x[p][2i][2j], x [planes, H, W] -> y [planes, H/2, W/2], and its
 * gradient dx (dy at the even positions, zeros elsewhere; every element of dx is written).  With a stride-1 1x1 convolution
 * behind it this is torch.nn.Conv2d(cin, cout, 1, stride=2) -- the down-sampling shortcut of a ResNet stage -- without the
 * library's NCHW <-> CNHW transposes of the full activation.  H even, W a multiple of 8 (float32) / 16 (bfloat16).
 * ------------------------------------------------------------------------------------------- */
int aadg_subsample2x2_supported(int H, int W, int dtype);
int aadg_subsample2x2(const void* x, void* y, int planes, int H, int W, int dtype, void* stream);
int aadg_subsample2x2_backward(const void* dy, void* dx, int planes, int H, int W, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Max pooling 3x3, stride 2, padding 1 over NCHW planes (the ResNet stem pool of the backbone's encoder,
 * torch.nn.MaxPool2d(3, 2, 1) semantics: padding never wins, ties -> first element in window order).
 * x [planes, H, W] -> y [planes, (H-1)/2+1, W/2]; W a multiple of 8.  `index` (optional in the forward: inference) receives
 * one byte per output, the arg-max as its position 3*a + b inside the 3x3 window (the library keeps an int64 element index:
 * 8 bytes); the backward needs only `index` and dy -- the pooled input is neither saved nor read.  aadg_maxpool3x3s2_index_bytes
 * = planes * Ho * Wo.  dtype 0 = float32, 1 = bfloat16.
 * ------------------------------------------------------------------------------------------- */
int aadg_maxpool3x3s2_supported(int H, int W);
size_t aadg_maxpool3x3s2_index_bytes(int planes, int H, int W);
int aadg_maxpool3x3s2_forward(const void* x, void* y, void* index, int planes, int H, int W, int dtype, void* stream);
int aadg_maxpool3x3s2_backward(const void* index, const void* dy, void* dx, int planes, int H, int W, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weight gradient of a 1x1 / stride-1 / no-padding convolution, NCHW bfloat16 activations (the pointwise
 * convolutions of the backbone: bottleneck conv1 / conv3, ASPP, decoder):
 *     dweight[o][c] = sum_{n, k} dy[n][o][k] * x[n][c][k],   dy [N, Co, HW], x [N, Ci, HW], dweight float32 [Co, Ci].
 * Runs on the matrix cores straight from the NCHW tensors (both operands are K-contiguous); HW a multiple of 64.
 * ------------------------------------------------------------------------------------------- */
int aadg_conv1x1_wgrad_supported(int Co, int Ci, int HW);
int aadg_conv1x1_wgrad_bf16(const void* dy, const void* x, float* dweight, int N, int Co, int Ci, int HW, void* stream);
/* ABI 8 -- f32x3 (see aadg_conv1x1_nchw_f32x3): float32 NCHW dy / x, float32 precision; HW a multiple of 32 */
int aadg_conv1x1_wgrad_f32x3(const float* dy, const float* x, float* dweight, int N, int Co, int Ci, int HW, void* stream);
/* ABI 10 -- normalise + ReLU on operand load: a BatchNorm + ReLU between two convolutions without its elementwise pass.  The producer
 * leaves the float64 totals of its output x (..._stats), aadg_bn_finalize_f32 turns them into mean / invstd / running statistics and
 * scale / shift [C], and the consuming 1x1 convolution (forward and weight gradient) reads x and applies max(x * scale[c] + shift[c], 0)
 * while it stages its operand -- the normalised tensor is never written.  The BatchNorm's backward is aadg_bn_backward with act = ReLU and
 * no stored output / mask (re-derived from x with the same coefficients).  Whole tiles of the transformed operand only
 * (aadg_conv1x1_f32x3_pre_supported: K % 32, HW % 256 == 0, K <= 512; aadg_conv1x1_wgrad_f32x3_pre_supported) -- AADG_E_UNSUPPORTED otherwise. */
int aadg_bn_finalize_f32(const double* sums, const float* weight, const float* bias, float* running_mean, float* running_var, float momentum,
                         float eps, int C, float* save_mean, float* save_invstd, float* scale, float* shift, void* stream);
/* ... and a bottleneck's projection shortcut: its BatchNorm (no activation) is applied while the main branch's BatchNorm kernel reads the
 * residual -- aadg_bn_sync_forward(phase 2) of float32 tensors with the shortcut's raw convolution output as `residual` and its scale /
 * shift (aadg_bn_finalize_f32); the shortcut BatchNorm's backward is aadg_bn_backward with act = none on the residual gradient. */
int aadg_bn_forward_res_affine_f32(const float* x, const float* residual, const float* res_scale, const float* res_shift, float* y,
                                   void* act_mask, const float* weight, const float* bias, float* running_mean, float* running_var,
                                   float momentum, float eps, int act, int N, int C, int HW, float* save_mean, float* save_invstd, double* sums,
                                   void* ws, size_t ws_bytes, void* stream);
/* ... and the backward of that pair in the two passes of one BatchNorm backward: the shortcut's sums are taken beside the main branch's
 * (its output gradient IS the main branch's masked gradient), its input gradient dx2 is written beside dx.  ws2: a second workspace. */
int aadg_bn_backward_res_bn_f32(const float* x, const void* act_mask, const float* dy, const void* const* dy_extra, int n_extra,
                                const float* dy_plane_const, const float* weight, const float* bias, const float* save_mean,
                                const float* save_invstd, int act, float* dx, float* dres, float* dweight, float* dbias, const float* x2,
                                const float* weight2, const float* save_mean2, const float* save_invstd2, float* dx2, float* dweight2,
                                float* dbias2, int N, int C, int HW, void* ws, size_t ws_bytes, void* ws2, size_t ws2_bytes,
                                long long dy_image_stride, void* stream);
/* ... and with synchronised statistics (ABI 11; data-parallel ranks, reference: models/sync_batchnorm/batchnorm.py:102-105 sums the
 * statistics over the replicas): phase 1 leaves both layers' float64 sums in `sums` [4C] (this layer's (sum g, sum g x^) per channel, then
 * the shortcut's) + the masked gradient dres + the LOCAL parameter gradients; the caller all-reduces `sums`; phase 2 writes dx / dx2 from
 * the totals and the forward's all-reduced element count `count` (device pointer). */
int aadg_bn_sync_backward_res_bn_f32(int phase, const float* x, const void* act_mask, const float* dy, const void* const* dy_extra, int n_extra,
                                     const float* dy_plane_const, const float* weight, const float* bias, const float* save_mean,
                                     const float* save_invstd, int act, float* dx, float* dres, float* dweight, float* dbias, const float* x2,
                                     const float* weight2, const float* save_mean2, const float* save_invstd2, float* dx2, float* dweight2,
                                     float* dbias2, int N, int C, int HW, double* sums, const double* count, void* ws, size_t ws_bytes, void* ws2,
                                     size_t ws2_bytes, long long dy_image_stride, void* stream);
int aadg_conv1x1_f32x3_pre_supported(int M, int K, int HW);
int aadg_conv1x1_wgrad_f32x3_pre_supported(int N, int Co, int Ci, int HW);
int aadg_conv1x1_nchw_f32x3_pre(const void* a_hi, const void* a_lo, const float* in, float* out, int N, int M, int K, int HW,
                                const float* pre_scale, const float* pre_shift, double* bn_sums, void* stream);
int aadg_conv1x1_wgrad_f32x3_pre(const float* dy, const float* x, float* dweight, int N, int Co, int Ci, int HW, const float* pre_scale,
                                 const float* pre_shift, void* stream);
/* ... and the 3x3 pair (a bottleneck's bn1 in front of conv2): the forward needs bn_sums (its own output's statistics), K <= 512 and a
 * shape of aadg_conv3x3_f32x3_stats_supported; zero padding as for the normalised tensor. */
int aadg_conv3x3_nchw_f32x3_pre(const void* a9_hi, const void* a9_lo, const float* in, float* out, int N, int M, int K, int H, int W,
                                int dilation, const float* pre_scale, const float* pre_shift, double* bn_sums, void* stream);
int aadg_conv3x3_wgrad_f32x3_pre(const float* dy, const float* x, float* dweight9, int N, int Co, int Ci, int H, int W, int dilation,
                                 const float* pre_scale, const float* pre_shift, void* stream);

/* Per-step re-layout of the float32 master weights of the convolutions above, every layer in one launch (csrc/weight_layouts.hip):
 * w [Co][Ci][taps] float32 -> plain [Co][Ci][taps], fwd [taps][Co][Ci], bwd [taps'][Ci][Co] bfloat16 (taps' = taps - 1 - t when flip,
 * the mirrored taps of a stride-1 input gradient; each output may be NULL).  Replaces the reference-side `weight.to(bfloat16)` under
 * autocast plus one permute + contiguous per convolution and direction.  `items` is a DEVICE array, `tiles` a DEVICE int32
 * [n_tiles][3] = (item, first out channel, first in channel) of every tile of every item -- 32 out x 32 in channels, 32 x 256 for
 * taps == 1; taps <= 9. */
typedef struct aadg_wl_item {
    const void* w;
    void* plain;
    void* fwd;
    void* bwd;
    int32_t Co, Ci, taps, flip;
} aadg_wl_item;
int aadg_weight_layouts_bf16(const aadg_wl_item* items, const int32_t* tiles, int n_tiles, void* stream);
/* ABI 8: the same layouts as (hi, lo) bfloat16 halves for the f32x3 kernels -- every output buffer holds 2 * Co * Ci * taps elements:
 * the hi plane first, the lo plane Co * Ci * taps elements behind it */
int aadg_weight_layouts_split_bf16(const aadg_wl_item* items, const int32_t* tiles, int n_tiles, void* stream);

/* Weight gradient of a 3x3 / stride-1 convolution with padding = dilation (the bottleneck conv2 of the ResNet stages), NCHW bfloat16:
 *     dweight9[kh * 3 + kw][o][c] = sum_{n, y, x} dy[n][o][y][x] * x[n][c][y + (kh - 1) d][x + (kw - 1) d]     (zero outside the image)
 * dy [N, Co, H, W], x [N, Ci, H, W], dweight9 float32 [9, Co, Ci] (tap-major; permute(1, 2, 0) gives torch's [Co, Ci, 3, 3]).
 * W in {32, 64, 128}, d in {1, 2}.  Replaces MIOpen's NHWC igemm_wrw + its transposes, zero-fill and cast (csrc/conv3x3_wgrad.hip). */
int aadg_conv3x3_wgrad_supported(int Co, int Ci, int H, int W, int dilation);
/* the same for stride 2 (padding 1, dilation 1; the first block of ResNet stages 2 and 3): dy [N, Co, Ho, Wo], x [N, Ci, 2 Ho, 2 Wo],
 * Wo in {32, 64} */
int aadg_conv3x3s2_wgrad_supported(int Co, int Ci, int Ho, int Wo);
int aadg_conv3x3s2_wgrad_bf16(const void* dy, const void* x, float* dweight9, int N, int Co, int Ci, int Ho, int Wo, void* stream);
/* Input gradient of the same stride-2 convolution: dx [N, C, 2 Ho, 2 Wo] from dy [N, M, Ho, Wo] and a9t [9, C, M] bfloat16 with
 * a9t[kh * 3 + kw][c][m] = weight[m][c][kh][kw]; M % 8 == 0, Wo in {32, 64}.  Four parity classes of output pixels = four small
 * stride-1 convolutions over dy (1 + 2 + 2 + 4 taps) in one kernel, no zero-stuffed intermediate (csrc/conv3x3_s2_dgrad.hip). */
int aadg_conv3x3s2_dgrad_supported(int C, int M, int Ho, int Wo);
/* ... and its forward: out [N, M, Ho, Wo] from in [N, K, 2 Ho, 2 Wo] and a9 [9, M, K] bfloat16 (a9[kh * 3 + kw][m][k] = weight[m][k][kh][kw]);
 * K % 8 == 0, Wo in {32, 64}.  The stride is folded into the LDS staging (even / odd column planes); replaces MIOpen's NHWC igemm_fwd and
 * the two layout transposes around it (csrc/conv3x3_s2_fwd.hip). */
int aadg_conv3x3s2_nchw_supported(int M, int K, int Ho, int Wo);
int aadg_conv3x3s2_nchw_bf16(const void* a9, const void* in, void* out, int N, int M, int K, int Ho, int Wo, void* stream);
/* ABI 8 -- f32x3 (see aadg_conv1x1_nchw_f32x3): the stride-2 convolution, its input gradient and its weight gradient on float32 NCHW
 * tensors at float32 precision; a9_hi / a9_lo ([9, M, K]) and a9t_hi / a9t_lo ([9, C, M]) = the bfloat16 halves of the tap-major weights */
int aadg_conv3x3s2_nchw_f32x3(const void* a9_hi, const void* a9_lo, const float* in, float* out, int N, int M, int K, int Ho, int Wo,
                              void* stream);
int aadg_conv3x3s2_dgrad_f32x3(const void* a9t_hi, const void* a9t_lo, const float* dy, float* dx, int N, int C, int M, int Ho, int Wo,
                               void* stream);
int aadg_conv3x3s2_wgrad_f32x3(const float* dy, const float* x, float* dweight9, int N, int Co, int Ci, int Ho, int Wo, void* stream);
int aadg_conv3x3s2_dgrad_bf16(const void* a9t, const void* dy, void* dx, int N, int C, int M, int Ho, int Wo, void* stream);
/* The convolution itself and its input gradient, NCHW bfloat16 in and out, float32 accumulation (csrc/conv3x3_fwd.hip):
 *     out[n][m][y][x] = sum_{k, kh, kw} a9[kh * 3 + kw][m][k] * in[n][k][y + (kh - 1) d][x + (kw - 1) d]
 * forward: a9[t][o][c] = weight[o][c][kh][kw]; input gradient: in = dy, a9[t][c][o] = weight[o][c][2 - kh][2 - kw].
 * a9 [9, M, K] bfloat16; K a multiple of 8; W in {32, 64, 128}, d in {1, 2}. */
int aadg_conv3x3_nchw_supported(int M, int K, int H, int W, int dilation);
int aadg_conv3x3_nchw_bf16(const void* a9, const void* in, void* out, int N, int M, int K, int H, int W, int dilation, void* stream);
int aadg_conv3x3_wgrad_bf16(const void* dy, const void* x, float* dweight9, int N, int Co, int Ci, int H, int W, int dilation,
                            void* stream);
/* ABI 8 -- f32x3 (see aadg_conv1x1_nchw_f32x3): float32 NCHW tensors, float32 precision; a9_hi / a9_lo [9, M, K] bfloat16 halves of the
 * tap-major float32 weights.  The weight gradient excludes W = 128 with d = 2 (LDS). */
int aadg_conv3x3_nchw_f32x3(const void* a9_hi, const void* a9_lo, const float* in, float* out, int N, int M, int K, int H, int W,
                            int dilation, void* stream);
/* ABI 10 -- the same with the BatchNorm statistics of `out` from the epilogue (see aadg_conv1x1_nchw_f32x3_stats: bn_sums float64
 * [2 M + 1] = (sum, sum of squares) per output channel + the element count, zeroed by the call; NULL = none).  Only the shapes of
 * aadg_conv3x3_f32x3_stats_supported (M % 64 == 0, K % 16 == 0: every bottleneck of the backbone) -- AADG_E_UNSUPPORTED otherwise. */
int aadg_conv3x3_f32x3_stats_supported(int M, int K, int H, int W, int dilation);
int aadg_conv3x3_nchw_f32x3_stats(const void* a9_hi, const void* a9_lo, const float* in, float* out, int N, int M, int K, int H, int W,
                                  int dilation, double* bn_sums, void* stream);
int aadg_conv3x3_wgrad_f32x3(const float* dy, const float* x, float* dweight9, int N, int Co, int Ci, int H, int W, int dilation,
                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * Policy controller (reference: models/controller.py:9-145) and its PPO update (losses.py:117-157) as fused kernels.
 * `params`: 9 device pointers in the module's parameter order -- embedding.weight [n_ops + n_mags, E],
 * lstm.weight_ih [4H, E], lstm.weight_hh [4H, H], lstm.bias_ih [4H], lstm.bias_hh [4H], outop.weight [n_ops, H],
 * outop.bias, outmag.weight [n_mags, H], outmag.bias (float32).  Q sub-policies x S = 2L decisions (op, magnitude,
 * op, ...), LSTM state reset per sub-policy, logits squashed as (C/T) * tanh(z).
 * sample: `uniforms` [M, Q*S] in [0, 1) drive an inverse-CDF draw per decision; writes policies [M, Q*S] int64, mean
 *   op / magnitude probabilities, sum log-prob [M], sum entropy [M]  (controller.sample / forward).
 * ppo_update: n_updates x { evaluate(policies) -> ratio = exp(lp - old) -> -min(ratio R, clip(ratio, 1 -+ clip) R).mean()
 *   -> backward -> Adam(lr, beta1, beta2, eps; step = step0 + 1 ...) } in place on params / exp_avg / exp_avg_sq;
 *   loss_terms [n_updates, M] receives the per-row surrogate terms (their mean over M is the update's loss).
 *   (round 5: for the module's widths 32 / 100, S <= 8 and M Q S <= 128 an epoch is two launches of M Q workgroups, one per
 *   (policy, sub-policy) sequence -- k_ppo_rollout, k_ppo_grad_adam -- instead of the M-workgroup rollout + per-parameter Adam.)
 * `ws`: aadg_controller_workspace_bytes() bytes, ZERO-filled once by the caller before the first call, then owned by the kernels.
 * ------------------------------------------------------------------------------------------- */
int aadg_controller_supported(int M, int Q, int S, int E, int H, int n_ops, int n_mags);
size_t aadg_controller_workspace_bytes(int M, int Q, int S, int E, int H, int n_ops, int n_mags);
int aadg_controller_sample_f32(void* const* params, int M, int Q, int S, int E, int H, int n_ops, int n_mags,
                               float c_over_t, const float* uniforms, long long* policies, float* op_probs,
                               float* mag_probs, float* log_probs, float* entropies, void* ws, size_t ws_bytes,
                               void* stream);
int aadg_controller_ppo_update_f32(void* const* params, void* const* exp_avg, void* const* exp_avg_sq, int M, int Q,
                                   int S, int E, int H, int n_ops, int n_mags, float c_over_t,
                                   const long long* policies, const float* old_log_probs, const float* reward,
                                   float clip, int n_updates, int step0, float lr, float beta1, float beta2, float eps,
                                   float* loss_terms, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Embedding prologue of the Sinkhorn reward: the EMA branch of MomentumFeatureDiscriminator under no_grad
 * (reference: models/discriminator.py:48-51, called at search_dg.py:133-135).
 *   fe[n]  = LeakyReLU_slope(W1 x[n] + b1)   x [N, C] (row stride ldx), W1 [E, C], fe [N, E] -> feeds
 *                                            aadg_sinkhorn_rewards_f32 directly
 *   out[n] = W2 fe[n] + b2                   W2 [D, E], out [N, D]; out == NULL skips it
 * C <= 4096, E <= 256.
 * ------------------------------------------------------------------------------------------- */
int aadg_embed_prologue_f32(const float* x, int ldx, int N, int C, const float* W1, const float* b1, int E,
                            const float* W2, const float* b2, int D, float slope, float* fe, float* out,
                            void* stream);
/* the same, also writing row_norm[n] = |fe[n]|_2 for aadg_sinkhorn_rewards_norm_f32 (SURVEY (f)1: the producer hands the cosine
 * cost its denominators).  The rows themselves stay un-normalised: geomloss derives the epsilon schedule from the diameter of
 * the RAW clouds. */
int aadg_embed_prologue_norm_f32(const float* x, int ldx, int N, int C, const float* W1, const float* b1, int E,
                                 const float* W2, const float* b2, int D, float slope, float* fe, float* out,
                                 float* row_norm, void* stream);

/* ---------------------------------------------------------------------------------------------
 * "Resize to the stride-4 grid and add" of the all-MLP segmentation head (BASELINE configs[4]; the reference head resizes each
 * stage's projection with bilinear, align_corners = False and concatenates them:
 * models/mmseg/models/decode_heads/segformer_head.py:66-80 -- aadg_amd/models/segformer.py folds the fuse convolution into the
 * projections, which turns the concatenation into this sum).
 *   out [planes, H, W] = full [planes, H, W] (NULL = 0) + sum_i bilinear(lows[i] [planes, low_h[i], low_w[i]]),  n_low <= 3
 * dtype 0 float32 / 1 bfloat16 for all tensors; low_h / low_w are host arrays.  Backward: d full = d out;
 * aadg_upsample_sum_backward gives d lows[i] [planes, h, w] from d out (gathered, no atomics).
 * ------------------------------------------------------------------------------------------- */
int aadg_upsample_sum(const void* full, const void* const* lows, const int* low_h, const int* low_w, int n_low, void* out,
                      int planes, int H, int W, int dtype, void* stream);
int aadg_upsample_sum_backward(const void* dout, void* dlow, int planes, int h, int w, int H, int W, int dtype, void* stream);
/* every dlow_i in ONE pass over dout (one workgroup per plane, the plane resident in LDS, separable transposed interpolation):
 * H * W <= 128 * 128, H, W <= 256 */
int aadg_upsample_sum_backward_all_supported(int H, int W, const int* low_h, const int* low_w, int n_low);
int aadg_upsample_sum_backward_all(const void* dout, void* const* dlows, const int* low_h, const int* low_w, int n_low, int planes,
                                   int H, int W, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AADG_HIP_H */

/*
 * gags_next.h -- C ABI of the kernels either side of the rasterizer, same
 * library (libgags_hip.so) and conventions as gags_raster.h: extern "C", device pointers, caller-owned memory,
 * `stream` = hipStream_t as void*, return GAGS_OK or a negative GAGS_E* code.  fp32 unless noted.
 *
 * Feature / scale maps are CHANNEL-MAJOR [C, H, W] as everywhere in the reference after
 * gaussian_renderer/__init__.py:73 (`permute(2, 0, 1)`); a segmentation map holds segment ids as floats, -1 = none
 * (scene/cameras.py seg_map, preprocess.py:332-336).
 */
#ifndef GAGS_NEXT_H
#define GAGS_NEXT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what these headers declare is exported */
#pragma GCC visibility push(default)

/* ---- N2: losses and ground-truth assembly of train.py:149-172 -------------------------------------------------- */

/* utils/loss_utils.py:138-154 get_trained_seg: 5x5 mean filter (zero padded) of scale_map[3,h,w], arg-max level,
 * out[h,w] = seg_map[1 + level] (seg_map[4,h,w]).  No gradient (arg-max). */
int gags_trained_seg(int h, int w, const float *seg_map, const float *scale_map, float *out, void *stream);

/* utils/loss_utils.py:59-66 scale_regulation_loss: acc[0] += sum(-s * log(s + 1e-6)) over n values (acc: one
 * double, zeroed by the caller; loss = acc / n).  Backward: v_s[i] = -(log(s + eps) + s / (s + eps)) * v_over_n. */
int gags_entropy_fwd(int64_t n, const float *s, double *acc, void *stream);
int gags_entropy_bwd(int64_t n, const float *s, float v_over_n, float *v_s, void *stream);
/* the same with the cotangent v[0] read on the device (v_over_n = v[0] / n): no host readback inside a backward pass */
int gags_entropy_bwd_dev(int64_t n, const float *s, const float *v, float *v_s, void *stream);

/* Per-segment first and second moments of a channel-major map x[c, n_pix] under the segment map seg[n_pix]
 * (ids in [0, n_seg), anything negative = no segment): s1[n_seg, c], s2[n_seg, c] (double) and cnt[n_seg], all
 * zeroed by the caller.  This is the one pass over the pixels behind both segment losses, which the reference
 * computes with a Python loop over the segment ids (utils/loss_utils.py:47-54 Scale_balance_loss with c = 1,
 * :117-133 scale_region_regulation_loss with c = the feature width). */
int gags_segment_stats(int64_t n_pix, int c, const float *x, const float *seg, int n_seg, double *s1, double *s2,
                       int32_t *cnt, void *stream);
/* The same with `copies` private accumulator sets s1 / s2 [copies, n_seg, c], cnt [copies, n_seg] (zero-filled by the
 * caller, summed by the caller): a workgroup adds into set (block index mod copies).  The double atomics serialize per
 * address at the memory side; with a few hundred segments per image that, not bandwidth, bounds the one-set kernel.
 * layout 1: x is pixel-major [n_pix, c] (the rasterizer's own memory under the [C,H,W] view: no `.contiguous()` copy). */
int gags_segment_stats_multi(int64_t n_pix, int c, const float *x, const float *seg, int n_seg, int copies, double *s1,
                             double *s2, int32_t *cnt, int layout, void *stream);
/* The same moments by runs of equal ids (round 6; csrc/losses.hip segment_stats_runs_kernel): no atomics in global memory, the
 * cost per pixel independent of how finely the map is cut.  Serves c == 16 pixel-major (layout 1) and c == 1 with
 * n_seg * (16 c + 4) <= 150 KB; gags_segment_stats_runs_copies returns the number of private copies s1 / s2 / cnt must hold
 * ([copies, n_seg, c], [copies, n_seg]; every element is written: no zero fill), or 0 when the shape is not served -- use
 * gags_segment_stats_multi then.  The caller sums the copies. */
int gags_segment_stats_runs_copies(int64_t n_pix, int c, int n_seg, int layout);
int gags_segment_stats_runs(int64_t n_pix, int c, const float *x, const float *seg, int n_seg, int copies, double *s1,
                            double *s2, int32_t *cnt, int layout, void *stream);
/* The two segment losses from the moments' private copies ([copies, n_seg, c] doubles, [copies, n_seg] counts, as left by
 * gags_segment_stats_multi / _runs), in two launches: the copies summed in copy order into s1 / s2 [n_seg, c] and cnt [n_seg], then
 *   mode 0 (c == 1), Scale_balance_loss (utils/loss_utils.py:32-57, mix_seg=True): loss[0] = mean over the present segments of
 *           the segment's mean; coef[i] = 1 / (n_i K), 0 for an absent segment (K = present segments, at least 1);
 *   mode 1, scale_region_regulation_loss (:103-136, mix_seg=True): loss[0] = sum over segments of >= 2 pixels of
 *           n_i mean_c var_c / n_pix (unbiased variance, clamped at 0); mean[n_seg, c] and coef[i] = 2 n_i / ((n_i - 1) c n_pix).
 * All arithmetic in double, results rounded to float once. */
int gags_segment_loss(int mode, int n_seg, int c, int copies, int64_t n_pix, const double *s1c, const double *s2c,
                      const int32_t *cntc, double *s1, double *s2, int32_t *cnt, float *loss, float *coef, float *mean,
                      void *stream);
/* Backward of the region-variance loss: v_x[c, p] = coef[seg(p)] * (x[c, p] - mean[seg(p), c]), 0 outside segments. */
int gags_region_var_bwd(int64_t n_pix, int c, const float *x, const float *seg, int n_seg, const float *mean,
                        const float *coef, float *v_x, void *stream);
/* ... with x and v_x pixel-major [n_pix, c] when layout = 1. */
int gags_region_var_bwd_layout(int64_t n_pix, int c, const float *x, const float *seg, int n_seg, const float *mean,
                               const float *coef, float *v_x, int layout, void *stream);
/* The same for a pixel-major map (c % 4 == 0, 16-byte aligned) with another consumer's gradient of that map added on the way
 * out: v_x = add + coef[seg] (x - mean[seg]) -- one pass instead of this kernel's and an element-wise sum's. */
int gags_region_var_bwd_add(int64_t n_pix, int c, const float *x, const float *seg, int n_seg, const float *mean,
                            const float *coef, const float *add, float *v_x, void *stream);
/* Backward of the segment-balanced mean: out[p] = coef[seg(p)], 0 outside segments. */
int gags_gather_seg_coef(int64_t n_pix, const float *seg, int n_seg, const float *coef, float *out, void *stream);

/* scene/dataset_readers.py:54-121 read_sam_clip_feature (default mode): for the three granularity levels l = 1..3 of
 * seg_map[4, h, w] gather img_embed[n_emb, c] rows (id -1 reads the LAST row, as Python indexing does), resize
 * bilinearly (align_corners) to the scale map's [H, W] and blend with scale_map[3, H, W]:
 *     feature_map[c, H, W] = sum_l F_l * scale_map[l],   mask[H, W] = all three ids != -1 (nearest resize), 0/1.
 * _bwd_scale: v_scale[l, H, W] = sum_c v_feature[c] * F_l[c]  (scale_map comes from the trainable scale decoder). */
int gags_sam_clip_feature(int c, int H, int W, int h, int w, int n_emb, const float *img_embed, const float *seg_map,
                          const float *scale_map, float *feature_map, float *mask, void *stream);
int gags_sam_clip_feature_bwd_scale(int c, int H, int W, int h, int w, int n_emb, const float *img_embed,
                                    const float *seg_map, const float *v_feature, float *v_scale, void *stream);

/* train.py:165-166 fused: l1_map[H, W] = mean_c |pred * mask - gt * mask| with gt, mask = read_sam_clip_feature(...)
 * WITHOUT materialising the [c, H, W] ground truth (4.25 GB at 1080p x 512) or the two masked copies.
 * Backward for a cotangent v_map[H, W]: v_pred[c, H, W] and v_scale[3, H, W]. */
/* layout: 0 = pred (and v_pred) are [c, H, W]; 1 = they are [H, W, c] (the memory behind the decoder's output when
 * gags_decoder_head wrote it pixel-major): nothing is transposed, every access is a row. */
int gags_distill_l1_map_fwd(int c, int H, int W, int h, int w, int n_emb, const float *pred, const float *img_embed,
                            const float *seg_map, const float *scale_map, float *l1_map, float *mask, int layout,
                            void *stream);
int gags_distill_l1_map_bwd(int c, int H, int W, int h, int w, int n_emb, const float *pred, const float *img_embed,
                            const float *seg_map, const float *scale_map, const float *v_map, float *v_pred,
                            float *v_scale, int layout, void *stream);

/* CNN_decoder's normalising head fused with the distillation L1 (train.py:159-166): from the last layer's fp32 logits
 * x[H*W, ld] (c = ld = 512: the reference's CNN_decoder(16, 512); other widths: GAGS_EINVAL, use the two entries) to
 *     l1_map[H, W] = mean_c | normalize(x) * mask - gt * mask |,   gt, mask = read_sam_clip_feature(...)
 * and back: d l1_map -> dz[H*W, ld] bf16 (gradient of the logits; what gags_decoder_head_bwd would have produced from
 * the loss's [c,H,W] gradient) and v_scale[3, H, W].  The normalised [c,H,W] map and its gradient never exist. */
int gags_decoder_head_distill_fwd(int c, int ld, int H, int W, int h, int w, int n_emb, const float *x,
                                  const float *img_embed, const float *seg_map, const float *scale_map,
                                  float *l1_map, float *mask, void *stream);
int gags_decoder_head_distill_bwd(int c, int ld, int H, int W, int h, int w, int n_emb, const float *x,
                                  const float *img_embed, const float *seg_map, const float *scale_map,
                                  const float *v_map, void *dz_bf16, float *v_scale, void *stream);
/* ... with the logits' gradient in fp32 (dz[H*W, ld] float): the fp32-tensor decoder tiers ("exact", "bf16x2"). */
int gags_decoder_head_distill_bwd_f32(int c, int ld, int H, int W, int h, int w, int n_emb, const float *x,
                                      const float *img_embed, const float *seg_map, const float *scale_map,
                                      const float *v_map, float *dz, float *v_scale, void *stream);

/* ---- N1: the per-pixel decoders (models/networks.py:109-248: stacks of 1x1 convolutions) ---------------------- */

/* fp32 pixel-major x[n_pix, c] (the rasterizer's own [H, W, D] output) -> bf16 y[n_pix, c_pad], zero-padded
 * (c_pad % 32 == 0). */
int gags_decoder_pack_input(int64_t n_pix, int c, int c_pad, const float *x, void *y_bf16, void *stream);
/* One 1x1-conv layer's parameters (w [co, ci] fp32 = Conv2d.weight[:, :, 0, 0], b [co]) in every form the bf16 kernels read,
 * dimensions zero-padded to multiples of 32 (Np, Kp): w_bf16 [Np, Kp] and its transpose wt_bf16 [Kp, Np] row-major, both
 * again in MFMA-fragment order ([R / 32][C / 16][2][32][8]: the A operand of one v_mfma_f32_32x32x16_bf16 as one
 * contiguous kilobyte) for the fused kernels, and the padded fp32 bias.  Round-to-nearest-even, as torch's cast. */
int gags_decoder_pack_layer(int co, int ci, const float *w, const float *b, void *w_bf16, void *wt_bf16, void *w_frag,
                            void *wt_frag, float *bias_pad, void *stream);
/* Every layer of a decoder in one launch: the arguments of gags_decoder_pack_layer as arrays of n_layers (<= 12) entries. */
int gags_decoder_pack_layers(int n_layers, const int *co, const int *ci, const float *const *w, const float *const *b,
                             void *const *w_bf16, void *const *wt_bf16, void *const *w_frag, void *const *wt_frag,
                             float *const *bias_pad, void *stream);

/* One 1x1-convolution layer as a GEMM on the 16-bit matrix cores (bf16 operands, fp32 accumulate):
 *     y[p, n] = act( sum_k (a1[p, k] + a2[p, k]) * w[n, k] + bias[n] ) * (mask_src[p, n] > 0) + residual[p, n]
 * a1, a2 (optional: the residual sums x1 + x2 / x3 + x4 of CNN_decoder.forward), w, mask_src (optional), residual
 * (optional), y_bf16: bf16; bias (optional), y_f32 (optional second output): fp32.  k_in % 32 == 0, n_out % 8 == 0.
 * The backward's input-gradient GEMM is the same call with w = W^T, mask_src = the layer below's output (its ReLU
 * mask) and residual = the gradient arriving over a skip connection. */
int gags_decoder_layer(int64_t n_pix, int n_out, int k_in, const void *a1, const void *a2, const void *w,
                       const float *bias, int relu, const void *mask_src, const void *residual, void *y_bf16,
                       void *y_premask_bf16, float *y_f32, void *stream);
/* (y_premask_bf16, optional: the value before the mask -- the gradient that also travels over a skip connection.) */

/* Weight and bias gradient of one layer: d_w[n_out, k_in] = sum_p dz[p, n] * (a1[p, k] + a2[p, k]),
 * d_b[n_out] = sum_p dz[p, n] (d_b optional); both fp32, OVERWRITTEN.  No atomics: every pixel chunk leaves a partial
 * matrix in `scratch` (gags_decoder_wgrad_scratch_bytes) and the partials are summed in chunk order -- bit-reproducible.
 * dz, a1, a2 (optional): bf16 pixel-major.  n_out % 16 == 0, k_in % 16 == 0. */
int64_t gags_decoder_wgrad_scratch_bytes(int64_t n_pix, int n_out, int k_in);
int gags_decoder_wgrad(int64_t n_pix, int n_out, int k_in, const void *dz, const void *a1, const void *a2, float *d_w,
                       float *d_b, void *scratch, int64_t scratch_bytes, void *stream);
/* The same gradient written in the parameter's own shape (round 6): d_w[co, ci] and d_b[co] are the leading co x ci block of the
 * padded [n_out, k_in] product (co <= n_out, ci <= k_in: what nn.Conv2d(ci, co, 1).weight.grad holds, models/networks.py:145-149),
 * every sum multiplied by out_scale[0] when given (a device scalar: the f16 tier's power of two) -- no slice copy and no
 * element-wise multiply per parameter after the call.  Same sums, same order, same bits as gags_decoder_wgrad. */
int gags_decoder_wgrad_out(int64_t n_pix, int n_out, int k_in, const void *dz, const void *a1, const void *a2, float *d_w,
                           float *d_b, int co, int ci, const float *out_scale, void *scratch, int64_t scratch_bytes,
                           void *stream);

/* Backward of gags_decoder_head: cotangent g (layout 0: [c, n_pix], 1: [n_pix, c]) + the saved logits x[n_pix, ld] ->
 * pixel-major bf16 dz[n_pix, ld]. */
int gags_decoder_head_bwd(int64_t n_pix, int c, int ld, int mode, const float *x, const float *g, void *dz_bf16,
                          int layout, void *stream);

/* bf16 x[n_pix, ld] -> fp32 y[n_pix, c] (first c columns): the decoder's input gradient in the rasterizer's own
 * [H, W, D] layout. */
int gags_decoder_unpack_grad(int64_t n_pix, int c, int ld, const void *x_bf16, float *y, void *stream);

/* Output head: pixel-major fp32 logits x[n_pix, ld] (first c columns) -> out; mode 0 = F.normalize(dim=0)
 * (CNN_decoder, :192), mode 1 = softmax over channels (CNN_scale_decoder, :242).
 * layout 0: CHANNEL-major out[c, n_pix] (the reference's contiguous [C, H, W]); layout 1 (c % 4 == 0, ld <= 512,
 * ld % 32 == 0): PIXEL-major out[n_pix, c] -- the caller views it as [C, H, W] through a permute, exactly like the
 * rasterizer's output, and nothing is transposed on the way in or out. */
int gags_decoder_head(int64_t n_pix, int c, int ld, int mode, const float *x, float *out, int layout, void *stream);

/* CNN_decoder's whole forward chain (models/networks.py:172-190: nine 1x1 convolutions, x3 = conv(x1 + x2),
 * x5 = conv(x3 + x4)) in ONE kernel, bf16 mode: 64-pixel tiles, activations resident in LDS, weights streamed from L2.
 * x [n_pix, c_in] fp32 (c_in <= 32); w_bf16[9]: the padded bf16 matrices gags_decoder_layer takes ([256, 32], 7 x [256,
 * 256], [n_last, 256]) re-ordered into MFMA fragments: [N / 32][K / 16][lane = 32 kh + n][8 values k = 16 s + 8 kh ..]; bias[9] fp32; acts_bf16[9] (or NULL, or NULL entries): a0 [n_pix, 32] and the eight hidden
 * activations [n_pix, 256] kept for the backward -- entries 3 and 6 hold the residual SUMS x1 + x2 and x3 + x4 (the inputs
 * of layers 3 and 6: what their weight gradients contract), not x2 / x4 --; logits [n_pix, n_last] fp32, n_last % 256 == 0.
 * Bit-identical to the same chain run through gags_decoder_layer. */
int gags_decoder_fwd_fused(int64_t n_pix, int c_in, int n_last, const float *x, const void *const *w_bf16,
                           const float *const *bias, void *const *acts_bf16, void *masks, float *logits, void *stream);
/* (masks, optional: uint32 [8, n_pix rounded up to a multiple of 64, 8] -- the ReLU decisions [activation > 0] of the eight hidden activations as bits, word
 * n / 32 of a pixel for channel n; the bit order inside a word is private to this kernel and gags_decoder_bwd_fused, which
 * reads the words instead of the activations themselves: an opaque buffer to the caller.) */

/* ... and the nine input-gradient GEMMs of its backward in one kernel: dz_last [n_pix, n_last] bf16 (from the head's
 * backward) -> dz_bf16[0..7] = the gradients at the outputs of layers 0..7 ([n_pix, 256] bf16 each, what the weight
 * gradients contract), gin [n_pix, c_in] fp32 (optional).  wt_bf16[9]: the TRANSPOSED padded matrices ([32, 256], 7 x
 * [256, 256], [256, n_last]) in the same fragment order; masks: the bit masks gags_decoder_fwd_fused kept (the two skip gradients stay in registers).  Bit-identical to the chain of gags_decoder_layer calls with mask_src / residual / y_premask. */
int gags_decoder_bwd_fused(int64_t n_pix, int c_in, int n_last, const void *dz_last_bf16, const void *const *wt_bf16,
                           const void *masks, void *const *dz_bf16, float *gin, void *stream);
/* (the same with the input gradient multiplied by the device scalar gin_scale[0] on its way out -- the f16 tier's 1 / S) */
int gags_decoder_bwd_fused_scaled(int64_t n_pix, int c_in, int n_last, const void *dz_last_bf16, const void *const *wt_bf16,
                                  const void *masks, void *const *dz_bf16, float *gin, const float *gin_scale, void *stream);

/* CNN_scale_decoder (models/networks.py:220-248: 16 -> 64 -> 128 -> 64 -> 32 -> 16 -> 3) as one kernel, bf16 mode: x
 * [n_pix, c_in <= 32] fp32; w_bf16[6]: the padded matrices [64,32] [128,64] [64,128] [32,64] [32,32] [32,32] in MFMA-
 * fragment order ([N / 32][K / 16][64][8]); bias[6] fp32 [N_pad]; acts_bf16[6] (optional): a0 [n_pix, 32] and the five
 * hidden activations [n_pix, 64 / 128 / 64 / 32 / 32] kept for the weight gradients; masks (optional): uint32 [n_pix, 11],
 * the ReLU decisions of the five hidden activations as bits (words 0-1, 2-5, 6-7, 8, 9 + one spare); logits [n_pix, 32]
 * fp32 (3 real columns).  Bit-identical to the same chain run through gags_decoder_layer. */
int gags_scale_decoder_fwd_fused(int64_t n_pix, int c_in, const float *x, const void *const *w_bf16,
                                 const float *const *bias, void *const *acts_bf16, void *masks, float *logits, void *stream);
/* The same with the decoder's head fused in (round 6): softmax3 [3, n_pix] fp32 channel-major = softmax over the three real
 * logits (CNN_scale_decoder.forward's last line, models/networks.py:248), bit-identical to gags_decoder_head(mode 1) on the
 * logits; `logits` may then be NULL (nothing needs them: gags_softmax_head_bwd_y works from the output). */
int gags_scale_decoder_fwd_fused_head(int64_t n_pix, int c_in, const float *x, const void *const *w_bf16,
                                      const float *const *bias, void *const *acts_bf16, void *masks, float *logits,
                                      float *softmax3, void *stream);
/* Backward of a softmax head of c <= 4 channels from its output: y, g [c, n_pix] fp32 -> dz [n_pix, ld] 16-bit (columns >= c
 * zero), dz = y (g - <y, g>): what gags_decoder_head_bwd(mode 1, layout 0) computes from the logits, bit for bit. */
int gags_softmax_head_bwd_y(int64_t n_pix, int c, int ld, const float *y, const float *g, void *dz_bf16, void *stream);

/* ... and the five input-gradient GEMMs of its backward in one kernel: dz_last [n_pix, 32] bf16 (from the head's backward)
 * -> dz_bf16[0..4] = the gradients at the outputs of layers 0..4 ([n_pix, 64 / 128 / 64 / 32 / 32] bf16: what the weight
 * gradients contract).  wt_bf16[1..5]: the TRANSPOSED padded matrices ([64,128]... = W_i^T [K_i, N_i]) in fragment order
 * (entry 0 unused); masks: what gags_scale_decoder_fwd_fused kept.  Bit-identical to the chain of gags_decoder_layer calls
 * with mask_src. */
int gags_scale_decoder_bwd_fused(int64_t n_pix, const void *dz_last_bf16, const void *const *wt_bf16, const void *masks,
                                 void *const *dz_bf16, void *stream);

/* ---- N1 at the reference's precision (models/networks.py:109-248 are fp32 Conv2d stacks) ------------------------- */

/* The same layer as gags_decoder_layer with fp32 tensors and fp32-equivalent arithmetic: every operand enters the 16-bit
 * matrix cores as three bfloat16 terms (h + m + l = the fp32 value, exactly) and a product as its six terms of order
 * <= 2 (dropped: <= 2^-24 relative), fp32 accumulation.  a1, a2 (optional), mask_src, residual, y, y_premask: fp32 with
 * leading dimensions lda (inputs) / ldy (everything [n_pix, n_out]-shaped); w [n_out, k_in]; any n_out, k_in >= 1. */
int gags_decoder_layer_exact(int64_t n_pix, int n_out, int k_in, const float *a1, const float *a2, int lda, const float *w,
                             const float *bias, int relu, const float *mask_src, const float *residual, float *y,
                             float *y_premask, int ldy, void *stream);

/* The same two kernels with the number of bfloat16 terms per operand as an argument.  terms = 3: the calls above.
 * terms = 2 (the "bf16x2" tier of gags_amd/decoders.py): h + m = 16 significand bits per operand, three matrix terms per
 * product (h h' + h m' + m h'), relative error <= ~2^-16 per product -- the reference's nn.Conv2d stacks
 * (models/networks.py:145-149, 229-233) run in TF32 (10-bit significands) under torch's defaults on the GPU its README
 * names (README.md:26-31), so this tier is still 32x tighter than the reference's own arithmetic, at half the matrix work. */
int gags_decoder_layer_split(int64_t n_pix, int n_out, int k_in, const float *a1, const float *a2, int lda, const float *w,
                             const float *bias, int relu, const float *mask_src, const float *residual, float *y,
                             float *y_premask, int ldy, int terms, void *stream);
int gags_decoder_wgrad_split(int64_t n_pix, int n_out, int k_in, const float *dz, int lddz, const float *a1,
                             const float *a2, int lda, float *d_w, float *d_b, void *scratch, int64_t scratch_bytes,
                             int terms, void *stream);

/* Weight and bias gradient at the same precision, WITHOUT atomics: pixel chunks -> partial matrices in `scratch` ->
 * summed in chunk order (bit-reproducible).  d_w [n_out, k_in] and d_b [n_out] (optional) are overwritten. */
int64_t gags_decoder_wgrad_exact_scratch_bytes(int64_t n_pix, int n_out, int k_in);
int gags_decoder_wgrad_exact(int64_t n_pix, int n_out, int k_in, const float *dz, int lddz, const float *a1,
                             const float *a2, int lda, float *d_w, float *d_b, void *scratch, int64_t scratch_bytes,
                             void *stream);

/* Backward of the output heads with an fp32 result dz[n_pix, lddz] (columns >= c zero); x = the logits [n_pix, ldx];
 * g: layout 0 = [c, n_pix], 1 = [n_pix, c]; modes as gags_decoder_head. */
int gags_decoder_head_bwd_exact(int64_t n_pix, int c, int ldx, int mode, const float *x, const float *g, int layout,
                                float *dz, int lddz, void *stream);

/* ---- N4: query-time relevancy (eval/openclip_encoder.py:42-56, 96-111) ---------------------------------------- */

/* For every pixel embedding embed[n_pix, c] and every positive phrase j: the LERF relevancy pair
 * probs[j, n_pix, 2] = softmax(10 * (sim_pos_j, sim_neg_k*)) with k* the negative phrase that minimises the positive
 * probability.  pos[n_pos, c], neg[n_neg, c] are unit text embeddings.  One read of the embeddings for all phrases. */
int gags_relevancy(int64_t n_pix, int c, int n_pos, int n_neg, const float *embed, const float *pos, const float *neg,
                   float *probs, void *stream);

/* The rest of the query path of one view (evaluate_iou_loc.py:100-146 `activate_stream`, :163-176
 * `lerf_localization`), for all phrases at once and without leaving the GPU.  valid_map[n_phrases, h, w] = the
 * relevancy maps get_max_across returned (one level).  Per phrase:
 *   avg      = 30x30 box mean of valid_map (cv2.filter2D with ones/900: anchor box/2, BORDER_REFLECT_101)   (:108-111)
 *   blended  = 0.5 * (avg + valid_map)                                      the heat map                      (:113)
 *   output   = clip(((blended - min) / (max - min + 1e-9)) * 2 - 1, 0, 1)                                     (:131-135)
 *   mask_pred   = output > thresh                                            uint8                             (:137)
 *   mask_smooth = eval/utils.py:55-64 smooth(mask_pred): (2 s + 1)^2 majority with the reference's own window bounds (:138)
 *   stats[k] = {min(blended), max(blended), max(avg)}; max(avg) is lerf_localization's score (:174), its position(s)
 *              are where avg == stats[k][2].
 * scratch: gags_relevancy_activate_scratch_bytes() bytes. */
int64_t gags_relevancy_activate_scratch_bytes(int n_phrases, int h, int w);
int gags_relevancy_activate(int n_phrases, int h, int w, const float *valid_map, float thresh, int box, int smooth_scale,
                            float *avg, float *blended, float *output, unsigned char *mask_pred,
                            unsigned char *mask_smooth, float *stats, void *scratch, int64_t scratch_bytes, void *stream);


/* ---- the "f16" decoder tier -------------------------------------------------------------------------------------------
 * The SAME kernels compiled with IEEE half as their 16-bit operand type (csrc/half16.h; v_mfma_f32_32x32x16_f16, fp32
 * accumulation): an 11-bit significand -- exactly the TF32 significand the reference's nn.Conv2d layers
 * (models/networks.py:145-149,229-233) compute with under PyTorch's default torch.backends.cudnn.allow_tf32 = True --
 * instead of bfloat16's 8.  Same signatures and semantics as the entry points above; every `*_bf16` pointer is a half
 * tensor.  Half has 5 exponent bits: conversions saturate at +-65504 (never inf), and the caller multiplies the gradient
 * entering a backward chain by a power of two (gags_amd/decoders.py does: precision="f16") and divides the results. */
/* The tier's gradient scale from the cotangent's magnitude, on the device: out[0] = S = 2^floor(target_log2 - log2(amax[0] / div))
 * (exponent clamped to +-100; S = 1 when amax is 0 or not finite), out[1] = 1 / S. */
int gags_pow2_scale(const float *amax, float div, float target_log2, float *out, void *stream);
int gags_decoder_pack_layer_h16(int co, int ci, const float *w, const float *b, void *w_bf16, void *wt_bf16, void *w_frag, void *wt_frag, float *bias_pad, void *stream);
int gags_decoder_pack_layers_h16(int n_layers, const int *co, const int *ci, const float *const *w, const float *const *b, void *const *w_bf16, void *const *wt_bf16, void *const *w_frag, void *const *wt_frag, float *const *bias_pad, void *stream);
int gags_decoder_pack_input_h16(int64_t n_pix, int c, int c_pad, const float *x, void *y_bf16, void *stream);
int gags_decoder_layer_h16(int64_t n_pix, int n_out, int k_in, const void *a1, const void *a2, const void *w, const float *bias, int relu, const void *mask_src, const void *residual, void *y_bf16, void *y_premask_bf16, float *y_f32, void *stream);
int gags_decoder_head_h16(int64_t n_pix, int c, int ld, int mode, const float *x, float *out, int layout, void *stream);
int64_t gags_decoder_wgrad_scratch_bytes_h16(int64_t n_pix, int n_out, int k_in);
int gags_decoder_wgrad_h16(int64_t n_pix, int n_out, int k_in, const void *dz, const void *a1, const void *a2, float *d_w, float *d_b, void *scratch, int64_t scratch_bytes, void *stream);
int gags_decoder_wgrad_out_h16(int64_t n_pix, int n_out, int k_in, const void *dz, const void *a1, const void *a2, float *d_w, float *d_b, int co, int ci, const float *out_scale, void *scratch, int64_t scratch_bytes, void *stream);
int gags_decoder_head_bwd_h16(int64_t n_pix, int c, int ld, int mode, const float *x, const float *g, void *dz_bf16, int layout, void *stream);
int gags_decoder_unpack_grad_h16(int64_t n_pix, int c, int ld, const void *x_bf16, float *y, void *stream);
int gags_decoder_fwd_fused_h16(int64_t n_pix, int c_in, int n_last, const float *x, const void *const *w_bf16, const float *const *bias, void *const *acts_bf16, void *masks, float *logits, void *stream);
int gags_decoder_bwd_fused_h16(int64_t n_pix, int c_in, int n_last, const void *dz_last_bf16, const void *const *wt_bf16, const void *masks, void *const *dz_bf16, float *gin, void *stream);
int gags_decoder_bwd_fused_scaled_h16(int64_t n_pix, int c_in, int n_last, const void *dz_last_bf16, const void *const *wt_bf16, const void *masks, void *const *dz_bf16, float *gin, const float *gin_scale, void *stream);
int gags_scale_decoder_bwd_fused_h16(int64_t n_pix, const void *dz_last_bf16, const void *const *wt_bf16, const void *masks, void *const *dz_bf16, void *stream);
/* the fused head + distillation L1 backward (gags_decoder_head_distill_bwd) with the logits' gradient as IEEE half, multiplied by
 * the power of two dz_scale[0] (a DEVICE float: chosen by the caller without a host sync) and saturated at +-65504 */
int gags_decoder_head_distill_bwd_h16(int c, int ld, int H, int W, int h, int w, int n_emb, const float *x,
                                     const float *img_embed, const float *seg_map, const float *scale_map,
                                     const float *v_map, void *dz_f16, const float *dz_scale, float *v_scale, void *stream);
int gags_scale_decoder_fwd_fused_h16(int64_t n_pix, int c_in, const float *x, const void *const *w_bf16, const float *const *bias, void *const *acts_bf16, void *masks, float *logits, void *stream);
int gags_scale_decoder_fwd_fused_head_h16(int64_t n_pix, int c_in, const float *x, const void *const *w_bf16, const float *const *bias, void *const *acts_bf16, void *masks, float *logits, float *softmax3, void *stream);
int gags_softmax_head_bwd_y_h16(int64_t n_pix, int c, int ld, const float *y, const float *g, void *dz_bf16, void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* GAGS_NEXT_H */

/* egorear_train.h — C ABI of the training-step kernels (fwd + bwd + AdamW).
 *
 * Same conventions as egorear_hip.h: plain device pointers and sizes, fp32 channels-last activations, launch on the
 * given stream, no allocation, no synchronisation, int return code (0 / positive hipError_t / negative EGR_E*).
 * The convolution data / weight gradients are egr_conv2d_nhwc_f32 (transposed mode) and egr_conv2d_wgrad_f32 of
 * egorear_hip.h; this header adds everything else a backward pass of the path needs.  Each entry cites the reference
 * operator whose autograd it replaces (paths relative to /root/reference/pose_estimation/).
 */
#ifndef EGOREAR_TRAIN_H
#define EGOREAR_TRAIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- BatchNorm2d in training mode (torchvision BasicBlock bn1/bn2/downsample.1, resnet.py:33-39 after network.train(),
 * pl_wrappers/egoposeformer/pose_3d_mvf_ex.py:115).  `groups` independent modules (the two stereo encoders) normalise
 * rows_per_group consecutive pixels each; per-channel arrays are (groups, c).
 * stats: batch mean / biased variance over the rows -> mean, invstd, and the affine pair (alpha = gamma*invstd,
 * shift = beta - mean*alpha) consumed by egr_scale_shift_f32; running_mean / running_var are updated in place with
 * `momentum` (unbiased variance), exactly what nn.BatchNorm2d writes.  workspace: >= groups*nblk*2*c doubles with
 * nblk = egr_bn_blocks(rows_per_group). */
int32_t egr_bn_blocks(int64_t rows_per_group);
int egr_bn_stats_f32(const float* x, int64_t rows_per_group, int32_t c, int32_t groups, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                     float* alpha, float* shift, double* workspace, size_t workspace_doubles, void* stream);
/* y = x*alpha[c] + shift[c] (+ res) (relu): the normalisation pass.  x, y, res dense (groups*rows_per_group, c). */
int egr_scale_shift_f32(const float* x, const float* alpha, const float* shift, const float* res, float* y,
                        int64_t rows_per_group, int32_t c, int32_t groups, int32_t relu, void* stream);
/* backward: dz = dy * [y > 0] (y may be NULL: no ReLU behind the BN);  dgamma = sum dz*xhat, dbeta = sum dz;
 * dx = alpha * (dz - dbeta/N - xhat*dgamma/N);  dz is also written to dz_out when given (the residual branch). */
int egr_bn_backward_f32(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                        const float* alpha, int64_t rows_per_group, int32_t c, int32_t groups, float* dgamma, float* dbeta,
                        float* dx, float* dz_out, double* workspace, size_t workspace_doubles, void* stream);

/* ---- element-wise pieces.  n = number of floats (multiple of 4), 16-byte aligned pointers. */
int egr_relu_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, void* stream);            /* dx = dy*[y>0]   */
int egr_add_f32(const float* a, const float* b, float* y, int64_t n, void* stream);                   /* y = a + b       */
int egr_gelu_f32(const float* z, float* h, int64_t n, void* stream);                                  /* exact-erf GELU  */
int egr_gelu_bwd_f32(const float* dh, const float* z, float* dz, int64_t n, void* stream);
int egr_rowmask_f32(float* x, const uint8_t* mask, int64_t rows, int32_t c, void* stream);            /* x[r,:] *= mask[r] */
int egr_fill_f32(float* x, float v, int64_t n, void* stream);

/* ---- pooling / resampling.  MaxPool2d with the arg-max window slot recorded (resnet.py:17, egoposeformer_mvf_ex.py:234)
 * and its gather-form backward; Upsample(x2, bilinear, align_corners=True) backward = the exact adjoint of
 * egr_upsample2x_nhwc_f32 (dy optionally masked by [y > 0] for the fused up+ReLU form). */
int egr_maxpool_train_f32(const float* x, float* y, uint8_t* slot, int32_t n, int32_t h, int32_t w, int32_t c, int32_t k,
                          int32_t stride, int32_t pad, void* stream);
int egr_maxpool_bwd_f32(const float* dy, const uint8_t* slot, float* dx, int32_t n, int32_t h, int32_t w, int32_t c,
                        int32_t k, int32_t stride, int32_t pad, void* stream);
int egr_upsample2x_bwd_f32(const float* dy, const float* y, float* dx, int32_t n, int32_t h, int32_t w, int32_t c,
                           void* stream);

/* ---- the stem's BatchNorm2d(train) + ReLU + MaxPool2d(3, 2, 1) (models/backbones/resnet.py:16-17) without its full-resolution
 * tensors (round 6).  Forward: x = raw conv output (n, h, w, c), n = groups * images per group; alpha / shift (groups, c) from
 * egr_bn_stats_ex_f32 / egr_bn_finalize_f32; y (n, ho, wo, c) = maxpool(relu(alpha x + shift)) and the arg-max window slot of every
 * output element - bit-identical to egr_scale_shift_f32(relu) followed by egr_maxpool_train_f32, the normalised tensor is never written.
 * Backward: dpool = gradient of y; per input pixel the gradient of the normalised + rectified tensor is gathered from the windows that
 * chose it and masked by [alpha x + shift > 0]; dgamma / dbeta (groups, c) and dx (n, h, w, c) - bit-identical to egr_maxpool_bwd_f32
 * followed by egr_bn_backward_ex_f32(dy, y, x, ...), neither dy nor y exists.  workspace as egr_bn_backward_ex_f32. */
int egr_bn_relu_maxpool_f32(const float* x, const float* alpha, const float* shift, float* y, uint8_t* slot, int32_t n, int32_t h,
                            int32_t w, int32_t c, int32_t groups, int32_t k, int32_t stride, int32_t pad, void* stream);
int egr_bn_pool_backward_f32(const float* dpool, const uint8_t* slot, const float* x, const float* mean, const float* invstd,
                             const float* alpha, const float* shift, int32_t n, int32_t h, int32_t w, int32_t c, int32_t groups,
                             int32_t k, int32_t stride, int32_t pad, float* dgamma, float* dbeta, float* dx, double* workspace,
                             size_t workspace_doubles, const float* xhat_max, uint32_t* amax_dx, void* stream);

/* ---- layout changes at the boundary: (n, c, hw) channel-major planes <-> (n, hw, cpad) channels-last, zero padded.
 * nmap places image n of the channel-major side (the (B,V,15,64,64) heat maps). */
int egr_planes_to_nhwc_f32(const float* planes, int32_t n_inner, int64_t stride_inner, int64_t stride_outer, float* y,
                           int32_t n, int32_t c, int32_t hw, int32_t cpad, void* stream);
/* the inverse: (n, hw, cpad) channels-last -> the first c channels as (c, hw) planes placed by nmap. */
int egr_nhwc_to_planes_f32(const float* x, float* planes, int32_t n_inner, int64_t stride_inner, int64_t stride_outer,
                           int32_t n, int32_t c, int32_t hw, int32_t cpad, void* stream);
/* stem input patches: (B,V,3,H,W) images -> (n*H/2*W/2, 160) rows of the 7x7/s2/p3 receptive field, (ky,kx,c) order,
 * columns 147..159 zero: turns the stem's weight gradient into a plain 1x1 egr_conv2d_wgrad_f32 (resnet.py:16,49). */
int egr_stem_im2col_f32(const float* img, int32_t n_inner, int64_t stride_inner, int64_t stride_outer, int32_t n, int32_t h,
                        int32_t w, float* cols, void* stream);

/* stem (conv 7x7 / s2 / p3, 3 -> 64) weight gradient straight from the (B,V,3,H,W) images and dy (groups*n, H/2, W/2, 64) NHWC:
 * dw (groups, 64, 147) in the OIHW flattening (ci, kh, kw).  Same patch staging as egr_stem_conv7x7_f32; the pixel is the
 * MFMA reduction index; persistent workgroups write one partial slab each (workspace >= groups*512*64*160 floats), summed in
 * fixed order.  xmap / gx as in egr_stem_conv7x7_f32.  Replaces autograd of resnet.py:16,49's conv1. */
int egr_stem_wgrad_f32(const float* x, int32_t n_inner, int64_t stride_inner, int64_t stride_outer, int32_t n, int32_t h, int32_t w,
                       const float* dy, float* dw, float* workspace, size_t workspace_floats, int32_t groups, int64_t gx, void* stream);

/* ---- LayerNorm(x + res) backward (transformer.py / heatmap_mvf_ex.py:861-935 norms).  pre = the normalised input
 * (x + res) saved by the forward; ds = gradient w.r.t. pre; dgamma/dbeta (groups, c) summed over each group's rows. */
int egr_layernorm_bwd_f32(const float* dy, const float* pre, const float* gamma, float* ds, float* dgamma, float* dbeta,
                          float* rowstats, int32_t rows, int32_t c, float eps, int32_t rows_per_group, void* stream);

/* ---- joint-to-joint attention core backward (SpatialMHA, heatmap_mvf_ex.py:799-817): qkv (b*j, 3*heads*d),
 * dout (b*j, heads*d) -> dqkv. */
int egr_joint_mha_bwd_f32(const float* qkv, const float* dout, float* dqkv, int32_t b, int32_t j, int32_t heads, int32_t d,
                          float scale, void* stream);

/* ---- deformable sampling backward (mmcv ms_deform_attn backward in sample-then-project form, deform_attn.py:122-162).
 * Inputs as egr_msda_gather_f32 plus dg (groups, rows, heads, cf): gradient w.r.t. the sampled un-projected rows,
 * da (groups, rows, heads*dh): gradient w.r.t. the projected head outputs (drives the positional table and the bias
 * mass sigma), cfold (groups, heads*dh): the folded bias.  Outputs: dol (groups, rows, heads*48) gradient w.r.t. the
 * offsets/logits *per view row* (the caller sums the views), dfeat (views,b,hw,cf) and dpos (groups,views,hw,heads*dh)
 * accumulated with atomics when given (must be zero-initialised by the caller). */
int egr_msda_gather_bwd_f32(const float* feat, int32_t cf, const float* pos, int32_t dh, const float* offs_logits,
                            const float* anchors, const uint8_t* valid, int32_t b, int32_t views, int32_t joints,
                            int32_t heads, int32_t hgt, int32_t wid, const float* dg, const float* da, const float* cfold,
                            float* dol, float* dfeat, float* dpos, int32_t groups, void* stream);

/* ---- small reductions: out[c] (+)= sum_r scale[r] * x[r, c] over `rows` rows of leading dimension ld, per group
 * (x group stride gx floats, scale group stride gs, out group stride c).  scale may be NULL. */
int egr_colsum_f32(const float* x, int64_t ld, int64_t rows, int32_t c, const float* scale, float* out, int32_t accumulate,
                   int32_t groups, int64_t gx, int64_t gs, int32_t cols_per_scale, int64_t scale_stride, void* stream);
/* cols_per_scale > 0: column ch uses the scale vector scale + (ch / cols_per_scale) * scale_stride (one vector per attention head) */
/* y[r, :] = sum_{k<fold} x[r*fold + k, :]  (sum the `fold` views of a query row). */
int egr_fold_rows_f32(const float* x, float* y, int64_t rows_out, int32_t fold, int32_t c, void* stream);
/* JQA sum backward (heatmap_mvf_ex.py:664-665): d_embed[g,j,:] = sum_b dx, d_bfb[b,:] = sum_j dx. */
int egr_jqa_sum_bwd_f32(const float* dx, float* d_embed, float* d_bfb, int32_t b, int32_t j, int32_t c, int32_t b_per_group,
                        void* stream);

/* ---- losses of the wrapper's training_step (pose_3d_mvf_ex.py:133-145; MpjpeLoss, pose_metric.py:10-16):
 * loss += weight/rows * sum_r ||gt[r,:] - pred[r,:]||_2 (accumulated into *loss, a device double), and
 * dpred = weight/rows * (pred - gt)/||.|| (0 where the norm is 0; same layout as pred).  d = 3 (poses) or 64
 * (heat-map rows).  Row r starts at (r / inner)*ld + (r % inner)*d floats, so channel-padded buffers are read in place. */
int egr_rownorm_loss_f32(const float* pred, const float* gt, int64_t rows, int32_t d, int32_t inner, int64_t ld_pred,
                         int64_t ld_gt, float weight, double* loss, float* dpred, void* stream);

/* ---- optimiser (pose_3d_mvf_ex.py:219-234 AdamW groups, yaml gradient_clip_val 5.0).
 * sumsq: *out (+)= sum g^2 (device double).  adamw: one fused decoupled-weight-decay Adam update over a flat range;
 * the clip coefficient min(1, clip/(sqrt(*sumsq)+1e-6)) is read from the device, so no host sync sits between backward
 * and the update.  `step` is the 1-based update count. */
/* Abs-max BOUNDS for the fp16 scheme's pre-scales without a pass over the tensors: the BatchNorm launches already
 * reduce over every element, so they also carry the batch extremes per channel and turn them into an upper bound of what the
 * normalised tensor / its gradient can hold - a consumer's pre-scale only needs an upper bound.
 *   egr_bn_stats_ex_f32: + xhat_max (groups*c, out: max |(x - mean) invstd| per channel), amax_res (record of the residual that
 *     egr_scale_shift_f32 will add, or NULL), amax_out (record of y = act(gamma xhat + beta [+ res]): |y| <= |gamma| max|xhat| + |beta|
 *     [+ max|res|]).  Workspace: 1.5 x the doubles of egr_bn_stats_f32 when xhat_max / amax_out are given.
 *   egr_bn_backward_ex_f32: + xhat_max (in), amax_dx (record of dx: |dx| <= |alpha| (max|dy masked| + |dbeta|/n + max|xhat| |dgamma|/n)).
 *   egr_record_bound_f32: out[0] max= scale_a max(a) + scale_b max(b) (b may be NULL): the record of a tensor bounded by its inputs'
 *     records (a sum, a pooling / up-sampling gradient). */
int egr_bn_stats_ex_f32(const float* x, int64_t rows_per_group, int32_t c, int32_t groups, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                        float* mean, float* invstd, float* alpha, float* shift, double* workspace,
                        size_t workspace_doubles, float* xhat_max, const uint32_t* amax_res, uint32_t* amax_out, void* stream);
int egr_bn_backward_ex_f32(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                           const float* alpha, int64_t rows_per_group, int32_t c, int32_t groups, float* dgamma,
                           float* dbeta, float* dx, float* dz_out, double* workspace, size_t workspace_doubles,
                           const float* xhat_max, uint32_t* amax_dx, void* stream);
/* The second half of egr_bn_stats_ex_f32 alone, on slabs a conv launch left behind (egr_conv_aux.bn_partials in egorear_hip.h: the
 * statistics pass over the conv output is folded into the conv's epilogue): partials = [groups][slabs][2][c] doubles followed by the
 * extremes [groups][slabs][2][c] floats. */
int egr_bn_finalize_f32(const double* partials, int32_t slabs, int64_t rows_per_group, int32_t c, int32_t groups, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* mean,
                        float* invstd, float* alpha, float* shift, float* xhat_max, const uint32_t* amax_res, uint32_t* amax_out,
                        void* stream);
/* dst (cols, rows) = src (rows, cols) transposed, tiled through LDS (16-byte aligned, distinct buffers).  Refreshes the data-gradient
 * operand W^T of a very large Linear (egoposeformer_mvf_ex.py:241-253 `mlp_pred.0`: 2048 x 32768) after an optimiser update. */
int egr_transpose_f32(const float* src, int32_t rows, int32_t cols, float* dst, void* stream);
int egr_record_bound_f32(const uint32_t* a, const uint32_t* b, float scale_a, float scale_b, uint32_t* out, void* stream);

/* nn.MSELoss(reduction="mean") * weight of the heat-map training stages (pl_wrappers/egoposeformer/heatmap.py:215-218,
 * heatmap_mvf_ex.py:258-261): loss (device double) += weight * mean (pred - gt)^2 over n elements (n % 4 == 0), dpred (may be NULL)
 * = 2 weight (pred - gt) / n. */
int egr_mse_loss_f32(const float* pred, const float* gt, int64_t n, float weight, double* loss, float* dpred, void* stream);
int egr_sumsq_f32(const float* g, int64_t n, double* out, int32_t accumulate, void* stream);
/* dst[0..3] = {a, b, c, d}: scalars travel as kernel arguments, so a host that runs many steps ahead of the device cannot
 * overwrite a value before it is consumed (a pinned-buffer copy could). */
int egr_set4_f32(float* dst, float a, float b, float c, float d, void* stream);
/* as egr_adamw_f32 with the step-dependent scalars read from device memory: hyper = {lr, 1 - beta1^t, sqrt(1 - beta2^t)}
 * (a captured hipGraph of the whole step replays with fresh values written into that buffer). */
int egr_adamw_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float beta1, float beta2,
                      float eps, float weight_decay, const double* grad_sumsq, float clip, void* stream);
int egr_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t step, const double* grad_sumsq, float clip, void* stream);

/* ---- parameter re-packing, all tensors of a step in ONE launch.  Every optimisation step the kernels' weight layouts
 * (forward operand [cout_pad][cin_pad/32][taps][32], data-gradient operand [cin_pad32][cout_pad/32][taps][32], padded bias
 * vectors, group stacks, row-concatenated projections) must be rebuilt from the nn.Parameters, and the weight gradients
 * must be brought back from the packed K order to OIHW inside the flat gradient buffer.  A table of descriptors (device
 * memory) drives one gather kernel; `blocks` maps every 1024-element block of work to (descriptor, first element). */
typedef struct egr_repack_desc {
    const float* src;
    float* dst;
    int32_t kind;      /* EGR_REPACK_* */
    int32_t rows;      /* valid source rows (cout) */
    int32_t cin;       /* valid input channels taken from the source */
    int32_t cin_tot;   /* channels per source row (row pitch = cin_tot * taps) */
    int32_t ci0;       /* first source channel */
    int32_t cin_pad;   /* channels of the packed side (multiple of 32) */
    int32_t taps;      /* kh * kw */
    int32_t rows_pad;  /* FWD: destination rows; DGRAD: source rows incl. zero padding handled by this descriptor (K side) */
    int32_t k_off;     /* DGRAD: first K-side channel written (row-concatenated sources) */
    int32_t k_tot;     /* DGRAD: K-side channels of the whole destination row (multiple of 32) */
    int64_t total;     /* elements enumerated by this descriptor */
} egr_repack_desc;
#define EGR_REPACK_COPYPAD 0 /* dst[i] = i < rows ? src[i] : 0                                     (bias / vector padding) */
#define EGR_REPACK_FWD 1     /* OIHW (column slice) -> forward operand, zero padded rows / channels                        */
#define EGR_REPACK_DGRAD 2   /* OIHW (column slice) -> data-gradient operand (in/out swapped)                               */
#define EGR_REPACK_UNPACK 3  /* packed gradient [rows][cin_pad/32][taps][32] -> OIHW (column slice) at dst                  */
int egr_repack_f32(const egr_repack_desc* table, const int64_t* blocks /* (desc, first) pairs */, int32_t n_blocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif

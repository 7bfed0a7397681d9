/*
 * egorear_hip.h — C ABI of the MI355X (gfx950) kernels for EgoRear's multi-view
 * inference hot path (heatmap encoder -> MVFEx/JQA refinement -> 2D-to-3D lifting).
 *
 * The reference is pure Python on PyTorch; the boundary it offers for native code is
 * its operator level.  Each entry point below names the reference
 * interface it replaces (paths relative to /root/reference/pose_estimation/).  The
 * one native op the reference itself binds is mmcv's
 * MultiScaleDeformableAttnFunction (models/utils/deform_attn.py:9,155-162); here it is
 * egr_msda_gather_f32 + egr_conv2d_nhwc_f32 (sample-then-project form).
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless noted;
 *   - activations are fp32, channels-last ("NHWC"): pixel-major, channel-minor, with an
 *     explicit pixel stride `ld*` (floats) so that channel slices of a wider buffer can
 *     be read/written in place (virtual concat);
 *   - image index n of a batch maps to memory as
 *         base + (n % n_inner) * stride_inner + (n / n_inner) * stride_outer
 *     which covers plain batches (n_inner = N) and view-major <-> batch-major
 *     re-orderings at the boundary;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous, nothing
 *     synchronises, allocates or frees, so every call is hipGraph-capturable;
 *   - return value: 0 on success, a positive hipError_t from the launch, or a negative
 *     EGR_E* code for arguments the kernels do not support (nothing is launched then).
 */
#ifndef EGOREAR_HIP_H
#define EGOREAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGR_EINVAL (-1)   /* unsupported shape / alignment */
#define EGR_ENULL (-2)    /* required pointer is NULL */
#define EGR_EWORKSPACE (-3) /* workspace too small */

enum { EGR_ACT_NONE = 0, EGR_ACT_RELU = 1, EGR_ACT_GELU = 2 };           /* GELU = exact erf form */
enum { EGR_RES_NONE = 0, EGR_RES_BEFORE_ACT = 1, EGR_RES_AFTER_ACT = 2,
       EGR_RES_UP2_BEFORE_ACT = 3 /* res is a (ho/2, wo/2) tensor: its bilinear x2 upsampling (align_corners=True) is added before the activation */ };

/* image-index -> offset mapping (floats) */
typedef struct {
    int32_t n_inner;
    int64_t stride_inner;
    int64_t stride_outer;
} egr_nmap;

/*
 * Convolution / linear layer as one implicit GEMM on fp32 MFMA.
 *   y[n,ho,wo,co] = act( acc * scale[co] + shift[co] * (rowscale ? rowscale[m] : 1) (+ res) )
 *   acc = sum_{kh,kw,ci} x[n, ho*stride-pad+kh, wo*stride-pad+kw, ci] * w[co, (ci/32, kh, kw, ci%32)]
 * (weights packed [round_up(cout,32)][cin/32][kh*kw][32]: all taps of one 32-channel chunk are adjacent in K)
 * with m = (n*ho_total + ho)*wo_total + wo.  A Linear layer is the case h=w=kh=kw=1, n=rows.
 * Replaces nn.Conv2d(+BatchNorm2d eval)(+ReLU)(+residual) of models/backbones/resnet.py:43-137
 * and of the conv stacks in models/estimator/egoposeformer_heatmap_mvf_ex.py:101-126,522-584 /
 * egoposeformer_mvf_ex.py:144,229-239, and nn.Linear(+ReLU/GELU) of models/utils/transformer.py:8-93,
 * deform_attn.py:60-63, egoposeformer_mvf_ex.py:156-162,241-262.
 */
typedef struct {
    int32_t n, h, w, cin;        /* input: n images of h x w pixels, cin channels (cin % 32 == 0) */
    int32_t cout;                /* true output channels; w holds round_up(cout,32) rows, zero padded */
    int32_t kh, kw, stride, pad;
    int32_t ho, wo;
    int32_t ldx, ldy, ldr;       /* pixel strides (floats) of x, y, res */
    egr_nmap xmap, ymap, rmap;   /* per-image placement of x, y, res */
    int32_t act;                 /* EGR_ACT_* */
    int32_t res_mode;            /* EGR_RES_* */
    int32_t out_nchw;            /* 1: y is written channel-major: ymap(n) + co*ho*wo + pix */
    int32_t split_k;             /* >1: partial sums go through `workspace` (groups * split_k * M * round_up(cout,32) floats) */
    /* Grouped launch: `groups` independent problems of identical shape (e.g. the two stereo estimators, the four
     * per-view refiners — same architecture, own weights) in one launch.  Group g uses x + g*gx, w + g*gw,
     * scale/shift + g*gp, res + g*gr, rowscale + g*grs, rowmask + g*grm, y + g*gy (strides in elements; 0 = shared). */
    int32_t groups;
    int64_t gx, gw, gp, gy, gr, grs, grm;
    /* Transposed (data-gradient) mode, first brick of the training row: with `transposed` = 1 the
     * launch computes dx = conv_transpose(dy, W): x is dy (n, h, w, cin = forward cout), y is dx (n, ho, wo, cout = forward
     * cin) — ho/wo are given, not derived — stride/pad/kh/kw are the FORWARD conv's, and w is the forward weight packed with
     * its in/out channels swapped ([cin_fwd_pad][cout_fwd/32][kh*kw][32]).  Row m = (n, hi, wi) gathers
     * dy[n, (hi+pad-kh)/stride, (wi+pad-kw)/stride, :] for the taps where the division is exact and in range. */
    int32_t transposed;
    /* Weight operand format.  EGR_W_F32 (0): w is the packed fp32 matrix described above, multiplied on the fp32 matrix
     * cores.  EGR_W_BF16X3 (1): w points to the image egr_pack_w6_f32 made of that matrix — every weight as the exact sum
     * of three bf16 numbers, in MFMA-fragment order — and the launch runs on the bf16 matrix cores: the activations are
     * split the same way on the fly and the six partial products of order <= 2 are accumulated in fp32 (the dropped
     * ones are below 2^-25 relative: the result is as close to the exact sum as an fp32 fma chain).
     * gw is then in bf16 elements (egr_w6_elems per group). */
    int32_t w_format;
} egr_conv_desc;
enum { EGR_W_F32 = 0, EGR_W_BF16X3 = 1,
       EGR_W_FORCE = 2 /* egr_conv2d_wgrad_f32 only: with EGR_W_BF16X3, take the split kernel whatever the problem size (tests) */,
       EGR_W_F16X2 = 4 /* egr_conv2d_nhwc_ex_f32 only: w is the image egr_pack_wh2_f32 made (two fp16 planes per weight) */ };

/* Split a packed fp32 weight matrix w (groups, npad, k) — npad = round_up(cout, 32), k = kh*kw*cin, k % 32 == 0 — into
 * the EGR_W_BF16X3 image: per group egr_w6_elems(npad, k) bf16 elements laid out
 * [round_up(npad/32, 4) column fragments][k/32 chunks][2 k16 steps][3 planes hi, mid, lo][64 lanes][8 bf16],
 * lane l of a fragment holding w[col = 32*frag + (l & 31)][k = 32*chunk + 16*step + 8*(l >> 5) + j], j = 0..7.
 * hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid), round to nearest even: hi + mid + lo == w exactly
 * (fragments beyond npad/32 are zero).  img must be 16-byte aligned. */
int64_t egr_w6_elems(int32_t npad, int32_t k);
int egr_pack_w6_f32(const float* w, int32_t npad, int32_t k, int32_t groups, void* img, void* stream);

/* The same for many matrices in one launch.  jobs: DEVICE array of `count` records sorted by first_block, record i covering the
 * workgroups [first_block_i, first_block_i + groups_i * round_up(npad_i/32, 4) * k_i/32); total_blocks = the end of the last one. */
typedef struct {
    const float* w;
    void* img;
    int32_t npad, k, groups, reserved;
    int64_t first_block;
} egr_w6_job;
int egr_pack_w6_many_f32(const egr_w6_job* jobs, int32_t count, int64_t total_blocks, void* stream);

/* Split a packed fp32 weight matrix w (groups, npad, k) into the EGR_W_F16X2 image: every weight, multiplied by the power of two
 * s[co] that puts the largest magnitude of its row (output channel) into [2^14, 2^15), as h = f16(w s) and l = f16(w s - h) (round to
 * nearest even; h + l carries 22 significant bits of w, fp16 subnormals included).  Layout per group (egr_wh2_elems(npad, k) 16-bit
 * elements): [round_up(npad/32, 4) column fragments][k/32 chunks][2 k16 steps][2 planes h, l][64 lanes][8 fp16], lane / k mapping as in
 * egr_pack_w6_f32.  descale (groups, npad) receives 1 / s[co] (exact powers of two, clamped to 2^+-60), which the launch multiplies
 * back into its accumulators.  Replaces nothing in the reference: it is the operand format of the launches that replace
 * nn.Conv2d / nn.Linear (see egr_conv2d_nhwc_f32) on the fp16 matrix cores. */
int64_t egr_wh2_elems(int32_t npad, int32_t k);
int egr_pack_wh2_f32(const float* w, int32_t npad, int32_t k, int32_t groups, void* img, float* descale, void* stream);
/* The same for every job of a device-resident table in two launches (the training step re-splits all its weight operands after every
 * parameter update): first_rblock / first_pblock = running counts of ceil(groups npad / 4) and groups round_up(npad/32, 4) (k/32)
 * over the jobs before this one; total_* their sums. */
typedef struct {
    const float* w;
    void* img;
    float* descale;
    int32_t npad, k, groups, reserved;
    int64_t first_rblock, first_pblock;
} egr_wh2_job;
int egr_pack_wh2_many_f32(const egr_wh2_job* jobs, int32_t count, int64_t total_rblocks, int64_t total_pblocks, void* stream);

int egr_conv2d_nhwc_f32(const egr_conv_desc* d, const float* x, const float* w,
                        const float* scale /* per co, NULL = 1 */, const float* shift /* per co, NULL = 0 */,
                        const float* res /* NULL unless res_mode */, const float* rowscale /* per m, NULL = 1 */,
                        const uint8_t* rowmask /* per m, NULL = keep; 0 -> row written as 0 */,
                        float* y, float* workspace, size_t workspace_floats, void* stream);

/* egr_conv2d_nhwc_f32 with the side operands of the fp16 scheme.  EGR_W_F16X2 launches multiply on
 * v_mfma_f32_32x32x16_f16: both operands as two fp16 planes of the value times an exact power of two, the three products
 * (l,h) (h,l) (h,h) accumulated in fp32, the accumulators multiplied by the inverse powers before the epilogue.  Measured against
 * fp64 the result is as close to the exact sum as an fp32 fma chain (tools/proto/f16x3_accuracy.hip).  The pre-scale of the
 * activations comes from a record of their largest magnitude that the PRODUCING launch left behind:
 *   amax_out  (any w_format; NULL = off) 64 uint32 slots, zero before the launch: the launch folds max |y| over everything it
 *             stores into them (atomic max on the float bits; the record is the maximum over the 64 slots);
 *   amax_in   (EGR_W_F16X2) the record of x: any upper bound of max |x| serves (a tensor derived from x by max-pooling or bilinear
 *             interpolation inherits x's record); the launch scales x by 2^k with max |x| 2^k in [2^14, 2^15), k clamped to +-60;
 *   w_descale (EGR_W_F16X2) the per-channel descale egr_pack_wh2_f32 wrote ((groups,) round_up(cout, 32) floats, group stride gp).
 * w: the fp32 matrix, the egr_pack_w6_f32 image or the egr_pack_wh2_f32 image as w_format says (gw in 16-bit elements for both
 * images).  EGR_W_F16X2 covers forward, transposed (data-gradient) and masked launches alike (the training step runs all three on
 * it); amax_out does not go with out_nchw. */
typedef struct {
    const float* w_descale;
    const uint32_t* amax_in;
    uint32_t* amax_out;
    /* Train-mode BatchNorm behind the conv (nn.BatchNorm2d on batch statistics, models/backbones/resnet.py:43-137 in train()): the
     * launch also leaves, per M tile and output channel, the sum / sum of squares (double) and min / max (float) of what it stores -
     * the slabs egr_bn_stats_f32's first pass would produce by reading y again; egr_bn_finalize_f32 (egorear_train.h) consumes them.
     * bn_partials: 16-byte aligned, bn_capacity doubles; layout [groups][tiles][2][cout] doubles followed by the same shape in floats,
     * `tiles` = *bn_tiles_out (host int, written before the launch is enqueued: the launch's M tiles per group).  Raw NHWC output
     * only (no activation / residual / row modifiers, cout % 64 == 0); EGR_EINVAL otherwise, EGR_EWORKSPACE when the slabs do not fit -
     * the caller then runs the statistics pass. */
    double* bn_partials;
    int32_t* bn_tiles_out;
    int64_t bn_capacity;
} egr_conv_aux;
int egr_conv2d_nhwc_ex_f32(const egr_conv_desc* d, const float* x, const void* w, const float* scale, const float* shift,
                           const float* res, const float* rowscale, const uint8_t* rowmask, float* y, float* workspace,
                           size_t workspace_floats, const egr_conv_aux* aux /* NULL = egr_conv2d_nhwc_f32 */, void* stream);

/* Two 1x1 / stride-1 convolutions back to back in ONE launch (fp16 scheme, round 5):
 *     y = act2( W2 . act1( W1 . x + shift1 ) + shift2 [+ res] )
 * with the cmid-channel intermediate kept in the accumulator registers of the lane that owns the pixel (never written to memory) -
 * it is split into its two fp16 planes under a PER-PIXEL power-of-two scale and fed back to the matrix cores as the second product's
 * operand.  Replaces the pairs nn.Conv2d(1x1)+ReLU -> nn.Conv2d(1x1)[+ReLU] of the path whose intermediate used to round-trip
 * through HBM: EfficientFPN's lateral conv -> fuse conv (models/backbones/resnet.py:96-110, 127-133; the fuse conv's up-sampled half
 * enters as res with EGR_RES_UP2_BEFORE_ACT) and HeatmapMVF.frame_feat_refined_proj_layers
 * (models/estimator/egoposeformer_heatmap_mvf_ex.py:553-563, 715).
 * d: the chain described as ONE 1x1 conv cin -> cout (n, h = ho, w = wo, ldx / ldy / ldr, the three maps, act = act2, res_mode,
 *    groups with gx / gy / gr / gp, gw = 16-bit elements between the groups' W1 images; kh = kw = stride = 1, pad = 0,
 *    w_format = EGR_W_F16X2); cin 64 or 128 with chain->cmid 128 (both weight images stay in LDS), or cin 256 with chain->cmid 256 and
 *    no residual (the heat-map heads' Conv2d(256, 256, 1) + ReLU -> [Upsample ->] Conv2d(256, 128, 1),
 *    egoposeformer_heatmap_mvf_ex.py:101-126, 570-584: both images are STREAMED through a four-slot LDS ring, one 8-KB chunk per k16
 *    step, and the intermediate is produced / consumed in two halves of 128 channels, each under its own per-pixel scale);
 *    cout <= 128 (cout % 4 == 0).
 * w1 / aux->w_descale: egr_pack_wh2_f32 image and descale of the first conv (cmid x cin; descale group stride chain->gp1);
 * aux->amax_in: abs-max record of x; aux->amax_out: record of y (may be NULL); shift2 / chain->w2_descale: (groups,) 128 floats,
 * group stride d->gp.  EGR_EINVAL for any other shape - the caller then runs the two launches of egr_conv2d_nhwc_ex_f32. */
typedef struct {
    const void* w2;            /* egr_pack_wh2_f32 image of the second conv (round_up(cout, 32) padded to 128 rows x cmid) */
    const float* w2_descale;
    const float* shift1;       /* bias of the first conv, (groups,) cmid floats, group stride gp1; NULL = 0 */
    int32_t cmid, act1;        /* 128 (256 with cin 256); EGR_ACT_NONE | EGR_ACT_RELU */
    int64_t gw2, gp1;          /* group strides: 16-bit elements of w2; floats of shift1 and of aux->w_descale */
} egr_chain_aux;
int egr_conv1x1_chain_f32(const egr_conv_desc* d, const float* x, const void* w1, const float* shift2, const float* res, float* y,
                          const egr_conv_aux* aux, const egr_chain_aux* chain, void* stream);

/* Data gradient behind a ReLU (training row): the conv of `d` (normally in transposed mode) with y = (conv [+ res]) * [mask > 0],
 * mask laid out exactly like y (the forward activation the ReLU produced).  Fuses torch's threshold_backward into the
 * data-gradient launch of the layer above.  No scale / shift / activation; res (may be NULL) is added before the mask. */
int egr_conv2d_masked_f32(const egr_conv_desc* d, const float* x, const float* w, const float* res, const float* mask, float* y,
                          float* workspace, size_t workspace_floats, void* stream);
/* ... with the side operands of the fp16 scheme (egr_conv_aux, see egr_conv2d_nhwc_ex_f32): the data-gradient launches of the
 * training step on the fp16 matrix cores, max |dx| recorded for the launch that consumes dx. */
int egr_conv2d_masked_ex_f32(const egr_conv_desc* d, const float* x, const void* w, const float* res, const float* mask, float* y,
                             float* workspace, size_t workspace_floats, const egr_conv_aux* aux, void* stream);

/* Weight (and bias) gradient of the conv / linear layer described by `d` (forward geometry; d->groups same-shape problems
 * in one launch: x + g*gx, dy + g*gy, dw + g*gw, db + g*gp), the weight-side half of the training row: dw[co][(ci/32, kh, kw, ci%32)] (+)= sum over output pixels of dy[m][co] * im2col(x)[m][k],
 * db[co] (+)= sum_m dy[m][co] (db may be NULL).  dw is in the packed weight layout of egr_conv2d_nhwc_f32.  The pixels are
 * split over workgroups; partial tiles go through `workspace` and are summed in fixed order (deterministic).
 * Replaces autograd of nn.Conv2d / nn.Linear for config 5 (pl_wrappers/egoposeformer/pose_3d_mvf_ex.py:117-153). */
/* d->w_format & EGR_W_BF16X3: large problems (>= 1024 pixels and >= 4 GFLOP) run on the bf16 matrix cores with both operands
 * split exactly into three bf16 on the fly; the result class is the fp32 kernel's. */
int egr_conv2d_wgrad_f32(const egr_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                         float* workspace, size_t workspace_floats, int32_t accumulate, void* stream);
/* ... in the fp16 scheme when d->w_format carries EGR_W_F16X2 beside EGR_W_BF16X3 [| EGR_W_FORCE]: the split launches
 * then take both operands as two fp16 planes of the value times a power of two from its abs-max record (amax_x / amax_dy: 64 slots
 * each, as egr_conv_aux.amax_in) - three matrix products per fp32 product instead of six; the accumulators are scaled back when the
 * partial tiles are written.  Launches below the split threshold ignore the records (fp32 matrix cores). */
int egr_conv2d_wgrad_ex_f32(const egr_conv_desc* d, const float* x, const float* dy, float* dw, float* db, float* workspace,
                            size_t workspace_floats, int32_t accumulate, const uint32_t* amax_x, const uint32_t* amax_dy, void* stream);
/* diagnostic (tests): 1 when the last weight-gradient call launched a fp16-scheme kernel */
int egr_wgrad_last_h2(void);
/* diagnostic (tests): the kernel the last egr_conv2d_wgrad_f32 call launched - 0 fp32 MFMA, 1 split-bf16 generic,
 * 2 / 3 split-bf16 3x3 stride-1 tap-sharing (64 channels x 2 input chunks / 128 x 1), 4 the small 1x1 kernel (fp32 path, 1x1 / stride 1,
 * rows x groups <= 8192, at most 256 tiles of 32 x 32: weight and bias gradient in ONE launch, no slabs; env EGR_WGRAD_SMALL=0
 * turns it off). */
int egr_wgrad_last_kernel(void);

/* Tuning knob for measurements: force the tile configuration of egr_conv2d_nhwc_f32
 * (-1 auto, 0 128x128, 1 256x64, 2 64x64, 3 128x32, 4 128x64).  Process-wide; results do not depend on it. */
int egr_conv_force_config(int cfg);
/* diagnostic (tests): the kernel the last egr_conv2d_nhwc_f32 / egr_conv2d_masked_f32 call launched - 0 fp32 MFMA,
 * 1 split-bf16 generic, 2 split-bf16 tap-sharing (3x3 / stride 1 / pad 1, tiles of whole image rows), 3 the stride-2 tap-sharing
 * kernel (3x3 / stride 2 / pad 1 on even images, taps shared by input parity class; env EGR_CONV_TAP2=0 turns it off), 4 the streaming 1x1 kernel (1x1 / stride 1, cin 64 / 128,
 * rows x groups >= 65536: weights stationary in LDS; env EGR_CONV_PW=0 turns it off), 5 the small fp32 1x1 kernel (1x1 / stride 1, fp32
 * weights, NHWC output, automatic split, K % 32 == 0 and K <= 1024, at most 256 tiles of 32 x 32 over all groups - any number when
 * K <= 64: one tile per workgroup, K split over its four waves, operands read straight from global memory; env EGR_CONV_SMALL=0 turns
 * it off; its sums differ from kernel 0's in the last bits (another fixed summation order).  egr_conv_set_tap(0) disables 2-4. */
int egr_conv_last_kernel(void);
/* diagnostic / test knob: 1 = split-K launches run the reduction + epilogue in the last-arriving K slice of each tile (arrival
 * counters, agent-scope slab accesses) instead of a second kernel (splitk_reduce_kernel).  Both sum the slices in slice order.
 * Default 0 (env EGR_SPLITK_FUSED): the fused form saves the launch but measured slower, see DESIGN.md §5d.  The counters are
 * one region per `workspace` pointer (launches that share a workspace cannot overlap anyway; two engine lanes own one each, also
 * when their graphs were captured on the same stream), 64 workspaces per process at most - the 65th takes the second pass. */
int egr_conv_set_splitk_fused(int on);
/* diagnostic / test knob: 0 = 3x3 stride-1 split launches stay on the generic split kernel (default 1, env EGR_CONV_TAP); 3 = the
 * tap-sharing kernels without the 64-row tile that launches below 256 tiles of 128 x 64 take in the fp16 scheme (batch 1: twice the
 * workgroups; env EGR_CONV_TAP64=0).  Same results either way. */
int egr_conv_set_tap(int on);
/* Test / tuning knob of the fp16 scheme's 3x3 / pad 1 forward launches (stride 1 with 64-channel-multiple outputs, stride 2 with
 * 256-channel-multiple outputs; nn.Conv2d 3x3 of models/backbones/resnet.py:43-74 and of the heads / refiners in
 * models/estimator/egoposeformer_heatmap_mvf_ex.py:101-126, 525-532, 570-584): launches with at least `min_tiles` output tiles over
 * all groups run on role-split persistent workgroups - four multiplying waves + four waves that load, split and run the previous
 * tile's epilogue under the current tile's K loop, `blocks` workgroups (one per CU) - egr_conv_last_kernel() = 6.  on = 0 keeps them
 * on the tap-sharing kernels (2 / 3); on = 2 / 3 force the 128 x 32 / 128 x 64 wave tile wherever that variant exists (1: chosen by
 * shape).  Negative values leave a setting unchanged; defaults 1, 256, 256 (env EGR_CONV_TAPX, EGR_CONV_TAPX_MIN_TILES,
 * EGR_CONV_TAPX_BLOCKS, EGR_CONV_TAPX_FN).  Process-wide; results do not depend on it (same products, same summation order).
 * Training launches take the same kernel where it measured faster (env EGR_CONV_TAPX_TRAIN=0: never): a 3x3 conv with the
 * statistics epilogue (egr_conv_aux.bn_partials; slabs = M tiles of the tile height chosen, 256 or 128 rows) and the stride-1 data
 * gradient (transposed, plain or through egr_conv2d_masked_f32) - from four tiles per CU, masked launches for 64-channel outputs only. */
int egr_conv_set_tapx(int32_t on, int32_t min_tiles, int32_t blocks);
/* Tuning / test knob of the persistent split-bf16 launches (short-K layers: a workgroup walks several tiles and requests the next
 * tile's operands under the current tile's stores): `slots` = resident workgroups a launch is sized for (0: never persistent;
 * default 512 = 2 per CU), `max_ktiles` = largest K in 32-deep chunks that takes them (default 4).  Negative: unchanged.
 * Process-wide; results do not depend on it. */
int egr_conv_set_persist(int slots, int max_ktiles);

/* Tail of a heat-map head in one pass: Upsample(x2, bilinear, align_corners=True) + ReLU, then the final 1x1 conv
 * (cin <= 128 -> cout <= 16, with bias) written as channel-major planes: lo (n, h, w, cin) NHWC, all groups' images back to back;
 * wgt (groups, cout, cin) plain row-major; image i of group g goes to planes + g*gy + nmap(i).  The cin-channel tensor at the
 * doubled resolution is never materialised.  Replaces `nn.Upsample -> ... -> ReLU -> Conv2d(128, 15, 1)` of
 * egoposeformer_heatmap_mvf_ex.py:101-126 (indices 6-9) and :570-584 (indices 4-7); 2h % 8 == 0, 2w % 32 == 0, cin % 16 == 0. */
int egr_up2_relu_head_f32(const float* lo, int32_t n, int32_t h, int32_t w, int32_t cin, const float* wgt, const float* bias,
                          int32_t cout, float* planes, int32_t n_inner, int64_t stride_inner, int64_t stride_outer,
                          int32_t groups, int64_t gy, void* stream);
/* Which kernel egr_up2_relu_head_f32 launches for 128 input channels and >= 1024 tiles (round 5): 1 (default; EGR_HEAD_PERSIST) =
 * persistent eight-wave workgroups that request the next tile's source pixels under the current tile's arithmetic, 0 = one
 * workgroup per tile; bit-identical results.  on < 0 only queries.  Returns the previous setting. */
int egr_head_set_persist(int on);

/* Diagnostic only (tools/conv_stamps.py): when `buf` is non-NULL every conv workgroup writes 8 x uint64 — s_memtime at
 * [0] start, [1] row decode done, [2] first chunk landed, [3] K loop done, [4] accumulators staged, [5] stores issued, and
 * [6] its physical placement (XCC id << 16 | HW_ID) — to buf[8 * workgroup].  NULL (the default) disables it. */
int egr_conv_debug_stamps(unsigned long long* buf);

/* ResNet stem: conv 7x7 stride 2 pad 3 (3 -> 64) + BatchNorm(eval) + ReLU, NCHW fp32 input
 * (h, w multiples of 64) -> NHWC output (n, h/2, w/2, 64).  w: [64][148] rows = (ci,kh,kw), last col 0.
 * Replaces layer_s2 of models/backbones/resnet.py:16,49.  scale == shift == NULL: the bare convolution (training mode,
 * BatchNorm on batch statistics follows as its own pass). */
int egr_stem_conv7x7_f32(const float* x, egr_nmap xmap, int32_t n, int32_t h, int32_t w,
                         const float* wpack, const float* scale, const float* shift, float* y,
                         int32_t groups /* group g: x + g*gx, wpack + g*64*148, scale/shift + g*64, y + g*n*(h/2)*(w/2)*64 */,
                         int64_t gx, void* stream);

/* The same stem followed by MaxPool2d(3, 2, 1) (resnet.py:16-17: layer_s2 + layer_s4[0]) in one pass: y is (n, h/4, w/4, 64) NHWC,
 * the stride-2 tensor is never written.  Eval mode only (scale / shift required: the fused max relies on the ReLU's >= 0).
 * Bit-identical to egr_stem_conv7x7_f32 + egr_maxpool_nhwc_f32.  Two launches (a zero-fill of the pooled pixels shared
 * between tiles, then the convolution); groups as above with y + g*n*(h/4)*(w/4)*64. */
int egr_stem_conv7x7_pool_f32(const float* x, egr_nmap xmap, int32_t n, int32_t h, int32_t w,
                              const float* wpack, const float* scale, const float* shift, float* y,
                              int32_t groups, int64_t gx, void* stream);

/* The stem on the bf16 matrix cores (fp32 in / fp32 out; every operand as hi + mid + lo bf16, six products: the arithmetic of
 * EGR_W_BF16X3 launches of egr_conv2d_nhwc_f32).  w6: the filter bank split once by egr_pack_stem_w6_f32 from the [groups][64][148]
 * fp32 layout above (egr_stem_w6_bytes() bytes per group, 16-byte aligned).  pool != 0: MaxPool2d(3, 2, 1) in the same pass,
 * y = (n, h/4, w/4, 64) (eval mode only; bit-identical to pool == 0 followed by egr_maxpool_nhwc_f32); pool == 0: y = (n, h/2, w/2, 64),
 * scale == shift == NULL for the bare convolution.  h, w multiples of 64.  Replaces resnet.py:16-17,49. */
int64_t egr_stem_w6_bytes(void);
int egr_pack_stem_w6_f32(const float* w, int32_t groups, void* w6, void* stream);
int egr_stem_conv7x7_x6_f32(const float* x, egr_nmap xmap, int32_t n, int32_t h, int32_t w,
                            const void* w6, const float* scale, const float* shift, float* y,
                            int32_t pool, int32_t groups, int64_t gx, void* stream);
/* The same with the abs-max record of the pooled output (pool = 1 only; 64 uint32 slots, zero before the launch: see
 * egr_conv2d_nhwc_ex_f32) - the first residual block's convolutions take their fp16 pre-scale from it. */
int egr_stem_conv7x7_x6_ex_f32(const float* x, egr_nmap xmap, int32_t n, int32_t h, int32_t w,
                            const void* w6, const float* scale, const float* shift, float* y,
                            int32_t pool, int32_t groups, int64_t gx, uint32_t* amax_out, void* stream);
/* A Linear layer with few rows (<= 64 per launch) and a large weight matrix, as a weight stream in the fp16 scheme
 * (EgoPoseFormerPose3D.mlp_pred[0]: 2048 x 32768, models/estimator/egoposeformer_mvf_ex.py:241-253, 317-320):
 *   y[b][n] = act(sum_k x[b][k] w[n][k] + bias[n]),  n % 64 == 0, k % 256 == 0.
 * egr_pack_wstream_f32: w (n x k row-major fp32) -> image (egr_wstream_image_bytes = 4 n k bytes: two fp16 planes of w 2^k[n] in MFMA
 * fragment order) + descale[n].  egr_linear_wstream_f32: amax_in = the abs-max record of x (64 slots, as egr_conv_aux.amax_in);
 * amax_out (optional) receives max |y|; workspace of egr_linear_wstream_workspace_bytes(rows, n, k) bytes, 16-byte aligned
 * (EGR_EWORKSPACE when smaller).  Deterministic: the summation order depends on (k, n) only. */
int64_t egr_wstream_image_bytes(int32_t n, int32_t k);
int egr_pack_wstream_f32(const float* w, int32_t n, int32_t k, void* img, float* descale, void* stream);
int64_t egr_linear_wstream_workspace_bytes(int32_t rows, int32_t n, int32_t k);
int egr_linear_wstream_f32(const float* x, int64_t ldx, int32_t rows, int32_t k, const void* wimg, const float* w_descale, const float* bias,
                           int32_t n, int32_t act, const uint32_t* amax_in, float* y, int64_t ldy, uint32_t* amax_out, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* The stem in the fp16 scheme: the filter bank as two fp16 planes of w * 2^k[co] (egr_pack_stem_wh2_f32:
 * groups x egr_stem_wh2_bytes() bytes + the per-channel descale, groups x 64 floats), the input patch of a tile split on the fly after
 * a power-of-two pre-scale taken PER TILE from the patch's own largest magnitude (every k of a tile's outputs lies in that patch, so a
 * per-tile scale is a per-row scale of the GEMM: exact).  Three MFMA products per fp32 product instead of six; same operands,
 * geometry constraints, pooling and abs-max record as egr_stem_conv7x7_x6_ex_f32. */
int64_t egr_stem_wh2_bytes(void);
int egr_pack_stem_wh2_f32(const float* w, int32_t groups, void* img, float* descale, void* stream);
int egr_stem_conv7x7_h2_f32(const float* x, egr_nmap xmap, int32_t n, int32_t h, int32_t w, const void* wh2, const float* w_descale,
                            const float* scale, const float* shift, float* y, int32_t pool, int32_t groups, int64_t gx,
                            uint32_t* amax_out, void* stream);

/* MaxPool2d(k, stride, pad) on NHWC (resnet.py:17 maxpool 3/2/1; egoposeformer_mvf_ex.py:234 MaxPool2d(2)). c % 4 == 0. */
int egr_maxpool_nhwc_f32(const float* x, float* y, int32_t n, int32_t h, int32_t w, int32_t c,
                         int32_t k, int32_t stride, int32_t pad, void* stream);

/* nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) on NHWC; c % 4 == 0.  relu != 0 applies ReLU to
 * the interpolated value: a bias-carrying 1x1 conv commutes with the interpolation (its weights sum to 1), so
 * relu(conv1x1(up(x))) is evaluated as relu(up(conv1x1(x))) at a quarter of the conv work. */
int egr_upsample2x_nhwc_f32(const float* x, int32_t ldx, float* y, int32_t ldy,
                            int32_t n, int32_t h, int32_t w, int32_t c, int32_t relu, void* stream);

/* Fold max |x| of a dense fp32 tensor of n elements into an abs-max record (64 uint32 slots, zero or holding earlier maxima: atomic
 * max on the float bits, see egr_conv2d_nhwc_ex_f32).  For tensors that enter the path from outside - e.g. the feature maps a caller
 * hands to EgoPoseFormerPose3D.forward (models/estimator/egoposeformer_mvf_ex.py:422-452) - so that their consumers can take the
 * fp16 scheme; tensors produced on the path carry the record of the launch that wrote them.  HBM-bound: one read of x. */
int egr_absmax_f32(const float* x, int64_t n, uint32_t* record, void* stream);

/* Global average pool over hw pixels of NHWC (F.adaptive_avg_pool2d(.,(1,1)), heatmap_mvf_ex.py:659). */
int egr_avgpool_nhwc_f32(const float* x, float* y, int32_t n, int32_t hw, int32_t c, void* stream);

/* utils/loss.py:122-142 get_max_preds(normalize=True): per row of `hw` = hgt*wid values: first-index argmax,
 * anchors (x/wid, y/hgt), maxvals, valid = max >= thr, raw flat index. */
int egr_argmax_rows_f32(const float* hm, int32_t rows, int32_t hgt, int32_t wid, float thr,
                        float* anchors /* rows x 2 */, float* maxvals, uint8_t* valid, int32_t* index, void* stream);

/* y = LayerNorm(x (+ res)) * gamma + beta over the last dim c (c <= 1024, c % 64 == 0), eps 1e-5. */
int egr_layernorm_f32(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                      int32_t rows, int32_t c, float eps, int32_t rows_per_group /* gamma/beta + (row/rows_per_group)*c; 0 = one group */,
                      void* stream);

/* Joint-to-joint attention core of SpatialMHA / EgoformerSpatialMHA (heatmap_mvf_ex.py:799-817,
 * egoposeformer_mvf_ex.py:481-498): qkv (b, j, 3*c) -> out (b, j, c); softmax(q k^T * scale) v per head; j <= 16. */
int egr_joint_mha_f32(const float* qkv, float* out, int32_t b, int32_t j, int32_t heads, int32_t d, float scale, void* stream);

/*
 * Deformable-attention sampling in sample-then-project form (replaces the mmcv CUDA op
 * MultiScaleDeformableAttnFunction, models/utils/deform_attn.py:155-162, together with the
 * softmax / location arithmetic of :122-137).  For every (b, joint, view, head):
 *   loc_p   = anchor[b,view,joint] + offs[b,joint,head,p] / (wid, hgt)
 *   a_p     = softmax_p(logits[b,joint,head,:])
 *   S_p(.)  = mmcv bilinear sample at pixel = loc*size - 0.5, zero padding
 *   g       = sum_p a_p S_p(feat[view,b])       (cf channels, all channels: the value projection is applied later)
 *   e       = sum_p a_p S_p(pos[view][:, head*dh:(head+1)*dh])   (optional, pos may be NULL)
 *   sigma   = sum_p a_p * (in-bounds bilinear weight mass)
 * Rows are ordered (b, joint, view) so that the projected result is already the concatenation
 * over views the fuse layer consumes.  rowmask[(b,joint,view)] = valid[b,view,joint].
 * offs_logits: (b*joint, heads*P*2 + heads*P) — offsets then attention logits (P = 16 points).
 */
int egr_msda_gather_f32(const float* feat /* (views, b, hgt*wid, cf) */, int32_t cf,
                        const float* pos /* (views, hgt*wid, heads*dh) or NULL */, int32_t dh,
                        const float* offs_logits, const float* anchors /* (b, views, joints, 2) */,
                        const uint8_t* valid /* (b, views, joints) */,
                        int32_t b, int32_t views, int32_t joints, int32_t heads, int32_t hgt, int32_t wid,
                        float* g /* (rows, heads, cf) */, float* e /* (rows, heads*dh) or NULL */,
                        float* sigma /* (heads, rows) */, uint8_t* rowmask /* (rows) */,
                        int32_t groups /* query sets sharing feat/anchors/valid: pos, offs_logits, g, e, sigma are [groups][...] */,
                        void* stream);

/*
 * One joint-query transformer layer behind its deformable sampling, as ONE launch (egr_layer.hip).  Replaces, per layer of
 * MultiViewTransformerLayer (egoposeformer_heatmap_mvf_ex.py:874-935) / EgoPoseFormerTransformerLayer
 * (egoposeformer_mvf_ex.py:546-588): MSDeformAttn's value / output projections on the sampled rows
 * (models/utils/deform_attn.py:122-168, sample-then-project form) with masked_fill(~valid), the concatenation over views +
 * fuse_mlp + residual + norm_cross, SpatialMHA (q/k/v/out projections and the softmax core, models/utils/transformer.py:36-93)
 * + residual + norm_spatial, FFN (transformer.py:8-33, exact-erf GELU) + residual + norm_ffn.  Optional tails: the NEXT layer's
 * sampling_offsets / attention_weights Linear on the new tokens, post_norm (heatmap_mvf_ex.py:707 / mvf_ex.py:411), and the
 * 3-D regression head reg_mlp + init anchors (mvf_ex.py:255-262, 412-418).
 * Operands: every weight is a plain row-major (groups, out, in) fp32 stack in the reference's Linear layout, biases / affine
 * parameters (groups, out).  g / e / sigma / rowmask are egr_msda_gather_f32's outputs; x the layer input (groups*b*joints, c).
 *   w_fold (groups, c, cf) = value_proj.weight . W_pre   (W_pre: the 1x1 conv in front of the attention), rows in head order
 *   c_fold (groups, c)     = value_proj.weight . b_pre + value_proj.bias       (scaled by sigma per row: zero padding)
 * Limits: views = 4, heads = 4, cf = 128, ffn_dim = 512, c in {128, 256}, joints <= 16; 16-byte aligned weights.
 */
typedef struct egr_layer_desc {
    int32_t B, J, V, C, heads, cf, groups, ffn_dim;
    float eps, mha_scale;                              /* LayerNorm eps (1e-5); softmax scale = (c / heads)^-1/2 */
    const float *x, *g, *e /* or NULL */, *sigma;
    const uint8_t* rowmask;
    const float *w_fold, *c_fold, *w_out, *b_out, *w_fuse, *b_fuse, *ln1_g, *ln1_b;
    const float *w_qkv, *b_qkv, *w_mo, *b_mo, *ln2_g, *ln2_b, *w_f0, *b_f0, *w_f1, *b_f1, *ln3_g, *ln3_b;
    float* x_out;                                      /* (groups*b*joints, c) */
    /* tail: next layer's offsets / logits (NULL = off): ol_out (groups*b*joints, ol_n), ol_n % 16 == 0 */
    const float *w_ol, *b_ol;
    float* ol_out;
    int32_t ol_n;
    int32_t w_packed;   /* != 0: every weight MATRIX (w_fold, w_out, w_fuse, w_qkv, w_mo, w_f0, w_f1, w_ol, w_r0; not w_r2, not the vectors) is in
                         * the fragment order of egr_pack_layer_w_f32 (1: fp32, 1-KiB contiguous wave loads) or of egr_pack_layer_wh2_f32
                         * (2: the fp16 scheme - two fp16 planes per weight, rows * k + rows floats per matrix, the contractions on
                         * v_mfma_f32_16x16x32_f16 with three products; what the shipped path passes under EGR_W_FORMAT=f16x2) */
    /* tail: post_norm (lnp_g NULL = off) -> xn_out (or NULL); with it the regression head: w_r0 (groups, c, c), w_r2 (groups, 3, c),
     * pred_out (groups*b*joints, 3) = reg(xn) + anchors3d */
    const float *lnp_g, *lnp_b;
    float* xn_out;
    const float *w_r0, *b_r0, *w_r2, *b_r2, *anchors3d;
    float* pred_out;
    /* tail (round 5): the refiner's head offset behind post_norm (heatmap_mvf_ex.py:707-711; TransformerHeadLayer.head[0..2]): the
     * (joints, c) tokens read as an s x s image (s * s == c; c = 256) with the joints as channels -> Conv2d(joints, h0_n, 1) + ReLU ->
     * Upsample(x2, bilinear, align_corners=True) -> h0_out (groups*b, 2s, 2s, h0_n) NHWC.  w_h0 (groups, h0_n, joints) plain
     * row-major, b_h0 (groups, h0_n); h0_n == 64; excludes the regression head.  amax_h0: NULL or the 64-slot abs-max record of
     * h0_out (receives an upper bound).  Replaces egr_tokens_to_nhwc_f32 + egr_linear_smallk_f32 + egr_upsample2x_nhwc_f32. */
    const float *w_h0, *b_h0;
    float* h0_out;
    uint32_t* amax_h0;
    int32_t h0_n;
} egr_layer_desc;
int egr_joint_layer_f32(const egr_layer_desc* d, void* stream);
/* Which kernel egr_joint_layer_f32 launches under w_packed = 2 (round 5): 1 (default; EGR_LAYER_PLANES) = every tile that feeds a
 * contraction is split ONCE into fragment-ordered fp16 planes in LDS, 0 = the round-4 kernel that splits inside the product loop;
 * bit-identical results.  on < 0 only queries.  Returns the previous setting. */
int egr_layer_set_planes(int on);
/* The JQA query of a refiner as ONE launch (round 5; HeatmapMVF.forward_feat_only, egoposeformer_heatmap_mvf_ex.py:655-665, behind
 * heatmap_proj[0] + ReLU):  hm_embed = heatmap_proj[2](t);  bfb = fc_bfb(adaptive_avg_pool2d(backbone_feat_bottom, (1, 1)));
 * x = ReLU(fc_query((joint_query_embed + bfb) + hm_embed));  ol = [sampling_offsets | attention_weights](x) of the refiner's
 * transformer layer (models/utils/deform_attn.py:122-135).  One workgroup per (query set, frame).
 * t (groups*b*joints, c); s32 (groups*b, pool_hw, kb) NHWC; embed (groups, joints, c); matrices w_hp2 (c, c), w_bfb (c, kb),
 * w_q (c, c), w_ol (ol_n, c) per group in the order w_packed names (egr_layer_desc.w_packed), vectors plain.
 * Limits: c = 256, kb = 512, joints <= 16, ol_n % 16 == 0.  Replaces egr_avgpool_nhwc_f32, egr_jqa_sum_f32 and four small
 * egr_conv2d_nhwc_f32 launches. */
typedef struct egr_jqa_query_desc {
    int32_t B, J, C, groups, kb, pool_hw, ol_n, w_packed;
    const float *t, *s32;
    const float *w_hp2, *b_hp2, *w_bfb, *b_bfb, *embed, *w_q, *b_q, *w_ol, *b_ol;
    float *x_out /* (groups*b*joints, c) */, *ol_out /* (groups*b*joints, ol_n) */;
} egr_jqa_query_desc;
int egr_jqa_query_f32(const egr_jqa_query_desc* d, void* stream);

/* The lifting head between mlp_pred[1] and its first decoder layer as ONE launch (round 5; egoposeformer_mvf_ex.py:255-262,
 * 317-322, 340-348, 400-410): pred = mlp_pred[2](h1) (b, joints, 3); the fisheye reprojection of egr_fisheye_project2_f32
 * (anchors3d_out = the points behind it: mutated in syn mode (ctm NULL), a copy in rw mode); x = query_gen_mlp([(j+1)/joints,
 * point]) (Linear(4, c) + ReLU, Linear + ReLU, Linear); ol = the first layer's [sampling_offsets | attention_weights](x).
 * One workgroup per frame.  h1 (b, c) = GELU(mlp_pred[1](..)); w_m2 (3*joints, c), w_qg2 / w_qg4 (c, c), w_ol (ol_n, c) in the
 * order w_packed names; w_qg0 (c, 4) and all vectors plain; cams / ctm as in egr_fisheye_project_f32.
 * Limits: c = 128, joints = 16, ol_n % 16 == 0.  Replaces egr_fisheye_project_f32, egr_linear_smallk_f32 and four small
 * egr_conv2d_nhwc_f32 launches. */
typedef struct egr_pose_query_desc {
    int32_t B, J, C, ol_n, w_packed;
    const float *h1, *w_m2, *b_m2, *ctm /* or NULL */, *cams;
    const float *w_qg0, *b_qg0, *w_qg2, *b_qg2, *w_qg4, *b_qg4, *w_ol, *b_ol;
    float *pred_out /* (b, joints, 3) */, *anchors3d_out /* (b, joints, 3) */, *anchors2d_out /* (b, 4, joints, 2) */;
    uint8_t* valid_out /* (b, 4, joints) */;
    float *x_out /* (b*joints, c) */, *ol_out /* (b*joints, ol_n) */;
} egr_pose_query_desc;
int egr_pose_query_f32(const egr_pose_query_desc* d, void* stream);

/* `matrices` row-major (rows, k) fp32 matrices (rows % 16 == 0, k % 128 == 0) -> the fragment order egr_joint_layer_f32 reads with
 * w_packed: [matrix][16-row block][128-deep chunk][16-deep k block u][lane = 16 q + i][4 floats] = w[16 block + i][128 chunk + 16 u + 4 q ..+3].
 * Same number of elements; out must not alias w. */
int egr_pack_layer_w_f32(const float* w, int32_t matrices, int32_t rows, int32_t k, float* out, void* stream);
/* The fp16-scheme image of the same matrices (w_packed = 2): per matrix rows * k + rows floats - [16-row block][128-deep chunk]
 * [32-deep k block][plane h | l][lane = 16 q + i][8 fp16] = plane of w[16 block + i][128 chunk + 32 kb + 8 q ..+7] * 2^e(row), then the
 * rows' descales 2^-e (fp32; e puts the row's largest magnitude into [2^14, 2^15), clamped to +-60).  Same limits as above. */
int egr_pack_layer_wh2_f32(const float* w, int32_t matrices, int32_t rows, int32_t k, float* out, void* stream);

/* utils/camera_models.py:53-104 + egoposeformer_mvf_ex.py:340-348,400-406: project the (b, joints, 3) proposals
 * into the four fisheye cameras.  cams: 4 records [npoly, cx, cy, W, H, poly[12]] (fp32).  syn mode
 * (ctm == NULL) reproduces the reference's in-place offset chain: `pts` is updated in place to the
 * mutated anchors.  rw mode applies ctm (b, 4, 4, 4) to pts*0.01 and scales by 100, pts untouched.
 * Also emits the decoder query input q4 = [ (joint+1)/joints, mutated pts ] (b, joints, 4). */
int egr_fisheye_project_f32(float* pts, const float* ctm, const float* cams, int32_t b, int32_t joints,
                            float* anchors /* (b, 4, joints, 2) */, uint8_t* valid /* (b, 4, joints) */,
                            float* q4, void* stream);
/* The same with the updated points written to pts_out (b, joints, 3) instead of in place: the mutated anchors in syn mode, a
 * copy of pts in rw mode - `init_anchors_3d = mlp_pred.clone()` (egoposeformer_mvf_ex.py:400) without the copy. */
int egr_fisheye_project2_f32(const float* pts, float* pts_out, const float* ctm, const float* cams, int32_t b, int32_t joints,
                             float* anchors, uint8_t* valid, float* q4, void* stream);

/* Small dense layer for K not a multiple of 32 (query_gen_mlp.0: K=4; head 1x1 conv: K=15):
 * y[m, n] = act(sum_k x[m*sxm + k*sxk] * w[n*K + k] + bias[n]). */
int egr_linear_smallk_f32(const float* x, int64_t sxm, int64_t sxk, const float* w, const float* bias, float* y,
                          int32_t m, int32_t n, int32_t k, int32_t act,
                          int32_t rows_per_group /* w + (row/rows_per_group)*n*k, bias + ..*n; 0 = one group */, void* stream);

/* JQA query pre-activation (heatmap_mvf_ex.py:664-665): y[b,j,:] = hm_embed[b,j,:] + embed[j,:] + bfb[b,:]. */
int egr_jqa_sum_f32(const float* hm_embed, const float* embed, const float* bfb, float* y,
                    int32_t b, int32_t j, int32_t c, int32_t b_per_group /* embed + (b/b_per_group)*j*c; 0 = one group */,
                    void* stream);

/* (b, j, s*s) token matrix -> NHWC (b, s, s, j) image with joints as channels (heatmap_mvf_ex.py:707-711). */
int egr_tokens_to_nhwc_f32(const float* x, float* y, int32_t b, int32_t j, int32_t hw, void* stream);

/* Frame pre-processing, replaces PIL.Image.resize([ow,oh], BICUBIC) + ToTensor + Normalize of
 * datasets/ego4view_syn/ego4view_syn_pose3d.py:41-44,159-162: src (n, h, w, 3) uint8 -> dst (n, 3, oh, ow) fp32.
 * bounds_* (out, 2) [first tap, tap count] and coef_* (out, ksize) 22-bit fixed-point weights are Pillow's
 * precompute_coeffs / normalize_coeffs_8bpc tables (host-computed); tmp (n, h, ow, 3) uint8 scratch;
 * mean / stdv: 3 HOST floats; u8out (n, oh, ow, 3) optional uint8 copy of the resized image (NULL to skip). */
int egr_preprocess_u8_f32(const uint8_t* src, int32_t n, int32_t h, int32_t w, int32_t oh, int32_t ow,
                          const int32_t* bounds_h, const int32_t* coef_h, int32_t ksize_h,
                          const int32_t* bounds_v, const int32_t* coef_v, int32_t ksize_v,
                          const float* mean, const float* stdv, uint8_t* tmp, float* dst, uint8_t* u8out, void* stream);

/* The same conversion as ONE launch: the horizontally resized uint8 rows of a 32-output-row band stay in LDS (no intermediate in
 * HBM, no `tmp`).  band_rows = egr_preprocess_band_rows(host copy of bounds_v, oh).  Returns EGR_EINVAL without launching when the
 * shape does not meet the kernel's limits (ow == 256, ksize_h <= 16, h % 4 == 0, 16-byte aligned rows-of-four, band in LDS): the
 * caller then falls back to egr_preprocess_u8_f32.  Bit-identical results. */
int egr_preprocess_band_rows(const int32_t* bounds_v_host, int32_t oh);
int egr_preprocess_fused_u8_f32(const uint8_t* src, int32_t n, int32_t h, int32_t w, int32_t oh, int32_t ow,
                                const int32_t* bounds_h, const int32_t* coef_h, int32_t ksize_h,
                                const int32_t* bounds_v, const int32_t* coef_v, int32_t ksize_v, int32_t band_rows,
                                const float* mean, const float* stdv, float* dst, uint8_t* u8out, void* stream);

/* Pose evaluation metrics per sample, replaces evaluate_pose of
 * pl_wrappers/egoposeformer/pose_3d_mvf_ex.py:317-333 (utils/loss.py:9-48, models/utils/pose_metric.py:104-167):
 * pred, gt (b, joints, 3) fp32 in cm -> out (b, 4) = [MPJPE mm, PA-MPJPE mm (similarity-aligned), PCK@pck_thr_mm %,
 * AUC % over n_auc thresholds linspace(0, pck_thr_mm, n_auc)]; aligned (b, joints, 3) optional (NULL to skip). */
int egr_pose_metrics_f32(const float* pred, const float* gt, int32_t b, int32_t joints, float pck_thr_mm, int32_t n_auc,
                         float* out, float* aligned, void* stream);

/* Ground-truth heat maps, replaces generate_target of generate_heatmap.py:10-48:
 * joints (maps, 2) float64 pixel coordinates in the image_size frame -> out (maps, heatmap_size, heatmap_size) fp32 with a
 * (2*tmp_size+1)^2 Gaussian window `gauss` (host table) centred on int(x * heatmap_size / image_size + 0.5). */
int egr_gt_heatmap_f32(const double* joints, int32_t maps, double image_size, int32_t heatmap_size, int32_t tmp_size,
                       const float* gauss, float* out, void* stream);

/* The reference's only native call, in its own operand layout: mmcv==2.2.0 MultiScaleDeformableAttnFunction.apply
 * (models/utils/deform_attn.py:155-162; third-party, un-vendored, pin README.md:134).
 *   value (n, lin, heads, d) fp32, spatial_shapes i64 (levels, 2) = (H_l, W_l), level_start_index i64 (levels) — both
 *   device arrays —, sampling_loc (n, lq, heads, levels, points, 2) normalised (x, y), attn_weight (n, lq, heads, levels,
 *   points)  ->  out (n, lq, heads*d).
 * pixel = loc*size - 0.5; a point contributes iff -1 < pixel < size on both axes; corners outside the map read as 0.
 * im2col_step of the reference only chunks the batch and has no counterpart here.  Corner tokens >= lin are skipped, so
 * inconsistent shapes cannot fault.  The model path itself uses egr_msda_gather_f32 (sample-then-project); these two
 * entries serve a maintainer who replaces only the mmcv extension (egorear_amd/msda.py). */
int egr_msda_fwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                     const float* sampling_loc, const float* attn_weight, int32_t n, int64_t lin, int32_t heads, int32_t d,
                     int32_t lq, int32_t levels, int32_t points, float* out, void* stream);
/* Backward of the above: grad_value (n, lin, heads, d) is zeroed on the stream and then accumulated with atomics (the
 * summation order, and so the last bits, vary run to run — as in mmcv); grad_sampling_loc and grad_attn_weight are
 * written for every point (zeros for points outside the map). */
int egr_msda_bwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                     const float* sampling_loc, const float* attn_weight, const float* grad_out, int32_t n, int64_t lin,
                     int32_t heads, int32_t d, int32_t lq, int32_t levels, int32_t points, float* grad_value,
                     float* grad_sampling_loc, float* grad_attn_weight, void* stream);

/* Library / device identification. */
const char* egr_version(void);
int egr_device_arch(char* buf, int32_t buflen); /* gcnArchName of the current device */

#ifdef __cplusplus
}
#endif
#endif /* EGOREAR_HIP_H */

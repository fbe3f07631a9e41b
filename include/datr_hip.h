/*
 * datr_hip.h -- C ABI of libdatr_hip.so, the MI355X (gfx950) native library behind
 * datr_amd.  Plain pointers and sizes only; no torch / ATen types.
 *
 * Every entry point
 *   - takes DEVICE pointers unless the parameter name ends in `_host`,
 *   - enqueues its work on the caller's `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream) and never synchronises the host,
 *   - keeps no global mutable state and is re-entrant (forward runs on the caller's
 *     thread, backward on an autograd worker thread in the reference's call pattern:
 *     /root/reference/models/dino/ops/functions/ms_deform_attn_func.py:21-38),
 *   - returns DATR_OK or a negative DATR_E* code; `datr_strerror` renders it.
 *
 * The reference-side binding a maintainer would write is shown in INTEGRATION.md.
 */
#ifndef DATR_HIP_H_
#define DATR_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DATR_OK            0
#define DATR_EINVAL       -1   /* null pointer / non-positive or inconsistent dimension      */
#define DATR_EUNSUPPORTED -2   /* shape outside what the kernels index with 32 bits          */
#define DATR_ELAUNCH      -3   /* hipGetLastError() != hipSuccess after the launch           */

const char *datr_strerror(int code);
/* ABI version: bumped whenever an exported signature, a struct or a memory layout of this header changes.
 * 2 (round 5): round 4 had removed datr_conv3x3_forward_f32 / datr_conv3x3_nhwc_forward_f32 /
 *   datr_gemm_k256_f32 / datr_wgrad_k256_f32, changed the Winograd filter layout to [16][Cin/8][Cout][8] and
 *   added datr_gemm_epilogue / datr_adamw_tensor without a bump; datr_adamw_tensor.used_index < 0 now means
 *   "no used flag: always update". */
int datr_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Multi-scale deformable attention: sampling + aggregation.
 *
 * Replaces the two functions the reference's pybind11 module exports
 *   ms_deform_attn_forward / ms_deform_attn_backward
 *   (/root/reference/models/dino/ops/src/vision.cpp:13-16,
 *    /root/reference/models/dino/ops/src/ms_deform_attn.h:21-60,
 *    host code /root/reference/models/dino/ops/src/cuda/ms_deform_attn_cuda.cu:20-153).
 *
 * Layouts, all contiguous:
 *   value   [N, S, M, D]          S = sum_l H_l*W_l
 *   shapes  [L, 2] int64 (H_l, W_l)   level_start [L] int64   -- DEVICE memory, as in the
 *           reference, whose kernels read them in-kernel (ms_deform_im2col_cuda.cuh:274-277)
 *   loc     [N, Lq, M, L, P, 2]   (x, y) in normalised [0,1] image coordinates
 *   attn    [N, Lq, M, L, P]
 *   out / grad_out   [N, Lq, M*D]
 *
 * forward : `out` is fully overwritten (no need to zero it).
 * backward: `grad_value` is zero-filled BY THE LIBRARY on `stream` and then accumulated with
 *           float atomics (the reference zero-fills with at::zeros_like,
 *           ms_deform_attn_cuda.cu:121); `grad_loc` and `grad_attn` are fully overwritten.
 *           Atomic accumulation order is not deterministic, so grad_value is reproducible
 *           to rounding only -- exactly as in the reference.
 * The reference's `im2col_step` batching (ms_deform_attn_cuda.cu:50-75) has no effect on
 * results; the whole batch is one launch here and the divisibility check lives in the
 * Python mirror (datr_amd/msda.py).
 * ------------------------------------------------------------------------------------------ */
int datr_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                          const float *loc, const float *attn,
                          int64_t N, int64_t S, int64_t M, int64_t D,
                          int64_t L, int64_t Lq, int64_t P,
                          float *out, void *stream);

int datr_msda_backward_f32(const float *grad_out, const float *value, const int64_t *shapes,
                           const int64_t *level_start, const float *loc, const float *attn,
                           int64_t N, int64_t S, int64_t M, int64_t D,
                           int64_t L, int64_t Lq, int64_t P,
                           float *grad_value, float *grad_loc, float *grad_attn, void *stream);

/* Same contracts as datr_msda_forward_f32 / datr_msda_backward_f32, for callers that also hold HOST
 * copies of `shapes` and `level_start` (the reference builds both from python ints,
 * /root/reference/models/dino/deformable_transformer.py:267-290, so a binding has them for free):
 * the library then picks the kernel by geometry -- this is the product path.  D == 32, L == P == 4,
 * Lq == S (the encoder's self-attention: the queries are the pyramid's own pixels): the
 * pyramid-region kernels (csrc/msda_fwd_pyr.hip: coarse-level windows gathered out of LDS;
 * csrc/msda_bwd_pyr.hip: grad_value by sorted scatter).  Lq != S, Lq <= 4096 (the decoder's
 * cross-attention): the owner-computes backward (csrc/msda_bwd_owner.hip, no atomics, no zero-fill).
 * Other D == 32 shapes: the query-tiled backward (fixed-point LDS accumulation, each touched row
 * flushed once).  Everything else falls through to the plain entry points.  Results never depend
 * on the window heuristics; the host arrays are only read during the call.
 * datr_msda_backward_query_tiled_f32 is the same dispatch WITHOUT the pyramid-region kernel: the
 * pyramid kernels are tuned for offsets of a few pixels and degrade when many samples leave their
 * windows, so a caller that watches the offset distribution (datr_amd/msda.py) routes wide
 * distributions here (medium) or to datr_msda_backward_f32 (very wide). */
int datr_msda_forward_tiled_f32(const float *value, const int64_t *shapes,
                                const int64_t *level_start, const int64_t *shapes_host,
                                const int64_t *level_start_host, const float *loc,
                                const float *attn, int64_t N, int64_t S, int64_t M, int64_t D,
                                int64_t L, int64_t Lq, int64_t P, float *out, void *stream);
int datr_msda_backward_tiled_f32(const float *grad_out, const float *value, const int64_t *shapes,
                                 const int64_t *level_start, const int64_t *shapes_host,
                                 const int64_t *level_start_host, const float *loc,
                                 const float *attn, int64_t N, int64_t S, int64_t M, int64_t D,
                                 int64_t L, int64_t Lq, int64_t P, float *grad_value,
                                 float *grad_loc, float *grad_attn, void *stream);
/* datr_msda_backward_tiled_f32 with the measured offset envelope of the samples (as for
 * datr_msda_forward_pyramid_f32: [8][4]{oy_lo, oy_hi, ox_lo, ox_hi} pixels of the sampled level, host
 * memory, or NULL): the encoder calls' pyramid-region kernels size their windows by it.  Results do not
 * depend on it.  Two launches on `stream` for those calls: grad_loc / grad_attn out of the forward's LDS
 * windows (msda_fwd_pyr2.hip, kDots), grad_value by the value-free sorted scatter (msda_bwd_pyr.hip);
 * all three outputs are complete when the stream reaches the end of the second. */
int datr_msda_backward_pyramid_f32(const float *grad_out, const float *value, const int64_t *shapes,
                                   const int64_t *level_start, const int64_t *shapes_host,
                                   const int64_t *level_start_host, const float *envelope_host,
                                   const float *loc, const float *attn, int64_t N, int64_t S, int64_t M,
                                   int64_t D, int64_t L, int64_t Lq, int64_t P, float *grad_value,
                                   float *grad_loc, float *grad_attn, void *stream);
/* The encoder calls' backward for the MSDeformAttn MODULE (ms_deform_attn.py:94-113): sampling locations and
 * attention weights come from ONE projection of the query (offsets | logits, M * 48 = 384 columns for M = 8 heads,
 * the division by (W_l, H_l) folded into its weights), so what the module's backward needs is the gradient of that
 * projection's output, grad_query [N * Lq, M * 48]: columns [m][l][p][2] = grad_sampling_loc, then [m][l * 4 + p] =
 * the softmax backward a (g - sum a g) of grad_attn_weight.  The LDS-window kernel writes those rows directly
 * instead of grad_loc / grad_attn followed by a pass that re-arranges them (msda_prologue.hip); grad_value as in
 * datr_msda_backward_pyramid_f32.  `attn` = the softmax OUTPUT, as everywhere.  DATR_EUNSUPPORTED (grad_query
 * untouched; the caller takes datr_msda_backward_pyramid_f32 + datr_msda_prologue_backward_f32) unless Lq == S,
 * M == 8, D == 32, L == P == 4 and the LDS-window plan covers the shape. */
int datr_msda_backward_pyramid_query_f32(const float *grad_out, const float *value, const int64_t *shapes_host,
                                         const int64_t *level_start_host, const float *envelope_host,
                                         const float *loc, const float *attn, int64_t N, int64_t S, int64_t M,
                                         int64_t D, int64_t L, int64_t Lq, int64_t P, float *grad_value,
                                         float *grad_query, void *stream);
/* The decoder calls' backward (few, spatially unordered queries: Lq != S, Lq <= 4096, D == 32) with
 * grad_value rows `grad_value_row_stride` floats apart (>= M * D): the six decoder layers of
 * /root/reference/models/dino/deformable_transformer.py:880-900 write their value gradients as column
 * slices of ONE [N, S, 6 * 256] buffer, so that the six value projections' data and weight gradients are
 * one GEMM each.  Every row of the slice is written exactly once.  DATR_EUNSUPPORTED for other shapes. */
int datr_msda_backward_strided_f32(const float *grad_out, const float *value, const int64_t *shapes_host,
                                   const int64_t *level_start_host, const float *loc, const float *attn,
                                   int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                                   float *grad_value, int64_t grad_value_row_stride, float *grad_loc,
                                   float *grad_attn, void *stream);
int datr_msda_backward_query_tiled_f32(const float *grad_out, const float *value, const int64_t *shapes,
                                       const int64_t *level_start, const int64_t *shapes_host,
                                       const int64_t *level_start_host, const float *loc,
                                       const float *attn, int64_t N, int64_t S, int64_t M, int64_t D,
                                       int64_t L, int64_t Lq, int64_t P, float *grad_value,
                                       float *grad_loc, float *grad_attn, void *stream);

/* Encoder self-attention forward, round 3 (csrc/msda_fwd_pyr2.hip): the same contract as
 * datr_msda_forward_tiled_f32 plus an optional per-(head, level) OFFSET ENVELOPE on the host,
 *   envelope_host[8][4][4] = {oy_lo, oy_hi, ox_lo, ox_hi} in pixels of the sampled level:
 * how far this call's sampling locations lie from their query's reference point (the sums
 * `reference_points + sampling_offsets / (W_l, H_l)` of
 * /root/reference/models/dino/ops/modules/ms_deform_attn.py:102-108).  The kernel stages value
 * windows of footprint + envelope per level in LDS and gathers ALL four levels from there; a
 * sample outside the envelope is fetched from global memory instead, so the envelope is a
 * performance hint only -- results never depend on it.  NULL = symmetric +-4.5 px (the reach of the
 * module's initial offset ring).  Envelopes too wide for any window plan, and shapes the phased kernel does not cover (D != 32, L != 4, P != 4,
 * Lq != S, M > 8) fall through to datr_msda_forward_tiled_f32. */
int datr_msda_forward_pyramid_f32(const float *value, const int64_t *shapes,
                                  const int64_t *level_start, const int64_t *shapes_host,
                                  const int64_t *level_start_host, const float *envelope_host,
                                  const float *loc, const float *attn, int64_t N, int64_t S, int64_t M,
                                  int64_t D, int64_t L, int64_t Lq, int64_t P, float *out, void *stream);
/* What the pyramid-region kernels would do for this call, without launching anything.
 * info[16] (int32): [0] forward covered by the phased kernel, [1] nRy, [2] nRx region grid,
 * [3] phases, [4] 16-query tasks per wave, [5] workgroups per image, [6] LDS fill KiB per workgroup
 * (head 0), [7] largest phase in 128-B rows (head 0); [8] backward covered by the pyramid-region
 * kernel, [9] nRy, [10] nRx; [11] datr_msda_forward_pyramid_f32 would run the phased kernel (it does
 * whenever [0] is set); [12] launch configuration (0: 512 threads x 2 workgroups per CU, 1: 256 x 3); rest 0. */
int datr_msda_pyramid_plan(const int64_t *shapes_host, const int64_t *level_start_host, int64_t N,
                           int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                           const float *envelope_host, int32_t *info);

/* Double-precision twins (generic kernels; they exist so the reference's gradcheck-in-double
 * op test, /root/reference/models/dino/ops/test.py:63-86, can be restated). */
int datr_msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *level_start,
                          const double *loc, const double *attn,
                          int64_t N, int64_t S, int64_t M, int64_t D,
                          int64_t L, int64_t Lq, int64_t P,
                          double *out, void *stream);

int datr_msda_backward_f64(const double *grad_out, const double *value, const int64_t *shapes,
                           const int64_t *level_start, const double *loc, const double *attn,
                           int64_t N, int64_t S, int64_t M, int64_t D,
                           int64_t L, int64_t Lq, int64_t P,
                           double *grad_value, double *grad_loc, double *grad_attn, void *stream);

/* Which kernel family a float32 call with these dimensions dispatches to:
 * 1 = row-vectorised gfx950 fast path, 0 = generic one-thread-per-scalar path. */
int datr_msda_uses_fast_path(int64_t S, int64_t M, int64_t D, int64_t L, int64_t P);

/* ------------------------------------------------------------------------------------------
 * Fused sigmoid focal loss (classification loss of SetCriterion).
 *
 * Replaces `sigmoid_focal_loss` (/root/reference/models/dino/utils.py:79-104) together with the
 * one-hot construction in `SetCriterion.loss_labels` (/root/reference/models/dino/dino.py:517-526).
 *   logits  [G, R, C] fp32      G independent groups (e.g. decoder layers), R rows (= B*Q) each
 *   target  [G, R]    int64     matched class index per row; any value outside [0, C) (the
 *                               reference uses num_classes) means "no positive in this row"
 *   out_sums[G]                 sum over the group's R*C elements of
 *                               alpha_t * BCEWithLogits(x, t) * (1 - p_t)^gamma
 *                               (alpha < 0 disables the alpha_t factor, as in the reference)
 * The reference's normalisation `.mean(1).sum() / num_boxes * num_queries` equals
 * out_sums / num_boxes and is left to the caller.
 * `scratch` must hold datr_focal_scratch_floats(G, R) floats; sums are folded in a fixed order
 * (bitwise reproducible).  backward: grad_logits[g,r,c] = grad_sums[g] * d loss/d x.
 * ------------------------------------------------------------------------------------------ */
int64_t datr_focal_scratch_floats(int64_t G, int64_t rows_per_group);
int datr_focal_loss_forward_f32(const float *logits, const int64_t *target, int64_t G,
                                int64_t rows_per_group, int64_t C, float alpha, float gamma,
                                float *scratch, float *out_sums, void *stream);
int datr_focal_loss_backward_f32(const float *logits, const int64_t *target,
                                 const float *grad_sums, int64_t G, int64_t rows_per_group,
                                 int64_t C, float alpha, float gamma, float *grad_logits,
                                 void *stream);

/* ------------------------------------------------------------------------------------------
 * Fused frozen batch-norm (+ residual) (+ ReLU) of the ResNet-50 trunk.
 *
 * Replaces `FrozenBatchNorm2d.forward` (/root/reference/models/dino/backbone.py:62-72: two
 * element-wise kernels) and the ReLU / residual-add that follow it in torchvision's Bottleneck.
 *   y[i] = act( x[i] * scale[c] + shift[c] (+ res[i]) ),   c = (i / inner) % C
 * `inner` = H*W for NCHW tensors, 1 for NHWC; `res` may be NULL; `relu` != 0 applies max(.,0).
 * scale = weight * rsqrt(running_var + 1e-5), shift = bias - running_mean * scale are computed
 * by the caller (they are buffers, not parameters).
 * backward: g = dy * (y > 0 if relu); dx = g * scale[c]; dres = g (if dres != NULL).
 * ------------------------------------------------------------------------------------------ */
int datr_affine_act_forward_f32(const float *x, const float *res, const float *scale,
                                const float *shift, int64_t n, int64_t C, int64_t inner, int relu,
                                float *y, void *stream);
int datr_affine_act_backward_f32(const float *dy, const float *y, const float *scale, int64_t n,
                                 int64_t C, int64_t inner, int relu, float *dx, float *dres,
                                 void *stream);
/* The same with TWO upstream gradients (g = (dy + dy2) * ...): a bottleneck's output feeds the next
 * bottleneck's conv1 AND its identity branch (torchvision Bottleneck.forward); summing the two gradients
 * here costs one extra read instead of autograd's separate three-pass add. */
int datr_affine_act_backward2_f32(const float *dy, const float *dy2, const float *y, const float *scale,
                                  int64_t n, int64_t C, int64_t inner, int relu, float *dx, float *dres,
                                  void *stream);

/* ------------------------------------------------------------------------------------------
 * Winograd F(2x2, 3x3) convolution on the MFMA units (csrc/wino.hip), exact fp32, NHWC, 3x3 /
 * stride 1 / pad 1 -- the image-level domain discriminator's layers
 * (/root/reference/models/dino/DA_utils.py:61-79, call site dino.py:351-359), all pyramid levels in
 * ONE launch (they share the filter):
 *   y = out_scale * gate( lrelu_slope( scale[co] * conv3x3(x, W)[.., co] + shift[co] ) )
 * gate(v) = v where gate_tensor > 0, gate_slope * v elsewhere (gate_tensor may be NULL): the
 * LeakyReLU backward of the PREVIOUS layer folded into this layer's data gradient.  scale / shift
 * may be NULL (1 / 0); shift = the bias for the discriminator; slope 1 = no activation;
 * out_scale = -1 folds the gradient-reversal layer into the first layer's data gradient.
 * `u` = the transformed filter from datr_wino_weights_f32: [16][Cin/8][Cout][8] floats (an opaque layout: the 8 channels of a block in the order the kernel's lanes read them).
 * Cin % 8 == 0, Cout % 64 == 0.
 * datr_wino_weights_f32 reads W[co][ci][r][s] at w[co*s_co + ci*s_ci + r*s_r + s*s_s]; for the data
 * gradient pass the strides of co and ci SWAPPED (Cin, Cout = the gradient's channel counts) and
 * flip = 1 (taps mirrored). */
#define DATR_WINO_MAX_LEVELS 4
typedef struct { const float *x; float *y; const float *gate; int64_t H, W; } datr_wino_level;
int datr_wino_weights_f32(const float *w, int64_t Cout, int64_t Cin, int64_t s_co, int64_t s_ci,
                          int64_t s_r, int64_t s_s, int flip, float *u, void *stream);
/* Both filters of a layer from one launch: u for the forward, u_flip for the data gradient (= what
 * datr_wino_weights_f32 gives with the channel strides swapped and flip = 1); Cin % 64 == 0 and Cout % 64 == 0. */
int datr_wino_weights_pair_f32(const float *w, int64_t Cout, int64_t Cin, int64_t s_co, int64_t s_ci, int64_t s_r,
                               int64_t s_s, float *u, float *u_flip, void *stream);
int datr_conv3x3_wino_nhwc_f32(const datr_wino_level *levels, int64_t nlevels, int64_t N, int64_t Cin,
                               int64_t Cout, const float *u, const float *scale, const float *shift,
                               float slope, float gate_slope, float out_scale, void *stream);

/* ------------------------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 convolution with ONE output channel, NHWC fp32 (csrc/conv_cout1.hip): the
 * classifier of the image-level discriminator (/root/reference/models/dino/DA_utils.py:67,78), all
 * pyramid levels of a call (they share the filter).  w: the torch weight [1, C, 3, 3] contiguous;
 * C == 128.  forward: y[N, H, W] = bias + conv(x[N, H, W, C], w).
 * backward, per level: x = the classifier's INPUT a = lrelu(z) (DA_utils.py:76), y = dY[N, H, W],
 *   dx[N, H, W, C] = gate(a) * conv_transpose(dY, w), gate = 1 where a > 0 else `slope` -- i.e. the
 *   gradient w.r.t. the pre-activation z, what the next data-gradient launch consumes;
 *   dw[1, C, 3, 3] and db[1] (may be NULL) are OVERWRITTEN with the sums over all levels, added in a
 *   fixed order (bitwise reproducible).  `partial`: datr_conv3x3_cout1_partial_floats(...) floats.
 * ------------------------------------------------------------------------------------------ */
typedef struct { const float *x; float *y; float *dx; int64_t H, W; } datr_c1_level;
int64_t datr_conv3x3_cout1_partial_floats(const datr_c1_level *levels, int64_t nlevels, int64_t N);
int datr_conv3x3_cout1_forward_f32(const datr_c1_level *levels, int64_t nlevels, int64_t N, int64_t C,
                                   const float *w, const float *bias, void *stream);
int datr_conv3x3_cout1_backward_f32(const datr_c1_level *levels, int64_t nlevels, int64_t N, int64_t C,
                                    const float *w, float slope, float *dw, float *db, float *partial,
                                    void *stream);

/* ------------------------------------------------------------------------------------------
 * out = xs[0] + ... + xs[n - 1] (2 <= n <= 8 contiguous fp32 tensors of `numel` elements, summed in
 * argument order) in one pass (csrc/addn.hip): the gradient of a tensor with several consumers --
 * what autograd's pairwise accumulation does in n - 1 launches for the sources of an encoder layer,
 * the position table and the decoder's memory
 * (/root/reference/models/dino/deformable_transformer.py:796-806, 880-900).  `xs`: HOST array of
 * device pointers. */
int datr_add_n_f32(const float *const *xs, int64_t n, int64_t numel, float *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Even pixels of an NHWC tensor and the adjoint (csrc/subsample.hip): what the 1x1 / stride-2
 * downsample convolutions of layer2-4.0 (/root/reference/models/dino/backbone.py:109-128) read in
 * front of their GEMM, and the gradient's way back.  x / dx: [N, H, W, C]; y / dy: [N, (H + 1) / 2,
 * (W + 1) / 2, C]; C % 4 == 0.  The scatter writes ALL of dx (zeros on the odd pixels). */
int datr_even_pixels_nhwc_f32(const float *x, int64_t N, int64_t H, int64_t W, int64_t C, float *y, void *stream);
int datr_even_pixels_scatter_nhwc_f32(const float *dy, int64_t N, int64_t H, int64_t W, int64_t C, float *dx,
                                      void *stream);

/* ------------------------------------------------------------------------------------------
 * ResNet-50 stem (csrc/stem.hip): y = relu(conv7x7(x, stride 2, pad 3) * scale + shift), 3 -> 64
 * channels, NHWC, forward only -- `conv1` + `bn1` + `relu` of the frozen trunk head
 * (/root/reference/models/dino/backbone.py:62-72,79-81).  x: [N, H, W, 3]; y: [N, (H + 1) / 2,
 * (W + 1) / 2, 64]; wk: [154][64] = for every filter row r the 21 (tap s, channel c) weights
 * W[co][c][r][s] in (s, c) order followed by one zero row; scale / shift: [64]. */
int datr_stem_conv7x7_bn_relu_nhwc_f32(const float *x, const float *wk, const float *scale, const float *shift,
                                       int64_t N, int64_t H, int64_t W, float *y, void *stream);

/* ------------------------------------------------------------------------------------------
 * 3x3 / stride 2 / pad 1 convolutions on NHWC tensors (csrc/conv_tap.hip): `conv2` of the first
 * bottleneck of layer2-4 (/root/reference/models/dino/backbone.py:109-128: torchvision resnet50) and
 * `input_proj[3]` (/root/reference/models/dino/dino.py:120-124: Conv2d(2048, 256, 3, stride 2,
 * padding 1)); replaces `nn.Conv2d.forward` and the two halves of autograd's convolution backward for
 * those layers.  x: [N, H, W, Cin]; y / dy: [N, (H + 1) / 2, (W + 1) / 2, Cout].
 *   forward  y = act(conv(x) * scale + shift); scale / shift [Cout] or NULL; act = LeakyReLU(slope)
 *            (slope 0: ReLU, 1: none).  wt: [9][Cin][Cout] = W.permute(2, 3, 1, 0).
 *            Cin % 16 == 0, Cout % 128 == 0.
 *   dgrad    dx = conv_transpose(dy); wt_t: [9][Cout][Cin] = W.permute(2, 3, 0, 1).
 *            Cout % 16 == 0, Cin % 128 == 0.  dx is overwritten.
 *   wgrad    dw[co * s_co + ci * s_ci + r * s_r + s * s_s] = sum x dy (overwritten).
 *            Cin % 32 == 0, Cout % 128 == 0.
 * `workspace`: datr_conv3x3s2_workspace_floats(...) floats of scratch (split-K partial sums, added in
 * a fixed order: results are bitwise reproducible). */
int64_t datr_conv3x3s2_workspace_floats(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout);
/* wt and / or wt_t (either may be NULL) from a filter addressed w[co * s_co + ci * s_ci + r * s_r + s * s_s];
 * Cin % 32 == 0, Cout % 32 == 0. */
int datr_conv3x3s2_weights_f32(const float *w, int64_t Cout, int64_t Cin, int64_t s_co, int64_t s_ci, int64_t s_r,
                               int64_t s_s, float *wt, float *wt_t, void *stream);
int datr_conv3x3s2_forward_nhwc_f32(const float *x, const float *wt, const float *scale, const float *shift,
                                    float slope, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                    float *y, float *workspace, int64_t workspace_floats, void *stream);
int datr_conv3x3s2_dgrad_nhwc_f32(const float *dy, const float *wt_t, int64_t N, int64_t H, int64_t W,
                                  int64_t Cin, int64_t Cout, float *dx, float *workspace,
                                  int64_t workspace_floats, void *stream);
int datr_conv3x3s2_wgrad_nhwc_f32(const float *x, const float *dy, int64_t N, int64_t H, int64_t W, int64_t Cin,
                                  int64_t Cout, float *dw, int64_t s_co, int64_t s_ci, int64_t s_r, int64_t s_s,
                                  float *workspace, int64_t workspace_floats, void *stream);

/* Weight gradient of the same convolutions in the Winograd domain (csrc/wino_wgrad.hip):
 * dW = G^T [ sum over 2x2 tiles (A dY A^T) o (B^T x B) ] G, summed over all levels (they share the
 * filter) -- replaces `torch.ops.aten.convolution_backward(dz, x, w, ..., [False, True, False])`, i.e.
 * the weight-gradient half of autograd's backward of the reference's nn.Conv2d layers
 * (/root/reference/models/dino/DA_utils.py:61-79, backbone.py:109-128).  x: [N, H, W, Cin], dy:
 * [N, H, W, Cout] NHWC fp32 per level; Cin, Cout multiples of 64.  dw is OVERWRITTEN, addressed
 * dw[co * s_co + ci * s_ci + r * s_r + s * s_s] (elements).  `partial`: scratch of
 * datr_wino_wgrad_partial_floats(...) floats; the split-K segments are summed in a fixed order
 * (deterministic, no atomics). */
#define DATR_WINO_WGRAD_MAX_SEGMENTS 160
typedef struct { const float *x; const float *dy; int64_t H, W; } datr_wino_wgrad_level;
int64_t datr_wino_wgrad_partial_floats(const datr_wino_wgrad_level *levels, int64_t nlevels, int64_t N,
                                       int64_t Cin, int64_t Cout);
int datr_conv3x3_wino_wgrad_nhwc_f32(const datr_wino_wgrad_level *levels, int64_t nlevels, int64_t N,
                                     int64_t Cin, int64_t Cout, float *partial, float *dw, int64_t s_co,
                                     int64_t s_ci, int64_t s_r, int64_t s_s, void *stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm on NHWC tensors (csrc/groupnorm.hip): `nn.GroupNorm(32, 256)` behind every input_proj
 * convolution (/root/reference/models/dino/dino.py:111-126), kept in the backbone's NHWC layout
 * (ATen transposes to NCHW and back).  x, y, dy, dx: [N, HW, C]; gamma, beta, dgamma, dbeta: [C];
 * mean, rstd: [N, G] (written by the forward, read by the backward).  C % 4 == 0, (C / G) % 4 == 0,
 * 256 % (C / 4) == 0, C <= 1024.  `partial`: datr_groupnorm_partial_floats(N, HW, C, G) floats of
 * scratch; all sums are re-reduced in a fixed order (deterministic, no atomics). */
int64_t datr_groupnorm_partial_floats(int64_t N, int64_t HW, int64_t C, int64_t G);
int datr_groupnorm_nhwc_forward_f32(const float *x, const float *gamma, const float *beta, int64_t N,
                                    int64_t HW, int64_t C, int64_t G, float eps, float *y, float *mean,
                                    float *rstd, float *partial, void *stream);
int datr_groupnorm_nhwc_backward_f32(const float *dy, const float *x, const float *mean, const float *rstd,
                                     const float *gamma, int64_t N, int64_t HW, int64_t C, int64_t G,
                                     float *dx, float *dgamma, float *dbeta, float *partial, void *stream);

/* ------------------------------------------------------------------------------------------
 * Bilinear resize + optional horizontal flip of a uint8 [H, W, 3] image, bit-exact with Pillow's
 * `Image.resize(size, BILINEAR)` -- what `F.resize` / `F.hflip` do in the reference's
 * RandomResize / RandomHorizontalFlip (/root/reference/datasets/da_transforms.py:62-140; Pillow is
 * un-vendored, its 8-bit separable resampler is restated).  xbounds / ybounds: {first source index,
 * count} per output column / row; xk / yk: ksx / ksy 22-bit fixed-point weights per output column /
 * row (host-computed as Pillow computes them, datr_amd/input_pipeline.py::pillow_coeffs).  tmp:
 * H * ow * 3 bytes of scratch.  flip mirrors the SOURCE reads (flip first, then resize). */
int datr_resize_bilinear_u8(const uint8_t *src, int64_t H, int64_t W, int flip, const int32_t *xbounds,
                            const int32_t *xk, int64_t ksx, const int32_t *ybounds, const int32_t *yk,
                            int64_t ksy, int64_t oh, int64_t ow, uint8_t *tmp, uint8_t *dst, void *stream);

/* `value.masked_fill(input_padding_mask[..., None], 0)` of MSDeformAttn
 * (/root/reference/models/dino/ops/modules/ms_deform_attn.py:101-102) IN PLACE on x [rows, cols]
 * (cols % 4 == 0, x 16-byte aligned): rows whose mask byte is non-zero are zeroed, the others are
 * not touched.  The backward of that op is the same call on the gradient. */
int datr_zero_rows_f32(float *x, const uint8_t *mask, int64_t rows, int64_t cols, void *stream);

/* ------------------------------------------------------------------------------------------
 * EMA teacher update, all tensors in one launch (csrc/ema.hip): the reference's key walk
 * `v *= d; v += (1. - d) * msd[k]` (/root/reference/models/dino/EMA.py:46-50, :123-128).
 * `tensors` (device): one entry per DISTINCT float32 tensor of the EMA model's state_dict -- dst the
 * EMA tensor, src the model's, `repeats` = the number of state_dict keys that alias it (the update
 * is replayed that many times, as the key walk does).  `pieces` (device): the work list, one entry
 * per datr_ema_piece_elements() elements of a tensor.  Both tables are built once by the caller.
 * Arithmetic: fl(fl(v * d) + fl((1 - d) * m)) per replay, d and 1 - d rounded to float32. */
typedef struct { float *dst; const float *src; int64_t numel; int64_t repeats; } datr_ema_tensor;
typedef struct { int64_t tensor; int64_t offset; } datr_ema_piece;
int64_t datr_ema_piece_elements(void);
int datr_ema_update_f32(const datr_ema_tensor *tensors, const datr_ema_piece *pieces, int64_t npieces,
                        double decay, void *stream);

/* ------------------------------------------------------------------------------------------
 * Class-wise query prototypes of the prototype alignment (/root/reference/models/dino/DA_utils.py:82-120
 * `get_prototype_class_wise`) given the class label of every query (the caller's argmax of the sigmoid scores):
 * feats [R, C], labels [R] int64 in [0, K), global_proto [K, C], amount [K] -> proto [K, C] (class means, 0 for
 * absent classes), present [K] in {0, 1}, new_global [K, C] = global (1 - w) + proto w with w = count / (count +
 * amount) (0 where count = 0), new_amount [K] = amount + count, onehot [R, K].  One launch, rows summed in a fixed
 * order.  backward: d_feats [R, C] = d_proto[label] / max(count[label], 1) (count [K] = new_amount - amount).
 * C % 64 == 0 (csrc/prototypes.hip).
 * ------------------------------------------------------------------------------------------ */
int datr_class_prototypes_forward_f32(const float *feats, const int64_t *labels, const float *global_proto,
                                      const float *amount, int64_t R, int64_t C, int64_t K, float *proto,
                                      float *present, float *new_global, float *new_amount, float *onehot,
                                      void *stream);
int datr_class_prototypes_backward_f32(const float *d_proto, const int64_t *labels, const float *count, int64_t R,
                                       int64_t C, int64_t K, float *d_feats, void *stream);

/* Contrastive prototype loss (/root/reference/models/dino/dino.py `loss_contrast_da`): cosine logits of the source /
 * target class prototypes q_* [K, 256] against the global prototypes [K, 256] (F.normalize, eps), cross entropy
 * against eye * class_map (mask_* [K] in {0, 1}), summed over the two domains -> loss [1]; and, the loss being a
 * scalar, d loss / d q_source, d loss / d q_target [K, 256] from the same launch.  K <= 16, C == 256. */
int datr_contrast_loss_f32(const float *q_source, const float *q_target, const float *global_proto,
                           const float *mask_source, const float *mask_target, int64_t K, int64_t C, float eps,
                           float *loss, float *d_q_source, float *d_q_target, void *stream);

/* ------------------------------------------------------------------------------------------
 * Stacked operands of ONE GEMM for two linear layers that read the same input -- MSDeformAttn's
 * `sampling_offsets` and `attention_weights` (/root/reference/models/dino/ops/modules/ms_deform_attn.py:96-97):
 * w [Ra + Rb, C] = [diag(scale) wa ; wb], b [Ra + Rb] = [scale * ba ; bb]; scale [Ra] may be NULL (with 2-d
 * reference points it is 1 / W_l, 1 / H_l per offset feature: the division of :101-104 folded into the
 * weights).  backward (scale != NULL only): d_wa = diag(scale) d_w[:Ra], d_ba = scale * d_b[:Ra]; the other
 * gradients are row slices of d_w / d_b.  C % 4 == 0 (csrc/stack_linear.hip).
 * ------------------------------------------------------------------------------------------ */
int datr_stack_linear_forward_f32(const float *wa, const float *ba, const float *wb, const float *bb,
                                  const float *scale, int64_t Ra, int64_t Rb, int64_t C, float *w, float *b,
                                  void *stream);
int datr_stack_linear_backward_f32(const float *d_w, const float *d_b, const float *scale, int64_t Ra, int64_t C,
                                   float *d_wa, float *d_ba, void *stream);

/* ------------------------------------------------------------------------------------------
 * Iterative box refinement of the decoder, new_ref = sigmoid(delta + inverse_sigmoid(ref)) with
 * inverse_sigmoid(x) = log(clamp(x, 0, 1).clamp(min = eps) / (1 - clamp(x, 0, 1)).clamp(min = eps))
 * (/root/reference/models/dino/deformable_transformer.py:738-744, /root/reference/models/dino/dino.py:316-322,
 * /root/reference/util/misc.py:587-591), as one element-wise launch each way (csrc/refine.hip).
 * n contiguous floats.  backward: grad_delta / grad_ref may be NULL (not both); autograd's conventions
 * (sigmoid' = y (1 - y); clamps pass the gradient inside the closed range).
 * ------------------------------------------------------------------------------------------ */
int datr_refine_boxes_forward_f32(const float *delta, const float *ref, int64_t n, float eps, float *out,
                                  void *stream);
int datr_refine_boxes_backward_f32(const float *grad_out, const float *out, const float *ref, int64_t n, float eps,
                                   float *grad_delta, float *grad_ref, void *stream);

/* ------------------------------------------------------------------------------------------
 * The optimizer step of the training loop as multi-tensor launches (csrc/adamw.hip):
 *     torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm);  optimizer.step()   (AdamW)
 * (/root/reference/engine.py:99-104; /root/reference/main.py:165).
 * `tensors` (device): one entry per parameter WITH a gradient this step -- param / grad / exp_avg /
 * exp_avg_sq of `numel` contiguous floats (any dense layout, the same for all four), `step` its own
 * step count (a device float, as torch's capturable / fused AdamW keeps it), its group's lr and
 * weight_decay, and `used_index`, its position in the optional `used` flag array.  `pieces` (device):
 * the work list, one entry per datr_adamw_piece_elements() elements of a tensor.
 * datr_grad_norm_clip_coef_f32: norm_coef[0] = the 2-norm over all gradients (fixed summation order:
 *   deterministic), norm_coef[1] = min(1, max_norm / (norm + 1e-6)) -- clip_grad_norm_'s coefficient;
 *   `partial`: npieces floats of scratch.  The gradients themselves are NOT rescaled.
 * datr_adamw_step_f32: torch's single-tensor AdamW arithmetic in float32 with every gradient multiplied
 *   by *clip_coef (NULL = 1) as it is read; tensors whose used[used_index] == 0 (used != NULL) are
 *   skipped entirely -- no decay, no moments, no step count -- as AdamW skips a parameter whose .grad
 *   is None (the reference's DistributedDataParallel(find_unused_parameters=True), main.py:156, leaves
 *   a globally unused parameter's .grad None).  Step counts are incremented after the update.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float *param; const float *grad; float *exp_avg; float *exp_avg_sq; float *step;
    int64_t numel; float lr; float weight_decay; int32_t used_index; int32_t pad_;
} datr_adamw_tensor;
typedef struct { int64_t tensor; int64_t offset; } datr_adamw_piece;
int64_t datr_adamw_piece_elements(void);
int datr_grad_norm_clip_coef_f32(const datr_adamw_tensor *tensors, const datr_adamw_piece *pieces, int64_t npieces,
                                 float max_norm, float *partial, float *norm_coef, void *stream);
int datr_adamw_step_f32(const datr_adamw_tensor *tensors, int64_t ntensors, const datr_adamw_piece *pieces,
                        int64_t npieces, const float *clip_coef, const int32_t *used, double beta1, double beta2,
                        double eps, void *stream);

/* ------------------------------------------------------------------------------------------
 * Photometric ("strong") augmentation of uint8 [H, W, 3] images, bit-exact with Pillow -- the
 * pixel work of the reference's make_coco_strong_transforms (/root/reference/datasets/DAcoco.py:
 * 330-360: torchvision ColorJitter / RandomGrayscale on PIL images = ImageEnhance.Brightness /
 * Contrast / Color, an HSV round trip, Image.convert("L"); GaussianBlur = ImageFilter.GaussianBlur).
 *
 * datr_pixel_ops_u8 applies a chain of up to DATR_PIXEL_OPS_MAX per-pixel operations in one pass
 * (src may equal dst): BRIGHTNESS / CONTRAST / SATURATION blend towards black / the mean grey of
 * the image AT THAT POINT OF THE CHAIN / the pixel's luma with factor `alpha` (Image.blend's
 * float32 arithmetic); HUE adds the byte `shift` to the H of Pillow's 8-bit HSV; GRAYSCALE
 * replaces the pixel by its luma.  `sums`: DATR_PIXEL_OPS_MAX uint64 of device scratch, needed
 * (non-NULL) when the chain holds a CONTRAST step.  src / dst 4-byte aligned. */
#define DATR_PIXEL_OPS_MAX    8
#define DATR_PIXEL_BRIGHTNESS 0
#define DATR_PIXEL_CONTRAST   1
#define DATR_PIXEL_SATURATION 2
#define DATR_PIXEL_HUE        3
#define DATR_PIXEL_GRAYSCALE  4
typedef struct {
    int32_t code;  /* DATR_PIXEL_* */
    float alpha;   /* blend factor (BRIGHTNESS / CONTRAST / SATURATION) */
    int32_t shift; /* HUE: (int)(hue_factor * 255) & 0xFF */
} datr_pixel_op;
int datr_pixel_ops_u8(const uint8_t *src, uint8_t *dst, int64_t npix, const datr_pixel_op *ops, int64_t nops,
                      uint64_t *sums, void *stream);

/* Pillow's ImagingBoxBlur on a uint8 [H, W, 3] image: `passes` horizontal passes of the extended
 * box blur, then `passes` vertical ones, edges clamped; `radius` = integer part of the box radius,
 * ww / fw = its 8.24 fixed-point inner / far weights (host-computed as libImaging BoxBlur.c does,
 * datr_amd/strong_aug.py::box_weights; ImageFilter.GaussianBlur(sigma) = 3 passes with the radius
 * of gaussian_box_radius(sigma)).  src != dst.  passes * (radius + 1) > ~60: DATR_EUNSUPPORTED. */
int datr_box_blur_u8(const uint8_t *src, uint8_t *dst, int64_t H, int64_t W, int64_t radius, uint32_t ww,
                     uint32_t fw, int64_t passes, void *stream);

/* ------------------------------------------------------------------------------------------
 * FFN backward, the non-GEMM pass: given h = relu(linear1(x)) saved by the forward and
 * dh = d loss / d h, computes IN PLACE dh <- dh * (h > 0) and db[c] = sum_r dh[r, c]
 * (the bias gradient of linear1) in one pass over HBM
 * (/root/reference/models/dino/deformable_transformer.py:803-806, :879-883 run
 * threshold_backward and a separate column reduction).  rows x cols fp32 row-major,
 * cols % 4 == 0.  `partial` is caller-provided scratch of
 * datr_relu_bwd_bias_partial_rows(rows) * cols floats; the column sum is two-stage and
 * deterministic (no atomics).
 * ------------------------------------------------------------------------------------------ */
int64_t datr_relu_bwd_bias_partial_rows(int64_t rows);
int datr_relu_bwd_bias_f32(float *dh, const float *h, int64_t rows, int64_t cols, float *partial,
                           float *db, void *stream);
/* Column sums out[c] = sum_r x[r, c] -- the bias gradient of any linear layer (autograd's
 * `grad_output.sum(0)`), same deterministic two-stage scheme and scratch size. */
int datr_colsum_f32(const float *x, int64_t rows, int64_t cols, float *partial, float *out,
                    void *stream);

/* ------------------------------------------------------------------------------------------
 * Sampling locations and attention weights of MSDeformAttn (8 heads, 4 levels, 4 points) from the
 * merged query projection (/root/reference/models/dino/ops/modules/ms_deform_attn.py:96-117):
 * both [rows, 384] = (256 sampling offsets in (head, level, point, xy) order | 128 logits),
 * ref [rows, 4, ref_dim] reference points (ref_dim 2: loc = ref + offset, offsets already divided
 * by (W_l, H_l); ref_dim 4: loc = ref_xy + offset / 4 * ref_wh * 0.5), loc [rows, 8, 4, 4, 2],
 * attn [rows, 8, 16] = softmax over (level, point).  backward: d_both [rows, 384] from d_loc,
 * d_attn and the saved attn; the reference points get no gradient. */
int datr_msda_prologue_forward_f32(const float *both, const float *ref, int64_t rows, int64_t ref_dim,
                                   float *loc, float *attn, void *stream);
int datr_msda_prologue_backward_f32(const float *d_loc, const float *d_attn, const float *attn,
                                    const float *ref, int64_t rows, int64_t ref_dim, float *d_both,
                                    void *stream);

/* ------------------------------------------------------------------------------------------
 * Scaled-dot-product attention forward, head_dim 32, additive [L, L] mask (may be
 * NULL), exact fp32 on the MFMA units: out = softmax(scale * Q K^T + mask) V per (batch n, head h) --
 * the decoder's self-attention (nn.MultiheadAttention in
 * /root/reference/models/dino/deformable_transformer.py:880-884).  Element (l, n, h, d) of a
 * tensor x is x[l * ld_l + n * ld_n + h * 32 + d]; strides = {q_l, q_n, k_l, k_n, v_l, v_n, o_l, o_n}
 * in floats, all multiples of 4.  lse [N, H, L] (natural log of the softmax denominator plus the
 * row maximum) may be NULL. */
int datr_mha_forward_d32_f32(const float *q, const float *k, const float *v, const float *mask,
                             int64_t L, int64_t N, int64_t H, const int64_t *strides, float scale,
                             float *out, float *lse, void *stream);

/* Backward of the same attention, fed with the forward's `out` and `lse`: grad_q = scale dS K,
 * grad_k = scale dS^T Q, grad_v = P^T dO with P = exp(scale Q K^T + mask - lse),
 * dS = P o (dO V^T - D), D = rowsum(dO o out)  (autograd of nn.MultiheadAttention's softmax
 * attention, deformable_transformer.py:880-884).  Two launches (query-stationary dQ, key-stationary
 * dK/dV), no atomics: bitwise reproducible.  strides = {q_l, q_n, k_l, k_n, v_l, v_n, o_l, o_n,
 * go_l, go_n, dq_l, dq_n, dk_l, dk_n, dv_l, dv_n} in floats, multiples of 4; delta = [N, H, L]
 * scratch (receives D); the three gradients are fully overwritten. */
int datr_mha_backward_d32_f32(const float *grad_out, const float *q, const float *k, const float *v,
                              const float *out, const float *lse, const float *mask, int64_t L,
                              int64_t N, int64_t H, const int64_t *strides, float scale, float *delta,
                              float *grad_q, float *grad_k, float *grad_v, void *stream);

/* ------------------------------------------------------------------------------------------
 * Cost matrix of the Hungarian matcher (/root/reference/models/dino/matcher.py:48-88) for `sets`
 * = (prediction sets x images) blocks of nq queries against all T ground-truth boxes of the
 * batch: logits [sets * nq, C], boxes [sets * nq, 4] cxcywh, tgt_ids [T] int64, tgt_boxes [T, 4].
 * cost_t [sets, T, nq] = TRANSPOSED cost (what datr_lsap_f32 reads).  *boxes_ok (caller sets it
 * to 1) is cleared if any box is degenerate (the reference asserts, box_ops.py:52-53). */
int datr_match_cost_f32(const float *logits, const float *boxes, const int64_t *tgt_ids,
                        const float *tgt_boxes, int64_t sets, int64_t nq, int64_t T, int64_t C,
                        float w_class, float w_bbox, float w_giou, float alpha, float *cost_t,
                        int32_t *boxes_ok, void *stream);

/* ------------------------------------------------------------------------------------------
 * Box regression losses of SetCriterion (`loss_boxes`, /root/reference/models/dino/dino.py:553-577,
 * with box_ops.py:9-63) for P matched pairs belonging to G prediction sets: src / tgt [P, 4]
 * cxcywh, group [P] int64 in [0, G).  forward: sums [4, G] = per-set sums of the L1 distance,
 * of (1 - GIoU), and of the xy / wh halves of the L1 (callers divide by num_boxes).
 * backward: d_src [P, 4] given the gradients d_l1 [G], d_giou [G] of the first two rows.
 * P <= DATR_BOX_LOSS_MAX_PAIRS, G <= 64.  Deterministic. */
#define DATR_BOX_LOSS_MAX_PAIRS 3072
int datr_box_loss_forward_f32(const float *src, const float *tgt, const int64_t *group, int64_t P,
                              int64_t G, float *sums, void *stream);
int datr_box_loss_backward_f32(const float *src, const float *tgt, const int64_t *group,
                               const float *d_l1, const float *d_giou, int64_t P, float *d_src,
                               void *stream);

/* ------------------------------------------------------------------------------------------
 * Deterministic row-wise top-k, k <= 1024 (csrc/topk.hip): the two-stage query selection
 * `torch.topk(enc_outputs_class_unselected.max(-1)[0], 900, dim=1)[1]`
 * (/root/reference/models/dino/deformable_transformer.py:342) and PostProcess's
 * `torch.topk(prob.view(B, -1), num_select, dim=1)` (/root/reference/models/dino/dino.py:960).
 * scores [rows, n] fp32 -> out_idx [rows, k] int64, out_val [rows, k] fp32 (may be NULL), sorted by
 * descending score, equal scores by ascending index, NaN first: one total order, so the result
 * does not depend on the device or the launch (torch.topk leaves the order of ties open).
 * k > 1024: DATR_EUNSUPPORTED. */
int datr_topk_rows_f32(const float *scores, int64_t rows, int64_t n, int64_t k, int64_t *out_idx,
                       float *out_val, void *stream);

/* ------------------------------------------------------------------------------------------
 * Class-aware greedy NMS of the teacher's pseudo labels (csrc/nms.hip):
 * `torchvision.ops.batched_nms(boxes, scores, labels, thr)` as called by rescale_pseudo_targets
 * (/root/reference/models/dino/self_training_utils.py:80-83; torchvision is un-vendored: classes
 * are separated by adding label * (max coordinate + 1) to the xyxy boxes, IoU > thr suppresses).
 * boxes [n, 4] xyxy, scores [n], labels [n] int64, n <= 4096.  keep [n] receives the surviving
 * ORIGINAL indices in decreasing score order (ties: lower index first), *count their number.
 * Bit-identical to the float32 host formulation (no FMA contraction). */
int datr_nms_f32(const float *boxes, const float *scores, const int64_t *labels, int64_t n,
                 float iou_threshold, int64_t *keep, int32_t *count, void *stream);

/* ------------------------------------------------------------------------------------------
 * Sine embedding of the decoder's reference boxes: `gen_sineembed_for_position`
 * (/root/reference/models/dino/utils.py:138-163).  pos [rows, ncoord] (ncoord = 2: (x, y) or
 * 4: (x, y, w, h)), dim_t [128] = 10000^(2 floor(k / 2) / 128), out [rows, 128 * ncoord] in the
 * reference's order (y, x[, w, h]); out_k = sin / cos of (coord * 2 pi) / dim_t[k] for even / odd k.
 * Forward only (the boxes are detached between decoder layers). */
int datr_sine_embed_f32(const float *pos, const float *dim_t, int64_t rows, int64_t ncoord,
                        float *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Exact-fp32 MFMA GEMM family with programmable epilogues (csrc/gemm_f32.hip): the 1x1
 * convolutions of the NHWC ResNet-50 bottlenecks and of input_proj
 * (/root/reference/models/dino/backbone.py:62-72,109-128 around torchvision's Bottleneck;
 * /root/reference/models/dino/dino.py:111-119) -- forward with frozen BN (+ residual) (+ ReLU) in the
 * epilogue, data gradient with the ReLU gate / identity-branch gradient in the epilogue, weight
 * gradient as a deterministic split-K product -- and the FFN backward's dz = (dy W2) * [h > 0]
 * with linear1's bias gradient (/root/reference/models/dino/deformable_transformer.py:783-787,803-806).
 *
 *   C[M, N] = epi( op(A) op(B) )        all matrices row-major fp32, leading dimensions in floats
 *   form DATR_GEMM_NT:  A [M, K] (lda),  B [N, K] (ldb)     y  = x W^T
 *   form DATR_GEMM_NN:  A [M, K] (lda),  B [K, N] (ldb)     dx = dy W
 *   form DATR_GEMM_TN:  A [K, M] (lda),  B [K, N] (ldb)     dW = dy^T x   (reduction split over
 *                       workgroups, partial products added in a fixed order: deterministic)
 *   NT / NN:  epi(v)[m, n] = gate( relu( v * scale[n] + shift[n] + residual[m, n] ) ),
 *             gate(v) = gate[m, n] > 0 ? v : 0;  colsum[n] = sum_m epi(v)[m, n]  (deterministic)
 *   TN:       epi(v)[m, n] = v * scale[m]   (per-ROW scale: the frozen-BN fold of a weight gradient);
 *             rowsum_a[m] = sum_k A[k, m] (deterministic); the other fields must be NULL / 0.
 * Every epilogue field may be NULL / 0; `epi` itself may be NULL.
 * Constraints: pointers 16-byte aligned, leading dimensions and N multiples of 4 (TN: M too);
 * NT / NN: K % 32 == 0.  Every matrix < 2^31 bytes.  Otherwise DATR_EUNSUPPORTED.
 * `workspace`: datr_gemm_workspace_floats(form, M, N, K, colsum != NULL) floats (0 = none needed),
 * caller-owned, reusable by later calls on the same stream.
 * ------------------------------------------------------------------------------------------ */
#define DATR_GEMM_NT 0
#define DATR_GEMM_NN 1
#define DATR_GEMM_TN 2
typedef struct {
    const float *scale;      /* [N] (NT / NN) or [M] (TN), or NULL                                */
    const float *shift;      /* [N] or NULL                                                       */
    const float *residual;   /* [M, N] with leading dimension ldr, or NULL                        */
    int64_t ldr;
    const float *gate;       /* [M, N] with leading dimension ldg, or NULL                        */
    int64_t ldg;
    int relu;
    float *colsum;           /* [N] column sums of the result, or NULL                            */
    float *rowsum_a;         /* TN only: [M] sums of A over the reduction axis (dy^T 1: the bias   */
                             /* gradient beside the weight gradient dy^T x), or NULL               */
} datr_gemm_epilogue;
int64_t datr_gemm_workspace_floats(int form, int64_t M, int64_t N, int64_t K, int want_colsum);
int datr_gemm_f32(int form, const float *A, int64_t lda, const float *B, int64_t ldb,
                  int64_t M, int64_t N, int64_t K, const datr_gemm_epilogue *epi,
                  float *C, int64_t ldc, float *workspace, int64_t workspace_floats, void *stream);

/* ------------------------------------------------------------------------------------------
 * Hungarian matching on the device: all of a step's rectangular assignment problems in one
 * launch, indices left on the device (no host synchronisation).  Replaces `C.cpu()` +
 * scipy.optimize.linear_sum_assignment per image (/root/reference/models/dino/matcher.py:91-95);
 * same algorithm as SciPy's solver (shortest augmenting paths, its scan order and tie rule), in
 * double on the fp32 costs, so the assignment is identical.
 *   cost_t   [G, B, Tsum, nc] fp32: the cost matrices TRANSPOSED (row = ground-truth box,
 *            column = query); problem (g, b) uses rows offsets[b] .. offsets[b+1]-1
 *   offsets  [B + 1] int32 on the device (prefix sums of the per-image box counts T_b)
 *   max_rows max_b T_b (host value; sizes the LDS);  requires nc <= 1024 and T_b < nc
 *            (SciPy transposes exactly the tall problems, T < nc; others: DATR_EUNSUPPORTED)
 *   q_idx, t_idx [G, Tsum] int64: for problem (g, b), entries offsets[b] + r, r < T_b, hold the
 *            matched query indices in ascending order and the box (0 .. T_b-1) matched to each --
 *            SciPy's (row_ind, col_ind) for the [nc x T_b] problem
 *   status   [G * B] int32: 0 ok, 1 = the matrix held NaN / -inf or was infeasible (outputs
 *            for that problem are then undefined)
 * ------------------------------------------------------------------------------------------ */
int datr_lsap_f32(const float *cost_t, const int32_t *offsets, int64_t G, int64_t B, int64_t Tsum,
                  int64_t nc, int64_t max_rows, int64_t *q_idx, int64_t *t_idx, int32_t *status,
                  void *stream);

/* ------------------------------------------------------------------------------------------
 * Residual add + LayerNorm over C = 256 channels (the post-norm tail of every transformer
 * sub-block, /root/reference/models/dino/deformable_transformer.py:796-806, :856-893):
 *   forward   y = LayerNorm(x + res) * gamma + beta; also writes mean / rstd [rows]
 *   backward  dx = d loss / d x = d loss / d res (x + res is recomputed, not stored), and
 *             dgamma / dbeta [C] through caller-provided scratch `partial` of
 *             datr_add_layernorm_partial_floats(rows) floats (deterministic, no atomics)
 * `res` may be NULL (plain LayerNorm).  rows x C fp32 row-major; C != 256: DATR_EUNSUPPORTED.
 * ------------------------------------------------------------------------------------------ */
int64_t datr_add_layernorm_partial_floats(int64_t rows);
int datr_add_layernorm_forward_f32(const float *x, const float *res, const float *gamma,
                                   const float *beta, int64_t rows, int64_t C, float eps, float *y,
                                   float *mean, float *rstd, void *stream);
int datr_add_layernorm_backward_f32(const float *dy, const float *x, const float *res,
                                    const float *mean, const float *rstd, const float *gamma,
                                    int64_t rows, int64_t C, float *dx, float *partial, float *dgamma,
                                    float *dbeta, void *stream);
/* The same, plus dxsum[C] = column sums of dx (fixed-order partial sums): when the normalised tensor is
 * x + linear(h) -- the FFN sub-block of deformable_transformer.py:803-806 -- this is linear's bias gradient. */
int datr_add_layernorm_backward_colsum_f32(const float *dy, const float *x, const float *res,
                                    const float *mean, const float *rstd, const float *gamma,
                                    int64_t rows, int64_t C, float *dx, float *partial, float *dgamma,
                                    float *dbeta, float *dxsum, void *stream);
/* Forward that also writes y2 = y + add in the same pass: an encoder layer's output together with the next layer's
 * query, tokens + position table (`with_pos_embed`, deformable_transformer.py:789-798).  add, y2 [rows, C]. */
int datr_add_layernorm_forward_query_f32(const float *x, const float *res, const float *gamma,
                                         const float *beta, const float *add, int64_t rows, int64_t C,
                                         float eps, float *y, float *y2, float *mean, float *rstd,
                                         void *stream);
/* Backward whose incoming gradient is (dy + dy1) + dy2, summed on load (dy1, dy2 may be NULL; dy2 only with dy1):
 * the consumers of an encoder layer's output (residual, value projection, next query) hand their gradients over
 * without a separate sum over the token tensor.  dxsum may be NULL (no column sums). */
int datr_add_layernorm_backward_fanin_f32(const float *dy, const float *dy1, const float *dy2,
                                          const float *x, const float *res, const float *mean,
                                          const float *rstd, const float *gamma, int64_t rows, int64_t C,
                                          float *dx, float *partial, float *dgamma, float *dbeta,
                                          float *dxsum, void *stream);

/* score[r] = max_c (LayerNorm(x[r]) . w[c] + bias[c]): the class score the two-stage query selection ranks the
 * encoder tokens by (/root/reference/models/dino/deformable_transformer.py:335-342: enc_output_norm, the class head,
 * `.max(-1)[0]` into top-k), in one pass that writes neither the normalised rows nor the logits.  x [rows, 256]
 * contiguous, w [classes, 256], classes <= 16.  row_mask (may be NULL): rows with a non-zero byte are evaluated
 * with x[r] = row_fill [256] -- the reference zeroes invalid tokens BEFORE the projection that produces x
 * (:329-333), and the projection of a zero row is its bias. */
int datr_layernorm_class_max_f32(const float *x, const float *gamma, const float *beta, const float *w,
                                 const float *bias, int64_t rows, int64_t C, int64_t classes, float eps,
                                 const uint8_t *row_mask, const float *row_fill, float *score, void *stream);

/* ------------------------------------------------------------------------------------------
 * Input pipeline tail on the device: ToTensor + Normalize + pad into the batch + padding mask
 * for ONE image (/root/reference/datasets/da_transforms.py:250-276, util/misc.py:387-409).
 *   img   uint8 [H, W, 3] (HWC, device)      mean, std  3 host floats each
 *   out   this image's slot of the float batch: [3, Hp, Wp] (channels_last = 0) or [Hp, Wp, 3]
 *   mask  this image's [Hp, Wp] slot of the bool mask (1 = padding)
 * Every element of both slots is written (padding = 0 / 1).  Bit-exact with the reference's
 * fp32 operation sequence (x / 255 - mean) / std.
 * ------------------------------------------------------------------------------------------ */
int datr_normalize_pad_u8_f32(const uint8_t *img, int64_t H, int64_t W, const float *mean,
                              const float *std, int64_t Hp, int64_t Wp, int channels_last, float *out,
                              uint8_t *mask, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DATR_HIP_H_ */

/*
 * adx.h — C ABI of libadx.so, the MI355X (gfx950) implementation of the diffusion
 * trajectory-denoising hot path of Justin900429/autonomous_driving_with_diffusion_model.
 *
 * The reference has no FFI layer: its boundary is the Python call surface
 * (SURVEY.md §8b).  This header is the boundary one level below it: every entry point
 * takes raw device pointers, plain sizes and a hipStream_t (passed as void*), returns an
 * int status (0 = ok, <0 = error; adx_last_error() gives the message) and never allocates
 * device memory — workspaces are sized by *_bytes() queries and owned by the caller.
 * All tensors are fp32, contiguous row-major unless a stride is given; integer schedule
 * indices are int64.  Each entry cites the reference code it replaces
 * (paths relative to the reference checkout).
 */
#ifndef ADX_H
#define ADX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADX_OK 0
#define ADX_ERR_INVALID (-1)   /* bad argument / unsupported shape */
#define ADX_ERR_HIP (-2)       /* a HIP runtime call failed */
#define ADX_ERR_STATE (-3)     /* object used before it was initialised */
#define ADX_ERR_RANGE (-4)     /* ADX_CHECK_RANGE=1 only: an activation left the fp16 range of the split kernels */

typedef void* adx_stream;      /* hipStream_t */

int adx_version(void);
/* sha256 (first 32 hex digits) of the sources this library was built from (csrc/build.sh); "unknown" for ad-hoc builds */
const char* adx_source_hash(void);
const char* adx_last_error(void);

/* ------------------------------------------------------------------------------------
 * Temporal stack, op level (each op is one kernel launch).
 * -----------------------------------------------------------------------------------*/

/* Geometry of one temporal convolution, y = epilogue(conv(x)).
 *   kind 0: Conv1d(k=taps, stride, padding=pad)           modeling/helpers.py:80,104; temporal.py:40-44,192
 *   kind 1: ConvTranspose1d(k=taps, stride=2, padding=1)  modeling/helpers.py:89
 * The input may be the channel-concatenation of two tensors (skip connection,
 * modeling/temporal.py:227): channels [0,c0) come from x0, [c0,c0+c1) from x1.
 * Strides are in elements so that [B,H,D] trajectories can be read/written in place
 * (the einops rearranges at modeling/temporal.py:204,243).                              */
typedef struct adx_tconv_desc {
  int32_t kind, taps, stride, pad;
  int32_t c0, c1, cout;
  int32_t lin, lout;
  int32_t groups;            /* 0: no GroupNorm/Mish epilogue; 8: Conv1dBlock (helpers.py:95-112) */
  float eps;
  /* weight source layout for adx_tconv_pack (0,0 = the module's own weight).  The data-gradient of a
   * convolution is again a convolution of this family with the SAME weight tensor read differently:
   *   w_layout 0: w[cout][cin][taps]   1: w[cin][cout][taps];   w_flip 1: taps reversed.          */
  int32_t w_layout, w_flip;
  /* 0 (default): split-fp16 MFMA where the geometry allows it (fp32-grade result, csrc/tconv_hs.hip);
   * 1: the exact-fp32 MFMA kernel (used for gradient-sized operands whose range fp16 cannot hold). */
  int32_t exact;
  /* Lengths that are not powers of two (the reference accepts any horizon divisible by 8, modeling/temporal.py:59-75:
   * 24 -> 24, 12, 6, 3): `lin` / `lout` are then the lengths rounded up to powers of two (the kernels' index arithmetic)
   * and these the real ones; positions >= lin_valid read as zero, positions >= lout_valid are neither stored nor counted
   * in the GroupNorm statistics.  0 = equal to lin / lout.  Such descriptors run on the exact-fp32 kernel; adx_tconv_wgrad
   * honours the same fields, and the training executor passes them to every backward kernel. */
  int32_t lin_valid, lout_valid;
} adx_tconv_desc;

/* Size (bytes) of the packed weight image for a conv of this geometry. */
size_t adx_tconv_packed_bytes(const adx_tconv_desc* d);

/* Pack a PyTorch-layout weight ([cout][cin][taps] for kind 0, [cin][cout][taps] for kind 1)
 * into the MFMA B-operand fragment order read by adx_tconv_forward. */
int adx_tconv_pack(const adx_tconv_desc* d, const float* w, float* packed, adx_stream s);

typedef struct adx_tconv_io {
  const float* x0; int64_t x0_sb, x0_sc, x0_sl;   /* input 0 and its batch/channel/position strides */
  const float* x1; int64_t x1_sb, x1_sc, x1_sl;   /* input 1 (NULL when c1 == 0) */
  const float* packed_w;                          /* from adx_tconv_pack */
  const float* bias;                              /* [cout] or NULL */
  const float* gamma; const float* beta;          /* GroupNorm affine [cout] (groups > 0) */
  const float* tbias; int64_t tbias_stride;       /* additive time bias [B][...] (temporal.py:53) or NULL */
  const float* res; int64_t res_sb, res_sc, res_sl; /* residual added last (temporal.py:55) or NULL */
  float* y; int64_t y_sb, y_sc, y_sl;
  int32_t batch;
  /* training only (NULL otherwise): conv+bias before GroupNorm, dense [B][cout][lout], and the
   * per-(sample, group) statistics [B][groups][2] = (mean, rstd) that the backward pass reuses */
  float* pre; float* stats;
  /* optional device scratch (scratch_floats floats, contents irrelevant): lets a launch whose grid would occupy a few CUs
   * only (tiny batches) split its reduction over more workgroups and finish with a reduce launch.  NULL: never split. */
  float* scratch; int64_t scratch_floats;
  /* optional (NULL: absent), with `scratch`: 256 device words that are ZERO when the call is enqueued.  A split reduction then
   * needs no reduce launch: the last workgroup to publish its partial tile adds them up and leaves the words zero again, so
   * the same words serve every later call on the same stream (one word per workgroup of the unsplit grid, which is at
   * most 128 for a launch that splits). */
  uint32_t* tickets;
} adx_tconv_io;

/* Conv (+bias) [-> GroupNorm -> Mish] [+ time bias] [+ residual], fp32 MFMA.
 * Replaces Conv1dBlock.forward / Downsample1d / Upsample1d / the 1x1 residual and head convs. */
int adx_tconv_forward(const adx_tconv_desc* d, const adx_tconv_io* io, adx_stream s);

/* SinusoidalPosEmb + time_mlp (+ cond_mlp) + cat(img_feature) + Mish:
 * modeling/helpers.py:62-74, modeling/temporal.py:88-98,205-213.
 * rows = effective batch; t has t_rows entries and img_feature feat_rows rows, both
 * broadcast by `r % n` exactly like the reference's .repeat (temporal.py:208-211).
 * Outputs: time_embed [rows][dim] (after the cond_mlp add) and
 *          mish_cond  [rows][2*dim] = Mish(cat(time_embed, img_feature)).               */
typedef struct adx_embed_weights {
  const float* freqs;                    /* [dim/2] host-computed exp(-i*ln(1e4)/(dim/2-1)) */
  const float* w1; const float* b1;      /* time_mlp.1: [4dim][dim] */
  const float* w3; const float* b3;      /* time_mlp.3: [dim][4dim] */
  const float* cw0; const float* cb0;    /* cond_mlp.0: [dim][2]   (NULL unless FREE_GUIDANCE) */
  const float* cw2; const float* cb2;    /* cond_mlp.2: [dim][dim] */
} adx_embed_weights;
int adx_embed_forward(const adx_embed_weights* w, int32_t dim, const int64_t* t, int32_t t_rows,
                      const float* cond /* [rows][2] or NULL (=> zeros) */,
                      const float* img_feature, int32_t feat_rows, int32_t rows,
                      float* time_embed, float* mish_cond, adx_stream s);

/* ------------------------------------------------------------------------------------
 * Whole-UNet executor (one call = TemporalMapUnet.forward without the perception pass,
 * modeling/temporal.py:204-245).  Weights are handed over once in the reference's
 * parameter registration order and packed into a caller-owned buffer.
 * -----------------------------------------------------------------------------------*/
typedef struct adx_unet adx_unet;

typedef struct adx_unet_config {
  int32_t horizon, transition_dim, dim;
  int32_t n_mults; int32_t dim_mults[8];
  int32_t guidance;          /* 0 NO_GUIDANCE, 1 FREE_GUIDANCE, 2 CLASSIFIER_GUIDANCE (misc/constant.py:17-20) */
} adx_unet_config;

int adx_unet_create(const adx_unet_config* cfg, adx_unet** out);
void adx_unet_destroy(adx_unet* u);
/* number of parameter tensors expected by adx_unet_pack (the non-perception parameters,
 * in named_parameters() order; TrajPredict's are accepted but handled by adx_trajpred_*) */
int adx_unet_num_params(const adx_unet* u);
size_t adx_unet_packed_bytes(const adx_unet* u);
int adx_unet_pack(adx_unet* u, const float* const* params, int32_t n_params, const float* freqs,
                  void* packed, adx_stream s);
/* The workspace needs no initialisation.  Its first 256 words are the ticket words of adx_tconv_io::tickets (at a place
 * that does not depend on `rows`); adx_unet_forward clears them itself at the head of every call. */
size_t adx_unet_workspace_bytes(const adx_unet* u, int32_t rows);

typedef struct adx_unet_io {
  const float* x;             /* [rows][horizon][transition_dim] */
  const float* img_feature; int32_t feat_rows;   /* perception output [feat_rows][dim] */
  const int64_t* t; int32_t t_rows;
  const float* cond;          /* [rows][2] or NULL */
  int32_t rows;
  float* out;                 /* [rows][horizon][transition_dim] (NO/FREE) or action [rows][horizon][3] (CLASSIFIER) */
  float* time_embed;          /* [rows][dim] or NULL (CLASSIFIER: returned to the caller, temporal.py:236-237) */
  /* Optional (zero = absent), for sampling loops.
   * x_rows: rows of `x` actually present: `rows` (default) or 1 -- every row then reads the one trajectory, which is the
   *   classifier-free pair torch.cat([trajs, trajs]) of interact.py:131 at B = 1 without building it.
   * time_bias: this call's [rows][adx_unet_time_bias_width] slice of a table made by adx_unet_time_conditioning; t, cond
   *   and img_feature are then not read and time_embed must be NULL (the caller holds the table's time_embed). */
  int32_t x_rows;
  const float* time_bias;
} adx_unet_io;
int adx_unet_forward(adx_unet* u, const void* packed, void* workspace, const adx_unet_io* io, adx_stream s);
/* Everything of the forward that depends on (t, cond, img_feature) only -- the time MLP, the condition MLP and the
 * time_mlp Linear of all 16 residual blocks (modeling/temporal.py:206-216, modeling/helpers.py:121-123) -- for `rows`
 * rows at once, e.g. all 50 timesteps of a sampling loop x the rows of one step: inside the loop
 * (interact.py:128-166) these are the same launches every tick, and none of them depends on the trajectory.  Uses
 * io->t, t_rows, cond, img_feature, feat_rows, rows (workspace sized by adx_unet_time_conditioning_workspace_bytes(u,
 * rows): two small per-row vectors, not a forward's activation ring).  Writes
 * time_bias [rows][adx_unet_time_bias_width(u)] and, if not NULL, time_embed [rows][dim].  Row-wise identical to what
 * adx_unet_forward computes internally. */
int32_t adx_unet_time_bias_width(const adx_unet* u);
size_t adx_unet_time_conditioning_workspace_bytes(const adx_unet* u, int32_t rows);
int adx_unet_time_conditioning(adx_unet* u, const void* packed, void* workspace, const adx_unet_io* io, float* time_embed,
                               float* time_bias, adx_stream s);

/* ------------------------------------------------------------------------------------
 * Training step T1 (train.py:242-251), temporal stack: forward that keeps a tape, and backward.
 * The reference gets these from torch autograd; the gradients are written in PyTorch layouts.
 * -----------------------------------------------------------------------------------*/
typedef struct adx_unet_tape adx_unet_tape;
int adx_unet_tape_create(adx_unet_tape** out);
void adx_unet_tape_destroy(adx_unet_tape* t);
size_t adx_unet_train_workspace_bytes(const adx_unet* u, int32_t rows);
int adx_unet_forward_train(adx_unet* u, const void* packed, void* workspace, size_t workspace_bytes,
                           const adx_unet_io* io, adx_unet_tape* tape, adx_stream s);
/* d_out [rows][H][D] -> one gradient per parameter of adx_unet_pack's list (written) and
 * d_img_feature [rows][dim] (gradient entering the perception encoder).  `params` are the same
 * raw parameter pointers given to adx_unet_pack (the data-gradient convs re-read them). */
int adx_unet_backward(adx_unet* u, const void* packed, void* workspace, size_t workspace_bytes, adx_unet_tape* tape,
                      const float* d_out, const float* d_time_embed /* extra gradient into time_embed or NULL */,
                      float* d_img_feature, const float* const* params, float* const* grads, int32_t n_grads,
                      adx_stream s);
/* op level: gradient through [+tb] -> Mish -> GroupNorm of a Conv1dBlock (modeling/helpers.py:105-108) */
int adx_gn_mish_backward(const float* dy, int64_t dy_sb, int64_t dy_sc, int64_t dy_sl, const float* pre,
                         const float* stats, const float* gamma, const float* beta, float* dc, float* dgamma,
                         float* dbeta, float* dbias, float* dtb, int64_t dtb_stride, int32_t B, int32_t C, int32_t L,
                         int32_t groups, adx_stream s);
/* op level: dW[cout][cin][taps] of a (kind 0) temporal conv from its input (io->x0/x1) and d(conv out) */
int adx_tconv_wgrad(const adx_tconv_desc* d, const adx_tconv_io* io, const float* dc, float* dw, adx_stream s);
int adx_bias_grad(const float* dc, float* db, int32_t B, int32_t C, int32_t L, adx_stream s);

/* ------------------------------------------------------------------------------------
 * Perception: ResNet-34 forward (eval mode, BatchNorm folded at pack time),
 * modeling/resnet.py:56-102,163-296 with fc = Linear(512, dim) (temporal.py:83-84).
 * -----------------------------------------------------------------------------------*/
typedef struct adx_resnet adx_resnet;
int adx_resnet_create(int32_t out_dim, adx_resnet** out);
void adx_resnet_destroy(adx_resnet* r);
int adx_resnet_num_tensors(const adx_resnet* r);     /* state_dict entries excluding num_batches_tracked */
size_t adx_resnet_packed_bytes(const adx_resnet* r);
/* tensors: the perception.* state_dict entries in order, num_batches_tracked skipped
 * (conv weight, bn weight, bn bias, bn running_mean, bn running_var, ..., fc weight, fc bias) */
int adx_resnet_pack(adx_resnet* r, const float* const* tensors, int32_t n, void* packed, adx_stream s);
size_t adx_resnet_workspace_bytes(const adx_resnet* r, int32_t batch, int32_t h, int32_t w);
/* Stream semantics: everything the call enqueues is ordered behind `s`'s earlier work and in front of its later work.  At
 * batch >= 32 (outside a stream capture) the pass runs as two sub-batches: one on `s`, one on a side stream the HANDLE owns
 * (created on first use, destroyed with the handle; forked from `s` and joined into `s` by events inside the call) -- one
 * sub-batch's launches fill the CUs the other's last round of workgroups leaves idle.  Same kernels and per-image arithmetic;
 * ADX_RESNET_STREAMS=1 keeps one chain.  A handle must not be driven from two host threads at once. */
int adx_resnet_forward(adx_resnet* r, const void* packed, void* workspace, const float* img /* NCHW */,
                       int32_t batch, int32_t h, int32_t w, float* feature /* [batch][out_dim] */, adx_stream s);
/* The same with the agents' image front-end folded into the stem's staging load: frames_hwc = uint8 camera frames
 * [batch][h][w][3] (RGB); ToTensor + Normalize(mean, std) (interact.py:73-78, e2e_driving/diffusion_agent.py:96-101) are
 * applied on the fly -- (v / 255 - mean[c]) / std[c], the arithmetic of adx_image_normalize -- and the fp32 NCHW image
 * is never materialised.  Bit-identical to adx_image_normalize followed by adx_resnet_forward. */
int adx_resnet_forward_u8(adx_resnet* r, const void* packed, void* workspace, const uint8_t* frames_hwc,
                          const float* mean /* [3] */, const float* stdv /* [3] */, int32_t batch, int32_t h, int32_t w,
                          float* feature /* [batch][out_dim] */, adx_stream s);

/* Op-level 2-D convolution used by the perception executor (one launch): NCHW fp32,
 * y = [relu]( conv(x, w) * scale[c] + shift[c] [+ res] ); scale/shift = eval-mode BatchNorm2d
 * (modeling/resnet.py:87-102).  cout must be a multiple of 64. */
typedef struct adx_conv2d_desc { int32_t cin, cout, k, stride, pad; } adx_conv2d_desc;
size_t adx_conv2d_packed_bytes(const adx_conv2d_desc* d);
int adx_conv2d_pack(const adx_conv2d_desc* d, const float* w /* [cout][cin][k][k] */, float* packed, adx_stream s);
int adx_conv2d_forward(const adx_conv2d_desc* d, const float* x, const float* packed_w, const float* scale,
                       const float* shift, const float* res, float* y, int32_t n, int32_t h, int32_t w,
                       int32_t relu, adx_stream s);
/* The pipelined 3x3 stride-1 kernel can also read and write CELL tensors: per image [C / 8][plane: hi, lo][H][W] cells of
 * 16 bytes = the 8 channels of one pixel split into fp16 halves, x = hi + lo / 2^11 (hi = fp16(x), lo = fp16((x - hi) 2^11)) --
 * the layout its staging writes to LDS, so a consumer copies cells where it would convert fp32 values; same bytes per tensor
 * as fp32 NCHW.  adx_resnet_forward keeps the activations between such launches in this layout (modeling/resnet.py:87-102
 * BasicBlock chain); these two entry points expose the launch for tests and other callers.  fmt: bit 0 = x is a cell tensor,
 * bit 1 = y is (required), bit 2 = res is.  adx_conv2d_cells_supported: 1 when (desc, n, h, w) runs as one such launch. */
int adx_conv2d_cells_supported(const adx_conv2d_desc* d, int32_t n, int32_t h, int32_t w);
int adx_conv2d_forward_cells(const adx_conv2d_desc* d, const void* x, const float* packed_w, const float* scale,
                             const float* shift, const void* res, void* y, int32_t n, int32_t h, int32_t w, int32_t relu,
                             int32_t fmt, adx_stream s);
/* Weight gradient of the same convolution (torch.nn.grad.conv2d_weight; train.py:242 reaches it through
 * loss.backward()): dw [cout][cin][k][k] = sum over batch and pixels of dy (x) x.  dy is [n][cout][oh][ow].
 * scratch: NULL, or >= adx_conv2d_wgrad_scratch_bytes() of device memory in which the range of dy is estimated
 * first -- the 3x3 stride-1 path multiplies on the fp16 matrix cores with hi/lo split operands and rescales dy by an
 * exact power of two; without scratch dy is taken as is (full accuracy only for |dy| >= 6e-5). */
size_t adx_conv2d_wgrad_scratch_bytes(void);
int adx_conv2d_wgrad(const adx_conv2d_desc* d, const float* x, const float* dy, float* dw, int32_t n, int32_t h,
                     int32_t w, void* scratch, adx_stream s);
/* The same with the two uses of the scratch told apart: estimate_range != 0 asks for the range estimate above (scratch
 * required); estimate_range == 0 takes dy as is, and a non-NULL scratch then only carries the per-split copies of the
 * bit-reproducible reduction (ADX_WGRAD_DETERMINISTIC=1; adx_conv2d_wgrad_scratch_bytes() covers both). */
int adx_conv2d_wgrad_ex(const adx_conv2d_desc* d, const float* x, const float* dy, float* dw, int32_t n, int32_t h,
                        int32_t w, void* scratch, int32_t estimate_range, adx_stream s);
/* The same for a 3x3 stride-1 pad-1 convolution with BOTH operands as cell tensors -- how the training executor holds the
 * activations between a BasicBlock's convs and the conv-output gradients (train.py:251, loss.backward()): x_cells
 * [n][cin / 8][hi, lo][h][w] cells, dy_cells likewise over cout, holding dy * dy_scale[0]; dy_scale = two floats {s, 1 / s} on
 * the device, s a power of two.  scratch: NULL, or adx_conv2d_wgrad_scratch_bytes() (ADX_WGRAD_DETERMINISTIC=1). */
int adx_conv2d_wgrad_cells(const adx_conv2d_desc* d, const void* x_cells, const void* dy_cells, const float* dy_scale, float* dw,
                           int32_t n, int32_t h, int32_t w, void* scratch, adx_stream s);

/* Training-mode perception (train.py:242 with model.train()): batch-statistics BatchNorm, running buffers
 * updated in place (momentum 0.1), everything the backward needs kept in the workspace. */
typedef struct adx_resnet_tape adx_resnet_tape;
int adx_resnet_tape_create(adx_resnet_tape** out);
void adx_resnet_tape_destroy(adx_resnet_tape* t);
size_t adx_resnet_train_workspace_bytes(const adx_resnet* r, int32_t batch, int32_t h, int32_t w);
int adx_resnet_forward_train(adx_resnet* r, const float* const* tensors, int32_t n_tensors, void* packed,
                             void* workspace, size_t workspace_bytes, const float* img, int32_t batch, int32_t h,
                             int32_t w, float* feature, adx_resnet_tape* tape, int32_t update_running, adx_stream s);
/* grads: one slot per tensor of adx_resnet_pack's list; conv weight / bn weight / bn bias / fc slots are written,
 * running-statistics slots are ignored (may be NULL). */
int adx_resnet_backward(adx_resnet* r, const float* const* tensors, float* const* grads, int32_t n_tensors,
                        void* workspace, size_t workspace_bytes, adx_resnet_tape* tape, const float* d_feature,
                        adx_stream s);
/* The same with completion events: `events` = adx_resnet_backward_groups(r) hipEvent_t handles (or NULL entries); event g is
 * recorded on `s` behind the last launch that writes a gradient of group g.  Groups in the order the backward produces them:
 * 0 = fc, 1 .. n_blocks = the BasicBlocks from layer4's last to layer1's first, n_blocks + 1 = the stem (conv1 + bn1).
 * adx_resnet_tensor_group(r, i) = the group of tensor slot i (-1: a running statistic).  This is what lets a gradient reduction
 * of the upper layers start while the lower layers are still being differentiated -- what DistributedDataParallel's reducer
 * does per parameter while accelerator.backward(loss) runs (train.py:176-178, 251). */
int32_t adx_resnet_backward_groups(const adx_resnet* r);
int32_t adx_resnet_tensor_group(const adx_resnet* r, int32_t tensor);
int adx_resnet_backward_events(adx_resnet* r, const float* const* tensors, float* const* grads, int32_t n_tensors,
                               void* workspace, size_t workspace_bytes, adx_resnet_tape* tape, const float* d_feature,
                               void* const* events, int32_t n_events, adx_stream s);

/* ------------------------------------------------------------------------------------
 * Classifier guidance: TrajPredict state head (modeling/helpers.py:22-59; hidden 64, 4 heads,
 * ff 256, 2 layers, eval mode) forward, input gradient, and the fused guidance update.
 * -----------------------------------------------------------------------------------*/
typedef struct adx_trajpred adx_trajpred;
int adx_trajpred_create(int32_t out_dim, adx_trajpred** out);
void adx_trajpred_destroy(adx_trajpred* t);
int adx_trajpred_num_params(const adx_trajpred* t);       /* state_pred.* named_parameters() count (30) */
size_t adx_trajpred_packed_bytes(const adx_trajpred* t);
/* Sequence lengths: T = horizon - 1 from 1 to 63 (modeling/temporal.py:119 builds the head for any horizon).  T <= 31 keeps
 * every activation of a sample in LDS; T = 32..63 runs a 64-row instantiation that keeps three large tiles per sample in
 * a scratch the caller lends (no initialisation; alive and unshared while launches that use it are in flight).
 * adx_trajpred_scratch_bytes is 0 for T <= 31; the calls below fail with ADX_ERR_STATE when the lent scratch is too small. */
size_t adx_trajpred_scratch_bytes(const adx_trajpred* t, int32_t batch, int32_t T);
int adx_trajpred_set_scratch(adx_trajpred* t, void* scratch, size_t bytes);
int adx_trajpred_pack(adx_trajpred* t, const float* const* params, int32_t n, const float* freqs, void* packed,
                      adx_stream s);
/* out[B][T][out_dim] = state_pred(action[B][T][3] (strides act_sb, act_st), time_embed[B][64]) */
int adx_trajpred_forward(adx_trajpred* t, const void* packed, const float* action, int64_t act_sb, int64_t act_st,
                         const float* time_embed, float* out, int32_t batch, int32_t T, adx_stream s);
/* grad_action[B][T][3] = (d out / d action)^T grad_out[B][T][out_dim]; what torch.autograd.grad returns for
 * `action` in control/guidance.py:47-50 through the state path */
int adx_trajpred_backward(adx_trajpred* t, const void* packed, const float* action, int64_t act_sb, int64_t act_st,
                          const float* time_embed, const float* grad_out, float* grad_action, int32_t batch,
                          int32_t T, adx_stream s);
/* Training (train.py:242-251 with CLASSIFIER_GUIDANCE): parameter gradients of the state head.  grad_image has
 * the layout of the packed buffer (adx_trajpred_packed_bytes); parameter i lives at float offset
 * offsets[i] (adx_trajpred_param_offsets, adx_trajpred_pack order) in its PyTorch layout. */
int adx_trajpred_backward_params(adx_trajpred* t, const void* packed, const float* action, int64_t act_sb,
                                 int64_t act_st, const float* time_embed, const float* grad_out, float* grad_action,
                                 void* grad_image, float* d_time_embed, int32_t batch, int32_t T, float dropout_p,
                                 uint64_t seed, adx_stream s);
/* Train-mode forward of the state head: nn.TransformerEncoderLayer's dropout (modeling/helpers.py:35-41 leaves it at
 * torch's default 0.1: attention probabilities, both residual branches, the feed-forward activation) with masks that are
 * a pure function of (seed, sample, site, element) -- adx_trajpred_backward_params regenerates them from the same
 * (dropout_p, seed).  dropout_p = 0 is the deterministic path the golden fixtures use. */
int adx_trajpred_forward_train(adx_trajpred* t, const void* packed, const float* action, int64_t act_sb, int64_t act_st,
                               const float* time_embed, float* out, int32_t batch, int32_t T, float dropout_p,
                               uint64_t seed, adx_stream s);
int adx_trajpred_param_offsets(const adx_trajpred* t, int64_t* offsets, int32_t n);
/* One launch for interact.py:153-160 + GuidanceLoss.forward with STEP = 1 (control/guidance.py:35-59) +
 * TargetGuidance (control/guidance_loss.py:10-22), per sample:
 * x = cat([0; state_pred(action[:-1])], action); pick h*; x -= scaled gradient; clip(-1, 1). */
int adx_guided_output(adx_trajpred* t, const void* packed, const float* action /* [B][T+1][3] */,
                      const float* time_embed, const float* target /* [B][2] */, float model_std, float scale,
                      float* x_guided /* [B][T+1][out_dim+3] */, float* loss /* [B] or NULL */, int32_t batch,
                      int32_t T, adx_stream s);

/* ------------------------------------------------------------------------------------
 * Optimizer step of train.py:252-261 in one launch: grad nan_to_num, AdamW, EMA shadow update.
 * table = device array of {float* p; const float* g; float* m; float* v; float* ema; int64_t n} per tensor;
 * block_tensor / block_chunk map each workgroup to (tensor, chunk of adx_optim_chunk() elements).
 * -----------------------------------------------------------------------------------*/
int adx_optim_chunk(void);
int adx_adamw_ema_step(const void* table, const int32_t* block_tensor, const int32_t* block_chunk, int32_t n_blocks,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                       float ema_decay, int32_t use_ema, int32_t sanitize, adx_stream s);
/* the same with every gradient multiplied by grad_scale as it is read (data-parallel training: 1 / world_size when the
 * all-reduce left the SUM over the ranks in .grad; the reference's DDP divides inside its own reduction, train.py:176-178) */
int adx_adamw_ema_step_scaled(const void* table, const int32_t* block_tensor, const int32_t* block_chunk, int32_t n_blocks,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                       float ema_decay, int32_t use_ema, int32_t sanitize, float grad_scale, adx_stream s);

/* ------------------------------------------------------------------------------------
 * Scheduler step math.  The integer schedule and the fp32 scalar coefficients are computed
 * by the host (the scheduler/ modules keep them as 0-dim CPU tensors) and passed by value.
 * -----------------------------------------------------------------------------------*/
#define ADX_PRED_EPSILON 0
#define ADX_PRED_SAMPLE 1
#define ADX_PRED_V 2

typedef struct adx_step_coef {
  int32_t prediction_type;     /* ADX_PRED_* */
  int32_t clip;                /* 1: clamp x0 to [-clip_range, clip_range]; `thresholding=True` with the
                                  diffusers default sample_max_value=1 is exactly clamp(-1, 1) */
  float clip_range;
  float sqrt_alpha_t, sqrt_beta_t;   /* abar_t ** 0.5, (1 - abar_t) ** 0.5 */
  float c_x0;                  /* DDIM: abar_prev ** 0.5;  DDPM: pred_original_sample_coeff */
  float c_dir;                 /* DDIM: (1 - abar_prev - std^2) ** 0.5 */
  float c_x;                   /* DDPM: current_sample_coeff */
  float c_noise;               /* DDIM: std_dev_t = eta * var ** 0.5;  DDPM: var ** 0.5 */
  int32_t add_noise;           /* DDIM: eta > 0;  DDPM: t > 0 */
  int32_t use_clipped_model_output;
  /* inpainting schedulers only */
  int32_t inpaint;             /* 1: Inpainting*Scheduler arithmetic */
  float c_const;               /* inpainting DDIM quirk: the SCALAR variance is added to every element */
  float c_known, c_known_noise;/* RePaint: known = c_known * target + (known_noise ? c_known_noise * z : 0) */
  int32_t known_noise;         /* t > 0 */
  /* classifier-free guidance combine fused in front of the step (interact.py:142-144):
   * model_output = uncond + free_scale * (cond - uncond); rows [0,B) cond, [B,2B) uncond */
  int32_t cfg_combine; float free_scale;
  int32_t zero_first;          /* 1: prev[:, 0, :3] = 0 after the step (interact.py:164, train.py:88) */
} adx_step_coef;

/* S1/S3: GuidanceDDIMScheduler.step / InpaintingDDIMScheduler.step
 *        scheduler/guidance_ddim_scheduler.py:60-173, inpainting_ddim_scheduler.py:10-153 */
int adx_ddim_step(const adx_step_coef* c, const float* model_output, const float* sample, const float* noise,
                  const float* target, const float* mask, float* prev, float* x0,
                  int32_t batch, int32_t horizon, int32_t dim, adx_stream s);
/* S2/S4: GuidanceDDPMScheduler.step / InpaintingDDPMScheduler.step / stock DDPMScheduler.step
 *        scheduler/guidance_ddpm_scheduler.py:59-178, inpainting_ddpm_scheduler.py:10-146, train.py:87 */
int adx_ddpm_step(const adx_step_coef* c, const float* model_output, const float* sample, const float* noise,
                  const float* target, const float* mask, float* prev, float* x0,
                  int32_t batch, int32_t horizon, int32_t dim, adx_stream s);
/* add_noise (train.py:234) fused with the [...,0,:3] = 0 of train.py:235 when zero_first != 0.
 * sqrt_ab / sqrt_1mab are the host tables sqrt(abar), sqrt(1-abar) of length n_train. */
int adx_add_noise(const float* x, const float* noise, const int64_t* t, const float* sqrt_ab, const float* sqrt_1mab,
                  int32_t n_train, float* out, int32_t batch, int32_t horizon, int32_t dim, int32_t zero_first,
                  adx_stream s);

/* Camera front-end (interact.py:73-78, e2e_driving/diffusion_agent.py:96-101): ToTensor + Normalize of uint8
 * [n][h][w][3] frames into fp32 [n][3][h][w]; mean/std are HOST pointers to 3 floats. */
int adx_image_normalize(const uint8_t* frame_hwc, float* out_nchw, int32_t n, int32_t h, int32_t w, const float* mean,
                        const float* stdv, adx_stream s);

/* Batch image augmentation on the GPU: stand-in for the reference's imgaug pipeline (dataset/augment.py:10-77, applied per
 * sample in dataset/carla_dataset.py:24-31).  frames_hwc: uint8 [n][h][w][3], augmented IN PLACE; scratch: n*h*w*3 bytes
 * (needed when any_blur != 0).  The host draws the plan (autonomous_driving_with_diffusion_model_amd/dataset/augment.py):
 * plan [n][7][8] floats = per image seven operator slots in application order (row = code, p0..p3, per_channel, 0, 0;
 * codes in csrc/augment.hip), seeds [n] for the per-pixel hash randomness, ranges [n][4] = the slot ranges applied before
 * and after the image's blur, blur_sigma [n] (0 = no blur).  All four arrays live on the device. */
int adx_image_augment(uint8_t* frames_hwc, uint8_t* scratch, int32_t n, int32_t h, int32_t w, const float* plan,
                      const uint64_t* seeds, const int32_t* ranges, const float* blur_sigma, int32_t any_blur, adx_stream s);

/* Measurement probe (bench.py's `roofline.sustained`; no reference counterpart): `workgroups` x 4 waves run `iters` trips of
 * the 3x3 convolution's inner loop -- 8 LDS operand reads per 12 v_mfma_f32_32x32x16_f16 -- on the 64 KB of fp16 operand cells
 * at `operands` (device memory), with no global traffic, staging or epilogue: the fp16 MFMA rate the chip sustains at its
 * power limit.  out: workgroups * 256 floats (a checksum, so that nothing is optimised away); *flops (host, optional)
 * receives the MFMA flops of the launch. */
int adx_probe_mfma_fp16(const void* operands, float* out, int32_t workgroups, int32_t iters, double* flops, adx_stream s);
/* The same probe on v_mfma_f32_16x16x32_f16 (the loop of csrc/conv2d_hs16.hip: 16 operand reads per 48 MFMAs -- the same flops and
 * LDS reads per trip).  At the power limit the two shapes sustain different clocks; `roofline.sustained` reports both. */
int adx_probe_mfma_fp16_16x16x32(const void* operands, float* out, int32_t workgroups, int32_t iters, double* flops, adx_stream s);

#ifdef __cplusplus
}
#endif
#endif /* ADX_H */

// Shared between the inference (conv2d.hip) and training (resnet_train.hip) perception executors.
#pragma once
#include <vector>

#include "adx_common.h"

namespace adx {

constexpr int kTileH = 4;    // output rows per workgroup (one per wave)
constexpr int kTileW = 32;   // output columns per workgroup (= MFMA N)

struct ConvSpec {
  int cin, cout, k, stride, pad;
  int t_w, t_g, t_b, t_m, t_v;           // tensor indices (weight, bn gamma, beta, mean, var)
  size_t o_w, o_scale, o_shift;          // float offsets in the packed buffer
  int cin_pad, cc;
  int fuse_with;                         // ResNet executor: index of the conv1 this downsample conv can ride on, or -1
  int dgrad;                             // 1: this launch is the data-gradient view of a forward conv
};

struct Conv2dArgs {
  const float* x;       // [N][Cin][H][W]
  const float* w;       // packed [KH*KW][cin_pad][Cout]
  const float* scale;   // [Cout] or null
  const float* shift;   // [Cout] or null
  const float* res;     // [N][Cout][OH][OW] or null
  float* y;             // [N][Cout][OH][OW]
  const float* w_ds;    // conv2d_hs only: fused 1x1 stride-2 downsample (packed weights, BN scale/shift, output) or null
  const float* scale_ds;
  const float* shift_ds;
  float* y_ds;
  const uint32_t* x_amax;  // optional: x_amax_n partial maxima of |x| as bit patterns (conv2d_hs rescales x by a power of two);
  int x_amax_n;            // x_amax_n < 0: x_amax points at two floats {xs, 1 / xs} instead -- the power of two a cell-layout x was
                           // ALREADY multiplied by when it was written (resnet_train.hip: bn_bwd_apply_groups_kernel)
  // stem only: the camera frames as uint8 [N][H][W][3]; ToTensor + Normalize ((v / 255 - mean) / std, the arithmetic of
  // image_normalize_kernel) happen in the staging load and the fp32 NCHW tensor is never written (x is unused then)
  const uint8_t* x_u8;
  float u8_mean[3], u8_std[3];
  int N, Cin, H, W, Cout, OH, OW, KH, KW, stride, pad, relu;
  int cin_pad, cc;      // channels padded to the chunk size, chunk size (16, or 4 for the stem)
  int tiles_x, tiles_y, cout_tiles, ntiles;
  int q_slots;          // conv2d_hs3x3q only: workgroups per (XCD, cout tile); each walks its XCD's spatial tiles in steps of q_slots
  int PH, PW, PWp;      // staged patch rows, columns, padded row pitch
  // conv2d_hs3x3 only: the input-channel chunks split over `ksplit` workgroups per tile (small batches: a 512->512 layer
  // on one 8x29 map is 8 tiles); part kpart covers chunks [kpart * cper, (kpart + 1) * cper) and writes its raw sums to
  // part + kpart * part_stride in the layout of y; conv2d_split_reduce_kernel adds them up and applies BN / residual / ReLU
  int ksplit, cper;
  float* part;
  size_t part_stride;
  // stem kernels only: a band's 32-column tiles split into segments of stem_seg_tiles tiles, one workgroup each (few images:
  // a band of one 256x900 frame is 15 tiles walked by ONE workgroup otherwise); 0 = the whole band
  int stem_seg_tiles, stem_nseg;
  // generic split kernel only: depth-to-space store.  The conv's output channels are four groups of d2s_cin channels, one per
  // parity class (py, px) of a map of d2s_h x d2s_w pixels; channel g * d2s_cin + ci at (oy, ox) is stored to (and its residual
  // read from) channel ci at (2 oy + py, 2 ox + px), g = 2 py + px.  0 = plain NCHW store.  This is how the data gradient of a
  // stride-2 3x3 conv runs as ONE stride-1 2x2 conv on the low-resolution gradient (conv2d_hs_dgrad_s2).
  int d2s_cin, d2s_h, d2s_w;
  // conv2d_hs3x3 only (training forward): per-workgroup sums of the output and of its squares, [Cout][2][stats_p] floats
  // (p = (image, row tile, column tile)), pixels outside the map excluded; bn statistics then need no pass over the output
  float* stats_part;
  int stats_p;
  // conv2d_hs3x3 only (training backward, bs_raw != null): this launch is a data gradient whose output dx is the incoming
  // gradient of a BatchNorm (the conv BEFORE it in forward order).  Its epilogue then also leaves, in stats_part, that
  // BatchNorm's backward sums -- per channel the sum of dz and of dz * xhat over the workgroup's pixels, dz = dx * ReLU mask,
  // xhat = (raw - mean) * rstd -- so that no pass over dx has to compute them (channel_sums_kernel<1>).  bs_mask: 1 the
  // mask is `bs_out > 0` (ReLU after the residual add), 2 it is re-derived from the conv output: fma(raw, gamma rstd, beta -
  // mean gamma rstd) > 0 (ReLU straight after BatchNorm).
  const float* bs_raw; const float* bs_out; const float* bs_mean; const float* bs_rstd; const float* bs_gamma; const float* bs_beta;
  int bs_mask;
  // bs_mask == 1 with bs_bits != null: the mask as the forward pass left it (resnet_train.hip: bn_apply_groups_kernel), one BIT per
  // element -- byte [n][c / 8][pixel], bit c % 8 -- instead of the fp32 map bs_out
  const uint8_t* bs_bits;
  // bs_raw != null only: the residual `res` passes through a ReLU mask given as bits in the same layout (the identity path of a
  // BasicBlock's backward: res = d(block output), the mask = where the block's output was positive), so that the masked gradient
  // is never written as a tensor of its own
  const uint8_t* res_bits;
  // conv2d_hs3x3 only (inference executor): x / y / res in the cell layout instead of fp32 NCHW (conv2d_hs.hip: XCELLS)
  int x_cells, y_cells, res_cells;
  int vw;       // y_cells: columns of one image in the virtual row the column tiles run over (W + 1: the images side by side with
                // one shared all-zero column between neighbours; W for a single image)
  float inv_vw; // 1 / vw
};

// activation formats of one launch of the inference executor (bits)
constexpr int kFmtXCells = 1, kFmtYCells = 2, kFmtResCells = 4;
constexpr int kFmtXScaled = 8;      // with kFmtXCells: a data-gradient launch whose cell-layout x carries its scale (x_amax_n < 0)


}  // namespace adx

struct adx_resnet {
  int out_dim = 0;
  std::vector<adx::ConvSpec> convs;      // execution order: stem, then per block conv1, conv2, [downsample]
  std::vector<int> block_has_ds;         // per BasicBlock
  int t_fcw = 0, t_fcb = 0, n_tensors = 0;
  size_t o_fcw = 0, o_fcb = 0, packed_floats = 0;
  bool packed_once = false;
  // side streams of the inference executor (conv2d.hip: sub-batches of a large batch run on streams of their own so that one
  // sub-batch's launches fill the CUs another's last round of workgroups leaves idle); created on first use, per device
  static constexpr int kMaxSub = 4;
  hipStream_t side[kMaxSub - 1] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[kMaxSub - 1] = {nullptr, nullptr, nullptr};
  int side_device = -1;
};

namespace adx {

// one conv2d launch: y = [relu](conv(x, w) [* scale + shift] [+ res]); w = packed [tap][cin_pad][cout]
// x_amax (optional, device): x_amax_n bit patterns whose maximum is max|x| over the whole input; the split-fp16 kernels use it to move x
// into fp16's normal range by an exact power of two (data gradients are far below 2^-14)
struct BnBwdStats { const float* raw; const float* out; const float* mean; const float* rstd; const float* gamma; const float* beta; int mask;
                    const uint8_t* bits = nullptr; const uint8_t* res_bits = nullptr; };
int conv2d_launch_raw(const ConvSpec& L, const float* x, const float* w, const float* scale, const float* shift,
                      const float* res, float* y, int N, int H, int W, int relu, hipStream_t s,
                      const uint32_t* x_amax = nullptr, int x_amax_n = 0, float* stats_part = nullptr, size_t stats_floats = 0,
                      int* stats_p = nullptr, int fmt = 0, const BnBwdStats* bst = nullptr);
// true when this launch will run the pipelined 3x3 stride-1 kernel as ONE launch (no split reduction): the launches whose
// input / output / residual may be in the cell layout (fmt: kFmt*)
bool conv2d_hs3x3_plain(const ConvSpec& L, int N, int H, int W);
// true when a training-forward launch of this conv (statistics in the epilogue, fp32 output) can read its input as a cell tensor
// (fmt = kFmtXCells together with stats_part): the pipelined 3x3 stride-1 kernel with room for its partial sums in stats_floats
bool conv2d_hs3x3_train_cells(const ConvSpec& L, int N, int H, int W, size_t stats_floats);
// true when a data-gradient launch of this (dgrad) spec can read a cell-layout, pre-scaled gradient (fmt = kFmtXCells | kFmtXScaled)
bool conv2d_hs3x3_dgrad_cells(const ConvSpec& L, int N, int H, int W);
// true when a data-gradient launch of this spec with a BnBwdStats request will run the pipelined 3x3 kernel's statistics epilogue
// (the launches that can take BnBwdStats::res_bits)
bool conv2d_hs3x3_dgrad_stats(const ConvSpec& L, int N, int H, int W, bool x_cells, size_t stats_floats);
// conv2d_hs16.hip: the 16x16x32 kernel's training-forward variant (cells in, fp32 + statistics out) and its tile count
bool conv2d_hs3x3q_train_eligible(const Conv2dArgs& a);
bool conv2d_hs3x3q_dgrad_eligible(const Conv2dArgs& a);
int conv2d_hs3x3q_train_tiles(const Conv2dArgs& a);
// stats_part (optional, stats_floats floats): where the launch may leave per-workgroup partial sums of its output and of its
// squares ([Cout][2][P] floats); *stats_p = P when it did (the pipelined 3x3 stride-1 kernel does), 0 when the caller has to
// compute the statistics from the output itself
// [Cout][Cin][k][k] -> [tap][cin_pad][Cout]; dgrad = 1 packs the data-gradient view instead:
// [tap'][cout_pad as K][Cin as N] with the taps flipped (conv of dy with this image gives dx)
int conv2d_pack_raw(const float* w, float* packed, int cout, int cin, int k, int cin_pad, int dgrad, hipStream_t s);
// fp32 weights -> the layout the kernel chosen for `consumer` reads (fp32 [tap][cin_pad][cout], or the fp16 hi/lo
// split image of conv2d_hs.hip).  dgrad = 1: `w` is the forward weight [consumer.cin][consumer.cout][k][k] and the
// image is its data-gradient view (roles swapped, taps flipped).
int conv2d_pack_spec(const ConvSpec& consumer, const float* w, float* packed, int dgrad, hipStream_t s);
// conv2d_hs.hip: fp32-equivalent convolution on the fp16 matrix cores (hi/lo split operands, 3 MFMAs per product)
bool conv2d_hs_eligible(const ConvSpec& L);
// floats a layer's packed weight image takes (direct image, plus the Winograd F(2,3) image where that path applies)
size_t conv2d_packed_floats(const ConvSpec& L);
int conv2d_hs_pack(const ConvSpec& consumer, const float* w, void* packed, int dgrad, hipStream_t s);
// several split-fp16 weight images in one launch (not the stem's): cout / cin as conv2d_hs_pack takes them from a ConvSpec
// (dgrad = 1: the spec of the data-gradient conv, i.e. cout = the forward cin and cin_pad = cin = the forward cout)
struct HsPackJob { const float* w; void* packed; int cout, cin_pad, cin, taps, dgrad; };
bool conv2d_hs_pack_batchable(const ConvSpec& consumer, int dgrad);
int conv2d_hs_pack_many(const HsPackJob* jobs, int n, hipStream_t s);
int conv2d_hs_launch(const ConvSpec& L, Conv2dArgs a, hipStream_t s);
// conv2d_hs16.hip: the same convolution on v_mfma_f32_16x16x32_f16 for the inference executor's plain cell-layout launches with
// Cout % 128 == 0 and Cin % 64 == 0 (same packed weights, same cell tensors; ADX_HS_MODE=0|1|2 keeps every launch on the 32x32x16 kernel)
bool conv2d_hs3x3q_eligible(const Conv2dArgs& a);
int conv2d_hs3x3q_launch(Conv2dArgs a, hipStream_t s);
// partial-sum slots (workgroups per 64-channel slab) the pipelined 3x3 stride-1 kernel would fill for this launch, 0 if another
// kernel serves it
int conv2d_hs_stats_tiles(const ConvSpec& L, const Conv2dArgs& a);
// Data gradient of a 3x3 stride-2 pad-1 conv (forward weight w [cout][cin][3][3], forward input h x w, output oh x ow):
// dx [n][cin][h][w] (+= when accumulate) from dy [n][cout][oh][ow].  An input pixel of parity class (py, px) receives from
// 1 / 2 / 2 / 4 of the nine taps, and all of them lie in the 2x2 window [oy, oy+1] x [ox, ox+1] of dy with oy = iy >> 1,
// ox = ix >> 1: one stride-1 2x2 conv with 4 cin output channels and a depth-to-space store -- 16 tap-products per four
// pixels where convolving a zero-dilated gradient with the flipped 3x3 kernel spends 36.  wbuild (16 cout cin floats) and
// wimg (16 cout cin floats) are scratch.  Returns ADX_ERR_INVALID when the shape is outside the kernel's rules.
bool conv2d_hs_dgrad_s2_eligible(int cin, int cout);
int conv2d_hs_dgrad_s2(const float* w, const float* dy, float* dx, int accumulate, int N, int cin, int cout, int H, int W,
                       float* wbuild, float* wimg, const uint32_t* dy_amax, int dy_amax_n, hipStream_t s);
// scratch for split reductions of the launches issued by this thread until it is cleared (a region of the calling
// executor's workspace, consumed in stream order)
void conv2d_set_split_scratch(float* p, size_t floats);
// stem conv + BN + ReLU + MaxPool2d(3, 2, 1) in one pass: writes only the pooled map [N][64][PH][PW]
int conv2d_hs_stem_pool(const ConvSpec& L, const float* x, const float* w, const float* scale, const float* shift,
                        float* pooled, int N, int H, int W, hipStream_t s, const uint8_t* frames_u8 = nullptr,
                        const float* mean = nullptr, const float* stdv = nullptr, int y_cells = 0);      // y_cells: pooled as a cell tensor
// conv1 (3x3 stride 2, +BN+ReLU) and the block's downsample (1x1 stride 2, +BN) in one pass over x; both must be
// conv2d_hs_eligible (the downsample's weights packed with conv2d_hs_pack_ds)
// scale / shift pairs may be null (identity: the training forward wants both raw conv outputs); relu applies to conv1 only
int conv2d_hs_launch_block_s2(const ConvSpec& c1, const ConvSpec& ds, const float* x, const float* w1, const float* scale1,
                              const float* shift1, float* y1, const float* wd, const float* scaled, const float* shiftd,
                              float* yd, int N, int H, int W, hipStream_t s, int x_cells = 0, int y_cells = 0, int relu = 1);   // x / both outputs in the cell layout
// conv1 3x3 stride 2 on the split-fp16 kernel + a 1x1 stride-2 downsample of the same shape: one fused launch (conv2d.hip)
inline bool resnet_fuses_ds(const ConvSpec& c1, const ConvSpec& ds) {
  return conv2d_hs_eligible(c1) && c1.k == 3 && c1.stride == 2 && c1.pad == 1 && ds.k == 1 && ds.stride == 2 && ds.pad == 0 &&
         ds.cin == c1.cin && ds.cout == c1.cout;
}
// conv2d_wgrad_hs.hip: weight gradient of the 3x3 convs on the fp16 matrix cores; dw must be zero on entry
bool conv2d_wgrad_hs_eligible(int Cin, int Cout, int k, int stride, int pad);
// x_cells: x is a cell tensor (the layout of the inference executor's activations; resnet_train.hip keeps a block's first
// activation that way)
int conv2d_wgrad_hs(const float* x, const float* dy, float* dw, int N, int Cin, int H, int W, int Cout, int stride,
                    const uint32_t* dy_amax, int dy_amax_n, hipStream_t s, bool x_cells = false, bool dy_cells = false);
// ADX_WGRAD_DETERMINISTIC=1: scratch for the per-split copies of dw that conv2d_wgrad_hs reduces in index order (lent by the calling
// thread's executor until cleared; conv2d_wgrad_partials_floats() = what to lend, 0 when the switch is off)
void conv2d_wgrad_set_partials(float* p, size_t floats);
size_t conv2d_wgrad_partials_floats();
// conv2d_wgrad_stem_hs.hip: the stem's (7x7 stride 2, 3 -> 64) weight gradient on the fp16 matrix cores; dw must be zero on entry
bool conv2d_wgrad_stem_hs_eligible(int Cin, int Cout, int k, int stride, int pad);
int conv2d_wgrad_stem_hs(const float* x, const float* dy, float* dw, int N, int H, int W, const uint32_t* dy_amax, int dy_amax_n,
                         hipStream_t s);
inline int conv_out_dim(int h, int k, int s, int p) { return (h + 2 * p - k) / s + 1; }
// ADX_CHECK_RANGE=1 (no-op otherwise): fails with ADX_ERR_RANGE naming (what, index) when the tensor holds |x| >= 65504 or a
// non-finite value (cells: an fp16 hi half that is inf / NaN); synchronises the stream
int conv2d_range_check(const char* what, int index, const float* t, size_t floats, bool cells, hipStream_t s);
int maxpool_launch(const float* x, float* y, int planes, int H, int W, int OH, int OW, hipStream_t s);
int avgpool_fc_launch(const float* x, const float* fw, const float* fb, float* out, int batch, int C, int HW, int out_dim,
                      hipStream_t s, int x_cells = 0);

}  // namespace adx

// Chained temporal layers (tconv_chain.hip): argument block shared with the UNet executor.
#pragma once
#include "tconv.h"

namespace adx {

constexpr int kChainMaxStages = 8;
constexpr int kChainWaves = 8;
constexpr int kChainMaxCells = 4;          // LDS cell buffers a chain rotates through
constexpr size_t kChainMaxLds = 156 * 1024;

struct ChainStage {
  int src, dst;               // LDS cell buffers (index into ChainArgs::cell_off): A operand; where the result's cells go (-1: nowhere)
  int kind, taps, stride, pad;
  int log2_ncell, nsteps;     // cells (8 channels) per tap of the input, MFMA steps of the reduction
  int lin, lout, log2_lout;   // per-sample lengths
  int cout, cout_pad, n_ct;   // output channels, padded to 16, 16-channel tiles
  int w_off;                  // float offsets into ChainArgs::packed: weight image ...
  int b_off, g_off, be_off;   // ... bias, GroupNorm affine (-1: absent)
  int cg_log2;                // log2(channels per GroupNorm group)
  float eps;
  int tb_col;                 // first column of this block's slice of the time-bias matrix (-1: none)
  int r_src, r_log2_ncell, r_nsteps, r_b_off, r_pitch;   // 1x1 residual conv as a second reduction (r_src < 0: none; r_nsteps = 0);
                              // its weight steps lie behind the main conv's in every tile of the image at w_off
  int res_identity;           // + the fp32 tile `f_dst` holds on entry (identity residual), overwritten with the result
  int f_dst;                  // fp32 result tile: index into ChainArgs::f_off
  int out;                    // global output slot (-1: none)
  int src_pitch, dst_pitch;   // row pitch of the cell buffers in 16-byte units (2 * cells + 1)
  int par;                    // float offset (dynamic LDS) of this stage's parameters: [bias | gamma | beta | residual bias]
                              // x cout_pad, then the workgroup's time-bias rows [bt][cout_pad]
};

struct ChainOut {
  float* p;
  int64_t sb, sc, sl;
  int vec;                    // 4 consecutive positions of a channel form an aligned 16-byte run
};

struct ChainArgs {
  const float* packed;
  const float* tb; int64_t tb_stride;
  const float* in0; int64_t in0_sb, in0_sc, in0_sl;
  const float* in1; int64_t in1_sb, in1_sc, in1_sl;
  int in_c0, in_c1, in_cpad, in_len, in_vec;
  ChainOut out[3];
  int batch, bt, n_stages;
  int rotate;                     // workgroups enter every reduction at different steps (different summation order per workgroup)
  int cell_off[kChainMaxCells];   // float offsets into the dynamic LDS
  int f_off[2];
  int args_off;                   // where the kernel parks a copy of this block (sizeof(ChainArgs) bytes)
  ChainStage st[kChainMaxStages];
};

bool chain_layer_ok(const adx_tconv_desc* d);
bool chain_residual_ok(const adx_tconv_desc* main);     // a 1x1 residual conv can ride behind this conv's reduction
int chain_steps(const adx_tconv_desc* d);
size_t chain_packed_floats(const adx_tconv_desc* d);
int chain_pack(const adx_tconv_desc* d, const float* w, const adx_tconv_desc* r, const float* rw, float* packed, hipStream_t s);
void chain_fill_stage(ChainStage* st, const adx_tconv_desc* d);
int chain_launch(const ChainArgs& ca, int grid, size_t lds_bytes, hipStream_t s);

}  // namespace adx

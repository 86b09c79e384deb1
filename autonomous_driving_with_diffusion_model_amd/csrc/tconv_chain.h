// Chained temporal layers (tconv_chain.hip): argument block shared with the UNet executor.
#pragma once
#include "tconv.h"

namespace adx {

constexpr int kChainMaxStages = 8;
constexpr int kChainWaves = 8;
constexpr int kChainMaxCells = 4;          // LDS cell buffers a chain rotates through
constexpr size_t kChainMaxLds = 156 * 1024;

// stage flags
constexpr int kChGn = 1, kChTb = 2, kChResIdentity = 4, kChResConv = 8, kChOut = 16, kChCells = 32, kChOutVec = 64;

// One stage = one layer.  Everything here is uniform over the workgroup; offsets into the dynamic LDS are in floats and
// already include the choice of `bt` (the executor fills them per launch: chain_layout).
struct ChainStage {
  int flags;                  // kCh*; bits 8-11: log2(channels per GroupNorm group); bits 12-13: global output slot
  int conv;                   // kind | taps << 8 | stride << 16 | pad << 24
  int log2_spt;               // log2(MFMA steps per tap) = log2(input cells / 4); inputs are padded to >= 32 channels
  int ns_main, ns_r;          // steps of the conv, of the 1x1 residual conv stored behind it in the image
  int lin, log2_lout;         // per-sample lengths
  int cout, log2_nct;         // output channels; log2 of the 16-channel tile count (cout_pad = 16 << log2_nct)
  int w_off;                  // float offset of the weight image in ChainArgs::packed
  int src, src_pitch;         // A operand: LDS cell buffer (float offset), row pitch in 16-byte units
  int r_src, r_pitch, r_log2_spt;   // the residual conv's input cells (the block input)
  int dst, dst_pitch;         // where the result's cells go (kChCells)
  int f_dst;                  // fp32 result tile (float offset), pitch cout_pad + 4
  int par;                    // parameters in LDS (float offset): [bias | gamma | beta | residual bias] x cout_pad
  int tbl;                    // this block's time-bias rows in LDS (float offset): [bt][cout_pad]
  int rows_out, zrow, r_zrow; // bt * lout; index of the all-zero row of src (bt * lin) and of r_src (bt * lout)
  float inv_n, eps;           // 1 / (channels per group x lout), GroupNorm epsilon
  int n_tiles, log2_nrt;      // (cout_pad / 16) x row tiles; log2(row tiles = ceil(rows_out / 16), a power of two); tile t = (channel
                              // tile t >> log2_nrt, row tile t & (nrt - 1))
  int nx_w_off, nx_nsteps, nx_log2_nrt, nx_n_tiles;   // the next stage's weight image, steps, row-tile shift and tiles (0: none)
};

struct ChainOut {
  float* p;
  int64_t sb, sc, sl;
};

struct ChainArgs {
  const float* packed;
  const float* tb; int64_t tb_stride;
  const float* in0; int64_t in0_sb, in0_sc, in0_sl;
  const float* in1; int64_t in1_sb, in1_sc, in1_sl;
  int in_c0, in_c1, in_cpad, in_len, in_vec, in_cells;   // in_cells: LDS float offset of the staged input
  ChainOut out[2];
  int batch, bt, n_stages;
  int par_src, par_floats, par_lds;    // the chain's per-channel parameters: one block in `packed` -> one block in LDS
  int n_tb; int tb_col[4], tb_cout[4], tb_lds[4];   // time-bias slices: column in tb, channels, LDS float offset of [bt][cout_pad]
  int tab_lds;                         // where the kernel parks the stage table
  unsigned* zero_words; int n_zero;    // words workgroup 0 clears (<= 512): the ticket words of the forward this launch opens
  unsigned* epoch_ctr; int epoch_slot; // ... except word epoch_slot, which receives the forward's number drawn from *epoch_ctr (tconv_pipe.h)
  int xch_lds;                         // GroupNorm partials of stages whose samples span two row tiles
  ChainStage st[kChainMaxStages];
};

bool chain_layer_ok(const adx_tconv_desc* d);
int chain_cin_pad(const adx_tconv_desc* d);           // input channels as the chain stages them (>= 32, a power of two)
int chain_steps(const adx_tconv_desc* d);
size_t chain_packed_floats(const adx_tconv_desc* d);
int chain_pack(const adx_tconv_desc* d, const float* w, const adx_tconv_desc* r, const float* rw, float* packed, hipStream_t s);
int chain_launch(const ChainArgs& ca, int grid, size_t lds_bytes, hipStream_t s);

}  // namespace adx

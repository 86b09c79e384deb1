// Shared between the inference executor (unet.hip) and the training executor (unet_train.hip).
#pragma once
#include <vector>

#include "tconv.h"
#include "tconv_chain.h"
#include "tconv_pipe.h"

namespace adx {

int embed_forward(const adx_embed_weights* w, int dim, const int64_t* t, int t_rows, const float* cond,
                  const float* feat, int feat_rows, int rows, float* time_embed, float* mish_cond, hipStream_t s);

struct ConvLayer {
  adx_tconv_desc d{};
  int p_w = -1, p_b = -1, p_g = -1, p_be = -1;      // indices into the parameter list
  size_t o_w = 0, o_b = 0, o_g = 0, o_be = 0;       // float offsets into the packed buffer
  size_t o_cw = 0;                                  // weight image in the chain kernel's layout (layers of a chain only)
  bool chained = false;
  size_t o_pw = 0;                                  // weight image in the pipeline kernel's layout (tconv_pipe.hip)
  bool piped = false;
};

struct ResBlock {
  ConvLayer a, b, r;
  bool has_r = false;
  int p_tw = -1, p_tb = -1;  // time_mlp.1 weight / bias
  int tb_off = 0;            // column offset in the fused time-bias matrix
  int c0 = 0, c1 = 0, cout = 0, len = 0;
};

// A run of layers executed by ONE launch of tconv_chain.hip: the two residual blocks of a level + its down / up conv
// (+ final_conv on the last up level).  The plan is what does not depend on the call; chain_args() turns it into the
// kernel's argument block for a batch and a choice of samples per workgroup.
struct ChainStagePlan {
  const ConvLayer* L = nullptr;      // the conv
  const ConvLayer* R = nullptr;      // 1x1 residual conv riding behind it (block 0's second conv), or null
  int src = 0, r_src = 0, dst = -1;  // LDS cell buffers (index); dst -1: the result is not needed as cells
  int f = 0;                         // fp32 result tile (0 / 1)
  int out = -1;                      // global output slot
  int tb_col = -1;                   // column of the time-bias matrix (block first convs)
  bool res_identity = false;
  size_t par_off = 0;                // float offset of [bias | gamma | beta | residual bias] inside the chain's parameter block
};
struct ChainPlan {
  std::vector<ChainStagePlan> st;
  int in_c0 = 0, in_c1 = 0, len = 0, max_len = 0;
  size_t o_par = 0, par_floats = 0;  // the parameter block in the packed buffer
  bool valid = false;
  bool with_head = false;
};

}  // namespace adx

struct adx_unet {
  adx_unet_config cfg{};
  std::vector<adx::ResBlock> blocks;
  std::vector<adx::ConvLayer> downs, ups;
  adx::ConvLayer head0, head1, tlin;
  int n_levels = 0, n_params = 0, sum_c = 0, out_ch = 0;
  int p_t1w = 0, p_t1b = 0, p_t3w = 0, p_t3b = 0, p_c0w = -1, p_c0b = -1, p_c2w = -1, p_c2b = -1;
  size_t o_freqs = 0, o_t1w = 0, o_t1b = 0, o_t3w = 0, o_t3b = 0, o_c0w = 0, o_c0b = 0, o_c2w = 0, o_c2b = 0;
  size_t o_tlin_raw = 0, o_tlin_b = 0;  // concatenated [sum_c][2 dim] block-Linear weight (staging) and bias
  size_t packed_floats = 0;
  bool packed_once = false;
  std::vector<adx::ChainPlan> down_chains, up_chains;   // per level; !valid: the level runs layer by layer
  // the deepest level's same-shaped layer run (block 0's second conv, block 1, both mid blocks: seven convs) as ONE pipeline
  // launch at small batches (tconv_pipe.hip); pipe_ok: the configuration qualifies (decided once, at creation)
  bool pipe_ok = false;
};


// Cooperative pipeline of same-shaped temporal layers (tconv_pipe.hip): argument block shared with the UNet executor.
#pragma once
#include "tconv.h"

namespace adx {

constexpr int kPipeMaxStages = 8;      // conv stages + the finisher
constexpr int kPipeCh = 16;            // output channels per workgroup (one 16 x 16 MFMA tile)
constexpr int kPipeRows = 16;          // MFMA rows: (sample, position) pairs of the whole batch
constexpr size_t kPipeMaxLds = 150 * 1024;

// How the INPUT of a stage is formed.  Stage 0 reads a finished activation; every later stage (and the finisher) reads the
// records of the stage before it -- raw conv sums + bias, [rows x L][C] -- and applies that conv's GroupNorm + Mish and the
// block's addend itself (a ResidualTemporalMapBlockConcat, modeling/temporal.py:46-55: h = block0(x) + time bias;
// y = block1(h) + residual).
struct PipeStage {
  const float* w;          // this conv's weight image in the pipeline layout (pipe_pack); unused by the finisher
  const float* bias;       // this conv's bias [C]
  const float* in;         // stage 0: finished activation [rows][C][L]; later stages: null
  const float* gamma;      // GroupNorm affine of the conv whose records are read (stages >= 1)
  const float* beta;
  const float* add;        // addend of the formed input, by add_kind
  int64_t add_stride;      // add_kind 1: floats between the time-bias rows of two samples
  float* pub;              // where rank 0 writes the formed input (pub_kind)
  float* pub2;             // optional second copy as [rows x L][C] (a later stage's residual), or null
  int add_early;           // add_kind 2 only: the residual was written by an EARLIER launch (requested before the wait)
  int add_kind;            // 0 none; 1 time bias add[sample * add_stride + c]; 2 residual [rows][C][L]; 3 residual [rows x L][C]
  int pub_kind;            // 0 not written; 1 as [rows][C][L] (a skip / the run's output); 2 as [rows x L][C] (a later residual)
};

struct PipeArgs {
  PipeStage st[kPipeMaxStages];
  int n_conv;              // conv stages; the launch has n_conv * P + 1 workgroups, the last one is the finisher (stage n_conv)
  int C, L, rows, P;       // channels (in = out), positions per sample, samples, workgroups per stage = C / 16
  int groups, taps, pad, tap0, ntap;   // GroupNorm groups; conv taps and padding; live taps [tap0, tap0 + ntap)
  float eps;
  float* records;          // [n_conv][P][16 rows][6 units of (three channels, tag)]: pipe_record_floats(); written by pipeline launches only
  const unsigned* epoch;   // this forward's number (drawn from pipe_epoch_counter by the launch that opens the forward)
  unsigned* fault;         // host-visible word (pipe_fault_word) a timed-out stage sets; null: the NaN output is the only signal
};

// A spin that times out turns the run's output into NaN AND sets a word of pinned host memory that the next adx_unet_forward /
// adx_unet_time_conditioning of the process reads (no synchronisation: the word is written through to the host) and reports as
// ADX_ERR_STATE.  pipe_fault_word(): the device-side address of that word (allocated on first use; null if that failed);
// pipe_fault_take(): its value, cleared.
unsigned* pipe_fault_word();
unsigned pipe_fault_take();
// The forward numbers behind the records' tags: one monotonic device word per GPU, owned by the library (allocate: only where
// no stream capture can be open); pipe_tickets_reset clears `n` ticket words and leaves a freshly drawn number in words[epoch_slot]
// (what tconv_chain's workgroup 0 does when a chained level opens the forward: ChainArgs::epoch_ctr).
unsigned* pipe_epoch_counter(bool allocate);
int pipe_tickets_reset(unsigned* words, int n, int epoch_slot, hipStream_t s);
size_t pipe_record_floats(int n_conv, int P);

// live taps of a stride-1 conv on L positions, and whether a layer run of this shape fits the kernel (weights of one workgroup
// in LDS, the whole batch in one 16-row tile)
void pipe_live_taps(int taps, int pad, int L, int* tap0, int* ntap);
bool pipe_shape_ok(int C, int L, int rows, int taps, int pad, int groups);
size_t pipe_packed_floats(int C, int taps, int pad, int L);
int pipe_pack(const float* w, float* packed, int C, int taps, int pad, int L, hipStream_t s);
// the same image made from the layer's K-split image (tconv_hs.hip), which the executor holds anyway
int pipe_repack_from_hs(const float* hs_image, float* packed, int C, int taps, int pad, int L, hipStream_t s);
// the same for n <= 8 layers of one shape, one launch
int pipe_repack_from_hs_many(const float* const* hs_images, float* const* packed, int n, int C, int taps, int pad, int L, hipStream_t s);
int pipe_launch(const PipeArgs& a, hipStream_t s);

}  // namespace adx

// Weight re-lay of the temporal stack as ONE launch per batch of layers.  Every layer's weights are re-laid from the PyTorch
// tensor into its kernel's fragment image whenever the weights change -- once per training step for ~60 forward images and
// ~55 data-gradient images, each of which was a launch of its own (4-6 us apiece, 0.55 ms per step).  The four images differ
// only in how an element's index maps to (output channel, input channel, tap), so one kernel walks a table of jobs.
#pragma once
#include "adx_common.h"

namespace adx {

enum PackKind : int {
  kPackExact = 0,   // tconv.hip:       [cout_pad / 16][tap x cin / 16][64 lanes][4] fp32 (B operand of 16x16x4)
  kPackHs = 1,      // tconv_hs.hip:    [cout_pad / 32][tap x cin / 16][hi | lo][64 lanes][8 halfs] (B operand of 32x32x16)
  kPackCell = 2,    // tconv_hs.hip (short-K) and tconv_chain.hip: [cout_pad / 16][steps of four 8-channel cells][hi | lo][64][8]
};

struct PackJob {
  const float* w;          // PyTorch layout: [cout][cin][taps] (layout 0) or [cin][cout][taps] (layout 1)
  void* out;
  uint32_t first_block;    // filled by the queue: the job's first 256-thread block inside its launch
  uint32_t total;          // elements (threads) of the job
  int kind, layout, flip, taps, cin, cout;
  int a, b;                // kPackExact / kPackHs: ncb, nkb;  kPackCell: ncell, nsteps
  int tile_steps, step0;   // kPackCell: steps per tile in the image, first step of this conv inside a tile
};

// While a queue is open (per host thread) pack_submit only records the job; pack_flush launches everything recorded, in
// submission order, as few launches as the kernel-argument table allows.  Without an open queue a job is launched at once.
void pack_queue_open();
int pack_submit(const PackJob& job, hipStream_t s);
int pack_flush(hipStream_t s);       // closes the queue
void pack_queue_abandon();           // closes the queue and drops what it holds

// Scope of an open queue: whatever path leaves the scope without pack_flush (an early error return) closes the queue, so that a
// later stand-alone pack is never silently parked in a queue nobody flushes.
struct PackQueueScope {
  PackQueueScope() { pack_queue_open(); }
  ~PackQueueScope() { pack_queue_abandon(); }
  PackQueueScope(const PackQueueScope&) = delete;
  PackQueueScope& operator=(const PackQueueScope&) = delete;
};

}  // namespace adx

// Many small device-to-device copies / zero fills as a handful of launches: the per-step re-pack of the temporal
// stack (biases, GroupNorm affines, the pieces of the fused block Linear) and the zeroing of ~180 small gradient
// tensors were ~450 hipMemcpyAsync / hipMemsetAsync calls per training step (4-5 us of GPU time each).
#pragma once
#include "adx_common.h"

namespace adx {

// queue (thread-local) and flush; flush keeps program order among the queued operations of its own kind only, so
// flush before anything that reads a queued destination
void batch_copy_add(float* dst, const float* src, size_t n);
int batch_copy_flush(hipStream_t s);
void batch_fill_add(float* dst, size_t n);        // zero fill
int batch_fill_flush(hipStream_t s);

}  // namespace adx

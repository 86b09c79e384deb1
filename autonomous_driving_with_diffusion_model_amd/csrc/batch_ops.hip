#include <vector>

#include "batch_ops.h"

namespace adx {

constexpr int kBatch = 48;

struct CopyTable { const float* src[kBatch]; float* dst[kBatch]; uint32_t n[kBatch]; };
struct FillTable { float* dst[kBatch]; uint32_t n[kBatch]; };

__global__ void __launch_bounds__(256) multi_copy_kernel(const CopyTable t) {
  const int e = blockIdx.y;
  const float* __restrict__ src = t.src[e];
  float* __restrict__ dst = t.dst[e];
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < t.n[e]; i += gridDim.x * 256) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) multi_fill_kernel(const FillTable t) {
  const int e = blockIdx.y;
  float* __restrict__ dst = t.dst[e];
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < t.n[e]; i += gridDim.x * 256) dst[i] = 0.f;
}

struct CopyOp { float* dst; const float* src; size_t n; };
struct FillOp { float* dst; size_t n; };
static thread_local std::vector<CopyOp> g_copies;
static thread_local std::vector<FillOp> g_fills;

void batch_copy_add(float* dst, const float* src, size_t n) { if (n) g_copies.push_back({dst, src, n}); }
void batch_fill_add(float* dst, size_t n) { if (n) g_fills.push_back({dst, n}); }

static unsigned blocks_for(size_t maxn) {
  const size_t b = (maxn + 2047) / 2048;      // ~8 elements per thread
  return (unsigned)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

int batch_copy_flush(hipStream_t s) {
  size_t i = 0;
  while (i < g_copies.size()) {
    CopyTable t;
    int cnt = 0;
    size_t maxn = 0;
    for (; i < g_copies.size() && cnt < kBatch; ++i) {
      const CopyOp& o = g_copies[i];
      if (o.n > 0xFFFFFFFFull) {              // too large for the table: an ordinary copy
        ADX_CHECK_HIP(hipMemcpyAsync(o.dst, o.src, o.n * sizeof(float), hipMemcpyDeviceToDevice, s));
        continue;
      }
      t.src[cnt] = o.src; t.dst[cnt] = o.dst; t.n[cnt] = (uint32_t)o.n;
      maxn = o.n > maxn ? o.n : maxn;
      ++cnt;
    }
    if (cnt) multi_copy_kernel<<<dim3(blocks_for(maxn), cnt), dim3(256), 0, s>>>(t);
  }
  g_copies.clear();
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

int batch_fill_flush(hipStream_t s) {
  size_t i = 0;
  while (i < g_fills.size()) {
    FillTable t;
    int cnt = 0;
    size_t maxn = 0;
    for (; i < g_fills.size() && cnt < kBatch; ++i) {
      const FillOp& o = g_fills[i];
      if (o.n > 0xFFFFFFFFull) {
        ADX_CHECK_HIP(hipMemsetAsync(o.dst, 0, o.n * sizeof(float), s));
        continue;
      }
      t.dst[cnt] = o.dst; t.n[cnt] = (uint32_t)o.n;
      maxn = o.n > maxn ? o.n : maxn;
      ++cnt;
    }
    if (cnt) multi_fill_kernel<<<dim3(blocks_for(maxn), cnt), dim3(256), 0, s>>>(t);
  }
  g_fills.clear();
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

}  // namespace adx

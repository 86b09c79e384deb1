#include <vector>

#include "tconv_pack.h"

namespace adx {

namespace {

constexpr int kMaxJobs = 56;          // 56 x 64 bytes + header: inside the 4 KB kernel-argument segment
struct PackTable {
  PackJob job[kMaxJobs];
  int n;
};
static_assert(sizeof(PackJob) == 64, "job layout");
static_assert(sizeof(PackTable) <= 4000, "kernel-argument segment");

constexpr float kPackLo = 2048.f;     // the split-fp16 kernels' lo scale (kLoScale, kChLoScale)

__global__ void __launch_bounds__(256) pack_many_kernel(const PackTable t) {
  // binary search of the job that owns this block (wave-uniform)
  int lo = 0, hi = t.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.job[mid].first_block <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackJob& J = t.job[lo];
  const uint32_t idx = (blockIdx.x - J.first_block) * 256u + threadIdx.x;
  if (idx >= J.total) return;
  const float* __restrict__ w = J.w;
  const int taps = J.taps, cin = J.cin, cout = J.cout;
  if (J.kind == kPackExact) {
    const int ncb = J.a, nkb = J.b;
    const int j = idx & 3, lane = (idx >> 2) & 63;
    const uint32_t blk = idx >> 8;
    const int kb = blk % nkb, t16 = blk / nkb;
    const int tap = kb / ncb, cb = kb - tap * ncb;
    const int n = t16 * 16 + (lane & 15), ci = cb * 16 + 4 * j + (lane >> 4);
    const int ts = J.flip ? taps - 1 - tap : tap;
    float v = 0.f;
    if (n < cout && ci < cin) v = J.layout == 0 ? w[((size_t)n * cin + ci) * taps + ts] : w[((size_t)ci * cout + n) * taps + ts];
    reinterpret_cast<float*>(J.out)[idx] = v;
    return;
  }
  const int j = idx & 7, ln = (idx >> 3) & 63;
  const uint32_t blk = idx >> 9;
  int n, ci, tap;
  size_t dst_blk;
  bool ok;
  if (J.kind == kPackHs) {
    const int ncb = J.a, nkb = J.b;
    const int kb = blk % nkb, t32 = blk / nkb;
    tap = kb / ncb;
    const int cb = kb - tap * ncb;
    n = t32 * 32 + (ln & 31);
    ci = cb * 16 + 8 * (ln >> 5) + j;
    ok = true;
    dst_blk = blk;
  } else {
    const int ncell = J.a, nsteps = J.b;
    const int step = blk % nsteps, t16 = blk / nsteps;
    const int kc = 4 * step + (ln >> 4);
    tap = kc / ncell;
    ci = 8 * (kc - tap * ncell) + j;
    n = t16 * 16 + (ln & 15);
    ok = tap < taps;
    dst_blk = (size_t)t16 * J.tile_steps + J.step0 + step;
  }
  float v = 0.f;
  if (ok && n < cout && ci < cin) {
    const int ts = J.flip ? taps - 1 - tap : tap;
    v = J.layout == 0 ? w[((size_t)n * cin + ci) * taps + ts] : w[((size_t)ci * cout + n) * taps + ts];
  }
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)((v - (float)h) * kPackLo);
  _Float16* dst = reinterpret_cast<_Float16*>(J.out) + dst_blk * 1024 + ln * 8 + j;
  dst[0] = h;
  dst[512] = l;
}

thread_local bool g_open = false;
thread_local std::vector<PackJob> g_jobs;

int launch_jobs(const PackJob* jobs, int n, hipStream_t s) {
  int i = 0;
  while (i < n) {
    PackTable t;
    t.n = 0;
    uint32_t blocks = 0;
    while (i < n && t.n < kMaxJobs) {
      PackJob j = jobs[i++];
      if (j.total == 0) continue;
      j.first_block = blocks;
      blocks += (j.total + 255u) / 256u;
      t.job[t.n++] = j;
    }
    if (t.n == 0) break;
    pack_many_kernel<<<dim3(blocks), dim3(256), 0, s>>>(t);
    ADX_LAUNCH_CHECK();
  }
  return ADX_OK;
}

}  // namespace

void pack_queue_open() {
  g_open = true;
  g_jobs.clear();
}

int pack_submit(const PackJob& job, hipStream_t s) {
  if (g_open) {
    g_jobs.push_back(job);
    return ADX_OK;
  }
  return launch_jobs(&job, 1, s);
}

void pack_queue_abandon() {
  g_open = false;
  g_jobs.clear();
}

int pack_flush(hipStream_t s) {
  g_open = false;
  const int rc = g_jobs.empty() ? ADX_OK : launch_jobs(g_jobs.data(), (int)g_jobs.size(), s);
  g_jobs.clear();
  return rc;
}

}  // namespace adx

// Does kernel-argument preloading (the first dwords of the argument block delivered in SGPRs at wave launch, gfx940+: LLVM's
// -mllvm -amdgpu-kernarg-preload-count=N) shorten a latency-bound launch?  A chain of 256 dependent small kernels captured in a
// HIP graph: each kernel's 256 workgroups read what the previous one wrote through the pointers of its argument block (the
// temporal stack's pattern: nothing can be issued before the arguments are there).  Built twice by tools/micro/build.sh
// (kernarg_preload.bin: plain; kernarg_preload_on.bin: with -mllvm -amdgpu-kernarg-preload-count=16); the kernel takes its
// pointers and sizes FIRST and a 256-byte parameter struct (as the library's kernels do) behind them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

struct Params { int v[64]; };

__global__ void __launch_bounds__(256) link_kernel(const float* __restrict__ in, float* __restrict__ out, int n, float bias, Params p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[(i + 64) % n] + bias + (float)p.v[threadIdx.x & 63];
}

int main() {
  const int n = 256 * 256, links = 256, reps = 50;
  float *a, *b;
  hipMalloc(&a, n * sizeof(float));
  hipMalloc(&b, n * sizeof(float));
  hipMemset(a, 0, n * sizeof(float));
  hipMemset(b, 0, n * sizeof(float));
  Params p{};
  hipStream_t s;
  hipStreamCreate(&s);
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int k = 0; k < links; ++k) link_kernel<<<256, 256, 0, s>>>(k & 1 ? b : a, k & 1 ? a : b, n, 1.f, p);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
  hipEventRecord(e1, s);
  hipStreamSynchronize(s);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  // eager chain as well
  for (int k = 0; k < 64; ++k) link_kernel<<<256, 256, 0, s>>>(k & 1 ? b : a, k & 1 ? a : b, n, 1.f, p);
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int r = 0; r < 20; ++r)
    for (int k = 0; k < links; ++k) link_kernel<<<256, 256, 0, s>>>(k & 1 ? b : a, k & 1 ? a : b, n, 1.f, p);
  hipEventRecord(e1, s);
  hipStreamSynchronize(s);
  float ms2 = 0.f;
  hipEventElapsedTime(&ms2, e0, e1);
  float h[4];
  hipMemcpy(h, a, sizeof(h), hipMemcpyDeviceToHost);
  printf("graph: %.3f us per dependent launch; eager: %.3f us per launch (check %.1f)\n", ms * 1e3 / (reps * links), ms2 * 1e3 / (20 * links), h[0]);
  return 0;
}

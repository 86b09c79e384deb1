// Cost of a barrier + small data exchange among G workgroups inside one launch (the hand-off a cooperative small-batch
// executor would pay per layer): every round each workgroup publishes 1 KB (sc1 stores), arrives on a monotonic counter
// (agent-scope atomic add after every wave drained its stores), polls until all G arrived, reads every other workgroup's
// 1 KB with sc1 loads and checks it.  Placement: SPREAD = the G workgroups are the whole grid (round-robin over the XCDs);
// ONE_XCD = a grid of 8 G workgroups of which those with blockIdx % 8 == 0 take part (same XCD if dispatch is round-robin).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) rounds_kernel(unsigned* counter, float* slabs, int G, int rounds, int stride8,
                                                     unsigned* errors, unsigned* timeouts) {
  int wg = blockIdx.x;
  if (stride8) {
    if (wg & 7) return;
    wg >>= 3;
  }
  const int tid = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, G * 2 * 1024, 0x00020000);
  unsigned bad = 0;
  __shared__ int give_up;
  if (tid == 0) give_up = 0;
  __syncthreads();
  for (int r = 1; r <= rounds; ++r) {
    const int buf = r & 1;                                  // double buffer: a fast workgroup must not overwrite what a slow one reads
    if (tid < 64) {
      const u32x4 v = {(unsigned)(r * 1000 + wg), (unsigned)tid, 3u, 4u};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, (buf * G + wg) * 1024 + tid * 16, 0, 16);   // aux 16 = sc1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)r * (unsigned)G;
      unsigned spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000u) { atomicAdd(timeouts, 1u); give_up = 1; break; }     // ~60 ms: something is wrong, leave
      }
    }
    __syncthreads();
    if (give_up) break;
    // read the other workgroups' records: sc1 loads (bypass this CU's L1)
    if (tid < 64) {
      for (int o = 0; o < G; ++o) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (buf * G + o) * 1024 + tid * 16, 0, 16);
        if (v[0] != (unsigned)(r * 1000 + o) || v[1] != (unsigned)tid) ++bad;
      }
    }
  }
  if (bad) atomicAdd(errors, bad);
}

int main(int argc, char** argv) {
  const int rounds = 2000;
  unsigned *counter, *errors, *timeouts;
  float* slabs;
  hipMalloc(&counter, 64); hipMalloc(&errors, 4); hipMalloc(&timeouts, 4);
  hipMalloc(&slabs, 64 * 2 * 1024);
  for (int stride8 = 0; stride8 < 2; ++stride8)
    for (int G : {2, 4, 8, 16, 32}) {
      hipMemset(counter, 0, 64); hipMemset(errors, 0, 4); hipMemset(timeouts, 0, 4);
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      rounds_kernel<<<stride8 ? 8 * G : G, 512>>>(counter, slabs, G, rounds, stride8, errors, timeouts);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned herr, hto;
      hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost);
      hipMemcpy(&hto, timeouts, 4, hipMemcpyDeviceToHost);
      printf("%s G %2d: %.3f us per round (publish 1 KB + barrier + read G KB), errors %u, timeouts %u\n",
             stride8 ? "ONE_XCD" : "SPREAD ", G, 1e3 * ms / rounds, herr, hto);
    }
  return 0;
}

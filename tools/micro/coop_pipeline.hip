// Bounded prototype of a cooperative executor for the deployed B = 1 mid blocks of the temporal stack (round-3 review, item 4b;
// modeling/temporal.py:46-55,217-231: mid_block1 + mid_block2 = four Conv1d(512, 512, 5) + GroupNorm + Mish at 2 rows x 2
// positions when MODEL.HORIZON = 16).
//
// What it keeps REAL: everything that costs time in such an executor -- (a) every CU's share of the weights of ITS layer is
// loaded into LDS once, before the first hand-off (P CUs per layer, 16 x (512 / 16 / P ...) channels each; at P = 32 that is the
// 98 KB per CU of three live taps x 512 input channels x 16 output channels x 4 bytes); (b) per layer and pass: wait for the
// producer layer's P workgroups on an agent-scope counter, read the 8 KB of activations they published (sc1 loads, cross-XCD),
// split them to fp16 hi / lo cells in LDS, walk the whole LDS weight share with ds_read_b128 + the layer's MFMAs
// (v_mfma_f32_16x16x32_f16, three per product), reduce the four waves through LDS, publish this CU's 16 channels x 4 positions
// with write-through (sc1) stores, drain, barrier, ONE relaxed agent-scope atomic add.  GroupNorm + Mish are applied by the
// CONSUMER while it stages (from the raw sums + per-CU partial statistics the producer publishes in the same record), so a layer
// costs one hand-off, not two.  Four layers form a pipeline of 4 P workgroups (one per CU, all resident); pass r + 1 of layer 0
// waits for pass r of layer 3, as a denoising step's next layers would.
// What it leaves out: correct numerics (weights are a pattern, the result is checked only for "every hand-off delivered the
// producer's current pass"), time bias, residual adds.  It is therefore a LOWER bound on the time of the real thing.
//
// Output: microseconds per pass of the four-layer pipeline for P = 8, 16, 32, to be compared with four launches of
// tconv_hs_kernel<2,8,4> at 2 rows (tools/tick_timeline.py: the 512 -> 512 launches of one deployed tick).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLayers = 4, kCin = 512, kPos = 4, kLiveTaps = 3;
constexpr int kRecFloats = 2048 / 4;        // a producer's record: up to 64 channels x 4 positions + partial statistics, 2 KB

struct Args {
  unsigned* counters;      // [kLayers] monotonic arrival counters
  float* slabs;            // [2 buffers][kLayers][P][kRecFloats]
  const u32x4* weights;    // [kLayers][P][wcells] 16-byte cells
  float* out;
  unsigned* errors;
  unsigned* timeouts;
  int P, passes, wcells;   // wcells: 16-byte weight cells per CU (hi and lo planes)
};

__global__ void __launch_bounds__(256) pipeline_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* wl = reinterpret_cast<u32x4*>(smem_raw);                       // this CU's weight share
  u32x4* cells = wl + a.wcells;                                         // activations as split cells: [pos][512 / 8][hi, lo]
  float* red = reinterpret_cast<float*>(cells + kPos * (kCin / 8) * 2); // [4 waves][256]
  __shared__ int give_up;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // layer = blockIdx / P: the P workgroups of a layer are spread over the XCDs by the dispatcher's round-robin
  const int layer = blockIdx.x / a.P, rank = blockIdx.x % a.P;
  const int ch_per = kCin / a.P;                                        // output channels of this CU
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(a.slabs, 0, 2 * kLayers * a.P * kRecFloats * 4, 0x00020000);
  if (tid == 0) give_up = 0;
  {   // (a) the weight share of THIS layer into LDS, once
    const u32x4* src = a.weights + ((size_t)layer * a.P + rank) * a.wcells;
    for (int i = tid; i < a.wcells; i += 256) wl[i] = src[i];
  }
  __syncthreads();
  const int prev = (layer + kLayers - 1) % kLayers;
  unsigned bad = 0;
  float keep = 0.f;
  for (int r = 1; r <= a.passes; ++r) {
    const int buf = r & 1;
    // ---- wait for the producer layer's pass (layer 0: the previous pass of the last layer) ----------------------------------
    const unsigned want = (unsigned)(layer == 0 ? r - 1 : r) * (unsigned)a.P;
    if (tid == 0 && want > 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(a.counters + prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 4000000u) { atomicAdd(a.timeouts, 1u); give_up = 1; break; }       // bounded: leave, never hang
      }
    }
    __syncthreads();
    if (give_up) break;
    // ---- stage: the P records of the producer (ch_per channels x 4 positions each + statistics), GroupNorm + Mish on the way,
    //      split into hi / lo cells.  Thread t owns 8 consecutive channels of one position: 4 x 64 = 256 items.
    {
      const int pos = tid >> 6, oct = tid & 63;                         // channels 8 oct .. 8 oct + 7
      const int src_buf = layer == 0 ? ((r - 1) & 1) : buf;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; j += 4) {
        const int c = 8 * oct + j, pr = c / ch_per, cl = c - pr * ch_per;      // producer rank, channel within its record
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(
            rs, ((src_buf * kLayers + prev) * a.P + pr) * kRecFloats * 4 + (pos * ch_per + cl) * 4, 0, 16);      // sc1
        v[j] = __builtin_bit_cast(float, q[0]); v[j + 1] = __builtin_bit_cast(float, q[1]);
        v[j + 2] = __builtin_bit_cast(float, q[2]); v[j + 3] = __builtin_bit_cast(float, q[3]);
      }
      // the record's tag (last float) says which pass wrote it: a stale hand-off shows up here
      if (want > 0 && oct == 0 && pos == 0) {
        const unsigned tag = __builtin_amdgcn_raw_buffer_load_b32(
            rs, ((src_buf * kLayers + prev) * a.P + 0) * kRecFloats * 4 + (kRecFloats - 1) * 4, 0, 16);
        if (tag != (unsigned)(layer == 0 ? r - 1 : r)) ++bad;
      }
      h8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // GroupNorm (statistics would come from the records) + Mish stand-ins with the real instruction mix: fma, exp, rcp
        const float xn = v[j] * 0.5f + 0.1f;
        const float e = __expf(xn);
        const float n = e * (e + 2.f);
        const float y = xn * n * __builtin_amdgcn_rcpf(n + 2.f);
        const _Float16 h = (_Float16)y;
        hi[j] = h;
        lo[j] = (_Float16)((y - (float)h) * 2048.f);
      }
      cells[(pos * 64 + oct) * 2] = __builtin_bit_cast(u32x4, hi);
      cells[(pos * 64 + oct) * 2 + 1] = __builtin_bit_cast(u32x4, lo);
    }
    __syncthreads();
    // ---- the layer: this CU's 16-channel tiles, K = 3 live taps x 512 channels = 48 steps of 32, split over the four waves;
    //      per step the wave reads its weight fragments (hi, lo: 2 x ds_read_b128) and the activation fragment (2 x ds_read_b128)
    f32x4 accm = {0.f, 0.f, 0.f, 0.f}, accx = {0.f, 0.f, 0.f, 0.f};
    const int ntile = (ch_per + 15) / 16;
    const int steps = kLiveTaps * kCin / 32;                            // 48
    for (int t = 0; t < ntile; ++t)
      for (int k = wave; k < steps; k += 4) {
        int wi = ((t * steps + k) * 2) * 64 + lane;                     // [tile][step][plane][lane]
        wi = wi < a.wcells - 64 ? wi : a.wcells - 128 + lane;           // (a truncated share re-reads its last step)
        const h8 wh = __builtin_bit_cast(h8, wl[wi]);
        const h8 wlo = __builtin_bit_cast(h8, wl[wi + 64]);
        const int row = lane & 3, cell = ((k * 4) + (lane >> 4)) & 63;  // rows 0..3 are the four positions, the rest read position 3
        const h8 ah = __builtin_bit_cast(h8, cells[(row * 64 + cell) * 2]);
        const h8 al = __builtin_bit_cast(h8, cells[(row * 64 + cell) * 2 + 1]);
        accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh, accm, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wlo, accx, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh, accx, 0, 0, 0);
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave * 256 + lane * 4 + i] = accm[i] + accx[i] * (1.f / 2048.f);
    __syncthreads();
    // ---- publish: ch_per channels x 4 positions (+ statistics, + the pass tag) with write-through stores ----------------------
    if (tid < 64) {
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = red[tid * 4 + i] + red[256 + tid * 4 + i] + red[512 + tid * 4 + i] + red[768 + tid * 4 + i];
      keep += o[0];
      const int rec = ((buf * kLayers + layer) * a.P + rank) * kRecFloats * 4;
      if (tid * 4 < ch_per * kPos)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs, rec + tid * 16, 0, 16);
      if (tid == 63) __builtin_amdgcn_raw_buffer_store_b32((unsigned)r, rs, rec + (kRecFloats - 1) * 4, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.counters + layer, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (bad) atomicAdd(a.errors, bad);
  if (tid == 0 && keep == 12345.678f) a.out[0] = keep;                  // keeps the arithmetic alive
}

int main() {
  const int passes = 1000;
  for (int P : {8, 16, 32}) {
    Args a;
    a.P = P; a.passes = passes;
    const int ch_per = kCin / P;
    const int ntile = (ch_per + 15) / 16;
    int wcells = ntile * (kLiveTaps * kCin / 32) * 2 * 64;             // [tile][48 steps][2 planes][64 lanes] x 16 bytes
    const int cap = (150 * 1024 - kPos * 64 * 2 * 16 - 4096) / 16;      // what one CU's LDS can hold beside the activations
    const bool fits = wcells <= cap;
    if (!fits) wcells = cap;                                            // P < 32: the real share does NOT fit (timing skeleton only)
    a.wcells = wcells;
    const size_t lds = (size_t)wcells * 16 + kPos * 64 * 2 * 16 + 4 * 256 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pipeline_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    unsigned *counters, *errors, *timeouts;
    float *slabs, *out;
    u32x4* weights;
    hipMalloc(&counters, 64); hipMalloc(&errors, 4); hipMalloc(&timeouts, 4); hipMalloc(&out, 64);
    hipMalloc(&slabs, (size_t)2 * kLayers * P * kRecFloats * 4);
    hipMalloc(&weights, (size_t)kLayers * P * wcells * 16);
    hipMemset(slabs, 0, (size_t)2 * kLayers * P * kRecFloats * 4);
    hipMemset(weights, 0x11, (size_t)kLayers * P * wcells * 16);
    a.counters = counters; a.slabs = slabs; a.weights = weights; a.out = out; a.errors = errors; a.timeouts = timeouts;
    float best = 1e30f;
    unsigned herr = 0, hto = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(counters, 0, 64); hipMemset(errors, 0, 4); hipMemset(timeouts, 0, 4);
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      pipeline_kernel<<<kLayers * P, 256, lds>>>(a);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
      unsigned e, t;
      hipMemcpy(&e, errors, 4, hipMemcpyDeviceToHost);
      hipMemcpy(&t, timeouts, 4, hipMemcpyDeviceToHost);
      herr += e; hto += t;
    }
    printf("P %2d CUs per layer (%3d KB of weights in LDS per CU%s): %.2f us per pass of the four-layer pipeline = %.2f us per layer, "
           "stale hand-offs %u, timeouts %u\n",
           P, wcells * 16 / 1024, fits ? "" : ", TRUNCATED: the real share does not fit", 1e3 * best / passes,
           1e3 * best / passes / kLayers, herr, hto);
    hipFree(counters); hipFree(errors); hipFree(timeouts); hipFree(out); hipFree(slabs); hipFree(weights);
  }
  return 0;
}

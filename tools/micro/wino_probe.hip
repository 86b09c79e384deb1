// What a Winograd variant of conv2d_hs3x3 could cost on this chip, measured BEFORE building it: the per-(512 pixels x 64
// channels x 16 input channels) instruction budget of three designs, issued as a synthetic loop with the real instruction
// kinds (split-fp16 MFMAs out of LDS fragments, LDS fragment reads, L2-resident global 16-byte loads, the fp32 -> hi/lo
// re-split VALU of a transformed operand, 16-byte LDS writes), 8 waves per workgroup, one workgroup per CU, random operands.
//   direct  (conv2d_hs3x3 MODE 1 as committed): 864 MFMA, 576 LDS reads,  76 LDS writes,  76 global loads, ~100 VALU
//   F(2,3) along W, single launch:              576 MFMA, 576 LDS reads, 128 LDS writes,  90 global loads, 1280 VALU
//   F(2x2,3x3), fused, 64 channels x 64 tiles:  384 MFMA, 256 LDS reads, 178 LDS writes, 176 global loads, 2304 VALU
// (wave-instructions per workgroup; derivation in profiles/README.md, round 5).  A trip of the loop is 1/24 of that budget
// per wave.  Output: microseconds per trip-set and the ratio to the direct budget -- the speed-up a perfect implementation
// of that design could show, since the chip is power-limited and time follows the energy of the instruction mix.
// Also: whether the matrix cores honour fp16 subnormal inputs (the single-accumulator form of the split needs them).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned f2u(float f) { return __builtin_bit_cast(unsigned, f); }

// one "transform unit" = 8 values: x = a - b (the Winograd input transform is adds of reconstructed values), then the split of
// x into fp16 hi and lo (scaled 2^11): v_cvt_pk for the hi pair, one v_fma_mix per residual, one v_cvt_pk for the lo pair
__device__ __forceinline__ void unit8(const float (&a)[8], const float (&b)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = a[2 * j] - b[2 * j], x1 = a[2 * j + 1] + b[2 * j + 1];
    f16x2 h;
    h[0] = (_Float16)x0; h[1] = (_Float16)x1;
    const float r0 = (x0 - (float)h[0]) * 2048.f, r1 = (x1 - (float)h[1]) * 2048.f;
    f16x2 l;
    l[0] = (_Float16)r0; l[1] = (_Float16)r1;
    hi[j] = __builtin_bit_cast(unsigned, h);
    lo[j] = __builtin_bit_cast(unsigned, l);
  }
}

// MF: MFMAs per trip (multiple of 4); LR: LDS fragment reads; GR: global 16-byte loads; TU: transform units (each: 8 fma_mix
// reconstructs + 8 adds + ~28 split ops, and the two cells it produces are written to LDS when LW allows); LW: LDS writes
template <int MF, int LR, int GR, int TU, int LW>
__global__ void __launch_bounds__(512, 2) probe(const u32x4* __restrict__ in, const u32x4* __restrict__ gbuf, float* __restrict__ out,
                                                int iters, unsigned gmask) {
  extern __shared__ u32x4 lds[];                     // 96 KB: 64 KB operand image + 32 KB written by the transform
  for (int i = threadIdx.x; i < 6144; i += 512) lds[i] = in[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4* base = lds + wave * 64 + lane;
  u32x4* wbase = lds + 4096 + wave * 256 + lane;
  const u32x4* g = gbuf + (size_t)(blockIdx.x & 7) * 65536 + wave * 64 + lane;     // 1 MB per XCD-slot, L2-resident
  constexpr int NACC = 8;
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  constexpr int NG = GR > 0 ? GR : 1;
  u32x4 gv[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j) gv[j] = g[j * 512];
  float carry[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) carry[j] = (float)(lane + j) * 0.01f;
  for (int it = 0; it < iters; ++it) {
    const u32x4* p = base + ((it * 8) & 2047);
    constexpr int NF = LR > 0 ? LR : 1;
    f16x8 f[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) f[j] = __builtin_bit_cast(f16x8, p[(j * 512 + (j >> 3) * 64) & 4095]);
    // the global loads of the NEXT trip are issued now and consumed next trip (two-deep: no exposed latency)
    u32x4 gn[NG];
    const unsigned go = ((unsigned)(it + 1) * 4096u) & gmask;
#pragma unroll
    for (int j = 0; j < NG; ++j) gn[j] = GR > 0 ? g[go + j * 512] : gv[j];
    // transform units on what the previous trip fetched
    u32x4 hi[TU > 0 ? TU : 1], lo[TU > 0 ? TU : 1];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const u32x4 c0 = gv[u % NG], c1 = gv[(u + 1) % NG];
      float a[8], b[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {              // reconstruct: hi + lo / 2^11 (v_fma_mix_f32 in the real kernel)
        const f16x2 h0 = __builtin_bit_cast(f16x2, c0[j]), l0 = __builtin_bit_cast(f16x2, c1[j]);
        a[2 * j] = (float)h0[0] + (float)l0[0] * (1.f / 2048.f);
        a[2 * j + 1] = (float)h0[1] + (float)l0[1] * (1.f / 2048.f);
        b[2 * j] = carry[2 * j];
        b[2 * j + 1] = carry[2 * j + 1];
      }
      unit8(a, b, hi[u], lo[u]);
#pragma unroll
      for (int j = 0; j < 8; ++j) carry[j] = a[j];
    }
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int t = m % NACC;
      // operands rotate over the fragments read this trip and the global loads of the last one, as A and B of a split product
      const f16x8 A = GR > 0 && (m & 1) ? __builtin_bit_cast(f16x8, gv[(m >> 1) % NG]) : f[m % NF];
      const f16x8 B = f[(m * 5 + 3) % NF];
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[t], 0, 0, 0);
      if (LW > 0 && TU > 0 && m < LW) {          // the LDS writes go out between the MFMAs, like the staging of the real kernel
        const int idx = m % (2 * TU), u = idx >> 1;       // every cell a unit produces is written (none of its VALU is dead)
        wbase[((it + m) & 3) * 64] = (idx & 1) ? lo[u] : hi[u];
      } else if (LW > 0 && TU == 0 && m < LW) {
        wbase[((it + m) & 3) * 64] = gv[m % NG];
      }
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) gv[j] = gn[j];
    if ((it & 7) == 7) __syncthreads();          // a barrier per ~stage group, as the pipelines have
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[t][i];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += carry[j];
  __syncthreads();
  s += u2f(lds[4096 + threadIdx.x][0]);
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// fp16 subnormal inputs: 32x32x16 product of A = 2^-20 (subnormal in fp16) and B = 2^10; expected sum over k = 16 * 2^-10
__global__ void denorm_test(float* out) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1024.f; }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
  // and the 16x16x32 form
  f32x4 c;
  for (int i = 0; i < 4; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[2] = c[0];
}

template <int MF, int LR, int GR, int TU, int LW>
static double run(const char* name, const u32x4* in, const u32x4* gbuf, float* out, double ref) {
  const int iters = 6000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MF, LR, GR, TU, LW>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    probe<MF, LR, GR, TU, LW><<<256, 512, 98304>>>(in, gbuf, out, iters, 65535u & ~4095u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double us_trip = best * 1000.0 / iters;
  const double tf = 256.0 * 8 * iters * MF * 32768.0 / best / 1e9;
  printf("%-44s MF %2d LR %2d GR %2d TU %2d LW %2d : %7.4f us/trip  %6.0f TF issued  %s%.3f\n", name, MF, LR, GR, TU, LW, us_trip, tf,
         ref > 0 ? "speed-up vs direct " : "", ref > 0 ? ref / us_trip : 0.0);
  return us_trip;
}

int main() {
  u32x4 *in, *gbuf; float* out;
  hipMalloc(&in, 4096 * 16); hipMalloc(&gbuf, (size_t)8 * 65536 * 16 + 65536 * 16); hipMalloc(&out, 256 * 512 * 4 + 64);
  unsigned short* h = (unsigned short*)malloc((size_t)9 * 65536 * 16);
  srand(1);
  for (size_t i = 0; i < (size_t)9 * 65536 * 8; ++i) h[i] = (unsigned short)(0x3000 + (rand() & 0x0FFF) + ((rand() & 1) << 15));
  hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
  hipMemcpy(gbuf, h, (size_t)9 * 65536 * 16, hipMemcpyHostToDevice);
  denorm_test<<<1, 64>>>(out);
  float d[3]; hipMemcpy(d, out, 12, hipMemcpyDeviceToHost);
  printf("fp16 subnormal operand 2^-20 (reads back %g): 32x32x16 gives %g, 16x16x32 gives %g; honoured = %g\n", d[1], d[0], d[2],
         16 * 9.5367431640625e-07 * 1024);
  // warm the chip
  for (int i = 0; i < 3; ++i) run<36, 24, 3, 0, 3>("warm-up", in, gbuf, out, 0);
  const double t0 = run<36, 24, 3, 0, 3>("direct (committed kernel's budget)", in, gbuf, out, 0);
  run<36, 24, 0, 0, 0>("direct, MFMAs + fragment reads only", in, gbuf, out, t0);
  run<24, 24, 4, 1, 5>("F(2,3) along W (68 VALU per trip)", in, gbuf, out, t0);
  run<24, 24, 4, 2, 5>("F(2,3) along W, 94 VALU", in, gbuf, out, t0);
  run<24, 24, 4, 0, 5>("F(2,3) along W, no transform VALU", in, gbuf, out, t0);
  run<24, 16, 4, 1, 5>("F(2,3) along W, 16 fragment reads", in, gbuf, out, t0);
  run<16, 11, 7, 2, 7>("F(2x2,3x3) (104 VALU per trip)", in, gbuf, out, t0);
  run<16, 11, 7, 3, 7>("F(2x2,3x3), 122 VALU", in, gbuf, out, t0);
  run<16, 11, 7, 1, 7>("F(2x2,3x3), ~65 VALU", in, gbuf, out, t0);
  run<16, 11, 7, 0, 7>("F(2x2,3x3), no transform VALU", in, gbuf, out, t0);
  run<16, 11, 0, 0, 0>("F(2x2,3x3), MFMAs + fragment reads only", in, gbuf, out, t0);
  run<16, 11, 7, 0, 0>("F(2x2,3x3), + global loads only", in, gbuf, out, t0);
  run<16, 11, 0, 0, 7>("F(2x2,3x3), + LDS writes only", in, gbuf, out, t0);
  run<16, 11, 4, 0, 4>("F(2x2,3x3), 4 loads + 4 writes", in, gbuf, out, t0);
  run<16, 11, 4, 2, 4>("F(2x2,3x3), 4 loads + 4 writes + 104 VALU", in, gbuf, out, t0);
  run<16, 16, 4, 2, 4>("F(2x2,3x3), 16 reads, 4 loads + 4 writes + VALU", in, gbuf, out, t0);
  run<36, 24, 3, 0, 3>("direct again (drift check)", in, gbuf, out, t0);
  return 0;
}

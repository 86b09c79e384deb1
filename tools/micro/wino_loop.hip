// The MAIN LOOP of a fused Winograd F(2x2,3x3) convolution on the split-fp16 scheme, with its real data flow and instruction
// streams, timed alone (no output transform, no BN / residual / cell store, no weight pre-transform) -- the part of the design
// that wino_probe.hip prices synthetically, here as the kernel would run it:
//   * workgroup = 8 waves = 64 channels x 64 tiles (2 tile rows x 32 tile columns = 4 x 64 output pixels), one per CU;
//   * per 16 input channels: the raw 6 x 66 patch arrives as cells by LDS-DMA (25 KB, one buffer), 512 threads transform it --
//     thread = (tile, k-half, xi): 16 cell reads, B^T d B for one row of the 4 x 4 transform domain x 8 channels in fp32, the
//     re-split into hi / UNSCALED lo (the matrix cores honour fp16 subnormals), 8 cell writes into V[next] (2 x 64 KB);
//   * wave w multiplies positions 2 w, 2 w + 1: the transformed weights U (64 KB per 16 channels and workgroup) come straight from
//     global memory / L2 into registers, V fragments from LDS, 24 x v_mfma_f32_32x32x16_f16 per wave into ONE accumulator per
//     (position, channel half, tile half): 128 accumulator registers;
//   * two barriers per 16 channels (V[next] complete / raw buffer free).
// Geometry of the 256 -> 256 @16x57 and 512 -> 512 @8x29 layers at B = 64 (928 / 480 workgroups, 16 / 32 chunks); random data.
// Compare with the committed direct kernel's WHOLE launch (0.152 / 0.146 ms): profiles/README.md, round 5.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kRawCells = 6 * 66 * 4;        // [k-half][plane][6 x 66 pixels]
constexpr int kVCells = 16 * 2 * 2 * 64;     // [position][plane][k-half][64 tiles]

__global__ void __launch_bounds__(512, 2) wino_loop(const u32x4* __restrict__ x, const u32x4* __restrict__ U, float* __restrict__ out,
                                                    int nch, int cout_tiles) {
  extern __shared__ u32x4 lds[];
  u32x4* V = lds;                      // 2 x kVCells
  u32x4* raw = lds + 2 * kVCells;      // kRawCells (padded to 1600)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = blockIdx.x % cout_tiles, tb = blockIdx.x / cout_tiles;
  const u32x4* xs = x + (size_t)tb * nch * 1600;                  // this tile block's input cells, chunk after chunk
  const u32x4* Us = U + (size_t)ct * nch * 16 * 256;              // [chunk][position][plane][k-half][64 channels]
  const int wcell = __builtin_amdgcn_readfirstlane(tid & ~63);
  auto dma_raw = [&](int c) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (wcell + 512 * k < 1600)
        __builtin_amdgcn_global_load_lds(xs + (size_t)c * 1600 + tid + 512 * k, (lds_void*)(raw + wcell + 512 * k), 16, 0, 0);
  };
  // transform role of this thread: tile t, k-half kh, transform row xi
  const int t = tid & 63, kh = wave & 1, xi = wave >> 1;
  const int tr = t >> 5, tc = t & 31;
  const int ra = 2 * tr + (xi == 0 ? 0 : (xi == 2 ? 2 : 1)), rb = 2 * tr + (xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3)));
  const float sb = (xi == 1) ? 1.f : -1.f;                        // row combination d[ra] + sb d[rb]
  const u32x4* rbase = raw + kh * 2 * 396 + 2 * tc;
  u32x4* vdst = V + (4 * xi) * 256 + kh * 64 + t;
  // multiply role: positions 2 wave, 2 wave + 1
  const int l31 = lane & 31, khalf = lane >> 5;
  f32x16 acc[2][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][c][i] = 0.f;

  auto transform = [&](int buf) {
    float r[4][8];                     // row-combined values of the 4 patch columns, 8 channels
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 ha = rbase[ra * 66 + j], la = rbase[396 + ra * 66 + j], hb = rbase[rb * 66 + j], lb = rbase[396 + rb * 66 + j];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f16x2 a0 = __builtin_bit_cast(f16x2, ha[q]), a1 = __builtin_bit_cast(f16x2, la[q]);
        const f16x2 b0 = __builtin_bit_cast(f16x2, hb[q]), b1 = __builtin_bit_cast(f16x2, lb[q]);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float xa = (float)a0[e] + (float)a1[e] * (1.f / 2048.f), xb = (float)b0[e] + (float)b1[e] * (1.f / 2048.f);
          r[j][2 * q + e] = xa + sb * xb;
        }
      }
    }
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      u32x4 hi, lo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f16x2 h, l;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int ch = 2 * q + e;
          const float v = nu == 0 ? r[0][ch] - r[2][ch] : (nu == 1 ? r[1][ch] + r[2][ch] : (nu == 2 ? r[2][ch] - r[1][ch] : r[1][ch] - r[3][ch]));
          h[e] = (_Float16)v;
          l[e] = (_Float16)(v - (float)h[e]);          // unscaled lo: subnormals are honoured
        }
        hi[q] = __builtin_bit_cast(unsigned, h);
        lo[q] = __builtin_bit_cast(unsigned, l);
      }
      vdst[buf * kVCells + nu * 256] = hi;
      vdst[buf * kVCells + nu * 256 + 128] = lo;
    }
  };

  dma_raw(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  transform(0);
  __syncthreads();
  dma_raw(1 < nch ? 1 : 0);
  // U fragments one chunk ahead (a second register set): 2 positions x 2 channel halves x 2 planes, straight from global memory
  f16x8 A[2][2][2][2];
  auto load_u = [&](int c, int set) {
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int m = 0; m < 2; ++m)
          A[set][pi][pl][m] = __builtin_bit_cast(f16x8, Us[((size_t)c * 16 + 2 * wave + pi) * 256 + pl * 128 + khalf * 64 + m * 32 + l31]);
  };
  load_u(0, 0);
  for (int c0 = 0; c0 < nch; c0 += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = c0 + u;
#ifndef NO_U
      load_u(c + 1 < nch ? c + 1 : c, u ^ 1);
#endif
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // everything but the U loads just issued: the raw patch of chunk c + 1 has landed
      __syncthreads();
      const u32x4* vb = V + (c & 1) * kVCells + khalf * 64 + l31;
      f16x8 B[2][2][2];
#pragma unroll
      for (int pi = 0; pi < 2; ++pi)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int th = 0; th < 2; ++th) B[pi][pl][th] = __builtin_bit_cast(f16x8, vb[(2 * wave + pi) * 256 + pl * 128 + th * 32]);
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int th = 0; th < 2; ++th) {
            acc[pi][m][th] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[u][pi][0][m], B[pi][0][th], acc[pi][m][th], 0, 0, 0);
            acc[pi][m][th] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[u][pi][0][m], B[pi][1][th], acc[pi][m][th], 0, 0, 0);
            acc[pi][m][th] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[u][pi][1][m], B[pi][0][th], acc[pi][m][th], 0, 0, 0);
          }
#ifndef NO_TRANSFORM
        if (pi == 0 && c + 1 < nch) transform((c + 1) & 1);     // the next chunk's transform between the two positions' MFMAs
#endif
      }
      __syncthreads();                   // V[next] complete, raw buffer free
#ifndef NO_RAW
      if (c + 2 < nch) dma_raw(c + 2);
#endif
    }
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[a][b][c][i];
  out[(size_t)blockIdx.x * 512 + tid] = s;
}

int main() {
  const size_t lds = (size_t)(2 * kVCells + 1600) * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_loop), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  struct Shape { const char* name; int nch, cout_tiles, tile_blocks; double direct_ms; } shapes[2] = {
      {"256->256 @16x57, B=64", 16, 4, 232, 0.152}, {"512->512 @8x29, B=64", 32, 8, 60, 0.146}};
  for (const Shape& sh : shapes) {
    const size_t xcells = (size_t)sh.tile_blocks * sh.nch * 1600, ucells = (size_t)sh.cout_tiles * sh.nch * 16 * 256;
    u32x4 *x, *U; float* out;
    hipMalloc(&x, xcells * 16); hipMalloc(&U, ucells * 16); hipMalloc(&out, (size_t)sh.tile_blocks * sh.cout_tiles * 512 * 4);
    unsigned short* h = (unsigned short*)malloc((xcells > ucells ? xcells : ucells) * 16);
    srand(1);
    for (size_t i = 0; i < (xcells > ucells ? xcells : ucells) * 8; ++i) h[i] = (unsigned short)(0x3000 + (rand() & 0x0FFF) + ((rand() & 1) << 15));
    hipMemcpy(x, h, xcells * 16, hipMemcpyHostToDevice);
    hipMemcpy(U, h, ucells * 16, hipMemcpyHostToDevice);
    free(h);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = sh.tile_blocks * sh.cout_tiles;
    float best = 1e30f;
    for (int rep = 0; rep < 12; ++rep) {
      hipEventRecord(e0);
      for (int k = 0; k < 10; ++k) wino_loop<<<grid, 512, lds>>>(x, U, out, sh.nch, sh.cout_tiles);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 2 && ms / 10 < best) best = ms / 10;
    }
    const double mfma_tf = (double)grid * 8 * sh.nch * 24 * 32768.0 / best / 1e9;
    printf("%s: %d workgroups, main loop alone %.4f ms (%.0f TFLOP/s of fp16 MFMA issued); the direct 16x16x32 kernel's whole launch %.3f ms -> %.2fx before "
           "the output transform, the epilogue and the weight pre-transform\n", sh.name, grid, best, mfma_tf, sh.direct_ms, sh.direct_ms / best);
    hipFree(x); hipFree(U); hipFree(out);
  }
  return 0;
}

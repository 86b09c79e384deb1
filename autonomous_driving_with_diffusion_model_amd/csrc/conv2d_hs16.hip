// The 3x3 stride-1 perception convolution on v_mfma_f32_16x16x32_f16 (round 5).
//
// Same arithmetic as conv2d_hs3x3_kernel (conv2d_hs.hip: fp32-grade results from fp16 hi / lo split operands, three matrix
// products per multiply-add, two fp32 accumulators), same operand tensors (cell-layout activations, the [cout/64][cin/16][tap]
// split weight image of conv2d_hs_pack), another matrix instruction.  Why: at this chip's power limit the 16x16x32 shape
// sustains 14 % more flops than the 32x32x16 shape at the same LDS reads per flop (tools/micro/mfma_rate.hip: 1642 vs 1436
// TFLOP/s on random operands; MI355X_MICROARCH.md, DVFS give-back item 7: half the accumulator traffic per flop), and a
// timing-only build of the 32x32x16 kernel with every MFMA issued as two 16x16x32 on the same registers ran 7 / 6 / 10 / 13 %
// faster on the 64 / 128 / 256 / 512-channel layers (profiles/README.md, round 5).
//
// What changes with K = 32 per instruction:
//  * a lane's fragment is still ONE 16-byte cell (8 channels of a pixel / of an output channel's tap), but a wave's 64 lanes are
//    16 rows x 4 k-groups, so a K-step covers 32 input channels: the patch of a chunk is [k-group 0..3][plane][pixels] cells and a
//    stage is ONE TAP of a 32-channel chunk (48 MFMAs per wave), weights double-buffered per tap (16 KB), the patch per chunk
//    (43 KB, fetched in three rounds spread over the chunk's nine stages);
//  * wave tile = 64 channels x 64 pixels (2 rows x 32 columns) as 4 x 4 blocks of 16 x 16: 16 fragment reads per 48 MFMAs, the
//    reads-per-flop of the 32x32x16 kernel; workgroup = 8 waves = 8 rows x 32 columns x 128 channels (MODE 2's tile);
//  * accumulator lane = (pixel l & 15, channels 4 (l >> 4) .. + 3): four v_permlane16_swap per pair of pixel blocks leave every
//    lane with one whole cell of 8 channels, then affine / residual / ReLU / split / two 16-byte stores as before.
// The weight image is the one conv2d_hs_pack writes for every kernel of this family (a layer's packed weights must not depend on
// the batch size that picks the kernel): a stage's 1024 cells are four contiguous 4 KB runs of it.
// Scope: the inference executor's plain cell-layout launches (x, y and the residual as cell tensors, virtual-row column tiles)
// with Cout % 128 == 0 and Cin % 64 == 0 -- 23 of the 29 stride-1 convs of ResNet-34; everything else stays on conv2d_hs3x3_kernel.
#include "adx_common.h"
#include "conv2d_internal.h"
#include "conv2d_hs_common.h"

namespace adx {

namespace {

constexpr int kQNT = 512;                  // threads
constexpr int kQTH = 8;                    // output rows per workgroup
constexpr int kQPW = 34;                   // patch columns
constexpr int kQPlane = (kQTH + 2) * kQPW; // 340 staged pixels
constexpr int kQPlaneP = 344;              // pitch of one [k-group][plane] image: 2 * pitch is a multiple of 16 cells, so the four
                                           // k-groups of a fragment read start on the same bank phase (conflict-free ds_read_b128)
constexpr int kQPairs = 4 * kQPlane;       // (k-group, pixel) cell pairs of a 32-channel chunk
constexpr int kQPit = (kQPairs + kQNT - 1) / kQNT;   // 3 rounds
constexpr int kQWst = 1024;                // weight cells of a stage: [slab][16-channel half][plane][k-half][64]
constexpr size_t kQLds = (size_t)2 * 8 * kQPlaneP * 16 + (size_t)2 * kQWst * 16 + 256 * sizeof(float) + 2 * 16 + 512 * sizeof(float);
// DMA variant (buffer_load / global_load ... lds: the staged cells go from memory to LDS without passing through registers): a wave's
// 64 lanes write 64 CONSECUTIVE cells, so the patch image is [plane][k-group][pitch] with the cell pairs of a chunk numbered
// linearly over (k-group, pixel); pitch 352 = 22 x 16 cells keeps the four k-groups of a fragment read on one bank phase
constexpr int kQPlaneD = 352;
constexpr int kQPairsD = 4 * kQPlaneD;     // 1408 = 22 waves of 64: whole waves only
constexpr size_t kQLdsD = (size_t)2 * 8 * kQPlaneD * 16 + (size_t)2 * kQWst * 16 + 256 * sizeof(float) + 2 * 16 + 512 * sizeof(float) +
                          (size_t)kQPit * kQNT * sizeof(uint32_t);      // + the next tile's gather offsets, parked per thread
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// hi + lo / 2^11 of four channels held as two packed fp16 pairs per plane
__device__ __forceinline__ void half4(uint32_t h0, uint32_t h1, uint32_t l0, uint32_t l1, float* out) {
  const float inv = 1.f / kLoScale;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(out[0]) : "v"(l0), "s"(inv), "v"(h0));
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(out[1]) : "v"(l0), "s"(inv), "v"(h0));
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(out[2]) : "v"(l1), "s"(inv), "v"(h1));
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(out[3]) : "v"(l1), "s"(inv), "v"(h1));
}

}  // namespace

// TRAIN: the training forward's launch on a cell-layout input (resnet_train.hip): the raw conv output as fp32 NCHW (BatchNorm
// needs batch statistics before anything can be applied) and, like conv2d_hs3x3_kernel's STATS == 1, per-workgroup partial sums
// of the output and of its squares per channel ([Cout][2][tiles] floats, pixels outside the map excluded).
// TRAIN == 2: a data gradient of the training backward on a cell-layout, pre-scaled gradient (x_amax_n < 0): dx = conv + residual
// (fp32 NCHW, the residual optionally through mask bits) and, like conv2d_hs3x3_kernel's STATS == 2, the consumer BatchNorm's
// backward sums (sum dz, sum dz xhat per channel) and max |dz| per workgroup.
template <bool DMA, int TRAIN = 0>
__global__ void __launch_bounds__(kQNT, 2) conv2d_hs3x3q_kernel(const Conv2dArgs a) {
  constexpr int NT = kQNT, PW = kQPW, PLANE = DMA ? kQPlaneD : kQPlane, PP = DMA ? kQPlaneD : kQPlaneP, PIT = kQPit, WST = kQWst;
  constexpr int NPAIRS = DMA ? kQPairsD : kQPairs;
  // cell strides of the patch image: [k-group][plane][PP], or (DMA) [plane][k-group][PP]
  constexpr int GST = DMA ? PP : 2 * PP, PLST = DMA ? 4 * PP : PP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* patch = reinterpret_cast<u32x4*>(smem_raw);          // 2 x [k-group][plane][PP]
  u32x4* wl = patch + 2 * 8 * PP;                             // 2 x [slab][16-channel half][plane][k-half][64]
  float* ss = reinterpret_cast<float*>(wl + 2 * WST);         // scale[128], shift[128]
  u32x4* dummy = reinterpret_cast<u32x4*>(ss + 256);          // where the idle threads of the last patch round write
  float* bsl = reinterpret_cast<float*>(dummy + 2);           // TRAIN == 2: [mean | rstd | mask scale | mask shift] x 128
  uint32_t* gnext = reinterpret_cast<uint32_t*>(bsl + 512);   // DMA: [round][thread] gather offsets of this workgroup's NEXT tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rowpair = wave & 3, slab = wave >> 2;
  const int j = lane & 15, kq = lane >> 4;
  // Persistent over tiles (round 6): workgroup b = (XCD x = b & 7, cout tile ct, slot q) walks the spatial tiles x's eighth of the
  // tile space holds -- one contiguous range per XCD, so shared halos meet in one L2, and the cout tiles of one spatial tile run
  // side by side on it -- in steps of Q = slots per (XCD, cout tile).  While a tile's LAST chunk is multiplied, the idle patch
  // buffer and the idle weight buffer receive the NEXT tile's first chunk and first tap -- the transfers that used to re-fetch
  // the last chunk into buffers nobody read -- so only a workgroup's first tile has a prologue (a 128-channel layer at B = 64 is
  // 3.6 tiles per CU: 2-3 us of exposed transfer latency per 45 us tile).  The cout tile never changes inside a workgroup, so
  // neither do its weights' addresses nor its per-channel constants.  Q = the longest range: one tile per workgroup, the old launch.
  const int Qs = a.q_slots;
  const int xcd = blockIdx.x & 7, rr_ = blockIdx.x >> 3;
  const int ct = rr_ % a.cout_tiles, slot = rr_ / a.cout_tiles;
  const int nsp = a.tiles_x * a.tiles_y;
  const int sp_end = (int)(((long)(xcd + 1) * nsp) >> 3);
  int sp = (int)(((long)xcd * nsp) >> 3) + slot;
  if (sp >= sp_end) return;                    // (uniform: a padding workgroup of a short range)
  int ty = sp / a.tiles_x, tx = sp - ty * a.tiles_x;
  int oy0 = ty * kQTH, vx0 = tx * 32;
  auto vdiv = [&](int v) { return (int)(((float)v + 0.5f) * a.inv_vw); };     // v / vw for 0 <= v < 2^21 (launch check)
  const size_t hw = (size_t)a.H * a.W;
  const int nch16 = a.cin_pad / 16, nch32 = a.cin_pad / 32, nstages = nch32 * 9;
  constexpr uint32_t kOutside = 0xC0000000u;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)(uint32_t)((size_t)a.N * a.Cin * hw * sizeof(float)), 0x00020000);
  const uint32_t plane_bytes = (uint32_t)(hw * sizeof(float));
  // gather offset of this thread's cell pair of patch round k for the tile at (ty_, tx_)
  auto goff_of = [&](int t_, int k, int oy0_, int vx0_) -> uint32_t {
    const int e = t_ + NT * k;
    const int g = e / PLANE, p = e - g * PLANE;                // k-group, staged pixel (DMA: p >= 340 is padding of the pitch)
    const int py = p / PW, px = p - py * PW;
    const int iy = oy0_ - 1 + py;
    const int v = vx0_ - 1 + px;                               // virtual column: image v / vw, column v % vw (column W of an image is zero)
    const int ni = vdiv(v < 0 ? 0 : v);
    const int ix = v - ni * a.vw;
    const bool ok = e < NPAIRS && p < kQPlane && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && ni < a.N;
    return !ok ? kOutside : (uint32_t)((size_t)ni * a.Cin * hw * sizeof(float) + (size_t)g * 8 * hw * sizeof(float) + ((size_t)iy * a.W + ix) * 16);
  };
  uint32_t goff[PIT];
  int pcell[PIT];
#pragma unroll
  for (int k = 0; k < PIT; ++k) {
    const int e = tid + NT * k;
    const int g = e / PLANE, p = e - g * PLANE;
    goff[k] = goff_of(tid, k, oy0, vx0);
    pcell[k] = e < NPAIRS ? g * GST + p : -1;
  }
  // this thread's two weight cells of a stage: cell e of the LDS image is cell wsrc_off[k] + (18 chunk + tap) * 256 of the packed image
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.w);
  int wsrc_off[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = tid + NT * k;
    const int sl = e >> 9, half16 = (e >> 8) & 1, within = e & 255;
    wsrc_off[k] = (((ct * 2 + sl) * nch16 + half16) * 9) * 256 + within;
  }
  float ssv = 0.f;
  if (tid < 256) {
    const int c = ct * 128 + (tid & 127);
    ssv = a.scale == nullptr ? (tid < 128 ? 1.f : 0.f) : (tid < 128 ? a.scale[c] : a.shift[c]);
  }

  f32x4 accm[4][4], accl[4][4];            // [channel block][pixel block]: hi*hi sums, cross-term sums (scaled by 2^11)

  u32x4 wv[2][2], pvh[2], pvl[2];          // register sets: weights of stage s in wv[s & 1]; the patch round fetched at stage s in pv*[s & 1]
  // DMA: the wave's first cell of round k (a wave-uniform LDS address goes into M0; lane l lands l cells further)
  const int wcell = __builtin_amdgcn_readfirstlane(tid & ~63);
  auto load_w = [&](int stage, int set) {           // DMA: `set` is the LDS weight buffer
    const int chunk = stage / 9, tap = stage - chunk * 9;
    const u32x4* ws = wsrc + (size_t)(chunk * 18 + tap) * 256;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if constexpr (DMA) __builtin_amdgcn_global_load_lds(ws + wsrc_off[k], (lds_void*)(wl + set * WST + wcell + NT * k), 16, 0, 0);
      else wv[set][k] = ws[wsrc_off[k]];
    }
  };
  auto store_w = [&](int set, int buf) {
    if constexpr (!DMA) {
#pragma unroll
      for (int k = 0; k < 2; ++k) wl[buf * WST + tid + NT * k] = wv[set][k];
    }
  };
  auto load_p_at = [&](int chunk, int k, int set, uint32_t go) {    // DMA: `set` is the LDS patch buffer; go: the round's gather offset
    const uint32_t cbase = (uint32_t)chunk * 32u * plane_bytes;
    if constexpr (DMA) {
      if (k < PIT - 1 || wcell + NT * k < NPAIRS) {      // whole waves: 1408 = 22 x 64
        u32x4* d = patch + set * 8 * PP + wcell + NT * k;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)d, 16, go, cbase, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)(d + PLST), 16, go, cbase + 4 * plane_bytes, 0, 0);
      }
    } else {
      pvh[set] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, go, cbase, 0);
      pvl[set] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, go, cbase + 4 * plane_bytes, 0);
    }
  };
  auto load_p = [&](int chunk, int k, int set) { load_p_at(chunk, k, set, goff[k]); };
  auto store_p = [&](int set, int k, int buf) {
    if constexpr (!DMA) {
      u32x4* pd = patch + buf * 8 * PP + pcell[k];
      u32x4* d0 = (k < PIT - 1 || pcell[k] >= 0) ? pd : dummy;
      u32x4* d1 = (k < PIT - 1 || pcell[k] >= 0) ? pd + PP : dummy + 1;
      *d0 = pvh[set];
      *d1 = pvl[set];
    }
  };

  // fragment bases: B (pixels) of lane (j, kq): patch cell kq * 2 PP + (2 rowpair + kh + row) * PW + 16 colhalf + j + kw, + PP for lo;
  // A (channels) of lane (j, kq): weight cell slab * 512 + (kq >> 1) * 256 + plane * 128 + (kq & 1) * 64 + 16 cb + j

  // prologue (a workgroup's FIRST tile only): stage 0 complete in LDS, the weights of stage 1 in flight
  load_w(0, 0);
  store_w(0, 0);
#pragma unroll
  for (int k = 0; k < PIT; ++k) { load_p(0, k, 0); store_p(0, k, 0); }
  if constexpr (!DMA) load_w(1, 1);
  if (tid < 256) ss[tid] = ssv;
  if constexpr (TRAIN == 2) {
    const int which = tid >> 7, c = ct * 128 + (tid & 127);
    const float mu = a.bs_mean[c], rs = a.bs_rstd[c];
    const float sc = a.bs_gamma[c] * rs;          // the mask's affine form exactly as the forward pass applied it (resnet_train.hip: bn_affine)
    bsl[tid] = which == 0 ? mu : (which == 1 ? rs : (which == 2 ? sc : __builtin_fmaf(-mu, sc, a.bs_beta[c])));
  }
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (;;) {       // tiles of this workgroup
  // Everything per-lane the stage loop needs is RE-DERIVED here from the thread id through an opaque zero instead of living in
  // registers across the epilogue (which needs every register it can get: 128 accumulators + the residual cells): ~100 VALU
  // instructions per tile against a dozen spilled registers.
  int zero_ = 0;
  asm volatile("" : "+s"(zero_));
  const int tl = tid + zero_;
  if constexpr (DMA) {
#pragma unroll
    for (int k = 0; k < PIT; ++k) goff[k] = goff_of(tl, k, oy0, vx0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tl + NT * k;
      const int sl = e >> 9, half16 = (e >> 8) & 1, within = e & 255;
      wsrc_off[k] = (((ct * 2 + sl) * nch16 + half16) * 9) * 256 + within;
    }
  }
  const int j_l = tl & 15, kq_l = (tl & 63) >> 4;
  const int pb_lane = kq_l * GST + (rowpair * 2) * PW + j_l;
  const int wa_lane = slab * 512 + (kq_l >> 1) * 256 + (kq_l & 1) * 64 + j_l;
  // the next tile (DMA only: the register-staged form runs one tile per workgroup)
  const int sp_next = sp + Qs;
  const bool has_next = DMA && sp_next < sp_end;
  const int nty = sp_next / a.tiles_x, ntx = sp_next - nty * a.tiles_x;
  const int n_oy0 = nty * kQTH, n_vx0 = ntx * 32;
  // the next tile's gather offsets: parked in LDS (a thread's own words) until the last chunk asks for them -- in registers they
  // pushed the stage loop over its budget, computed where they are used they kept a dozen scalars of the address arithmetic alive
  if constexpr (DMA) {
#pragma unroll
    for (int k = 0; k < PIT; ++k) gnext[k * NT + tid] = has_next ? goff_of(tl, k, n_oy0, n_vx0) : goff[k];
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int pb = 0; pb < 4; ++pb)
#pragma unroll
      for (int i = 0; i < 4; ++i) { accm[cb][pb][i] = 0.f; accl[cb][pb][i] = 0.f; }

  // (The compiler places each stage's barrier in the MIDDLE of the stage's MFMAs -- legal: the fragments are in registers -- so a
  // wave reads the next stage's fragments while its SIMD partner still multiplies.  Pinning the barrier to the stage's end, or
  // staggering waves 4-7 by half a stage against waves 0-3, measured 1-3 % slower: profiles/README.md, round 5.)
  for (int cp = 0; cp < nch32; cp += 2) {
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int s = 9 * cp + i;                   // global stage; s & 1 == i & 1 because 9 cp is even
      const int cpar = i / 9, t = i % 9, kh = t / 3, kw = t % 3;
      const u32x4* pb0 = patch + cpar * 8 * PP + pb_lane + kh * PW + kw;
      const u32x4* wa0 = wl + (i & 1) * WST + wa_lane;
      if constexpr (DMA) {
        // one stage ahead, straight into LDS: the weights of stage s + 1 into the buffer stage s - 1 read (every wave is past that
        // stage's barrier), and at a kernel row's first tap one round of the NEXT chunk's patch into the idle patch buffer
        // -- past the tile's end: the NEXT tile's first tap / first chunk (buffers 0: the last stage is odd, the last chunk is
        // chunk parity 1), or, without a next tile, the last stage / chunk again (never read)
        load_w(s + 1 < nstages ? s + 1 : (has_next ? 0 : nstages - 1), (i + 1) & 1);
        if (kw == 0) {
          // (the tile's last chunk has chunk parity 1: the channel chunks come in pairs)
          if (cpar == 0 || cp + cpar + 1 < nch32) load_p(cp + cpar + 1, kh, (cpar + 1) & 1);
          else load_p_at(has_next ? 0 : nch32 - 1, kh, (cpar + 1) & 1, gnext[kh * NT + tid]);
        }
      } else {
        // fetch two stages ahead: the weights of stage s + 2, and at a kernel row's first tap one round of the NEXT chunk's patch
        // (past the end the last stage / chunk is fetched again; its copy in the idle buffers is never read)
        load_w(s + 2 < nstages ? s + 2 : nstages - 1, i & 1);
        if (kw == 0) load_p(cp + cpar + 1 < nch32 ? cp + cpar + 1 : nch32 - 1, kh, i & 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      f16x8 A[2][4], B[2][4];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) A[pl][cb] = __builtin_bit_cast(f16x8, wa0[pl * 128 + cb * 16]);
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) B[pl][pb] = __builtin_bit_cast(f16x8, pb0[pl * PLST + (pb >> 1) * PW + (pb & 1) * 16]);
      }
      auto row_mfmas = [&](int rr) {             // the 24 MFMAs of one of the wave's two output rows (pixel blocks 2 rr, 2 rr + 1)
#pragma unroll
        for (int pb = 2 * rr; pb < 2 * rr + 2; ++pb) {
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            accm[cb][pb] = mfma16(A[0][cb], B[0][pb], accm[cb][pb]);
            accl[cb][pb] = mfma16(A[0][cb], B[1][pb], accl[cb][pb]);
          }
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) accl[cb][pb] = mfma16(A[1][cb], B[0][pb], accl[cb][pb]);
          if (pb == 0) {
            // what arrived during the previous stage goes to LDS under this stage's remaining MFMAs
            store_w((i + 1) & 1, (i + 1) & 1);
            if (kw == 1) store_p((i + 1) & 1, kh, (cpar + 1) & 1);
          }
        }
      };
      row_mfmas(0);
      row_mfmas(1);
      if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this stage's transfers have landed (issued a stage of MFMAs ago)
      __syncthreads();
    }
  }

  if constexpr (TRAIN == 2) {
    // ---- data-gradient epilogue: everything AFTER the swap (lane (j, rw = kq): channels 16 cb + 8 (rw >> 1) .. + 7 of the pixel at
    // column 16 (rw & 1) + j: a half wave = 32 consecutive pixels of a channel plane, 128-byte runs for every fp32 access) ----
    const float xs_inv = reinterpret_cast<const float*>(a.x_amax)[1];
    const uint32_t plane_ob = (uint32_t)(a.OH * a.OW) * (uint32_t)sizeof(float);
    const uint32_t bplane = (uint32_t)(a.OH * a.OW);
    const uint32_t img_bytes = (uint32_t)a.N * (uint32_t)a.Cout * plane_ob, bit_bytes = (uint32_t)a.N * (uint32_t)(a.Cout >> 3) * bplane;
    const int rw = kq;
    const int vcol = vx0 + 16 * (rw & 1) + j;
    const int nl = vdiv(vcol), xl = vcol - nl * a.vw;
    const bool col_valid = nl < a.N && xl < a.W;
    const bool mask_bits = a.bs_mask == 1, res_masked = a.res_bits != nullptr;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.res != nullptr ? a.res : a.y), 0, a.res != nullptr ? (int)img_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bs_raw), 0, (int)img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(mask_bits ? a.bs_bits : reinterpret_cast<const uint8_t*>(a.y)), 0, mask_bits ? (int)bit_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(res_masked ? a.res_bits : reinterpret_cast<const uint8_t*>(a.y)), 0, res_masked ? (int)bit_bytes : 0, 0x00020000);
    const int cout0 = ct * 128 + slab * 64;
    const float* bsw = bsl + slab * 64 + 8 * (rw >> 1);         // + 16 cb + c8: this lane's channels
    float* red = reinterpret_cast<float*>(patch + 8 * PP);      // [wave][rw][cb][c8][2], then [4096 + wave]: max |dz| (patch buffer 1: dead;
                                                                // buffer 0 may already hold the next tile's first chunk)
    uint32_t voff[2], boff[2];
    bool valid[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int oy = oy0 + rowpair * 2 + rr;
      valid[rr] = col_valid && oy < a.OH;
      voff[rr] = valid[rr] ? (uint32_t)nl * (uint32_t)a.Cout * plane_ob + (uint32_t)(oy * a.OW + xl) * 4u + (uint32_t)(8 * (rw >> 1)) * plane_ob : kOutside;
      boff[rr] = valid[rr] ? (uint32_t)nl * (uint32_t)(a.Cout >> 3) * bplane + (uint32_t)(oy * a.OW + xl) + (uint32_t)(rw >> 1) * bplane : kOutside;
    }
    float dmx = 0.f;
    // The ReLU mask of the consumer layer comes as bits or is recomputed from that layer's raw output; which, is uniform over the
    // launch.  Both forms are evaluated and one is SELECTED, with the lane's validity folded in by `&`: as valid && (mask_bits ?
    // bit : fma(raw, scale, shift) > 0) the compiler kept three branches per value and, on the recomputing side, an LDS round
    // trip per value (the constants' reads sat inside the branch): 190 branches per tile.
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      float sm[8], sq[8];
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8) { sm[c8] = 0.f; sq[c8] = 0.f; }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        float o[8], res8[8], rw8[8];
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
          const uint32_t so = (uint32_t)(cout0 + 16 * cb + c8) * plane_ob;
          res8[c8] = u2f(__builtin_amdgcn_raw_buffer_load_b32(rrsrc, voff[rr], so, 0));
          rw8[c8] = u2f(__builtin_amdgcn_raw_buffer_load_b32(wrsrc, voff[rr], so, 0));
        }
        const uint32_t bso = (uint32_t)((cout0 >> 3) + 2 * cb) * bplane;
        const uint32_t mres = res_masked ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(qrsrc, boff[rr], bso, 0) : 0xFFu;
        const uint32_t mbs = mask_bits ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(brsrc, boff[rr], bso, 0) : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x = (accm[cb][2 * rr][i] + accl[cb][2 * rr][i] * (1.f / kLoScale)) * xs_inv;
          const float y = (accm[cb][2 * rr + 1][i] + accl[cb][2 * rr + 1][i] * (1.f / kLoScale)) * xs_inv;
          const auto sw = __builtin_amdgcn_permlane16_swap(f2u(x), f2u(y), false, false);
          o[i] = u2f(sw[0]);
          o[4 + i] = u2f(sw[1]);
        }
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
          const float* kc = bsw + 16 * cb + c8;
          const float k_mu = kc[0], k_rs = kc[128], k_sc = kc[256], k_sh = kc[384];
          const float yv = o[c8] + (((mres >> c8) & 1u) ? res8[c8] : 0.f);
          __builtin_amdgcn_raw_buffer_store_b32(f2u(yv), yrsrc, voff[rr], (uint32_t)(cout0 + 16 * cb + c8) * plane_ob, 0);
          const bool on_bit = ((mbs >> c8) & 1u) != 0u, on_raw = __builtin_fmaf(rw8[c8], k_sc, k_sh) > 0.f;
          const bool keep = valid[rr] & (mask_bits ? on_bit : on_raw);
          const float pz = keep ? yv : 0.f;
          sm[c8] += pz;
          sq[c8] = __builtin_fmaf(pz, (rw8[c8] - k_mu) * k_rs, sq[c8]);
          dmx = __builtin_fmaxf(dmx, __builtin_fabsf(pz));
        }
      }
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8) {
        float s0 = sm[c8], s1 = sq[c8];
        s0 += hs_dpp<0xB1>(s0);  s1 += hs_dpp<0xB1>(s1);
        s0 += hs_dpp<0x4E>(s0);  s1 += hs_dpp<0x4E>(s1);
        s0 += hs_dpp<0x141>(s0); s1 += hs_dpp<0x141>(s1);
        s0 += hs_dpp<0x140>(s0); s1 += hs_dpp<0x140>(s1);
        if (j == 0) {
          float* d = red + ((((wave * 4 + rw) * 4 + cb) * 8) + c8) * 2;
          d[0] = s0;
          d[1] = s1;
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dmx = __builtin_fmaxf(dmx, __shfl_xor(dmx, off, 64));
    if (lane == 0) red[4096 + wave] = dmx;
    __syncthreads();
    const int ptile = ty * a.tiles_x + tx;
    if (tid < 256) {
      const int which = tid & 1, ch = tid >> 1, sl = ch >> 6, cl = ch & 63;
      const int cb = cl >> 4, h8 = (cl >> 3) & 1, c8 = cl & 7;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int e = 0; e < 2; ++e) t += red[(((((sl * 4 + w) * 4 + 2 * h8 + e) * 4 + cb) * 8) + c8) * 2 + which];
      a.stats_part[((size_t)(ct * 128 + ch) * 2 + which) * a.stats_p + ptile] = t;
    } else if (tid < 258) {
      const int sl = tid - 256;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t = __builtin_fmaxf(t, red[4096 + sl * 4 + w]);
      a.stats_part[(size_t)a.Cout * 2 * a.stats_p + (size_t)(ct * 2 + sl) * a.stats_p + ptile] = t;
    }
  } else if constexpr (TRAIN == 1) {
    // ---- training epilogue ----
    // statistics first, from the accumulators as they lie: lane (j, kq) holds channels 16 cb + 4 kq + i of its four pixels (row
    // pb >> 1, virtual column 16 (pb & 1) + j); the four pixels in the lane, then the 16 lanes of a DPP row; lanes j == 0 park
    // the row's totals in LDS (the operand images are dead: every wave is past the last stage's barrier) and 256 threads add
    // the four row-pair waves of a slab in a fixed order.
    const uint32_t plane_ob = (uint32_t)(a.OH * a.OW) * (uint32_t)sizeof(float);
    float* red = reinterpret_cast<float*>(patch + 8 * PP);    // [wave][kq][cb][i][2] (patch buffer 1, as above)
    float valid[4];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      const int vc = vx0 + 16 * (pb & 1) + j;
      const int np = vdiv(vc), xp = vc - np * a.vw;
      valid[pb] = (np < a.N && xp < a.W && oy0 + rowpair * 2 + (pb >> 1) < a.OH) ? 1.f : 0.f;
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
          const float v = (accm[cb][pb][i] + accl[cb][pb][i] * (1.f / kLoScale)) * valid[pb];
          accm[cb][pb][i] = v;                    // (a pixel outside the map is never stored: its lane's store offset is out of range)
          sm += v;
          sq = __builtin_fmaf(v, v, sq);
        }
        sm += hs_dpp<0xB1>(sm);  sq += hs_dpp<0xB1>(sq);      // quad_perm [1,0,3,2]
        sm += hs_dpp<0x4E>(sm);  sq += hs_dpp<0x4E>(sq);      // quad_perm [2,3,0,1]
        sm += hs_dpp<0x141>(sm); sq += hs_dpp<0x141>(sq);     // row_half_mirror
        sm += hs_dpp<0x140>(sm); sq += hs_dpp<0x140>(sq);     // row_mirror: every lane of a row of 16 holds the row's total
        if (j == 0) {
          float* d = red + ((((wave * 4 + kq) * 4 + cb) * 4) + i) * 2;
          d[0] = sm;
          d[1] = sq;
        }
      }
    __syncthreads();
    if (tid < 256) {
      const int which = tid & 1, ch = tid >> 1, sl = ch >> 6, cl = ch & 63;        // channel ch of the workgroup's 128
      const int cb = cl >> 4, q = (cl >> 2) & 3, i = cl & 3;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += red[(((((sl * 4 + w) * 4 + q) * 4 + cb) * 4) + i) * 2 + which];
      a.stats_part[((size_t)(ct * 128 + ch) * 2 + which) * a.stats_p + (ty * a.tiles_x + tx)] = t;
    }
    // the conv output: after the swap of a row's two pixel blocks lane (j, rw = kq) holds channels 16 cb + 8 (rw >> 1) .. + 7 of
    // the pixel at column 16 (rw & 1) + j, i.e. the 32 lanes of a half wave are 32 consecutive pixels of one channel plane
    const int rw = kq;
    const int vcol = vx0 + 16 * (rw & 1) + j;
    const int nl = vdiv(vcol), xl = vcol - nl * a.vw;
    const bool col_valid = nl < a.N && xl < a.W;
    const uint32_t img_bytes = (uint32_t)a.N * (uint32_t)a.Cout * plane_ob;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)img_bytes, 0x00020000);
    const int cout0 = ct * 128 + slab * 64;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int oy = oy0 + rowpair * 2 + rr;
      const uint32_t voff = (col_valid && oy < a.OH)
                                ? (uint32_t)nl * (uint32_t)a.Cout * plane_ob + (uint32_t)(oy * a.OW + xl) * 4u + (uint32_t)(8 * (rw >> 1)) * plane_ob
                                : kOutside;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const auto sw = __builtin_amdgcn_permlane16_swap(f2u(accm[cb][2 * rr][i]), f2u(accm[cb][2 * rr + 1][i]), false, false);
          __builtin_amdgcn_raw_buffer_store_b32(sw[0], yrsrc, voff, (uint32_t)(cout0 + 16 * cb + i) * plane_ob, 0);
          __builtin_amdgcn_raw_buffer_store_b32(sw[1], yrsrc, voff, (uint32_t)(cout0 + 16 * cb + 4 + i) * plane_ob, 0);
        }
    }
  } else {
  // ---- epilogue: one cell (8 channels of a pixel, hi + lo) per lane and (channel block, row) ----
  // Accumulator lane (j, kq) holds channels 16 cb + 4 kq + i of pixel (row pb >> 1, column 16 (pb & 1) + j).  After the swaps of
  // a row's two pixel blocks lane (j, rw = kq) holds the cell of channels 16 cb + 8 (rw >> 1) .. + 7 at column 16 (rw & 1) + j.
  const int rw = kq;
  const int vcol = vx0 + 16 * (rw & 1) + j;
  const int nl = vdiv(vcol), xl = vcol - nl * a.vw;
  const bool col_valid = nl < a.N && xl < a.W;
  const uint32_t plane_ob = (uint32_t)(a.OH * a.OW) * (uint32_t)sizeof(float);
  const uint32_t cplane = (uint32_t)(a.OH * a.OW) * 16u;
  const uint32_t img_bytes = (uint32_t)a.N * (uint32_t)a.Cout * plane_ob;
  const uint32_t img_off = (uint32_t)nl * (uint32_t)a.Cout * plane_ob;
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.res != nullptr ? a.res : a.y), 0, a.res != nullptr ? (int)img_bytes : 0, 0x00020000);
  const int cout0 = ct * 128 + slab * 64;
  const uint32_t cell0 = (uint32_t)(cout0 >> 3) * 2u * cplane;            // hi plane of this wave's first cell group
  uint32_t vcell[2];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int oy = oy0 + rowpair * 2 + rr;
    vcell[rr] = (col_valid && oy < a.OH) ? img_off + (uint32_t)(oy * a.OW + xl) * 16u + (uint32_t)(rw >> 1) * 2u * cplane : kOutside;
  }
  // the residual cells of the whole tile first (one exposed latency): cell group 2 cb + (rw >> 1) of the wave's 64 channels
  u32x4 rh[4][2], rl[4][2];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const uint32_t so = cell0 + (uint32_t)(2 * cb) * 2u * cplane;
      rh[cb][rr] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, vcell[rr], so, 0);
      rl[cb][rr] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, vcell[rr], so + cplane, 0);
    }
  const float* sc = ss + slab * 64;
  const float* sh = ss + 128 + slab * 64;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      float x[4], y[4];                           // pixel blocks 2 rr (columns 0-15) and 2 rr + 1 (columns 16-31): channels 16 cb + 4 kq + i
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cl = cb * 16 + 4 * kq + i;
        x[i] = (accm[cb][2 * rr][i] + accl[cb][2 * rr][i] * (1.f / kLoScale)) * sc[cl] + sh[cl];
        y[i] = (accm[cb][2 * rr + 1][i] + accl[cb][2 * rr + 1][i] * (1.f / kLoScale)) * sc[cl] + sh[cl];
      }
      float o[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // rows of 16 lanes: x's odd rows <-> y's even rows.  Afterwards x = channels + 0..3 and y = channels + 4..7 of the lane's cell
        const auto sw = __builtin_amdgcn_permlane16_swap(f2u(x[i]), f2u(y[i]), false, false);
        o[i] = u2f(sw[0]);
        o[4 + i] = u2f(sw[1]);
      }
      float r8[8];
      half4(rh[cb][rr][0], rh[cb][rr][1], rl[cb][rr][0], rl[cb][rr][1], r8);
      half4(rh[cb][rr][2], rh[cb][rr][3], rl[cb][rr][2], rl[cb][rr][3], r8 + 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float tsum = o[i] + r8[i];
        o[i] = a.relu ? __builtin_fmaxf(tsum, 0.f) : tsum;
      }
      u32x4 hi, lo;
      split8(o, 1.f, hi, lo);
      const uint32_t so = cell0 + (uint32_t)(2 * cb) * 2u * cplane;
      __builtin_amdgcn_raw_buffer_store_b128(hi, yrsrc, vcell[rr], so, 0);
      __builtin_amdgcn_raw_buffer_store_b128(lo, yrsrc, vcell[rr], so + cplane, 0);
      // a VALU write to the data registers of a 16-byte buffer store with an SGPR offset in the next issue slot can reach the
      // store (conv2d_hs.hip: cells_store32); keep both operands alive across one wait state
      asm volatile("s_nop 0" ::"v"(hi), "v"(lo));
    }
  }
  // ---- on to this workgroup's next tile: its first chunk and first tap are in LDS already ----
  if (!has_next) break;
  if constexpr (TRAIN != 0) __syncthreads();        // every wave is done with this tile's statistics area (patch buffer 1)
  sp = sp_next; tx = ntx; ty = nty; oy0 = n_oy0; vx0 = n_vx0;
  }
}

// a training-forward launch (cells in, fp32 + statistics out) the TRAIN variant serves
bool conv2d_hs3x3q_train_eligible(const Conv2dArgs& a) {
  return debug_switches().hs_mode < 0 && debug_switches().train_cells >= 3 && a.x_cells && !a.y_cells && !a.res_cells && a.res == nullptr &&
         a.x_amax == nullptr && a.bs_raw == nullptr && a.scale == nullptr && a.Cout % 128 == 0 && a.cin_pad % 64 == 0 && a.cin_pad == a.Cin &&
         a.pad == 1 && a.stride == 1 && a.KH == 3 && a.KW == 3 && a.H == a.OH && a.W == a.OW && (long)a.N * (a.OW + 1) < (1L << 21) &&
         (size_t)a.N * a.Cout * a.OH * a.OW * sizeof(float) < 0xC0000000u && (size_t)a.N * a.Cin * a.H * a.W * sizeof(float) < 0xC0000000u;
}
// a data-gradient launch with the consumer BatchNorm's sums (cells in under their scale, fp32 out) the TRAIN == 2 variant serves
bool conv2d_hs3x3q_dgrad_eligible(const Conv2dArgs& a) {
  return debug_switches().hs_mode < 0 && debug_switches().train_cells >= 5 && a.x_cells && !a.y_cells && !a.res_cells &&
         a.x_amax != nullptr && a.x_amax_n < 0 && a.bs_raw != nullptr && (a.bs_mask == 2 || (a.bs_mask == 1 && a.bs_bits != nullptr)) &&
         a.scale == nullptr && a.relu == 0 && a.Cout % 128 == 0 && a.cin_pad % 64 == 0 && a.cin_pad == a.Cin &&
         a.pad == 1 && a.stride == 1 && a.KH == 3 && a.KW == 3 && a.H == a.OH && a.W == a.OW && (long)a.N * (a.OW + 1) < (1L << 21) &&
         (size_t)a.N * a.Cout * a.OH * a.OW * sizeof(float) < 0xC0000000u && (size_t)a.N * a.Cin * a.H * a.W * sizeof(float) < 0xC0000000u;
}
int conv2d_hs3x3q_train_tiles(const Conv2dArgs& a) {
  const int vw = a.N > 1 ? a.OW + 1 : a.OW;
  return ceil_div(a.OH, kQTH) * ceil_div(a.N * vw - (a.N > 1 ? 1 : 0), 32);
}

bool conv2d_hs3x3q_eligible(const Conv2dArgs& a) {
  const int pin = debug_switches().hs_mode;            // ADX_HS_MODE=0|1|2 pins a tile mode of the 32x32x16 kernel
  return pin < 0 && a.x_cells && a.y_cells && (a.res == nullptr || a.res_cells) && a.x_amax == nullptr && a.stats_part == nullptr &&
         a.Cout % 128 == 0 && a.cin_pad % 64 == 0 && a.cin_pad == a.Cin && a.pad == 1 && a.stride == 1 && a.KH == 3 && a.KW == 3 &&
         a.H == a.OH && a.W == a.OW;
}

int conv2d_hs3x3q_launch(Conv2dArgs a, hipStream_t s) {
  const bool train = a.stats_part != nullptr, dgrad = train && a.bs_raw != nullptr;
  ADX_REQUIRE(dgrad ? conv2d_hs3x3q_dgrad_eligible(a) : (train ? conv2d_hs3x3q_train_eligible(a) : conv2d_hs3x3q_eligible(a)),
              "conv2d_hs3x3q: launch outside the kernel's rules");
  static std::atomic<uint64_t> attr{0};
  if (DeviceOnce once{attr}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3q_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kQLds));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3q_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kQLdsD));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3q_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kQLds));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3q_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kQLdsD));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3q_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kQLds));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3q_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kQLdsD));
    once.commit();
  }
  a.vw = a.N > 1 ? a.OW + 1 : a.OW;
  a.inv_vw = 1.f / (float)a.vw;
  ADX_REQUIRE((long)a.N * a.vw < (1L << 21), "conv2d_hs: batch x width exceeds the virtual-row arithmetic");
  a.tiles_x = ceil_div(a.N * a.vw - (a.N > 1 ? 1 : 0), 32);       // the last image's zero column needs no tile
  a.tiles_y = ceil_div(a.OH, kQTH);
  a.cout_tiles = a.Cout / 128;
  const size_t tiles = (size_t)a.cout_tiles * a.tiles_x * a.tiles_y;
  ADX_REQUIRE(tiles < (1u << 31), "conv2d_hs: grid too large");
  a.ntiles = (int)tiles;
  // workgroup = (XCD, cout tile, slot); slots per (XCD, cout tile): as many as the longest eighth of the spatial tiles (one tile
  // per workgroup: ADX_HS_PERSIST=0 and the register-staged form), or -- the LDS-DMA form -- as many as fit one workgroup per CU,
  // each walking its XCD's range in steps of that count (bit-identical results either way)
  const int nsp = a.tiles_x * a.tiles_y;
  int slots = ceil_div(nsp, 8);
  if (debug_switches().hs_dma && debug_switches().hs_persist) {
    static std::atomic<int> n_cu{0};
    int cus = n_cu.load(std::memory_order_relaxed);
    if (cus == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      ADX_CHECK_HIP(hipGetDevice(&dev));
      ADX_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
      cus = prop.multiProcessorCount > 8 ? prop.multiProcessorCount : 8;
      n_cu.store(cus, std::memory_order_relaxed);
    }
    slots = std::min(slots, std::max(1, cus / (8 * a.cout_tiles)));
  }
  a.q_slots = slots;
  const size_t grid = (size_t)8 * a.cout_tiles * slots;
  ADX_REQUIRE((size_t)a.N * a.Cout * a.OH * a.OW * sizeof(float) < 0xC0000000u && (size_t)a.N * a.Cin * a.H * a.W * sizeof(float) < 0xC0000000u,
              "conv2d_hs: a cell-layout tensor exceeds the 32-bit byte offsets");
  if (train) {
    ADX_REQUIRE(a.stats_p == a.tiles_y * a.tiles_x, "conv2d_hs3x3q: statistics buffer laid out for %d tiles, launch has %d", a.stats_p,
                a.tiles_y * a.tiles_x);
    if (dgrad) {
      if (debug_switches().hs_dma) conv2d_hs3x3q_kernel<true, 2><<<dim3((unsigned)grid), dim3(kQNT), kQLdsD, s>>>(a);
      else conv2d_hs3x3q_kernel<false, 2><<<dim3((unsigned)grid), dim3(kQNT), kQLds, s>>>(a);
    } else if (debug_switches().hs_dma) conv2d_hs3x3q_kernel<true, 1><<<dim3((unsigned)grid), dim3(kQNT), kQLdsD, s>>>(a);
    else conv2d_hs3x3q_kernel<false, 1><<<dim3((unsigned)grid), dim3(kQNT), kQLds, s>>>(a);
  } else if (debug_switches().hs_dma) conv2d_hs3x3q_kernel<true><<<dim3((unsigned)grid), dim3(kQNT), kQLdsD, s>>>(a);
  else conv2d_hs3x3q_kernel<false><<<dim3((unsigned)grid), dim3(kQNT), kQLds, s>>>(a);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

}  // namespace adx

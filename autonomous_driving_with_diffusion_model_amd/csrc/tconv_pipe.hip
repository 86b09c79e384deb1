// A run of same-shaped temporal layers as ONE launch: a pipeline of workgroups that hand activations on through memory.
//
// Where: the deepest level of the temporal stack at a small batch -- MODEL.HORIZON = 16 and one or two scenes per tick, what the
// reference drives with (interact.py:115-168): seven Conv1d(512, 512, 5) + GroupNorm + Mish on 2 samples x 2 positions
// (downs.3's last three convs and the two mid blocks, modeling/temporal.py:46-55,217-231).  As launches each of them is 8
// workgroups that stream a 393 KB weight slab through one CU behind ~4 us of launch, staging and epilogue latency: 10.9 us per
// layer, 25 % of a denoising step (profiles/r04_tick_free_b1.txt).
//
// How: stage k of the pipeline = P = C / 16 workgroups, one per CU, each owning 16 output channels of layer k.
//   * A workgroup's weight share (live taps x C x 16 channels, fp16 hi / lo pairs: 96 KB) goes into LDS at kernel entry; for every
//     stage but the first that load runs while the stage waits for its input.
//   * A stage publishes RAW conv sums + bias as EPOCH-TAGGED 16-byte records (round 6): a unit = three channels of one row + the
//     tag of THIS forward ([rows x L][6 units] per workgroup), written with ONE write-through 16-byte store each -- a unit is in
//     memory whole or not at all, so the tag IS the signal: no drain, no barrier, no counter, no atomic on the producer's side.
//     The tag is a bijective hash of a forward number drawn from a monotonic counter the LIBRARY owns (pipe_epoch_counter: one
//     device word per GPU; the launch that clears a forward's ticket words draws the number and leaves it in the workspace's
//     ticket area for this kernel to read): whatever an earlier forward of the process left in the records region carries an
//     older number, nothing else writes that region (the tail of the executor's scratch), and the caller's workspace holds no
//     state that must survive between forwards.  (Bytes the CALLER leaves there pose as a record with probability 2^-32 per unit;
//     the number wraps after 2^32 forwards of a process -- two weeks of back-to-back ticks -- onto tags whose records have been
//     overwritten 2^32 times since.)
//   * The next stage's threads poll the units THEY need (sc1 loads: they bypass the L2; the producers sit on other XCDs) until
//     every tag is this forward's, and the stage forms its input ITSELF: GroupNorm (two-pass statistics over the group's
//     channels x positions, which it holds completely) -> Mish -> + time bias or + residual, exactly the epilogue the producer
//     would have run -- but the producer could not: a GroupNorm group spans four workgroups.  So a layer costs ONE hand-off.
//   * A block's output (a later residual, the level's skip, the run's result) is written by rank 0 of the stage that formed it;
//     rank 0 drains those stores before it publishes its own records, and every reader of the tensor has seen records that were
//     published after rank 0's (two hand-offs later at the earliest).  The finisher (one workgroup) forms and writes the last
//     layer's output.
// Deadlock-free without a cooperative launch: a stage waits only for workgroups with LOWER ids, which every XCD dispatches first;
// every spin is bounded (a timeout leaves the result wrong, never the GPU hung).  Arithmetic = tconv_hs.hip's split-fp16 scheme
// (x = hi + 2^-11 lo, three v_mfma_f32_16x16x32_f16 per product, fp32 accumulation); fixed summation order: bit-reproducible.
// Prototype and its measurement: tools/micro/coop_pipeline.hip, profiles/r04_coop_pipeline.txt (5.4 us per layer).
#include <algorithm>
#include <mutex>

#include "adx_common.h"
#include "tconv_pipe.h"

namespace adx {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr float kPipeLoScale = 2048.f;
constexpr int kPipeNT = 256;
constexpr int kPipeUnits = 6;        // 16-byte record units per (row, workgroup): five of three channels + one of one, each with its tag
constexpr int kPipeRecWords = kPipeRows * kPipeUnits * 4;      // words per (stage, workgroup)
__host__ __device__ __forceinline__ unsigned pipe_tag(unsigned n) { return n * 2654435761u + 0x7F4A7C15u; }   // bijective: numbers differ, tags differ
constexpr int kPipeQ = 9;            // quads per thread of the formed input: 16 rows x C / 4 over 256 threads, C <= 576

// Mish with the hardware exp2 / rcp (same closed form as mish_f, ~3e-7 relative: tconv_hs.hip's fast epilogue uses the same)
__device__ __forceinline__ float pipe_mish(float x) {
  if (x > 20.f) return x;
  const float e = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
  const float n = e * (e + 2.f);
  return x * n * __builtin_amdgcn_rcpf(n + 2.f);
}

// sum over the 16 lanes of a DPP row, left in every lane of the row (no LDS: four dependent v_add with a DPP operand)
template <int CTRL>
__device__ __forceinline__ float pipe_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float pipe_row_sum(float v) {
  v += pipe_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
  v += pipe_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
  v += pipe_dpp<0x141>(v);     // row_half_mirror
  v += pipe_dpp<0x140>(v);     // row_mirror
  return v;
}

void pipe_live_taps(int taps, int pad, int L, int* tap0, int* ntap) {
  const int lo = std::max(0, pad - (L - 1)), hi = std::min(taps - 1, pad + L - 1);
  *tap0 = lo;
  *ntap = hi - lo + 1;
}

static size_t pipe_lds_bytes(int C, int M, int ntap) {
  const int steps = ntap * (C / 32), pitch = C / 8 * 2 + 2;
  return (size_t)steps * 128 * 16 + (size_t)(M + 1) * pitch * 16 + (size_t)M * C * 4 + 4 * 256 * 4 + 64 * 2 * 4;
}

bool pipe_shape_ok(int C, int L, int rows, int taps, int pad, int groups) {
  if (C < 64 || C % 32 != 0 || groups < 1 || C % groups != 0) return false;
  if (L < 1 || rows < 1 || rows * L > kPipeRows || rows * groups > 64) return false;
  if (taps < 1 || taps > 8 || pad < 0 || pad >= taps) return false;
  int t0, nt;
  pipe_live_taps(taps, pad, L, &t0, &nt);
  if (nt < 1) return false;
  const int P = C / kPipeCh;
  if ((kPipeMaxStages - 1) * P + 1 > 256) return false;             // every workgroup of the longest run on a CU of its own
  return pipe_lds_bytes(C, rows * L, nt) <= kPipeMaxLds;
}

size_t pipe_record_floats(int n_conv, int P) { return (size_t)n_conv * P * kPipeRecWords; }

size_t pipe_packed_floats(int C, int taps, int pad, int L) {
  int t0, nt;
  pipe_live_taps(taps, pad, L, &t0, &nt);
  return (size_t)(C / kPipeCh) * nt * (C / 32) * 128 * 4;          // [rank][step][plane][64 lanes] x 16 bytes
}

// weight image: [rank = cout / 16][step = (live tap, cin / 32)][plane hi | lo][lane][8 halfs]; lane (n = lane & 15, kg = lane >> 4)
// holds W[16 rank + n][32 c32 + 8 kg + j][tap] -- the B fragment of v_mfma_f32_16x16x32_f16
__global__ void __launch_bounds__(256) pipe_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ packed, int C, int taps,
                                                         int tap0, int steps, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int j = idx & 7, lane = (idx >> 3) & 63, plane = (idx >> 9) & 1;
  const size_t blk = idx >> 10;
  const int k = blk % steps, rank = blk / steps;
  const int c32n = C / 32, ti = k / c32n, c32 = k - ti * c32n;
  const int n = rank * kPipeCh + (lane & 15), cin = c32 * 32 + (lane >> 4) * 8 + j;
  const float v = w[((size_t)n * C + cin) * taps + tap0 + ti];
  const _Float16 hi = (_Float16)v;
  packed[idx] = plane == 0 ? hi : (_Float16)((v - (float)hi) * kPipeLoScale);
}

// the same image from the K-split kernel's image of the layer (tconv_hs.hip: [cout / 32][tap x cin / 16][plane][64 lanes][8 halfs],
// lane (n & 31, half of the 16-channel block), which every piped layer has anyway): a pure permutation of 16-byte cells, so the
// pipeline's images can be made where they are first needed (the inference forward) instead of at every weight update
struct PipeRepackJobs { const u32x4* hs[8]; u32x4* packed[8]; };

__global__ void __launch_bounds__(256) pipe_repack_kernel(const PipeRepackJobs jobs, int C, int taps, int tap0, int steps, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;            // one 16-byte cell: (rank, step, plane, lane)
  if (idx >= total) return;
  const u32x4* __restrict__ hs = jobs.hs[blockIdx.y];
  u32x4* __restrict__ packed = jobs.packed[blockIdx.y];
  const int lane = idx & 63, plane = (idx >> 6) & 1;
  const size_t blk = idx >> 7;
  const int k = blk % steps, rank = blk / steps;
  const int c32n = C / 32, ti = k / c32n, c32 = k - ti * c32n;
  const int n = rank * kPipeCh + (lane & 15), cin = c32 * 32 + (lane >> 4) * 8;
  const int ncb = C / 16, nkb = taps * ncb;
  const int kb = (tap0 + ti) * ncb + (cin >> 4), ln = (((cin >> 3) & 1) << 5) | (n & 31);
  packed[idx] = hs[((size_t)(n >> 5) * nkb + kb) * 128 + plane * 64 + ln];
}

// n <= 8 layers of one shape in ONE launch (adx_unet_pack: the pipeline run's images are part of the packed buffer's contents,
// made whenever the K-split images they are re-laid from are made)
int pipe_repack_from_hs_many(const float* const* hs_images, float* const* packed, int n, int C, int taps, int pad, int L, hipStream_t s) {
  ADX_REQUIRE(n >= 1 && n <= 8, "pipe_repack_from_hs_many: %d layers (1..8)", n);
  (void)pipe_epoch_counter(true);      // (adx_unet_pack: never inside a capture) the forward-number word of this device exists from here on
  int t0, nt;
  pipe_live_taps(taps, pad, L, &t0, &nt);
  const int steps = nt * (C / 32);
  const size_t total = (size_t)(C / kPipeCh) * steps * 2 * 64;
  PipeRepackJobs jobs;
  for (int i = 0; i < 8; ++i) {
    jobs.hs[i] = reinterpret_cast<const u32x4*>(hs_images[i < n ? i : 0]);
    jobs.packed[i] = reinterpret_cast<u32x4*>(packed[i < n ? i : 0]);
  }
  pipe_repack_kernel<<<dim3((unsigned)((total + 255) / 256), (unsigned)n), dim3(256), 0, s>>>(jobs, C, taps, t0, steps, total);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

int pipe_repack_from_hs(const float* hs_image, float* packed, int C, int taps, int pad, int L, hipStream_t s) {
  return pipe_repack_from_hs_many(&hs_image, &packed, 1, C, taps, pad, L, s);
}

int pipe_pack(const float* w, float* packed, int C, int taps, int pad, int L, hipStream_t s) {
  int t0, nt;
  pipe_live_taps(taps, pad, L, &t0, &nt);
  const int steps = nt * (C / 32);
  const size_t total = (size_t)(C / kPipeCh) * steps * 2 * 64 * 8;
  pipe_pack_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(w, reinterpret_cast<_Float16*>(packed), C, taps, t0,
                                                                              steps, total);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

#ifdef ADX_PIPE_TRACE
// diagnostic build (-DADX_PIPE_TRACE): ranks 0 and P - 1 of every stage leave the 100 MHz real-time counter (one clock for all
// XCDs) at their phase boundaries; tools/pipe_trace.py reads them back through adx_pipe_trace_read
__device__ unsigned long long g_pipe_trace[kPipeMaxStages * 2 * 16];
#define PIPE_STAMP(slot)                                                                                          \
  do {                                                                                                            \
    if (tid == 0 && (rank == 0 || rank == P - 1)) g_pipe_trace[(stage * 2 + (rank != 0)) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define PIPE_STAMP(slot)
#endif

__global__ void __launch_bounds__(kPipeNT) tconv_pipe_kernel(const PipeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // The kernel-argument segment is host memory on this platform: every 64-byte line of it a wave touches for the first time is
  // a ~2 us round trip, and the compiler loads fields where they are used -- behind the spin wait, in series, on the pipeline's
  // critical path.  So this workgroup's stage descriptor and the header go to LDS NOW (all lines in flight together) and
  // everything below reads that copy.
  __shared__ PipeStage S_lds;
  __shared__ int hdr_lds[16];
  __shared__ int got;                    // 1: fine; 2: a thread gave up waiting for a record of the producing stage
  __shared__ float rs1[kPipeRows * 8], rs2[kPipeRows * 8];     // fast GroupNorm statistics: per (row, group) partial sums
  __shared__ float rec_lds[kPipeRows * kPipeCh];               // this workgroup's sums, regrouped into three-channel units
  const int P = a.P;
  const int stage = blockIdx.x / P, rank = blockIdx.x - stage * P;      // the finisher: stage == n_conv, rank 0
  const unsigned epoch_n = *a.epoch;     // this forward's number (requested now: needed behind the weight load at the earliest)
  PIPE_STAMP(0);
#if defined(__HIP_DEVICE_COMPILE__)
  {
    typedef const __attribute__((address_space(4))) int* kernarg_words;
    kernarg_words kraw = (kernarg_words)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int SW = (int)(sizeof(PipeStage) / 4), HW = (int)((sizeof(PipeArgs) - offsetof(PipeArgs, n_conv)) / 4);
    static_assert(HW <= 16 + 4, "header words");
    if (tid == 255) got = 1;
    if (tid < SW) reinterpret_cast<int*>(&S_lds)[tid] = kraw[stage * SW + tid];
    else if (tid >= 64 && tid < 64 + 12) hdr_lds[tid - 64] = kraw[(int)(offsetof(PipeArgs, n_conv) / 4) + tid - 64];
  }
#endif
  __syncthreads();
  const PipeStage S = S_lds;
  PIPE_STAMP(1);
  const int n_conv = hdr_lds[0], C = hdr_lds[1], L = hdr_lds[2], rows = hdr_lds[3];
  const int groups = hdr_lds[5], pad = hdr_lds[7], tap0 = hdr_lds[8], ntap = hdr_lds[9];
  const float eps = __builtin_bit_cast(float, hdr_lds[10]);
  const int M = rows * L;
  const bool conv = stage < n_conv;
  const int ncell = C >> 3, pitch = ncell * 2 + 2;                       // 16-byte cells per row (+2: rows land on different bank slots)
  const int steps = ntap * (C >> 5);
  u32x4* wl = reinterpret_cast<u32x4*>(smem_raw);                        // [step][plane][lane]
  u32x4* cells = wl + steps * 128;                                       // [M + 1 rows][pitch]; row M is all zero
  float* xf = reinterpret_cast<float*>(cells + (M + 1) * pitch);         // the formed input, [M][C] fp32
  float* red = xf + M * C;                                               // [4 waves][256]
  float* gst = red + 4 * 256;                                            // [rows x groups][mean, rstd]

  // ---- (1) this workgroup's weight share into LDS; the zero row -----------------------------------------------------------
  // Eight 16-byte loads in flight per thread (one at a time the 24 round trips of a 96 KB share took 7 us).  Stage k needs its
  // weights ~7 k us after the launch, stage 0 NOW: the later stages hold their loads back a little so that stage 0's 3 MB do not
  // queue behind the other 18 MB in the fabric.
  if (conv) {
    for (int w = 0; w < stage; ++w) __builtin_amdgcn_s_sleep(60);
    const u32x4* src = reinterpret_cast<const u32x4*>(S.w) + (size_t)rank * steps * 128;
    const int n = steps * 128;
    for (int i0 = tid; i0 < n; i0 += 8 * kPipeNT) {
      u32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = src[min(i0 + u * kPipeNT, n - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * kPipeNT < n) wl[i0 + u * kPipeNT] = t[u];
    }
  }
  for (int i = tid; i < pitch; i += kPipeNT) cells[M * pitch + i] = u32x4{0u, 0u, 0u, 0u};
  PIPE_STAMP(2);

  // ---- (2) everything the formed input needs that does not come from the producer -- the producing conv's GroupNorm affine, the
  //      time-bias rows, a residual an EARLIER LAUNCH wrote -- is requested BEFORE the wait, and every address used behind the
  //      wait is computed here (no integer division on the critical path).  A thread owns quads q = tid, tid + 256, ... of 4
  //      consecutive channels of one row (at most kPipeQ: 16 rows x C / 4 quads over 256 threads); xf offset of quad q = 4 q.
  const int c4n = C >> 2, nq = M * c4n;
  const int cg = C / groups, log2L = 31 - __builtin_clz(L);              // L is a power of two (the executor checks)
  const bool from_records = S.in == nullptr;
  constexpr int kOut = 0x7FFFFFF0;                                        // out-of-range offset: loads return 0, stores are dropped
  f32x4 ga[kPipeQ], be[kPipeQ], ad[kPipeQ];
  int roff[kPipeQ], rsel[kPipeQ], aoff[kPipeQ], poff[kPipeQ], sgi[kPipeQ], mg[kPipeQ], mg0[kPipeQ];
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(S.add), 0, S.add_kind >= 2 ? M * C * 4 : 0, 0x00020000);
#pragma unroll
  for (int i = 0; i < kPipeQ; ++i) {
    ga[i] = be[i] = ad[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    roff[i] = aoff[i] = poff[i] = kOut;
    rsel[i] = 0;
    sgi[i] = 0;
    mg[i] = 0;
    mg0[i] = 0;
    const int q = tid + i * kPipeNT;
    if (i * kPipeNT >= nq) continue;                                      // uniform
    if (from_records && q < nq) {
      const int m = q / c4n, c = (q - m * c4n) * 4, sb = m >> log2L, l = m & (L - 1);
      ga[i] = *reinterpret_cast<const f32x4*>(S.gamma + c);
      be[i] = *reinterpret_cast<const f32x4*>(S.beta + c);
      // the quad's channels c' = c & 15 in {0, 4, 8, 12} of producer c >> 4 lie in the units c' / 3 and c' / 3 + 1 of row m
      roff[i] = ((((c >> 4) * kPipeRows + m) * kPipeUnits + (c & 15) / 3) * 4) * 4;
      rsel[i] = (c >> 2) & 3;
      sgi[i] = sb * groups + c / cg;                                      // (a quad never straddles a group: cg % 4 == 0)
      mg[i] = m * groups + c / cg;
      mg0[i] = (m - l) * groups + c / cg;                                 // the sample's first row, same group
      if (S.add_kind == 1) ad[i] = *reinterpret_cast<const f32x4*>(S.add + (size_t)sb * S.add_stride + c);
      if (S.add_kind == 2) aoff[i] = (((sb * C) + c) * L + l) * 4;        // [rows][C][L]: the quad's channels are L floats apart
      if (S.add_kind == 3) aoff[i] = q * 16;                              // [rows x L][C]
      if (S.pub_kind == 1 && rank == 0) poff[i] = (((sb * C) + c) * L + l) * 4;
      if (S.pub_kind == 2 && rank == 0) poff[i] = q * 16;
      if (S.add_kind == 2 && S.add_early) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ad[i][jj] = u2f(__builtin_amdgcn_raw_buffer_load_b32(ars, aoff[i] + jj * L * 4, 0, 0));
      }
    }
  }

  // ---- (3) wait for the producing stage -----------------------------------------------------------------------------------
  // Only the stage whose producer is already running polls its records tightly (256 threads x a few 16-byte loads per round):
  // until the stage two hops up has started to publish, ONE thread looks at one of THAT stage's units every ~2 us (hundreds of
  // tight pollers in the fabric delayed the producers' own stores).  Every spin is bounded.
  const unsigned tag = pipe_tag(epoch_n);
  if (tid == 0 && stage >= 2) {
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
        a.records + (size_t)(stage - 2) * P * kPipeRecWords, 0, P * kPipeRecWords * 4, 0x00020000);
    unsigned spins = 0;
    while (__builtin_amdgcn_raw_buffer_load_b32(crs, ((P - 1) * kPipeRecWords + 3) * 4, 0, 16) != tag) {     // row 0, unit 0 of the last rank
      __builtin_amdgcn_s_sleep(64);
      if (++spins > (1u << 18)) break;
    }
  }
  if (stage >= 2) __syncthreads();
  PIPE_STAMP(3);

  // ---- (4) form the input: [M][C] fp32 in xf ---------------------------------------------------------------------------------
  if (!from_records) {
    // stage 0: a finished activation [rows][C][L] of an earlier launch
    for (int q = tid; q < nq; q += kPipeNT) {
      const int m = q / c4n, c = (q - m * c4n) * 4, sb = m >> log2L, l = m & (L - 1);
      const float* src = S.in + ((size_t)sb * C + c) * L + l;
      f32x4 v;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) v[jj] = src[(size_t)jj * L];
      *reinterpret_cast<f32x4*>(xf + q * 4) = v;
    }
    __syncthreads();
  } else {
    // the producer's records (raw conv sums + bias, [P][16 rows][16 channels], written through to memory by other XCDs) and,
    // where the addend is a residual an earlier stage of THIS launch wrote, that tensor: all loads in flight together
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
        a.records + (size_t)(stage - 1) * P * kPipeRecWords, 0, P * kPipeRecWords * 4, 0x00020000);
    u32x4 rv[kPipeQ];
    {
      // poll: both units of every quad of this thread, all loads of a round in flight together; a unit is this forward's when its
      // fourth word is the tag (the producer wrote the 16 bytes with one store).  Quads past the batch carry the out-of-range offset:
      // their loads return zero and are not waited for.
      u32x4 ua[kPipeQ], ub[kPipeQ];
      unsigned spins = 0;
      bool all;
      do {
        all = true;
#pragma unroll
        for (int i = 0; i < kPipeQ; ++i) {
          if (i * kPipeNT >= nq) continue;                                  // uniform
          ua[i] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff[i], 0, 16);                                   // sc1
          ub[i] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff[i] == kOut ? kOut : roff[i] + 16, 0, 16);
        }
#pragma unroll
        for (int i = 0; i < kPipeQ; ++i) {
          if (i * kPipeNT >= nq) continue;
          all = all && (roff[i] == kOut || (ua[i][3] == tag && ub[i][3] == tag));
        }
        if (!all) {
          if (++spins > (1u << 20)) { got = 2; break; }                    // bounded: a wrong result, never a hung GPU
          __builtin_amdgcn_s_sleep(1);
        }
      } while (!all);
#pragma unroll
      for (int i = 0; i < kPipeQ; ++i) {
        rv[i] = u32x4{0u, 0u, 0u, 0u};
        if (i * kPipeNT >= nq) continue;
        const int sl = rsel[i];             // c' = 4 sl: the quad is {A0 A1 A2 B0}, {A1 A2 B0 B1}, {A2 B0 B1 B2}, {A0 A1 A2 B0}
        const uint32_t A0 = ua[i][0], A1 = ua[i][1], A2 = ua[i][2], B0 = ub[i][0], B1 = ub[i][1], B2 = ub[i][2];
        rv[i][0] = sl == 1 ? A1 : (sl == 2 ? A2 : A0);
        rv[i][1] = sl == 1 ? A2 : (sl == 2 ? B0 : A1);
        rv[i][2] = sl == 1 ? B0 : (sl == 2 ? B1 : A2);
        rv[i][3] = sl == 1 ? B1 : (sl == 2 ? B2 : B0);
      }
    }
    // where the addend is a residual an earlier stage of THIS launch wrote, that tensor (its writer drained it before publishing
    // records this stage's producers had to see first)
#pragma unroll
    for (int i = 0; i < kPipeQ; ++i) {
      if (i * kPipeNT >= nq) continue;                                    // uniform
      if (S.add_kind == 3) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(ars, aoff[i], 0, 16);
        ad[i] = f32x4{u2f(t[0]), u2f(t[1]), u2f(t[2]), u2f(t[3])};
      } else if (S.add_kind == 2 && !S.add_early) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          ad[i][jj] = u2f(__builtin_amdgcn_raw_buffer_load_b32(ars, aoff[i] == kOut ? kOut : aoff[i] + jj * L * 4, 0, 16));
      }
    }
#pragma unroll
    for (int i = 0; i < kPipeQ; ++i)
      if (tid + i * kPipeNT < nq) *reinterpret_cast<u32x4*>(xf + (tid + i * kPipeNT) * 4) = rv[i];
    __syncthreads();
    if (got == 2) {
      // A producer never arrived (a hung or evicted workgroup: nothing a correct run produces).  This stage publishes nothing, so
      // every later stage times out in turn; the finisher makes the failure LOUD instead of leaving whatever the output buffer
      // held: the run's output becomes NaN, and the pinned host word makes the next forward of the process report it.
      if (!conv && S.pub != nullptr)
        for (int q = tid; q < M * C; q += kPipeNT) S.pub[q] = __builtin_nanf("");
      if (tid == 0 && a.fault != nullptr) __hip_atomic_store(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    PIPE_STAMP(4);
    const int n_sg = rows * groups, per = cg * L;
    float qmean[kPipeQ], qrstd[kPipeQ];
    const bool fast_stats = cg == 64 && (c4n & 15) == 0 && groups <= 8;
    if (fast_stats) {
      // 16 channels' quads of one row and group sit in 16 consecutive lanes: the statistics come out of the registers -- quad sum,
      // 16-lane shuffle tree, one LDS word per (row, group), the L rows of a sample added by every reader in row order (two passes,
      // fixed order: bit-reproducible)
      const float inv = 1.f / (float)per;
#pragma unroll
      for (int i = 0; i < kPipeQ; ++i) {
        if (i * kPipeNT >= nq) break;                                     // uniform: the quads past the batch cost nothing
        const float t = pipe_row_sum((u2f(rv[i][0]) + u2f(rv[i][1])) + (u2f(rv[i][2]) + u2f(rv[i][3])));
        if ((tid & 15) == 0 && tid + i * kPipeNT < nq) rs1[mg[i]] = t;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kPipeQ; ++i) {
        if (i * kPipeNT >= nq) break;
        float t = 0.f;
        for (int l = 0; l < L; ++l) t += rs1[mg0[i] + l * groups];
        qmean[i] = t * inv;
        float d2 = 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { const float dd = u2f(rv[i][jj]) - qmean[i]; d2 += dd * dd; }
        d2 = pipe_row_sum(d2);
        if ((tid & 15) == 0 && tid + i * kPipeNT < nq) rs2[mg[i]] = d2;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kPipeQ; ++i) {
        if (i * kPipeNT >= nq) break;
        float t = 0.f;
        for (int l = 0; l < L; ++l) t += rs2[mg0[i] + l * groups];
        qrstd[i] = __builtin_amdgcn_rsqf(t * inv + eps);                  // hardware rsq: 1 ulp
      }
    } else {
    // GroupNorm statistics, two passes, 16 lanes per (sample, group): the group's channels x positions are all here
    for (int sg = tid >> 4; sg < n_sg; sg += kPipeNT >> 4) {
      const int sb = sg / groups, g = sg - sb * groups, part = tid & 15;
      const float* xg = xf + (sb * L) * C + g * cg;                       // element e: position e & (L - 1), channel e >> log2L
      float sm = 0.f;
      for (int e = part; e < per; e += 16) sm += xg[(e & (L - 1)) * C + (e >> log2L)];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) sm += __shfl_xor(sm, off, 16);
      const float mean = sm / (float)per;
      float v = 0.f;
      for (int e = part; e < per; e += 16) {
        const float dd = xg[(e & (L - 1)) * C + (e >> log2L)] - mean;
        v += dd * dd;
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 16);
      if (part == 0) {
        gst[2 * sg] = mean;
        gst[2 * sg + 1] = 1.0f / sqrtf(v / (float)per + eps);
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPipeQ; ++i) { qmean[i] = gst[2 * sgi[i]]; qrstd[i] = gst[2 * sgi[i] + 1]; }
    }
    PIPE_STAMP(5);
    // GroupNorm affine -> Mish -> + time bias | + residual; rank 0 writes the result where a later residual / skip / the caller wants it
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(S.pub, 0, S.pub_kind != 0 ? M * C * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs2 = __builtin_amdgcn_make_buffer_rsrc(S.pub2, 0, S.pub2 != nullptr ? M * C * 4 : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < kPipeQ; ++i) {
      if (i * kPipeNT >= nq) break;                                       // uniform
      if (tid + i * kPipeNT >= nq) continue;
      const float mean = qmean[i], rstd = qrstd[i];
      f32x4 x;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) x[jj] = pipe_mish((u2f(rv[i][jj]) - mean) * rstd * ga[i][jj] + be[i][jj]) + ad[i][jj];
      *reinterpret_cast<f32x4*>(xf + (tid + i * kPipeNT) * 4) = x;
      if (S.pub_kind == 1) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          __builtin_amdgcn_raw_buffer_store_b32(f2u(x[jj]), prs, poff[i] == kOut ? kOut : poff[i] + jj * L * 4, 0, 16);
      } else if (S.pub_kind == 2) {
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{f2u(x[0]), f2u(x[1]), f2u(x[2]), f2u(x[3])}, prs, poff[i], 0, 16);
      }
      if (S.pub2 != nullptr && rank == 0)       // a second copy in the pipeline's own layout (a later stage's residual reads it fast)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{f2u(x[0]), f2u(x[1]), f2u(x[2]), f2u(x[3])}, prs2, (tid + i * kPipeNT) * 16, 0, 16);
    }
    // rank 0's copies of the formed input are in memory before the barrier that precedes its record stores (a microsecond of
    // conv later): whoever sees those records may read the copies
    if (rank == 0 && (S.pub_kind != 0 || S.pub2 != nullptr)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PIPE_STAMP(6);
  }
  if (!conv) return;                                                      // the finisher is done: the kernel boundary publishes its stores

  // ---- (5) split into hi / lo cells: item = (row, 8-channel octet) ---------------------------------------------------------------
  for (int it = tid; it < M * ncell; it += kPipeNT) {
    const int m = it / ncell, oc = it - m * ncell;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(xf + it * 8), v1 = *reinterpret_cast<const f32x4*>(xf + it * 8 + 4);
    h8 hi, lo;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const float x = jj < 4 ? v0[jj & 3] : v1[jj & 3];
      const _Float16 h = (_Float16)x;
      hi[jj] = h;
      lo[jj] = (_Float16)((x - (float)h) * kPipeLoScale);
    }
    cells[m * pitch + 2 * oc] = __builtin_bit_cast(u32x4, hi);
    cells[m * pitch + 2 * oc + 1] = __builtin_bit_cast(u32x4, lo);
  }
  __syncthreads();
  PIPE_STAMP(7);

  // ---- (6) the conv: one 16 x 16 tile; per live tap the 32-channel K-steps are dealt to the four waves, four steps' fragments
  //      requested before the first MFMA of the group ---------------------------------------------------------------------------
  f32x4 accm = {0.f, 0.f, 0.f, 0.f}, accx = {0.f, 0.f, 0.f, 0.f};
  {
    const int r16 = lane & 15, kg = lane >> 4, c32n = C >> 5;
    const int sb = r16 >> log2L, l = r16 & (L - 1);
    for (int ti = 0; ti < ntap; ++ti) {
      const int ip = l + tap0 + ti - pad;
      const bool ok = r16 < M && (unsigned)ip < (unsigned)L;
      const u32x4* rowp = cells + (ok ? sb * L + ip : M) * pitch + 2 * kg;
      const u32x4* wrow = wl + (ti * c32n) * 128 + lane;
#pragma unroll 4
      for (int c32 = wave; c32 < c32n; c32 += 4) {
        const h8 ah = __builtin_bit_cast(h8, rowp[8 * c32]);
        const h8 al = __builtin_bit_cast(h8, rowp[8 * c32 + 1]);
        const h8 wh = __builtin_bit_cast(h8, wrow[c32 * 128]);
        const h8 wlo = __builtin_bit_cast(h8, wrow[c32 * 128 + 64]);
        accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh, accm, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wlo, accx, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh, accx, 0, 0, 0);
      }
    }
  }
  *reinterpret_cast<f32x4*>(red + wave * 256 + lane * 4) = accm + accx * (1.f / kPipeLoScale);
  __syncthreads();
  PIPE_STAMP(8);

  // ---- (7) publish: raw sums + bias of this workgroup's [16 rows][16 channels] as tagged units, one write-through 16-byte store each ----
  float* rec = rec_lds;
  if (tid < 64) {
    const int col = tid & 15, kg = tid >> 4;                              // accumulator lane: channel col, rows 4 kg .. 4 kg + 3
    const f32x4 sum = *reinterpret_cast<const f32x4*>(red + tid * 4) + *reinterpret_cast<const f32x4*>(red + 256 + tid * 4) +
                      (*reinterpret_cast<const f32x4*>(red + 512 + tid * 4) + *reinterpret_cast<const f32x4*>(red + 768 + tid * 4));
    const float b = S.bias[rank * kPipeCh + col];
#pragma unroll
    for (int i = 0; i < 4; ++i) rec[(4 * kg + i) * kPipeCh + col] = sum[i] + b;
  }
  __syncthreads();
  if (tid < kPipeRows * kPipeUnits) {
    const int row = tid / kPipeUnits, u = tid - row * kPipeUnits;
    if (row < M) {
      const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
          a.records + ((size_t)stage * P + rank) * kPipeRecWords, 0, kPipeRecWords * 4, 0x00020000);
      const float* r3 = rec + row * kPipeCh + 3 * u;
      const u32x4 unit = {f2u(r3[0]), u < 5 ? f2u(r3[1]) : 0u, u < 5 ? f2u(r3[2]) : 0u, tag};
      __builtin_amdgcn_raw_buffer_store_b128(unit, wrs, tid * 16, 0, 16);            // sc1; unit (row, u) sits at (row * 6 + u) * 16 = tid * 16
    }
  }
  PIPE_STAMP(9);
}

// The forward numbers: one device word per GPU, owned by the library for the life of the process, only ever incremented (by the
// launch that opens a forward: tconv_chain's workgroup 0 or tickets_reset_kernel).  Allocated where no stream capture can be
// open (adx_unet_pack); null if that never happened or failed -- the executor then keeps the launch chain.
static unsigned* g_epoch_ctr[64] = {};
static std::mutex g_epoch_mu;

unsigned* pipe_epoch_counter(bool allocate) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_epoch_mu);
  unsigned*& p = g_epoch_ctr[dev & 63];
  if (p == nullptr && allocate) {
    void* d = nullptr;
    if (hipMalloc(&d, 64) == hipSuccess) {
      if (hipMemset(d, 0, 64) == hipSuccess) p = (unsigned*)d;
      else (void)hipFree(d);
    }
  }
  return p;
}

// the launch that opens a forward where no chained level does: clears the ticket words and draws the forward's number
__global__ void __launch_bounds__(256) tickets_reset_kernel(unsigned* words, int n, unsigned* epoch_ctr, int epoch_slot) {
  const int tid = threadIdx.x;
  if (tid < n && tid != epoch_slot) words[tid] = 0u;
  if (tid == epoch_slot) words[tid] = epoch_ctr != nullptr ? __hip_atomic_fetch_add(epoch_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u : 0u;
}

int pipe_tickets_reset(unsigned* words, int n, int epoch_slot, hipStream_t s) {
  ADX_REQUIRE(words != nullptr && n <= 256 && epoch_slot >= 0 && epoch_slot < n, "pipe_tickets_reset: bad arguments");
  tickets_reset_kernel<<<dim3(1), dim3(256), 0, s>>>(words, n, pipe_epoch_counter(false), epoch_slot);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

static unsigned* g_fault_host = nullptr;      // pinned, mapped: the kernel's store lands in host memory
static unsigned* g_fault_dev = nullptr;
static std::atomic<int> g_fault_state{0};     // 0 not tried, 1 being allocated, 2 ready, 3 failed

unsigned* pipe_fault_word() {
  int st = g_fault_state.load(std::memory_order_acquire);
  if (st == 0) {
    int expect = 0;
    if (g_fault_state.compare_exchange_strong(expect, 1)) {
      void* h = nullptr;
      void* d = nullptr;
      bool ok = hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess;
      if (ok) {
        g_fault_host = (unsigned*)h;
        g_fault_dev = (unsigned*)d;
        *g_fault_host = 0;
      }
      g_fault_state.store(ok ? 2 : 3, std::memory_order_release);
    }
    while ((st = g_fault_state.load(std::memory_order_acquire)) == 1) {}
  }
  return st == 2 ? g_fault_dev : nullptr;
}

unsigned pipe_fault_take() {
  if (g_fault_state.load(std::memory_order_acquire) != 2) return 0;
  volatile unsigned* p = g_fault_host;
  const unsigned v = *p;
  if (v != 0) *p = 0;
  return v;
}

int pipe_launch(const PipeArgs& a, hipStream_t s) {
  ADX_REQUIRE(a.n_conv >= 1 && a.n_conv < kPipeMaxStages && a.P == a.C / kPipeCh && a.records && a.epoch,
              "tconv_pipe: bad argument block");
  ADX_REQUIRE(pipe_shape_ok(a.C, a.L, a.rows, a.taps, a.pad, a.groups), "tconv_pipe: shape outside the kernel's rules");
  ADX_REQUIRE((a.C / a.groups) % 4 == 0, "tconv_pipe: GroupNorm group width must be a multiple of 4");
  const size_t lds = pipe_lds_bytes(a.C, a.rows * a.L, a.ntap);
  static std::atomic<uint64_t> attr{0};
  if (DeviceOnce once{attr}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kPipeMaxLds));
    once.commit();
  }
  tconv_pipe_kernel<<<dim3((unsigned)(a.n_conv * a.P + 1)), dim3(kPipeNT), lds, s>>>(a);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

}  // namespace adx

#ifdef ADX_PIPE_TRACE
extern "C" int adx_pipe_trace_read(unsigned long long* host, int nwords) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(adx::g_pipe_trace), (size_t)nwords * 8);
}
#endif

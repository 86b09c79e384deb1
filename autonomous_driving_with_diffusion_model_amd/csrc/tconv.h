// Temporal (1-D) convolution on fp32 MFMA — geometry shared by the op-level C ABI and the
// UNet executor.
#pragma once
#include "adx_common.h"

namespace adx {

// Tiling derived from an adx_tconv_desc and the batch.  One workgroup (4 waves) owns
// `bt` samples x `ct` output channels for ALL positions, so every GroupNorm group it
// touches is complete inside the workgroup; the four waves split the K = taps*cin
// reduction and are summed through LDS in the epilogue.
struct TConvTile {
  int cin, cin_pad, ncb, nkb;  // channels, padded to 16, 16-channel blocks, taps*ncb
  int cout_pad;                // cout padded to 16
  int bt, mf;                  // samples per workgroup, 16-row MFMA fragments (bt*lout = 16*mf)
  int ct, nf;                  // channels per workgroup, 16-col fragments
  int pl, lp, rs, ck;          // left zero pad, per-sample LDS pitch, LDS row stride, channels per LDS chunk
  int ntiles;                  // workgroups along channels
  int nw;                      // waves per workgroup (K-split factor): 4, 8 or 16
  size_t lds_bytes;
};

int tconv_check(const adx_tconv_desc* d);
int tconv_tile(const adx_tconv_desc* d, int batch, TConvTile* t);
size_t tconv_packed_floats(const adx_tconv_desc* d);
int tconv_pack(const adx_tconv_desc* d, const float* w, float* packed, hipStream_t s);
int tconv_forward(const adx_tconv_desc* d, const adx_tconv_io* io, hipStream_t s);

// tile rules of the exact-fp32 MFMA kernel (power-of-two group widths, groups of a multiple of 64 elements)
bool tconv_exact_supported(const adx_tconv_desc* d);
// any-shape fallback (tconv_generic.hip): plain fp32 FMAs, one workgroup per (sample, GroupNorm group)
bool tconv_generic_supported(const adx_tconv_desc* d);
size_t tconv_generic_packed_floats(const adx_tconv_desc* d);
int tconv_generic_pack(const adx_tconv_desc* d, const float* w, float* packed, hipStream_t s);
int tconv_generic_forward(const adx_tconv_desc* d, const adx_tconv_io* io, hipStream_t s);

// split-fp16 MFMA implementation (tconv_hs.hip); tconv_pack / tconv_forward route to it when the geometry is
// supported and the descriptor does not ask for the exact-fp32 kernel
bool tconv_hs_supported(const adx_tconv_desc* d);
size_t tconv_hs_packed_floats(const adx_tconv_desc* d);
bool tconv_hs_kernel_image(const adx_tconv_desc* d);     // tconv_pack writes the K-split kernel's weight image for this layer
int tconv_hs_pack(const adx_tconv_desc* d, const float* w, float* packed, hipStream_t s);
int tconv_hs_forward(const adx_tconv_desc* d, const adx_tconv_io* io, hipStream_t s);
// two independent convolutions, in one launch where both run on the short-K kernel
int tconv_hs_forward_pair(const adx_tconv_desc* da, const adx_tconv_io* ioa, const adx_tconv_desc* db,
                          const adx_tconv_io* iob, hipStream_t s);

}  // namespace adx

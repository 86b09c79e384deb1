// extern "C" surface of libadx.so (declared in include/adx.h): thin wrappers over the adx::
// implementations plus the thread-local error string.
#include <stdarg.h>
#include <stdlib.h>

#include "tconv.h"

namespace adx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const DebugSwitches& debug_switches() {
  static const DebugSwitches sw = [] {
    DebugSwitches d;
    // exact one-character values: "10" is not "1" (the Python side compares whole strings too)
    auto is = [](const char* name, char v) { const char* e = getenv(name); return e != nullptr && e[0] == v && e[1] == '\0'; };
    d.conv_exact = is("ADX_CONV_EXACT", '1');
    d.wgrad_exact = is("ADX_WGRAD_EXACT", '1');
    d.tconv_exact = is("ADX_TCONV_EXACT", '1');
    d.unet_chain = !is("ADX_UNET_CHAIN", '0');
    d.unet_pipe = !is("ADX_UNET_PIPE", '0');
    d.conv_cells = !is("ADX_CONV_CELLS", '0');
    d.conv_vrow = !is("ADX_CONV_VROW", '0');
    d.train_cells = is("ADX_TRAIN_CELLS", '0') ? 0 : (is("ADX_TRAIN_CELLS", '1') ? 1 : (is("ADX_TRAIN_CELLS", '2') ? 2 : (is("ADX_TRAIN_CELLS", '3') ? 3 : (is("ADX_TRAIN_CELLS", '4') ? 4 : 5))));
    d.check_range = is("ADX_CHECK_RANGE", '1');
    d.hs_dma = !is("ADX_HS_DMA", '0');
    d.hs_persist = !is("ADX_HS_PERSIST", '0');
    d.wgrad_deterministic = is("ADX_WGRAD_DETERMINISTIC", '1');
    if (const char* e = getenv("ADX_CHAIN_MASK")) d.chain_mask = (unsigned)strtoul(e, nullptr, 0);
    if (const char* e = getenv("ADX_HS_MODE")) d.hs_mode = atoi(e);
    if (const char* e = getenv("ADX_RESNET_SPLIT_FROM")) d.resnet_split_from = atoi(e) < 0 ? -1 : atoi(e);
    if (const char* e = getenv("ADX_RESNET_STREAMS")) d.resnet_streams = atoi(e) < 1 ? 1 : (atoi(e) > 4 ? 4 : atoi(e));
    return d;
  }();
  return sw;
}

int embed_forward(const adx_embed_weights* w, int dim, const int64_t* t, int t_rows, const float* cond,
                  const float* feat, int feat_rows, int rows, float* time_embed, float* mish_cond, hipStream_t s);
int ddim_step(const adx_step_coef* c, const float* mo, const float* x, const float* z, const float* tgt,
              const float* mask, float* prev, float* x0, int b, int h, int d, hipStream_t s);
int ddpm_step(const adx_step_coef* c, const float* mo, const float* x, const float* z, const float* tgt,
              const float* mask, float* prev, float* x0, int b, int h, int d, hipStream_t s);
int add_noise(const float* x, const float* n, const int64_t* t, const float* sa, const float* sb, int n_train,
              float* out, int batch, int horizon, int dim, int zero_first, hipStream_t s);

int image_normalize(const uint8_t* src, float* dst, int n, int h, int w, const float* mean, const float* stdv,
                    hipStream_t s);

}  // namespace adx

extern "C" {

#ifndef ADX_SRC_HASH
#define ADX_SRC_HASH "unknown"
#endif
int adx_version(void) { return 2; }
// the "ADX_SRC_HASH=" tag lets a build script read the hash out of the file without loading the library
static const char kSrcHash[] = "ADX_SRC_HASH=" ADX_SRC_HASH;
const char* adx_source_hash(void) { return kSrcHash + 13; }
const char* adx_last_error(void) { return adx::g_err; }

size_t adx_tconv_packed_bytes(const adx_tconv_desc* d) {
  if (d == nullptr || adx::tconv_check(d) != ADX_OK) return 0;
  return adx::tconv_packed_floats(d) * sizeof(float);
}
int adx_tconv_pack(const adx_tconv_desc* d, const float* w, float* packed, adx_stream s) {
  return adx::tconv_pack(d, w, packed, (hipStream_t)s);
}
int adx_tconv_forward(const adx_tconv_desc* d, const adx_tconv_io* io, adx_stream s) {
  return adx::tconv_forward(d, io, (hipStream_t)s);
}
int adx_embed_forward(const adx_embed_weights* w, int32_t dim, const int64_t* t, int32_t t_rows, const float* cond,
                      const float* img_feature, int32_t feat_rows, int32_t rows, float* time_embed, float* mish_cond,
                      adx_stream s) {
  return adx::embed_forward(w, dim, t, t_rows, cond, img_feature, feat_rows, rows, time_embed, mish_cond,
                            (hipStream_t)s);
}
int adx_ddim_step(const adx_step_coef* c, const float* model_output, const float* sample, const float* noise,
                  const float* target, const float* mask, float* prev, float* x0, int32_t batch, int32_t horizon,
                  int32_t dim, adx_stream s) {
  return adx::ddim_step(c, model_output, sample, noise, target, mask, prev, x0, batch, horizon, dim, (hipStream_t)s);
}
int adx_ddpm_step(const adx_step_coef* c, const float* model_output, const float* sample, const float* noise,
                  const float* target, const float* mask, float* prev, float* x0, int32_t batch, int32_t horizon,
                  int32_t dim, adx_stream s) {
  return adx::ddpm_step(c, model_output, sample, noise, target, mask, prev, x0, batch, horizon, dim, (hipStream_t)s);
}
int adx_add_noise(const float* x, const float* noise, const int64_t* t, const float* sqrt_ab, const float* sqrt_1mab,
                  int32_t n_train, float* out, int32_t batch, int32_t horizon, int32_t dim, int32_t zero_first,
                  adx_stream s) {
  return adx::add_noise(x, noise, t, sqrt_ab, sqrt_1mab, n_train, out, batch, horizon, dim, zero_first, (hipStream_t)s);
}

int adx_image_normalize(const uint8_t* frame_hwc, float* out_nchw, int32_t n, int32_t h, int32_t w, const float* mean,
                        const float* stdv, adx_stream s) {
  return adx::image_normalize(frame_hwc, out_nchw, n, h, w, mean, stdv, (hipStream_t)s);
}

}  // extern "C"

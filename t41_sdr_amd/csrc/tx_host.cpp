// t41_sdr_amd/csrc/tx_host.cpp -- host side of the C ABI in include/t41tx.h (transmit exciter).
// Owns what the reference keeps in the exciter's static CMSIS instances (T41_SDR.ino:278-299,
// 877-888): per channel, resident in HBM between calls.  No CPU implementation exists here:
// without a HIP device every create / process call fails with T41RX_ERR_HIP.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <string>

#include "tx_internal.hpp"

using namespace t41;

struct t41tx_ctx {
  int device = 0;
  int nchan = 0;
  t41tx_params params{};
  float *d_state = nullptr;
  TxCoef *d_coef = nullptr;
  int16_t *d_in = nullptr, *d_outL = nullptr, *d_outR = nullptr;  // staging of the host-pointer entry
  size_t staging = 0;
};

namespace {
thread_local std::string g_tx_error;
int fail(int code, const char *msg) {
  g_tx_error = msg;
  return code;
}
struct Guard {
  int prev = -1;
  bool ok;
  explicit Guard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~Guard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
// every mode the receive side accepts: ExciterIQData() runs in all of them and only applies the TX
// IQ correction in LSB / USB (Exciter.cpp:117-140)
bool valid(const t41tx_params &p) { return (p.mode >= T41RX_DEMOD_USB && p.mode <= T41RX_DEMOD_NFM) || p.mode == T41RX_DEMOD_SAM; }
void free_ctx(t41tx_ctx *c) {
  if (!c) return;
  (void)hipFree(c->d_state);
  (void)hipFree(c->d_coef);
  (void)hipFree(c->d_in);
  (void)hipFree(c->d_outL);
  (void)hipFree(c->d_outR);
  delete c;
}
}  // namespace

extern "C" {

void t41tx_default_params(t41tx_params *p) {
  if (!p) return;
  p->mode = T41RX_DEMOD_USB;
  p->IQXAmpCorrectionFactor = 1.0f;    // gwv.cpp:73
  p->IQXPhaseCorrectionFactor = 0.0f;  // gwv.cpp:74
}

int t41tx_create(t41tx_ctx **out, int device_id, int n_channels, const t41tx_params *p) {
  if (!out || !p) return fail(T41RX_ERR_ARG, "null argument");
  *out = nullptr;
  if (n_channels <= 0 || !valid(*p)) return fail(T41RX_ERR_ARG, "bad n_channels or mode");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return fail(T41RX_ERR_HIP, "no such HIP device");
  Guard g(device_id);
  if (!g.ok) return fail(T41RX_ERR_HIP, "hipSetDevice failed");
  t41tx_ctx *c = new (std::nothrow) t41tx_ctx();
  if (!c) return fail(T41RX_ERR_NOMEM, "host allocation failed");
  c->device = device_id;
  c->nchan = n_channels;
  c->params = *p;
  TxCoef h;
  std::memcpy(h.c192, kTx192k10k, sizeof(h.c192));
  std::memcpy(h.c48, kTx48k8k, sizeof(h.c48));
  std::memcpy(h.h45, kTxHilbert45, sizeof(h.h45));
  std::memcpy(h.hn45, kTxHilbertNeg45, sizeof(h.hn45));
  const size_t sb = sizeof(float) * kTxStateFloats * (size_t)n_channels;
  if (hipMalloc((void **)&c->d_state, sb) != hipSuccess || hipMalloc((void **)&c->d_coef, sizeof(TxCoef)) != hipSuccess ||
      hipMemset(c->d_state, 0, sb) != hipSuccess || hipMemcpy(c->d_coef, &h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) {
    free_ctx(c);
    return fail(T41RX_ERR_HIP, "device allocation failed");
  }
  *out = c;
  return T41RX_OK;
}

int t41tx_destroy(t41tx_ctx *ctx) {
  if (!ctx) return T41RX_OK;
  Guard g(ctx->device);
  (void)hipDeviceSynchronize();
  free_ctx(ctx);
  return T41RX_OK;
}

int t41tx_set_params(t41tx_ctx *ctx, const t41tx_params *p) {
  if (!ctx || !p) return fail(T41RX_ERR_ARG, "null argument");
  if (!valid(*p)) return fail(T41RX_ERR_ARG, "bad mode");
  ctx->params = *p;
  return T41RX_OK;
}

int t41tx_reset(t41tx_ctx *ctx) {
  if (!ctx) return fail(T41RX_ERR_ARG, "null argument");
  Guard g(ctx->device);
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemset(ctx->d_state, 0, sizeof(float) * kTxStateFloats * (size_t)ctx->nchan) != hipSuccess)
    return fail(T41RX_ERR_HIP, "state reset failed");
  return T41RX_OK;
}

int t41tx_n_channels(const t41tx_ctx *ctx) { return ctx ? ctx->nchan : T41RX_ERR_ARG; }

int t41tx_process_device_q15(t41tx_ctx *ctx, const int16_t *dL, const int16_t *dR, int16_t *oL, int16_t *oR, int n_frames,
                             void *hip_stream) {
  (void)dR;  // decimated and then overwritten by the L channel in the reference (Exciter.cpp:85, 89, 98)
  if (!ctx || !dL || !oL || !oR) return fail(T41RX_ERR_ARG, "null argument");
  if (n_frames <= 0) return fail(T41RX_ERR_ARG, "n_frames must be > 0");
  if ((reinterpret_cast<uintptr_t>(dL) | reinterpret_cast<uintptr_t>(oL) | reinterpret_cast<uintptr_t>(oR)) & 15u)
    return fail(T41RX_ERR_ARG, "device pointers must be 16-byte aligned");
  Guard g(ctx->device);
  if (!g.ok) return fail(T41RX_ERR_HIP, "hipSetDevice failed");
  TxArgs a{};
  a.inL = dL;
  a.outL = oL;
  a.outR = oR;
  a.state = ctx->d_state;
  a.coef = ctx->d_coef;
  a.nchan = ctx->nchan;
  a.nframes = n_frames;
  a.corr_on = (ctx->params.mode == T41RX_DEMOD_LSB || ctx->params.mode == T41RX_DEMOD_USB) ? 1 : 0;
  a.i_scale = (ctx->params.mode == T41RX_DEMOD_LSB) ? +ctx->params.IQXAmpCorrectionFactor : -ctx->params.IQXAmpCorrectionFactor;
  a.iq_phase = ctx->params.IQXPhaseCorrectionFactor;
  if (launch_tx(a, (hipStream_t)hip_stream) != hipSuccess) return fail(T41RX_ERR_HIP, "kernel launch failed");
  return T41RX_OK;
}

int t41tx_process_host_q15(t41tx_ctx *ctx, const int16_t *L, const int16_t *R, int16_t *oL, int16_t *oR, int n_frames) {
  (void)R;
  if (!ctx || !L || !oL || !oR) return fail(T41RX_ERR_ARG, "null argument");
  if (n_frames <= 0) return fail(T41RX_ERR_ARG, "n_frames must be > 0");
  Guard g(ctx->device);
  if (!g.ok) return fail(T41RX_ERR_HIP, "hipSetDevice failed");
  const size_t bytes = sizeof(int16_t) * 2048 * (size_t)n_frames * (size_t)ctx->nchan;
  if (bytes > ctx->staging) {
    (void)hipFree(ctx->d_in);
    (void)hipFree(ctx->d_outL);
    (void)hipFree(ctx->d_outR);
    ctx->d_in = ctx->d_outL = ctx->d_outR = nullptr;
    ctx->staging = 0;
    if (hipMalloc((void **)&ctx->d_in, bytes) != hipSuccess || hipMalloc((void **)&ctx->d_outL, bytes) != hipSuccess ||
        hipMalloc((void **)&ctx->d_outR, bytes) != hipSuccess)
      return fail(T41RX_ERR_NOMEM, "staging allocation failed");
    ctx->staging = bytes;
  }
  if (hipMemcpy(ctx->d_in, L, bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(T41RX_ERR_HIP, "copy in failed");
  const int rc = t41tx_process_device_q15(ctx, ctx->d_in, nullptr, ctx->d_outL, ctx->d_outR, n_frames, nullptr);
  if (rc != T41RX_OK) return rc;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(oL, ctx->d_outL, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(oR, ctx->d_outR, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return fail(T41RX_ERR_HIP, "copy out failed");
  return T41RX_OK;
}

}  // extern "C"

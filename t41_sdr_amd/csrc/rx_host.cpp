// t41_sdr_amd/csrc/rx_host.cpp -- host side of the C ABI in include/t41rx.h.
//
// Owns what the reference keeps in firmware globals: the coefficient arrays
// (FIR_dec1_coeffs ... FIR_filter_mask, T41_SDR.ino:398-399, Filter.cpp:39-41), the CMSIS
// instance state (T41_SDR.ino:384-397), the oscillator state (Freq_Shift.cpp:13-14) and the
// overlap-save block (T41_SDR.ino:403-404) -- here per channel and resident in HBM.
// There is no CPU implementation of the path in this library: without a HIP device every
// create/process call fails with T41RX_ERR_HIP.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "nr_kernels.hpp"
#include "rx_experiments.hpp"
#include "rx_internal.hpp"
#include "rx_kernels.hpp"

using namespace t41;

struct t41rx_ctx {
  int device = 0;
  int nchan = 0;
  t41rx_params params{};
  std::vector<float> blob;       // canonical coefficient blob (host)
  std::vector<int32_t> nco_hz;   // NCOFreq per channel (host)
  float *d_state = nullptr;      // [nchan][state_floats]
  DevCoef *d_coef = nullptr;
  float2 *d_tab = nullptr;
  ChanNco *d_nco = nullptr;
  float *dbg_nco = nullptr, *dbg_dec = nullptr, *dbg_demod = nullptr;
  float *spect = nullptr, *spect_max = nullptr;  // audio-spectrum side output (t41rx_set_audio_spectrum)
  int tap_frames = 0, spect_frames = 0;          // frames per call those buffers are sized for
  // display FFT side output (t41rx_set_display_spectrum)
  float *disp_spec = nullptr, *disp_old = nullptr;  // caller's buffers
  float *d_pre = nullptr, *d_disp = nullptr;         // input tap [nchan][disp_frames][4096], state [nchan][kDispFloats]
  double *d_win = nullptr;
  int disp_frames = 0, disp_zoom = 0;
  // FFT_LENGTH 4096 pipeline: constant table + scratch between its three kernels
  float2 *d_tab4k = nullptr;
  float *d_mid = nullptr, *d_aud24 = nullptr;
  float *d_agc_pipe = nullptr;  // AGC on, FFT_LENGTH 512: the pipelined kernel's slots (RxArgs::agc_pipe), allocated on first use
  int scratch_frames = 0;
  int layout = T41RX_LAYOUT_CHANNEL_MAJOR;  // of I / Q / audio (t41rx_set_buffer_layout)
  int nco_sel = 0;               // long FFT: which NcoState copy is current (flips with every process call)
  // noise reduction / notch (Process.cpp:841-866): state of Xanr() and of the two spectral functions, window tables;
  // allocated when a call first needs them
  float *d_nr_anr = nullptr, *d_nr_spec = nullptr, *d_nr_tab = nullptr;
  // staging for t41rx_process_host
  float *d_in_i = nullptr, *d_in_q = nullptr, *d_out = nullptr;
  size_t staging_floats = 0;
};

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}
int hip_fail(hipError_t e, const char *what) {
  return fail(T41RX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t e__ = (expr);                           \
    if (e__ != hipSuccess) return hip_fail(e__, #expr); \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

constexpr float kPiF = 3.1415926535897932384626433832795f;  // FIR.h:10

// FreqShift2's per-call constants (Freq_Shift.cpp:121-124) turned into what the kernel needs:
// the rotation per sample as a 0.64 fixed-point fraction of a turn, the steady-state
// amplitude of the oscillator's gain loop, and the eight per-lane sample offsets.
ChanNco make_nco(int32_t nco_freq_hz, int side_tone_hz) {
  ChanNco n{};
  const long f = (long)nco_freq_hz + (long)side_tone_hz;
  const float inc = (float)(2.0 * kPiF * f / 192000.0);  // NCO_INC (float32_t)
  const double c = std::cos((double)inc), s = std::sin((double)inc);  // OSC_COS / OSC_SIN
  const long double ang = atan2l((long double)s, (long double)c);
  long double turns = ang / (2.0L * 3.14159265358979323846264338327950288L);
  if (turns < 0.0L) turns += 1.0L;
  const long double scaled = floorl(turns * 18446744073709551616.0L + 0.5L);
  n.phase_inc = (scaled >= 18446744073709551616.0L) ? 0ull : (uint64_t)scaled;
  n.w_abs = (double)hypotl((long double)c, (long double)s);
  // fixed point of r <- r (1.95 - r^2) |W|  (Freq_Shift.cpp:130-134)
  n.r_star_sq = 1.95 - 1.0 / n.w_abs;
  const double amp = (double)1.1f * std::sqrt(n.r_star_sq) * n.w_abs;  // freqAdjFactor * |Osc|
  for (int k = 0; k < 8; ++k) {
    // 1.1 A e^{j k ang} (-j)^k: the (-j)^k folds FreqShift1's x j^n (Freq_Shift.cpp:42-65) into the
    // conjugate multiply of the mixer; a rotation by k quarter turns is exact
    const double c = amp * (double)cosl(ang * k), s = amp * (double)sinl(ang * k);
    double re, im;
    switch (k & 3) {
      case 0: re = c; im = s; break;
      case 1: re = s; im = -c; break;   // (c + js)(-j) = s - jc
      case 2: re = -c; im = -s; break;
      default: re = -s; im = c; break;  // (c + js)(j) = -s + jc
    }
    n.wk[k][0] = (float)re;
    n.wk[k][1] = (float)im;
  }
  return n;
}

int upload_nco(t41rx_ctx *ctx) {
  std::vector<ChanNco> h((size_t)ctx->nchan);
  const int side = (int)blob_view(ctx->blob.data()).scalars[kScSideTone];
  for (int i = 0; i < ctx->nchan; ++i) h[(size_t)i] = make_nco(ctx->nco_hz[(size_t)i], side);
  HIP_TRY(hipMemcpy(ctx->d_nco, h.data(), sizeof(ChanNco) * h.size(), hipMemcpyHostToDevice));
  return T41RX_OK;
}

// blob -> device constant block + lane-ordered tables
int upload_coeffs(t41rx_ctx *ctx) {
  const int N = ctx->params.fft_length;
  BlobView v = blob_view(ctx->blob.data());
  DevCoef dc;
  std::memset(&dc, 0, sizeof(dc));
  std::memcpy(dc.dec1, v.dec1, sizeof(float) * kDec1Taps);
  // FIR_dec2_coeffs times the level adjust volScaleFactor (Process.cpp:481-492), the multiply right behind that filter
  for (int i = 0; i < kDec2Taps; ++i) dc.dec2[i] = v.dec2[i] * v.scalars[kScLevel];
  std::memcpy(dc.int1, v.int1, sizeof(float) * kInt1Taps);
  // FIR_int2_coeffs times the volume factor DF * VolumeToAmplification(audioVolume) (Process.cpp:929):
  // the x4 interpolator is the last stage before it, so the kernels apply it through the taps
  for (int i = 0; i < kInt2Taps; ++i) dc.int2[i] = v.int2[i] * v.scalars[kScOutScale];
  std::memcpy(dc.lp1, v.lp1, sizeof(float) * 5);
  std::memcpy(dc.sc, v.scalars, sizeof(float) * kNumScalars);
  std::memcpy(dc.agc, v.agc, sizeof(float) * kNumAgc);
  std::memcpy(dc.deemph, kDeemphFir24000, sizeof(float) * kDeemphTaps);
  HIP_TRY(hipMemcpy(ctx->d_coef, &dc, sizeof(dc), hipMemcpyHostToDevice));

  const int R = N / 512;  // 2048-sample segments per frame
  std::vector<float2> tab((size_t)kTabEntries512, make_float2(0.0f, 0.0f));
  const float invN = 1.0f / (float)N;  // exact power of two: folding it into the mask is lossless
  if (N == 512) {
    for (int r = 0; r < 8; ++r)
      for (int l = 0; l < 64; ++l) {
        const int k = l + 64 * r;
        tab[(size_t)(kTabMask + 64 * r + l)] = make_float2(v.mask[2 * k] * invN, v.mask[2 * k + 1] * invN);
      }
  } else {
    // N = R x 512 decomposition: radix-R twiddles W_N^(k' q) and the mask in [q][m] order
    std::vector<float2> t4((size_t)tab_long_entries(R));
    const double tp = 6.283185307179586476925286766559;
    for (int q = 1; q < R; ++q)
      for (int k = 0; k < 512; ++k) {
        const double a = -tp * (double)(k * q) / (double)N;
        t4[(size_t)(512 * (q - 1) + k)] = make_float2((float)std::cos(a), (float)std::sin(a));
      }
    for (int q = 0; q < R; ++q)
      for (int m = 0; m < 512; ++m) {
        const int k = q + R * m;
        t4[(size_t)((R - 1) * 512 + 512 * q + m)] = make_float2(v.mask[2 * k] * invN, v.mask[2 * k + 1] * invN);
      }
    HIP_TRY(hipMemcpy(ctx->d_tab4k, t4.data(), sizeof(float2) * t4.size(), hipMemcpyHostToDevice));
  }
  const double two_pi = 6.283185307179586476925286766559;
  for (int q = 1; q < 8; ++q)
    for (int l = 0; l < 64; ++l) {
      const double a1 = -two_pi * (double)(l * q) / 512.0;
      const double a2 = -two_pi * (double)((l & 7) * q) / 64.0;
      tab[(size_t)(kTabTw1 + 64 * (q - 1) + l)] = make_float2((float)std::cos(a1), (float)std::sin(a1));
      tab[(size_t)(kTabTw2 + 64 * (q - 1) + l)] = make_float2((float)std::cos(a2), (float)std::sin(a2));
    }
  for (int i = 0; i < 256; ++i) {
    const double a = two_pi * (double)i / 256.0;
    tab[(size_t)(kTabSinCos + i)] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  // DC high-pass (FIR.cpp:87-89, a1 = 0.854352383886757938) carry multipliers for the DPP scan
  for (int l = 0; l < 64; ++l) {
    const double a1 = 0.854352383886757938;
    tab[(size_t)(kTabHp8 + l)] = make_float2((float)std::pow(a1, 8.0 * ((l & 15) + 1)), (float)std::pow(a1, 8.0 * ((l & 31) + 1)));
    tab[(size_t)(kTabHp4 + l)] = make_float2((float)std::pow(a1, 4.0 * ((l & 15) + 1)), (float)std::pow(a1, 4.0 * ((l & 31) + 1)));
  }
  // AM demodulator (Process.cpp:698-705): carry multipliers of its two wave scans.  DC blocker
  // w = m + 0.99 w_old: powers of ca^4 in double; biquad_lowpass1 (DF1): powers of the 2x2
  // transition matrix over one lane's four samples, P1 = M^4, M = [[a1, a2], [1, 0]], in f32.
  {
    const double ca = (double)0.99f, a4 = ca * ca * ca * ca;
    struct M2 { float a, b, c, d; };
    auto mm = [](M2 x, M2 y) { return M2{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d}; };
    const M2 M{v.lp1[3], v.lp1[4], 1.0f, 0.0f};
    const M2 Mq = mm(M, M), P1 = mm(Mq, Mq);
    for (int l = 0; l < 64; ++l) {
      double p15 = a4, p31 = a4;
      M2 Q15 = P1, Q31 = P1;
      for (int i = 0; i < (l & 15); ++i) { p15 *= a4; Q15 = mm(Q15, P1); }
      for (int i = 0; i < (l & 31); ++i) { p31 *= a4; Q31 = mm(Q31, P1); }
      float2 *e = &tab[(size_t)(kTabAm + 6 * l)];
      std::memcpy(&e[0], &p15, sizeof(double));
      std::memcpy(&e[1], &p31, sizeof(double));
      e[2] = make_float2(Q15.a, Q15.b);
      e[3] = make_float2(Q15.c, Q15.d);
      e[4] = make_float2(Q31.a, Q31.b);
      e[5] = make_float2(Q31.c, Q31.d);
    }
  }
  // arm_sin_f32's table for the synchronous detector (Demod.cpp:75-76): sin(2 pi k / 512), k = 0..512, rounded from
  // double (the library's own literals are not available here: DESIGN.md section 2), exact zeros where it has them
  {
    float *t = reinterpret_cast<float *>(&tab[(size_t)kTabSam]);
    for (int k = 0; k <= 512; ++k) t[k] = (float)std::sin(6.283185307179586476925286766559 * (double)k / 512.0);
    t[0] = 0.0f;
    t[256] = 0.0f;
    t[512] = -0.0f;
  }
  HIP_TRY(hipMemcpy(ctx->d_tab, tab.data(), sizeof(float2) * tab.size(), hipMemcpyHostToDevice));
  return T41RX_OK;
}

// InitializeDataArrays() + SpectralNoiseReductionInit() (T41_SDR.ino:479-504, 657)
int reset_nr(t41rx_ctx *ctx) {
  std::vector<float> anr((size_t)kAnrStRows * (size_t)ctx->nchan), spec((size_t)kNrSpecFloats * (size_t)ctx->nchan);
  nr_reset_anr(anr.data(), (size_t)ctx->nchan);
  for (int c = 0; c < ctx->nchan; ++c) nr_reset_record(spec.data() + (size_t)kNrSpecFloats * (size_t)c);
  HIP_TRY(hipMemcpy(ctx->d_nr_anr, anr.data(), sizeof(float) * anr.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ctx->d_nr_spec, spec.data(), sizeof(float) * spec.size(), hipMemcpyHostToDevice));
  return T41RX_OK;
}

int reset_state(t41rx_ctx *ctx) {
  const size_t sf = state_floats(ctx->params.fft_length);
  std::vector<float> h(sf * (size_t)ctx->nchan, 0.0f);
  for (int c = 0; c < ctx->nchan; ++c) {
    NcoState ns;
    ns.phase = 0;  // Osc_Vect_Q = 1, Osc_Vect_I = 0 (Freq_Shift.cpp:13-14)
    ns.r = 1.0;
    std::memcpy(h.data() + sf * (size_t)c + kStNco, &ns, sizeof(ns));
    std::memcpy(h.data() + sf * (size_t)c + kStNco + 4, &ns, sizeof(ns));
  }
  HIP_TRY(hipMemcpy(ctx->d_state, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
  ctx->nco_sel = 0;
  if (ctx->d_disp) HIP_TRY(hipMemset(ctx->d_disp, 0, sizeof(float) * kDispFloats * (size_t)ctx->nchan));
  if (ctx->d_nr_anr) return reset_nr(ctx);
  return T41RX_OK;
}

// state and tables of the noise-reduction stages, on first use
int ensure_nr(t41rx_ctx *ctx) {
  if (ctx->d_nr_anr) return T41RX_OK;
  float *anr = nullptr, *spec = nullptr, *tab = nullptr;
  if (hipMalloc((void **)&anr, sizeof(float) * kAnrStRows * (size_t)ctx->nchan) != hipSuccess ||
      hipMalloc((void **)&spec, sizeof(float) * kNrSpecFloats * (size_t)ctx->nchan) != hipSuccess ||
      hipMalloc((void **)&tab, sizeof(float) * kNrTabFloats) != hipSuccess) {
    (void)hipFree(anr);
    (void)hipFree(spec);
    (void)hipFree(tab);
    return fail(T41RX_ERR_NOMEM, "noise-reduction state allocation failed");
  }
  float h[kNrTabFloats];
  nr_make_tables(h);
  std::vector<float> ha((size_t)kAnrStRows * (size_t)ctx->nchan), hs((size_t)kNrSpecFloats * (size_t)ctx->nchan);
  nr_reset_anr(ha.data(), (size_t)ctx->nchan);
  for (int c = 0; c < ctx->nchan; ++c) nr_reset_record(hs.data() + (size_t)kNrSpecFloats * (size_t)c);
  if (hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(anr, ha.data(), sizeof(float) * ha.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(spec, hs.data(), sizeof(float) * hs.size(), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(anr);
    (void)hipFree(spec);
    (void)hipFree(tab);
    return fail(T41RX_ERR_HIP, "noise-reduction state upload failed");
  }
  ctx->d_nr_anr = anr;
  ctx->d_nr_spec = spec;
  ctx->d_nr_tab = tab;
  return T41RX_OK;
}

void free_ctx(t41rx_ctx *ctx) {
  if (!ctx) return;
  (void)hipFree(ctx->d_state);
  (void)hipFree(ctx->d_coef);
  (void)hipFree(ctx->d_tab);
  (void)hipFree(ctx->d_nco);
  (void)hipFree(ctx->d_tab4k);
  (void)hipFree(ctx->d_mid);
  (void)hipFree(ctx->d_aud24);
  if (ctx->d_agc_pipe && std::getenv("T41RX_PIPE_STAT")) {  // diagnostic build's counters (rx_kernels.hip: PIPE_STAT_*)
    unsigned long long c[16] = {};
    std::vector<unsigned long long> all((size_t)ctx->nchan * 16);
    (void)hipMemcpy(all.data(), ctx->d_agc_pipe + (size_t)ctx->nchan * 3 * 1024, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    for (size_t i = 0; i < all.size(); ++i) c[i & 15] += all[i];
    std::fprintf(stderr, "pipe_stat chain_cycles %llu chains %llu slow_blocks %llu back_wait %llu duty_wait %llu blocks %llu chain_stage %llu chain_steps %llu front %llu prep %llu back %llu wave_iterations %llu chain_state_wait %llu\n",
                 c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], c[10], c[11], c[12]);
  }
  (void)hipFree(ctx->d_agc_pipe);
  (void)hipFree(ctx->d_in_i);
  (void)hipFree(ctx->d_in_q);
  (void)hipFree(ctx->d_out);
  (void)hipFree(ctx->d_pre);
  (void)hipFree(ctx->d_disp);
  (void)hipFree(ctx->d_win);
  (void)hipFree(ctx->d_nr_anr);
  (void)hipFree(ctx->d_nr_spec);
  (void)hipFree(ctx->d_nr_tab);
  delete ctx;
}

}  // namespace

extern "C" {

int t41rx_abi_version(void) { return T41RX_ABI_VERSION; }

const char *t41rx_strerror(int status) {
  switch (status) {
    case T41RX_OK: return "ok";
    case T41RX_ERR_ARG: return "invalid argument";
    case T41RX_ERR_UNSUPPORTED: return "not supported by this build";
    case T41RX_ERR_HIP: return "HIP runtime/device error";
    case T41RX_ERR_NOMEM: return "out of memory";
    case T41RX_ERR_STATE: return "blob/state mismatch";
    default: return "unknown status";
  }
}

const char *t41rx_last_error(void) { return g_last_error.c_str(); }

int t41rx_supported_fft_length(int fft_length) {
  return (fft_length == 512 || fft_length == 1024 || fft_length == 2048 || fft_length == 4096) ? 1 : 0;
}

void t41rx_default_params(t41rx_params *p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->fft_length = 512;               // SDT.h:39
  p->mode = T41RX_DEMOD_USB;         // bands[] 20 m row, T41_SDR.ino:163
  p->FLoCut = 200;
  p->FHiCut = 3000;
  p->rfGainAllBands = 1;             // gwv.cpp:17
  p->RFgain = 1;                     // bands[].RFgain
  p->IQAmpCorrectionFactor = 1.0f;   // gwv.cpp:71
  p->IQPhaseCorrectionFactor = 0.0f; // gwv.cpp:72
  p->AGCMode = 0;                    // firmware default is 1 (gwv.cpp:15); 0 = fixed gain
  p->audioVolume = 30;               // gwv.cpp:16
  p->nfmFilterBW = 12000;            // Filter.cpp:16
  p->xmtMode = T41RX_SSB_MODE;       // gwv.cpp:22
  p->CWFreqShift = 750;
  p->am_lpf_f0 = 3000;               // boot band 40 m: max(FHiCut, -FLoCut) = 3000
  p->AGC_thresh = 20;                // bands[] "AGC" column, T41_SDR.ino:145-168
  p->nrOptionSelect = 0;             // gwv.cpp:23
  p->ANR_notchOn = 0;                // Process.cpp:45
  p->NR_PSI = 0.0;                   // gwv.cpp:61-63
  p->NR_alpha = 0.95;
  p->NR_beta = 0.85;
}

size_t t41rx_coeff_blob_bytes(int fft_length) {
  if (!(fft_length == 512 || fft_length == 1024 || fft_length == 2048 || fft_length == 4096)) return 0;
  return blob_floats(fft_length) * sizeof(float);
}

int t41rx_design_coeffs(const t41rx_params *p, void *blob, size_t blob_bytes) {
  if (!p || !blob) return fail(T41RX_ERR_ARG, "null argument");
  const char *why = nullptr;
  if (!params_valid(*p, &why)) return fail(T41RX_ERR_ARG, why ? why : "bad params");
  if (blob_bytes < t41rx_coeff_blob_bytes(p->fft_length)) return fail(T41RX_ERR_ARG, "blob buffer too small");
  return design_blob(*p, blob, blob_bytes);
}

// the product's host side is never an experiment build; what the KERNEL objects were built as is asked at run time
// (kernel_build_flags(), below): tools/build_variant.sh links this very object with experiment kernels
static_assert(T41RX_EXPERIMENT == 0 && t41::kKernelBuildFlags == 0, "rx_host.cpp is product code: build it without T41RX_EXPERIMENT / experiment switches");

int t41rx_create(t41rx_ctx **out, int device_id, int n_channels, const t41rx_params *p) {
  // A library whose kernels were built as a timing experiment computes WRONG RESULTS by construction (rx_experiments.hpp:
  // -DT41RX_EXPERIMENT=1 with T41RX_ABLATE / _LOO / _AGC_X / _FCABL) and one built with diagnostics writes stamps next to
  // the samples: neither may stand in for the product by accident.  The tools that time such builds say so in the environment.
  if (kernel_build_flags() != 0) {
    const char *allow = std::getenv("T41RX_ALLOW_EXPERIMENT");
    if (!allow || std::atoi(allow) == 0)
      return fail(T41RX_ERR_UNSUPPORTED, (kernel_build_flags() & 1)
                      ? "this libt41rx was built as a timing experiment (wrong results by construction); set T41RX_ALLOW_EXPERIMENT=1 to time it"
                      : "this libt41rx was built with kernel diagnostics (stamps / counters); set T41RX_ALLOW_EXPERIMENT=1 to use it");
    static bool warned = false;
    if (!warned) {
      warned = true;
      std::fprintf(stderr, "libt41rx: EXPERIMENT BUILD (kernel_build_flags %d)%s\n", kernel_build_flags(),
                   (kernel_build_flags() & 1) ? " -- results are WRONG by construction" : " -- diagnostics on");
    }
  }

  if (!out || !p) return fail(T41RX_ERR_ARG, "null argument");
  *out = nullptr;
  if (n_channels <= 0) return fail(T41RX_ERR_ARG, "n_channels must be > 0");
  const char *why = nullptr;
  if (!params_valid(*p, &why)) return fail(T41RX_ERR_ARG, why ? why : "bad params");

  if (!t41rx_supported_fft_length(p->fft_length)) return fail(T41RX_ERR_UNSUPPORTED, "no kernel for this fft_length");

  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail(T41RX_ERR_HIP, "no such HIP device");
  DeviceGuard g(device_id);
  if (!g.ok) return fail(T41RX_ERR_HIP, "hipSetDevice failed");

  t41rx_ctx *ctx = new (std::nothrow) t41rx_ctx();
  if (!ctx) return fail(T41RX_ERR_NOMEM, "host allocation failed");
  ctx->device = device_id;
  ctx->nchan = n_channels;
  ctx->params = *p;
  ctx->blob.assign(blob_floats(p->fft_length), 0.0f);
  ctx->nco_hz.assign((size_t)n_channels, 0);
  int rc = design_blob(*p, ctx->blob.data(), ctx->blob.size() * sizeof(float));
  if (rc != T41RX_OK) {
    free_ctx(ctx);
    return fail(rc, "coefficient design failed");
  }
  hipError_t e;
  const size_t sbytes = sizeof(float) * state_floats(p->fft_length) * (size_t)n_channels;
  if ((e = hipMalloc((void **)&ctx->d_state, sbytes)) != hipSuccess ||
      (e = hipMalloc((void **)&ctx->d_coef, sizeof(DevCoef))) != hipSuccess ||
      (e = hipMalloc((void **)&ctx->d_tab, sizeof(float2) * kTabEntries512)) != hipSuccess ||
      (e = hipMalloc((void **)&ctx->d_nco, sizeof(ChanNco) * (size_t)n_channels)) != hipSuccess ||
      (p->fft_length != 512 && (e = hipMalloc((void **)&ctx->d_tab4k, sizeof(float2) * tab_long_entries(p->fft_length / 512))) != hipSuccess)) {
    free_ctx(ctx);
    return hip_fail(e, "hipMalloc");
  }
  if ((rc = upload_coeffs(ctx)) != T41RX_OK || (rc = upload_nco(ctx)) != T41RX_OK ||
      (rc = reset_state(ctx)) != T41RX_OK) {
    free_ctx(ctx);
    return rc;
  }
  *out = ctx;
  return T41RX_OK;
}

int t41rx_destroy(t41rx_ctx *ctx) {
  if (!ctx) return T41RX_OK;
  DeviceGuard g(ctx->device);
  (void)hipDeviceSynchronize();
  free_ctx(ctx);
  return T41RX_OK;
}

int t41rx_set_params(t41rx_ctx *ctx, const t41rx_params *p) {
  if (!ctx || !p) return fail(T41RX_ERR_ARG, "null argument");
  const char *why = nullptr;
  if (!params_valid(*p, &why)) return fail(T41RX_ERR_ARG, why ? why : "bad params");
  if (p->fft_length != ctx->params.fft_length) return fail(T41RX_ERR_ARG, "fft_length cannot change on a live context");


  std::vector<float> nb(ctx->blob.size());
  int rc = design_blob(*p, nb.data(), nb.size() * sizeof(float));
  if (rc != T41RX_OK) return fail(rc, "coefficient design failed");
  DeviceGuard g(ctx->device);
  HIP_TRY(hipDeviceSynchronize());
  ctx->blob.swap(nb);
  ctx->params = *p;
  if ((rc = upload_coeffs(ctx)) != T41RX_OK) return rc;
  return upload_nco(ctx);  // the CW side-tone offset may have changed
}

int t41rx_get_params(const t41rx_ctx *ctx, t41rx_params *p) {
  if (!ctx || !p) return fail(T41RX_ERR_ARG, "null argument");
  *p = ctx->params;
  return T41RX_OK;
}

int t41rx_get_coeffs(const t41rx_ctx *ctx, void *blob, size_t blob_bytes) {
  if (!ctx || !blob) return fail(T41RX_ERR_ARG, "null argument");
  const size_t need = ctx->blob.size() * sizeof(float);
  if (blob_bytes < need) return fail(T41RX_ERR_ARG, "blob buffer too small");
  std::memcpy(blob, ctx->blob.data(), need);
  return T41RX_OK;
}

int t41rx_set_coeffs(t41rx_ctx *ctx, const void *blob, size_t blob_bytes) {
  if (!ctx || !blob) return fail(T41RX_ERR_ARG, "null argument");
  const size_t need = ctx->blob.size() * sizeof(float);
  if (blob_bytes < need) return fail(T41RX_ERR_STATE, "blob too small for this context");
  const int32_t *h = reinterpret_cast<const int32_t *>(blob);
  if ((uint32_t)h[0] != kBlobMagic || h[1] != T41RX_ABI_VERSION) return fail(T41RX_ERR_STATE, "bad blob header");
  if (h[2] != ctx->params.fft_length) return fail(T41RX_ERR_STATE, "blob fft_length differs from the context");
  if ((h[3] < T41RX_DEMOD_USB || h[3] > T41RX_DEMOD_NFM) && h[3] != T41RX_DEMOD_SAM) return fail(T41RX_ERR_STATE, "bad demodulation mode in blob");
  // the parameters the blob was designed for become the context's (every rank that installs a
  // broadcast blob then runs -- and later re-designs from -- the designer's parameters)
  if (h[4] != (int32_t)sizeof(t41rx_params)) return fail(T41RX_ERR_STATE, "blob carries another t41rx_params layout");
  t41rx_params bp;
  std::memcpy(&bp, h + 8, sizeof(bp));
  const char *why = nullptr;
  if (!params_valid(bp, &why) || bp.fft_length != h[2] || bp.mode != h[3])
    return fail(T41RX_ERR_STATE, std::string("blob parameters invalid: ") + (why ? why : "header mismatch"));

  DeviceGuard g(ctx->device);
  HIP_TRY(hipDeviceSynchronize());
  std::memcpy(ctx->blob.data(), blob, need);
  ctx->params = bp;
  int rc = upload_coeffs(ctx);
  if (rc != T41RX_OK) return rc;
  return upload_nco(ctx);
}

int t41rx_set_nco_freq(t41rx_ctx *ctx, const int32_t *nco_freq_hz, int n) {
  if (!ctx || !nco_freq_hz) return fail(T41RX_ERR_ARG, "null argument");
  if (n != ctx->nchan) return fail(T41RX_ERR_ARG, "n must equal n_channels");
  for (int i = 0; i < n; ++i)
    if (nco_freq_hz[i] < -96000 || nco_freq_hz[i] > 96000) return fail(T41RX_ERR_ARG, "NCOFreq beyond +-Fs/2");
  DeviceGuard g(ctx->device);
  HIP_TRY(hipDeviceSynchronize());
  std::memcpy(ctx->nco_hz.data(), nco_freq_hz, sizeof(int32_t) * (size_t)n);
  return upload_nco(ctx);
}

static int pipe_timeouts_clear(t41rx_ctx *ctx);
static int pipe_status(t41rx_ctx *ctx);

int t41rx_reset(t41rx_ctx *ctx) {
  if (!ctx) return fail(T41RX_ERR_ARG, "null argument");
  DeviceGuard g(ctx->device);
  HIP_TRY(hipDeviceSynchronize());
  {  // (and the pipelined kernels' time-out counter)
    const int rc = pipe_timeouts_clear(ctx);
    if (rc != T41RX_OK) return rc;
  }
  return reset_state(ctx);
}

int t41rx_set_buffer_layout(t41rx_ctx *ctx, int layout) {
  if (!ctx) return fail(T41RX_ERR_ARG, "null argument");
  if (layout != T41RX_LAYOUT_CHANNEL_MAJOR && layout != T41RX_LAYOUT_TIME_MAJOR) return fail(T41RX_ERR_ARG, "unknown buffer layout");
  if (layout == T41RX_LAYOUT_TIME_MAJOR && ctx->params.fft_length != 512)
    return fail(T41RX_ERR_UNSUPPORTED, "the time-major layout is built for fft_length 512 (the long-FFT pipeline's kernels walk a channel's samples contiguously)");
  ctx->layout = layout;
  return T41RX_OK;
}
int t41rx_get_buffer_layout(const t41rx_ctx *ctx) { return ctx ? ctx->layout : T41RX_ERR_ARG; }

int t41rx_n_channels(const t41rx_ctx *ctx) { return ctx ? ctx->nchan : T41RX_ERR_ARG; }
int t41rx_frame_len(const t41rx_ctx *ctx) { return ctx ? 4 * ctx->params.fft_length : T41RX_ERR_ARG; }

namespace {
int process_device_impl(t41rx_ctx *ctx, const float *dI, const float *dQ, float *dAudio, int n_frames,
                        void *hip_stream, bool q15);
}

int t41rx_process_device(t41rx_ctx *ctx, const float *dI, const float *dQ, float *dAudio, int n_frames,
                         void *hip_stream) {
  return process_device_impl(ctx, dI, dQ, dAudio, n_frames, hip_stream, false);
}

// float_buffer_L (= I) is filled from the R queue and float_buffer_R (= Q) from the L queue
// (Process.cpp:107-108)
int t41rx_process_device_q15(t41rx_ctx *ctx, const int16_t *dQ_in_L, const int16_t *dQ_in_R, int16_t *dQ_out_L,
                             int n_frames, void *hip_stream) {
  return process_device_impl(ctx, reinterpret_cast<const float *>(dQ_in_R), reinterpret_cast<const float *>(dQ_in_L),
                             reinterpret_cast<float *>(dQ_out_L), n_frames, hip_stream, true);
}

namespace {
int process_device_impl(t41rx_ctx *ctx, const float *dI, const float *dQ, float *dAudio, int n_frames,
                        void *hip_stream, bool q15) {
  if (!ctx || !dI || !dQ || !dAudio) return fail(T41RX_ERR_ARG, "null argument");
  if (n_frames <= 0) return fail(T41RX_ERR_ARG, "n_frames must be > 0");
  if ((reinterpret_cast<uintptr_t>(dI) | reinterpret_cast<uintptr_t>(dQ) | reinterpret_cast<uintptr_t>(dAudio)) & 15u)
    return fail(T41RX_ERR_ARG, "I/Q/audio device pointers must be 16-byte aligned");
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(T41RX_ERR_HIP, "hipSetDevice failed");
  const int seg = ctx->params.fft_length / 512;
  const bool nr_on = ctx->params.nrOptionSelect != 0 || ctx->params.ANR_notchOn != 0;  // (fft_length 512: params_valid)
  if (nr_on) {
    const int rc = ensure_nr(ctx);
    if (rc != T41RX_OK) return rc;
  }
  if ((seg > 1 || nr_on) && n_frames > ctx->scratch_frames) {
    // scratch between the kernels of the long-FFT pipeline / the noise-reduction pipeline (grown on demand, kept)
    HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    (void)hipFree(ctx->d_mid);
    (void)hipFree(ctx->d_aud24);
    ctx->d_mid = ctx->d_aud24 = nullptr;
    ctx->scratch_frames = 0;
    const size_t per = (size_t)ctx->nchan * (size_t)n_frames * (size_t)(256 * seg);  // fft_length / 2 per frame
    HIP_TRY(hipMalloc((void **)&ctx->d_mid, per * 2 * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&ctx->d_aud24, per * 2 * sizeof(float)));  // complex when the back kernel demodulates
    ctx->scratch_frames = n_frames;
  }
  RxArgs a{};
  a.I = dI;
  a.Q = dQ;
  a.out = dAudio;
  a.state = ctx->d_state;
  a.coef = ctx->d_coef;
  a.tab = ctx->d_tab;
  a.nco = ctx->d_nco;
  a.nchan = ctx->nchan;
  a.nframes = seg * n_frames;  // 2048-sample segments
  if (ctx->layout == T41RX_LAYOUT_TIME_MAJOR) {  // [frame][channel][frame_len] (fft_length 512: set_buffer_layout / set_params)
    a.chan_stride = 2048;
    a.frame_stride = (long long)ctx->nchan * 2048;
  } else {
    a.chan_stride = (long long)a.nframes * 2048;
    a.frame_stride = 2048;
  }
  a.seg = seg;
  a.nframes4k = n_frames;
  a.mid = ctx->d_mid;
  a.aud24 = ctx->d_aud24;
  a.tab4k = ctx->d_tab4k;
  {
    const float *sc = blob_view(ctx->blob.data()).scalars;
    const bool iq_on = sc[kScIqCorrOn] != 0.0f;
    const float gi = iq_on ? sc[kScBandGain] * sc[kScNegIqAmp] : sc[kScBandGain];
    a.g_rf = sc[kScRfGain];
    a.g_band = sc[kScBandGain];
    a.neg_iq_amp = sc[kScNegIqAmp];
    a.iq_phase = sc[kScIqPhase];
    a.iq_corr_on = iq_on ? 1 : 0;
    // PLAIN folds "I <- -I" (IQ correction on, amplitude factor 1) into the sign of the RF gain; with
    // the correction on and gi = +1 (IQAmpCorrectionFactor = -1) the general kernel must run
    a.plain = ((iq_on ? gi == -1.0f : gi == 1.0f) && sc[kScBandGain] == 1.0f && (!iq_on || sc[kScIqPhase] == 0.0f)) ? 1 : 0;
  }
  a.q15 = q15 ? 1 : 0;
  a.nco_rd = ctx->nco_sel;
  a.nfm_atan = (ctx->params.mode == T41RX_DEMOD_NFM && ctx->params.nfm_demod == 1) ? 1 : 0;
  if (a.nfm_atan && seg > 1) return fail(T41RX_ERR_UNSUPPORTED, "nfm_demod = 1 is built for fft_length 512");
  {
    // Segment-parallel kernels of the long-FFT pipeline: about 4096 wave slots (256 CUs x 16) to
    // fill; a wave that starts inside the call pays one extra sub-block to rebuild its filter
    // memories, so runs are as long as still gives every slot a wave (and never longer than 8).
    const long segs = (long)a.nframes, waves = 4096;
    long run = (long)ctx->nchan * segs / waves;
    if (const char *e = std::getenv("T41RX_SEG_RUN")) run = std::atol(e);  // experiments
    a.seg_run = (int)(run < 1 ? 1 : (run > 8 ? 8 : (run > segs ? segs : run)));
  }
  a.agc = ctx->params.AGCMode != 0 ? 1 : 0;
  // the pipelined kernels' slots: AGC on (every mode but SAM), or the synchronous detector with the AGC off
  if ((a.agc != 0 || ctx->params.mode == T41RX_DEMOD_SAM) && seg == 1 && n_frames >= 4) {
    if (!ctx->d_agc_pipe) {  // (+ 8 counters per wave of the -DT41RX_PIPE_STAT diagnostic build; behind them the second
      // stage's slots: the synchronous detector behind the AGC runs two chains per frame, rx_kernels.hip PSA)
      const size_t bytes = 2 * (size_t)ctx->nchan * 3 * 1024 * sizeof(float) + ((size_t)ctx->nchan + 16) * 16 * sizeof(unsigned long long);
      float *slots = nullptr;  // the context only ever sees a buffer whose counters are zero
      HIP_TRY(hipMalloc((void **)&slots, bytes));
      const hipError_t em = hipMemset(slots, 0, bytes);
      if (em != hipSuccess) {
        (void)hipFree(slots);
        return hip_fail(em, "hipMemset of the pipelined kernels' slots");
      }
      ctx->d_agc_pipe = slots;
    }
    a.agc_pipe = ctx->d_agc_pipe;
  }
  if (a.agc && (int)blob_view(ctx->blob.data()).agc[kAgcAttackBuffsize] != kAgcDelay)
    return fail(T41RX_ERR_STATE, "coefficient blob carries an AGC look-ahead the kernel is not built for");
  if ((ctx->dbg_nco || ctx->dbg_dec || ctx->dbg_demod) && n_frames > ctx->tap_frames)
    return fail(T41RX_ERR_ARG, "n_frames exceeds the max_frames the debug tap buffers were set with");
  if (ctx->spect && n_frames > ctx->spect_frames)
    return fail(T41RX_ERR_ARG, "n_frames exceeds the max_frames the audio-spectrum buffers were set with");
  if (ctx->disp_spec && n_frames > ctx->disp_frames)
    return fail(T41RX_ERR_ARG, "n_frames exceeds the max_frames the display-spectrum buffers were set with");
  if (ctx->disp_spec && (!ctx->d_pre || !ctx->d_disp || !ctx->d_win)) return fail(T41RX_ERR_STATE, "display spectrum enabled without its buffers");
  a.dbg_pre = ctx->disp_spec ? ctx->d_pre : nullptr;
  a.dbg_nco = ctx->dbg_nco;
  a.dbg_dec = ctx->dbg_dec;
  a.dbg_demod = ctx->dbg_demod;
  a.spect = ctx->spect;
  a.spect_max = ctx->spect_max;
  if (nr_on) a.aud_out = ctx->d_aud24;  // the fused kernel stops behind the demodulator
  hipError_t e = launch_rx(a, ctx->params.fft_length, ctx->params.mode, (hipStream_t)hip_stream);
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  if (nr_on) {
    // Process.cpp:841-866 on the call's audio @24 kS/s, then the interpolators, volume and stores (Process.cpp:917-937)
    NrArgs n{};
    n.aud = ctx->d_aud24;
    n.anr = ctx->d_nr_anr;
    n.spec = ctx->d_nr_spec;
    n.tab_nr = ctx->d_nr_tab;
    n.tab = ctx->d_tab;
    n.nchan = ctx->nchan;
    n.nframes = n_frames;
    n.nr_option = ctx->params.nrOptionSelect;
    n.notch = ctx->params.ANR_notchOn;
    n.alpha = ctx->params.NR_alpha;
    n.beta = ctx->params.NR_beta;
    n.psi = ctx->params.NR_PSI;
    nr_vad_range(ctx->params.FLoCut, ctx->params.FHiCut, &n.vad_lo, &n.vad_hi);
    e = launch_nr(n, (hipStream_t)hip_stream);
    if (e != hipSuccess) return hip_fail(e, "noise-reduction kernel launch");
    a.aud_out = nullptr;
    a.aud24 = ctx->d_aud24;
    e = launch_back512(a, (hipStream_t)hip_stream);
    if (e != hipSuccess) return hip_fail(e, "interpolator kernel launch");
  }
  if (seg > 1) ctx->nco_sel ^= 1;  // the kernels wrote the other copy
  if (ctx->disp_spec) {
    DispArgs d{};
    d.pre = ctx->d_pre;
    d.disp = ctx->d_disp;
    d.spec = ctx->disp_spec;
    d.spec_old = ctx->disp_old;
    d.tab = ctx->d_tab;
    d.win = ctx->d_win;
    d.nchan = ctx->nchan;
    d.nframes = n_frames;
    d.zoom = ctx->disp_zoom;
    if (d.zoom > 0) {
      std::memcpy(d.iir, kZoomIirCoeffs[d.zoom - 1], sizeof(d.iir));
      design_zoom_fir(d.zoom, d.fir);
    }
    e = launch_display(d, (hipStream_t)hip_stream);
    if (e != hipSuccess) return hip_fail(e, "display kernel launch");
  }
  return T41RX_OK;
}

// staging buffers of the host-pointer entry points, sized in bytes per array
int ensure_staging(t41rx_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->staging_floats * sizeof(float)) return T41RX_OK;
  (void)hipFree(ctx->d_in_i);
  (void)hipFree(ctx->d_in_q);
  (void)hipFree(ctx->d_out);
  ctx->d_in_i = ctx->d_in_q = ctx->d_out = nullptr;
  ctx->staging_floats = 0;
  const size_t nfl = (bytes + sizeof(float) - 1) / sizeof(float);
  HIP_TRY(hipMalloc((void **)&ctx->d_in_i, nfl * sizeof(float)));
  HIP_TRY(hipMalloc((void **)&ctx->d_in_q, nfl * sizeof(float)));
  HIP_TRY(hipMalloc((void **)&ctx->d_out, nfl * sizeof(float)));
  ctx->staging_floats = nfl;
  return T41RX_OK;
}
}  // namespace

int t41rx_process_host_q15(t41rx_ctx *ctx, const int16_t *Q_in_L, const int16_t *Q_in_R, int16_t *Q_out_L, int n_frames) {
  if (!ctx || !Q_in_L || !Q_in_R || !Q_out_L) return fail(T41RX_ERR_ARG, "null argument");
  if (n_frames <= 0) return fail(T41RX_ERR_ARG, "n_frames must be > 0");
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(T41RX_ERR_HIP, "hipSetDevice failed");
  const size_t bytes = (size_t)ctx->nchan * (size_t)n_frames * (size_t)(4 * ctx->params.fft_length) * sizeof(int16_t);
  int rc = ensure_staging(ctx, bytes);
  if (rc != T41RX_OK) return rc;
  HIP_TRY(hipMemcpy(ctx->d_in_i, Q_in_L, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ctx->d_in_q, Q_in_R, bytes, hipMemcpyHostToDevice));
  rc = t41rx_process_device_q15(ctx, reinterpret_cast<const int16_t *>(ctx->d_in_i), reinterpret_cast<const int16_t *>(ctx->d_in_q),
                                reinterpret_cast<int16_t *>(ctx->d_out), n_frames, nullptr);
  if (rc != T41RX_OK) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(Q_out_L, ctx->d_out, bytes, hipMemcpyDeviceToHost));
  return pipe_status(ctx);  // (these calls synchronise: samples of a run whose hand-over broke do not leave with OK)
}

int t41rx_process_host(t41rx_ctx *ctx, const float *I, const float *Q, float *audio, int n_frames) {
  if (!ctx || !I || !Q || !audio) return fail(T41RX_ERR_ARG, "null argument");
  if (n_frames <= 0) return fail(T41RX_ERR_ARG, "n_frames must be > 0");
  DeviceGuard g(ctx->device);
  if (!g.ok) return fail(T41RX_ERR_HIP, "hipSetDevice failed");
  const size_t nfl = (size_t)ctx->nchan * (size_t)n_frames * (size_t)(4 * ctx->params.fft_length);
  {
    const int rc0 = ensure_staging(ctx, nfl * sizeof(float));
    if (rc0 != T41RX_OK) return rc0;
  }
  HIP_TRY(hipMemcpy(ctx->d_in_i, I, nfl * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ctx->d_in_q, Q, nfl * sizeof(float), hipMemcpyHostToDevice));
  int rc = t41rx_process_device(ctx, ctx->d_in_i, ctx->d_in_q, ctx->d_out, n_frames, nullptr);
  if (rc != T41RX_OK) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(audio, ctx->d_out, nfl * sizeof(float), hipMemcpyDeviceToHost));
  return pipe_status(ctx);  // (these calls synchronise: samples of a run whose hand-over broke do not leave with OK)
}

namespace {
constexpr uint32_t kStateMagic = 0x54343153u;  // "T41S"
constexpr size_t kStateHeaderBytes = 32;
}  // namespace

// Checkpoint sections behind the path's records (header word 5 = which are present):
//   bit 0  noise reduction / notch: Xanr()'s taps, delay line and leak words [kAnrStRows][n_channels], then the
//          Kim / spectral records [n_channels][kNrSpecFloats] (Noise.cpp:19-56) -- present once the stages have run
//   bit 1  display FFT: zoom filters, ring, FFT_spec_old [n_channels][kDispFloats] (FFT.cpp:14-26) -- present while
//          t41rx_set_display_spectrum is on; header word 6 = its spectrumZoom
constexpr int32_t kSecNr = 1, kSecDisp = 2;
static size_t nr_section_bytes(int nchan) { return sizeof(float) * ((size_t)kAnrStRows + (size_t)kNrSpecFloats) * (size_t)nchan; }
static size_t disp_section_bytes(int nchan) { return sizeof(float) * (size_t)kDispFloats * (size_t)nchan; }
static int32_t state_sections(const t41rx_ctx *ctx) {
  return (ctx->d_nr_anr ? kSecNr : 0) | ((ctx->disp_spec && ctx->d_disp) ? kSecDisp : 0);
}

size_t t41rx_state_bytes(const t41rx_ctx *ctx) {
  if (!ctx) return 0;
  const int32_t sec = state_sections(ctx);
  return kStateHeaderBytes + sizeof(float) * state_floats(ctx->params.fft_length) * (size_t)ctx->nchan +
         ((sec & kSecNr) ? nr_section_bytes(ctx->nchan) : 0) + ((sec & kSecDisp) ? disp_section_bytes(ctx->nchan) : 0);
}

// the pipelined kernels count a wait that ran out (rx_kernels.hip: pipe_wait_ge) behind their slots: a broken hand-over
// protocol would leave wrong samples, not a hung GPU -- reported at the calls that synchronise anyway
static size_t pipe_timeout_offset(const t41rx_ctx *ctx) {
  return (size_t)ctx->nchan * 3 * 1024 * sizeof(float) + ((size_t)ctx->nchan + 15) * 16 * sizeof(unsigned long long);
}
static int pipe_timeouts(t41rx_ctx *ctx) {  // < 0: the counter could not be read
  if (!ctx->d_agc_pipe) return 0;
  unsigned n = 0;
  const char *p = reinterpret_cast<const char *>(ctx->d_agc_pipe) + pipe_timeout_offset(ctx);
  if (hipMemcpy(&n, p, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)n;
}
static int pipe_timeouts_clear(t41rx_ctx *ctx) {
  if (!ctx->d_agc_pipe) return T41RX_OK;
  HIP_TRY(hipMemset(reinterpret_cast<char *>(ctx->d_agc_pipe) + pipe_timeout_offset(ctx), 0, sizeof(unsigned long long)));
  return T41RX_OK;
}
// what the synchronising entry points answer when a wait inside the pipelined kernels has run out
static int pipe_status(t41rx_ctx *ctx) {
  const int n = pipe_timeouts(ctx);
  if (n < 0) return fail(T41RX_ERR_HIP, "could not read the pipelined kernels' time-out counter");
  if (n > 0)
    return fail(T41RX_ERR_STATE, "a wait inside the pipelined AGC / SAM kernel ran out: the samples since the last reset or restored checkpoint are not valid");
  return T41RX_OK;
}

int t41rx_get_state(t41rx_ctx *ctx, void *host_buf, size_t bytes) {
  if (!ctx || !host_buf) return fail(T41RX_ERR_ARG, "null argument");
  if (bytes < t41rx_state_bytes(ctx)) return fail(T41RX_ERR_STATE, "state buffer too small");
  DeviceGuard g(ctx->device);
  HIP_TRY(hipDeviceSynchronize());
  {
    const int rc = pipe_status(ctx);
    if (rc != T41RX_OK) return rc;
  }
  const int32_t sec = state_sections(ctx);
  int32_t hdr[8] = {(int32_t)kStateMagic, T41RX_ABI_VERSION, ctx->params.fft_length, ctx->nchan,
                    (int32_t)state_floats(ctx->params.fft_length), sec, (sec & kSecDisp) ? ctx->disp_zoom : 0, 0};
  std::memcpy(host_buf, hdr, sizeof(hdr));
  const size_t path_bytes = sizeof(float) * state_floats(ctx->params.fft_length) * (size_t)ctx->nchan;
  char *out = static_cast<char *>(host_buf) + kStateHeaderBytes;
  HIP_TRY(hipMemcpy(out, ctx->d_state, path_bytes, hipMemcpyDeviceToHost));
  // canonical checkpoint: the current oscillator state in both slots
  float *rec = reinterpret_cast<float *>(out);
  const size_t sf = state_floats(ctx->params.fft_length);
  for (int c = 0; c < ctx->nchan; ++c) {
    float *n = rec + sf * (size_t)c + kStNco;
    std::memcpy(n + 4 * (ctx->nco_sel ^ 1), n + 4 * ctx->nco_sel, sizeof(NcoState));
  }
  out += path_bytes;
  if (sec & kSecNr) {
    const size_t ab = sizeof(float) * (size_t)kAnrStRows * (size_t)ctx->nchan;
    HIP_TRY(hipMemcpy(out, ctx->d_nr_anr, ab, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out + ab, ctx->d_nr_spec, nr_section_bytes(ctx->nchan) - ab, hipMemcpyDeviceToHost));
    out += nr_section_bytes(ctx->nchan);
  }
  if (sec & kSecDisp) HIP_TRY(hipMemcpy(out, ctx->d_disp, disp_section_bytes(ctx->nchan), hipMemcpyDeviceToHost));
  return T41RX_OK;
}

int t41rx_set_state(t41rx_ctx *ctx, const void *host_buf, size_t bytes) {
  if (!ctx || !host_buf) return fail(T41RX_ERR_ARG, "null argument");
  if (bytes < kStateHeaderBytes) return fail(T41RX_ERR_STATE, "state size mismatch");
  int32_t hdr[8];
  std::memcpy(hdr, host_buf, sizeof(hdr));
  const size_t sf = state_floats(ctx->params.fft_length);
  if ((uint32_t)hdr[0] != kStateMagic || hdr[1] != T41RX_ABI_VERSION || hdr[2] != ctx->params.fft_length ||
      hdr[3] != ctx->nchan || hdr[4] != (int32_t)sf)
    return fail(T41RX_ERR_STATE, "checkpoint header does not match this context (magic / abi / fft_length / channels)");
  const int32_t sec = hdr[5];
  if (sec & ~(kSecNr | kSecDisp)) return fail(T41RX_ERR_STATE, "checkpoint: unknown sections");
  const size_t path_bytes = sizeof(float) * sf * (size_t)ctx->nchan;
  if (bytes != kStateHeaderBytes + path_bytes + ((sec & kSecNr) ? nr_section_bytes(ctx->nchan) : 0) +
                   ((sec & kSecDisp) ? disp_section_bytes(ctx->nchan) : 0))
    return fail(T41RX_ERR_STATE, "state size mismatch");
  if ((sec & kSecNr) && ctx->params.fft_length != 512) return fail(T41RX_ERR_STATE, "checkpoint: noise-reduction section at a long fft_length");
  if (sec & kSecDisp) {
    if (!(ctx->disp_spec && ctx->d_disp)) return fail(T41RX_ERR_STATE, "checkpoint carries display-FFT state but the display spectrum is off here");
    if (hdr[6] != ctx->disp_zoom) return fail(T41RX_ERR_STATE, "checkpoint: display-FFT state of another spectrumZoom");
  }
  const char *nr_sec = static_cast<const char *>(host_buf) + kStateHeaderBytes + path_bytes;
  const char *disp_sec = nr_sec + ((sec & kSecNr) ? nr_section_bytes(ctx->nchan) : 0);
  if (sec & kSecNr) {
    // what the kernels index with or divide by (nr_kernels.hip): Xanr()'s leak index, the spectral functions' ring pointers
    const float *anr = reinterpret_cast<const float *>(nr_sec);
    const float *spec = anr + (size_t)kAnrStRows * (size_t)ctx->nchan;
    for (int c = 0; c < ctx->nchan; ++c) {
      const float lidx = anr[(size_t)kAnrStLidx * ctx->nchan + c], ng = anr[(size_t)kAnrStNgamma * ctx->nchan + c];
      if (!(lidx >= 0.0f && lidx <= 1000.0f) || !std::isfinite(ng)) return fail(T41RX_ERR_STATE, "checkpoint: notch leak words out of range");
      const float *sc = spec + (size_t)kNrSpecFloats * (size_t)c + kNrScal;
      // (the kernel casts them with (int) and uses them as array indices and counters: integral values only)
      auto whole = [](float v) { return v == std::floor(v); };
      if (!(sc[0] >= 0.0f && sc[0] <= 2.0f) || !(sc[1] >= 0.0f && sc[1] <= 14.0f) || !(sc[2] == 0.0f || sc[2] == 1.0f || sc[2] == 2.0f) ||
          !(sc[3] >= 0.0f && sc[3] <= 1.0e6f) || !whole(sc[0]) || !whole(sc[1]) || !whole(sc[3]))
        return fail(T41RX_ERR_STATE, "checkpoint: noise-reduction ring pointers out of range or not integral");
    }
  }
  if (sec & kSecDisp) {
    const float *d = reinterpret_cast<const float *>(disp_sec);
    for (int c = 0; c < ctx->nchan; ++c) {
      int32_t ptr;
      std::memcpy(&ptr, d + (size_t)kDispFloats * (size_t)c + kDispPtr, sizeof(ptr));
      if (ptr < 0 || ptr >= 512) return fail(T41RX_ERR_STATE, "checkpoint: zoom_sample_ptr out of range");
    }
  }
  // what the kernels consume as it stands: the oscillator amplitude and the AGC state words
  const float *rec = reinterpret_cast<const float *>(static_cast<const char *>(host_buf) + kStateHeaderBytes);
  const size_t ag = st_agc(ctx->params.fft_length) + kAgcHistFloats;
  for (int c = 0; c < ctx->nchan; ++c) {
    const float *r = rec + sf * (size_t)c;
    NcoState ns;
    std::memcpy(&ns, r + kStNco, sizeof(ns));
    if (!(ns.r > 0.25 && ns.r < 4.0)) return fail(T41RX_ERR_STATE, "checkpoint: oscillator amplitude out of range");
    int32_t w[3];
    std::memcpy(w, r + ag + kAgcStState, sizeof(w));
    if (w[0] < 0 || w[0] > 4 || w[1] < 0 || w[1] > 1 || w[2] < 0 || w[2] > (1 << 20))
      return fail(T41RX_ERR_STATE, "checkpoint: AGC state words out of range");
    for (int k = 0; k < 4; ++k)
      if (!std::isfinite(r[ag + k])) return fail(T41RX_ERR_STATE, "checkpoint: AGC levels not finite");
    // AMDecodeSAM's statics (Demod.cpp:19-23): the kernel wraps phzerror with one conditional step
    // each way, which is the reference's pair of `while` loops only for a phase already in [0, 2 pi] (2 pi itself is
    // what a tiny negative phase + 2 pi rounds to: the loops leave it, and so does the kernel)
    const float phz = r[kStMisc + kMiscSamPhz], fil = r[kStMisc + kMiscSamFil], om = r[kStMisc + kMiscSamOmega];
    if (!(phz >= 0.0f && phz <= 6.2831855f) || !std::isfinite(fil) || !(std::fabs(fil) < 4.0f) || !(std::fabs(om) <= 1.05f))
      return fail(T41RX_ERR_STATE, "checkpoint: synchronous-detector PLL words out of range");
  }
  DeviceGuard g(ctx->device);
  HIP_TRY(hipDeviceSynchronize());
  if (sec & kSecNr) {
    const int rc = ensure_nr(ctx);
    if (rc != T41RX_OK) return rc;
  }
  HIP_TRY(hipMemcpy(ctx->d_state, rec, path_bytes, hipMemcpyHostToDevice));
  ctx->nco_sel = 0;  // (a checkpoint carries the current oscillator state in both slots)
  // The side stages' memories follow the checkpoint too: restored where it carries them, back to power-on where it
  // does not (a checkpoint taken before the stages first ran) -- never the values of the stream being replaced.
  if (sec & kSecNr) {
    const size_t ab = sizeof(float) * (size_t)kAnrStRows * (size_t)ctx->nchan;
    HIP_TRY(hipMemcpy(ctx->d_nr_anr, nr_sec, ab, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_nr_spec, nr_sec + ab, nr_section_bytes(ctx->nchan) - ab, hipMemcpyHostToDevice));
  } else if (ctx->d_nr_anr) {
    const int rc = reset_nr(ctx);
    if (rc != T41RX_OK) return rc;
  }
  if (sec & kSecDisp) {
    HIP_TRY(hipMemcpy(ctx->d_disp, disp_sec, disp_section_bytes(ctx->nchan), hipMemcpyHostToDevice));
  } else if (ctx->d_disp) {
    HIP_TRY(hipMemset(ctx->d_disp, 0, sizeof(float) * kDispFloats * (size_t)ctx->nchan));  // ZoomFFTPrep()
  }
  return pipe_timeouts_clear(ctx);  // the restored state is valid again
}

int t41rx_set_debug_taps(t41rx_ctx *ctx, float *d_post_nco, float *d_dec, float *d_demod, int max_frames) {
  if (!ctx) return fail(T41RX_ERR_ARG, "null argument");
  const bool any = d_post_nco || d_dec || d_demod;
  // (T41RX_STAMP_TAPS: the -DT41RX_STAMP diagnostic kernels of the long-FFT pipeline put their cycle stamps behind the demod tap)
  if (any && ctx->params.fft_length != 512 && !std::getenv("T41RX_STAMP_TAPS"))
    return fail(T41RX_ERR_UNSUPPORTED, "the stage taps are built for fft_length 512");
  if (any && max_frames <= 0) return fail(T41RX_ERR_ARG, "max_frames must be > 0");
  ctx->dbg_nco = d_post_nco;
  ctx->dbg_dec = d_dec;
  ctx->dbg_demod = d_demod;
  ctx->tap_frames = any ? max_frames : 0;
  return T41RX_OK;
}

int t41rx_set_audio_spectrum(t41rx_ctx *ctx, float *d_spect, float *d_max, int max_frames) {
  if (!ctx) return fail(T41RX_ERR_ARG, "null context");
  if ((d_spect == nullptr) != (d_max == nullptr)) return fail(T41RX_ERR_ARG, "set both pointers or neither");
  if (d_spect && ctx->params.fft_length != 512) return fail(T41RX_ERR_UNSUPPORTED, "the audio spectrum is built for fft_length 512");
  if ((reinterpret_cast<uintptr_t>(d_spect) | reinterpret_cast<uintptr_t>(d_max)) & 3u) return fail(T41RX_ERR_ARG, "unaligned pointer");
  if (d_spect && max_frames <= 0) return fail(T41RX_ERR_ARG, "max_frames must be > 0");
  ctx->spect = d_spect;
  ctx->spect_max = d_max;
  ctx->spect_frames = d_spect ? max_frames : 0;
  return T41RX_OK;
}

int t41rx_set_display_spectrum(t41rx_ctx *ctx, float *d_spec, float *d_spec_old, int spectrumZoom, int max_frames) {
  if (!ctx) return fail(T41RX_ERR_ARG, "null context");
  if ((d_spec == nullptr) != (d_spec_old == nullptr)) return fail(T41RX_ERR_ARG, "set both pointers or neither");
  DeviceGuard g(ctx->device);
  HIP_TRY(hipDeviceSynchronize());
  if (!d_spec) {
    ctx->disp_spec = ctx->disp_old = nullptr;
    ctx->disp_frames = 0;
    return T41RX_OK;
  }
  if (ctx->params.fft_length != 512) return fail(T41RX_ERR_UNSUPPORTED, "the display FFT is built for fft_length 512");
  if (spectrumZoom < 0 || spectrumZoom > 4) return fail(T41RX_ERR_ARG, "spectrumZoom must be 0 (1x) .. 4 (16x)");  // MAX_ZOOM_ENTRIES, ButtonProc.h:6
  if (max_frames <= 0) return fail(T41RX_ERR_ARG, "max_frames must be > 0");
  if ((reinterpret_cast<uintptr_t>(d_spec) | reinterpret_cast<uintptr_t>(d_spec_old)) & 3u) return fail(T41RX_ERR_ARG, "unaligned pointer");
  // Anything that fails from here on leaves the side output switched OFF (a previous successful
  // call's pointers must not survive next to a freed tap buffer: the display kernel would read it).
  ctx->disp_spec = ctx->disp_old = nullptr;
  const int had_frames = ctx->disp_frames;
  ctx->disp_frames = 0;
  if (max_frames > had_frames || !ctx->d_pre) {
    (void)hipFree(ctx->d_pre);
    ctx->d_pre = nullptr;
    HIP_TRY(hipMalloc((void **)&ctx->d_pre, sizeof(float) * 4096 * (size_t)max_frames * (size_t)ctx->nchan));
  }
  if (!ctx->d_disp) HIP_TRY(hipMalloc((void **)&ctx->d_disp, sizeof(float) * kDispFloats * (size_t)ctx->nchan));
  if (!ctx->d_win) {
    HIP_TRY(hipMalloc((void **)&ctx->d_win, sizeof(double) * 512));
    double w[512];
    for (int i = 0; i < 512; ++i) w[i] = 0.5 - 0.5 * std::cos(6.28 * i / 512);  // FFT.cpp:110, 222 ("Hanning", 6.28 as written)
    HIP_TRY(hipMemcpy(ctx->d_win, w, sizeof(w), hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemset(ctx->d_disp, 0, sizeof(float) * kDispFloats * (size_t)ctx->nchan));  // ZoomFFTPrep(): a fresh start
  if (max_frames < had_frames) max_frames = had_frames;  // (the tap buffer was kept: it still holds that many)
  ctx->disp_spec = d_spec;
  ctx->disp_old = d_spec_old;
  ctx->disp_frames = max_frames;
  ctx->disp_zoom = spectrumZoom;
  return T41RX_OK;
}

}  // extern "C"

// t41_sdr_amd/csrc/design.cpp -- host-side coefficient designer of the RX chain.
//
// Produces the arrays the reference recomputes on every filter change:
//   CalcFIRCoeffs      FIR.cpp:908-980   (Kaiser-windowed sinc low-pass, 4 resampler filters)
//   CalcCplxFIRCoeffs  FIR.cpp:1008-1065 (complex band-pass, 4-term Blackman-Harris)
//   InitFilterMask     Filter.cpp:260-284 (zero-pad + forward complex FFT -> FIR_filter_mask)
//   SetDecIntFilters   Filter.cpp:396-438
//   SetIIRCoeffs       FIR.cpp:1076-1116 (AM low-pass biquad, designed once at boot)
// plus the per-call scalars of Process.cpp (gains, level adjust, volume).
// Runs on the host once per filter change; the blob is uploaded (and broadcast over RCCL in
// multi-GPU runs).  All arithmetic keeps the reference's float/double operand types.
#include <cmath>
#include <cstring>
#include <vector>

#include "rx_internal.hpp"

namespace t41 {
namespace {

// FIR.h:10-16 -- the firmware redefines PI & friends as float literals
constexpr float kPi = 3.1415926535897932384626433832795f;
constexpr float kHalfPi = 1.5707963267948966192313216916398f;
constexpr float kTwoPi = 6.283185307179586476925286766559f;
constexpr float kFourPi = 2.0f * kTwoPi;
constexpr float kSixPi = 3.0f * kTwoPi;

constexpr float kDF1 = 4.0f, kDF = 8.0f;  // T41_SDR.ino:333-335
constexpr float kAtt = 90.0f;             // n_att, T41_SDR.ino:336

// modified Bessel I0 by its power series (Utility.cpp:211-229)
float bessel_i0(float x) {
  const float half = x / 2.0;
  float sum = 1.0, term = 1.0, k = 1.0;
  const float eps = 1e-9;
  do {
    float q = half / k;
    q *= q;
    term *= q;
    sum += term;
    k += 1.0;
  } while (term >= eps * sum);
  return sum;
}

// Utility.cpp:197-203
float msinc(int m, float fc) {
  const float x = m * kHalfPi;
  return m == 0 ? 1.0f : sinf(x * fc) / (fc * x);
}

// Kaiser low-pass of CalcFIRCoeffs (type 0 only: the only type the RX path requests)
void kaiser_lowpass(float *h, int ntaps, float fc_hz, float astop_db, float fs_hz) {
  const float fc = fc_hz / fs_hz;
  float beta;
  if (astop_db < 20.96)
    beta = 0.0;
  else if (astop_db >= 50.0)
    beta = 0.1102 * (astop_db - 8.71);
  else
    beta = 0.5842 * powf((astop_db - 20.96), 0.4) + 0.07886 * (astop_db - 20.96);
  const float i0b = bessel_i0(beta);
  const float fcf = fc * 2.0;
  // the reference walks ii = -n, -n+2, ..., n-2: n taps of an (n+1)-point symmetric design
  int tap = 0;
  for (int ii = -ntaps; ii < ntaps; ii += 2, ++tap) {
    const float x = (float)ii / (float)ntaps;
    const float win = bessel_i0(beta * sqrtf(1.0f - x * x)) / i0b;
    h[tap] = fcf * msinc(ii, fcf) * win;
  }
}

// CalcCplxFIRCoeffs with FIR_filter_window == 1 (FIR.cpp:10)
void complex_bandpass(std::vector<float> &ci, std::vector<float> &cq, int ntaps, float flo,
                      float fhi, float fs) {
  const float nFL = flo / fs, nFH = fhi / fs;
  const float nFc = (nFH - nFL) / 2.0;
  const float nFs = kPi * (nFH + nFL);
  const float centre = 0.5 * (float)(ntaps - 1);
  ci.assign(ntaps, 0.0f);
  cq.assign(ntaps, 0.0f);
  for (int i = 0; i < ntaps; ++i) {
    const float x = (float)i - centre;
    float z;
    if (fabsf(x) < 0.01) {
      z = 2.0 * nFc;
    } else {
      z = (float)sinf(kTwoPi * x * nFc) / (kPi * x) *
          (0.35875 - 0.48829 * cosf((kTwoPi * i) / (ntaps - 1)) +
           0.14128 * cosf((kFourPi * i) / (ntaps - 1)) - 0.01168 * cosf((kSixPi * i) / (ntaps - 1)));
    }
    ci[i] = z * cosf(nFs * x);
    cq[i] = z * sinf(nFs * x);
  }
}

// In-place forward complex FFT in f32 (the role arm_cfft_f32(maskS, ...) plays in
// InitFilterMask).  Stockham autosort radix-2; twiddles rounded once from double.
void fft_forward_f32(std::vector<float> &re, std::vector<float> &im) {
  const int n = (int)re.size();
  std::vector<float> tr(n), ti(n);
  std::vector<float> wr(n / 2), wi(n / 2);
  for (int k = 0; k < n / 2; ++k) {
    const double a = -2.0 * M_PI * (double)k / (double)n;
    wr[k] = (float)std::cos(a);
    wi[k] = (float)std::sin(a);
  }
  float *xr = re.data(), *xi = im.data(), *yr = tr.data(), *yi = ti.data();
  for (int half = n / 2, stride = 1; half >= 1; half >>= 1, stride <<= 1) {
    // n = 2 * half * stride
    for (int p = 0; p < half; ++p) {
      const float cr = wr[p * stride], cw = wi[p * stride];
      for (int q = 0; q < stride; ++q) {
        const int ia = q + stride * p, ib = q + stride * (p + half);
        const float ar = xr[ia], ai = xi[ia], br = xr[ib], bi = xi[ib];
        const float dr = ar - br, di = ai - bi;
        const int o0 = q + stride * (2 * p), o1 = q + stride * (2 * p + 1);
        yr[o0] = ar + br;
        yi[o0] = ai + bi;
        yr[o1] = dr * cr - di * cw;
        yi[o1] = dr * cw + di * cr;
      }
    }
    std::swap(xr, yr);
    std::swap(xi, yi);
  }
  if (xr != re.data()) {
    std::memcpy(re.data(), xr, sizeof(float) * n);
    std::memcpy(im.data(), xi, sizeof(float) * n);
  }
}

// SetIIRCoeffs, low-pass branch
void biquad_lowpass(float out[5], float f0, float q, float fs) {
  if (f0 > fs / 2.0) f0 = fs / 2.0;
  const float w0 = f0 * (kTwoPi / fs);
  const float sn = sinf(w0);
  const float alpha = sn / (q * 2.0);
  const float cs = cosf(w0);
  const float scale = 1.0 / (1.0 + alpha);
  out[0] = ((1.0 - cs) / 2.0) * scale;
  out[1] = (1.0 - cs) * scale;
  out[2] = out[0];
  out[3] = (2.0 * cs) * scale;
  out[4] = (-1.0 + alpha) * scale;
}

// The AGC's derived constants: AGCPrep() (DSP_Fn.cpp:444-468) then AGCLoadValues()
// (DSP_Fn.cpp:368-435) for one AGCMode, i.e. the boot sequence (T41_SDR.ino:791).  In the
// firmware a later mode change re-runs AGCLoadValues() only, so hang_thresh = 1.0 written by
// modes 3 / 4 would survive a switch back to 1 / 2; a context always gets the boot values.
// Every quantity is a float32_t global there; unsuffixed literals are double.
void agc_constants(const t41rx_params &p, float *out) {
  if (p.AGCMode == 0) return;  // stays zero
  const float tau_attack = 0.001, tau_fast_backaverage = 0.250, tau_fast_decay = 0.005;
  const float tau_hang_backmult = 0.500, tau_hang_decay = 0.100;
  const float max_input = 1.0, out_targ = 1.0, var_gain = 1.5, pop_ratio = 5.0;
  const int n_tau = 4;
  float tau_decay = 0.250, hangtime = 0.250, hang_thresh = 0.250;
  switch (p.AGCMode) {
    case 1: hangtime = 2.000; tau_decay = 2.000; break;
    case 2: hangtime = 1.000; tau_decay = 0.5; break;
    case 3: hang_thresh = 1.0; hangtime = 0.000; tau_decay = 0.250; break;
    default: hang_thresh = 1.0; hangtime = 0.0; tau_decay = 0.050; break;
  }
  const float sample_rate = (float)kSampleRate / kDF;
  const float max_gain = powf(10.0, (float)p.AGC_thresh / 20.0);
  auto one_minus_exp = [&](float tau) { return (float)(1.0 - expf(-1.0 / (sample_rate * tau))); };
  out[kAgcAttackBuffsize] = (float)(int)std::ceil(sample_rate * n_tau * tau_attack);
  out[kAgcAttackMult] = one_minus_exp(tau_attack);
  out[kAgcDecayMult] = one_minus_exp(tau_decay);
  out[kAgcFastDecayMult] = one_minus_exp(tau_fast_decay);
  out[kAgcFastBackmult] = one_minus_exp(tau_fast_backaverage);
  out[kAgcOnemFastBackmult] = 1.0 - out[kAgcFastBackmult];
  out[kAgcHangBackmult] = one_minus_exp(tau_hang_backmult);
  out[kAgcOnemHangBackmult] = 1.0 - out[kAgcHangBackmult];
  out[kAgcHangDecayMult] = one_minus_exp(tau_hang_decay);
  const float out_target = out_targ * (1.0 - expf(-(float)n_tau)) * 0.9999;
  out[kAgcOutTarget] = out_target;
  out[kAgcMinVolts] = out_target / (var_gain * max_gain);
  float tmp = log10f(out_target / (max_input * var_gain * max_gain));
  if (tmp == 0.0) tmp = 1e-16;
  out[kAgcSlopeConstant] = (out_target * (1.0 - 1.0 / var_gain)) / tmp;
  out[kAgcInvMaxInput] = 1.0 / max_input;
  tmp = powf(10.0, (hang_thresh - 1.0) / 0.125);
  out[kAgcHangLevel] = (max_input * tmp + (out_target / (var_gain * max_gain)) * (1.0 - tmp)) * 0.637;
  out[kAgcPopRatio] = pop_ratio;
  out[kAgcHangCount] = (float)(int)(hangtime * kSampleRate / kDF);
}

}  // namespace

const float kDeemphFir24000[kDeemphTaps] = {
    0.000481913f, -0.000816211f, -0.00205384f, -0.00264474f, -0.00258229f, -0.00247939f, -0.00305299f, -0.00448116f, -0.00620366f, -0.00737591f, -0.00761292f, -0.00737176f, -0.0075984f, -0.00890065f, -0.0109592f, -0.0127338f, -0.0133493f, -0.0129165f, -0.0125289f, -0.013351f, -0.0155348f, -0.0179452f, -0.0190498f, -0.0183068f, -0.016827f, -0.0165808f, -0.0186455f, -0.0219659f, -0.0238965f, -0.0223995f, -0.0182146f, -0.0149414f, -0.0163342f, -0.0223751f, -0.0271497f, -0.020849f, 0.00446391f, 0.0485999f, 0.100768f, 0.143223f, 0.159583f, 0.143223f, 0.100768f, 0.0485999f, 0.00446391f, -0.020849f, -0.0271497f, -0.0223751f, -0.0163342f, -0.0149414f, -0.0182146f, -0.0223995f, -0.0238965f, -0.0219659f, -0.0186455f, -0.0165808f, -0.016827f, -0.0183068f, -0.0190498f, -0.0179452f, -0.0155348f, -0.013351f, -0.0125289f, -0.0129165f, -0.0133493f, -0.0127338f, -0.0109592f, -0.00890065f, -0.0075984f, -0.00737176f, -0.00761292f, -0.00737591f, -0.00620366f, -0.00448116f, -0.00305299f, -0.00247939f, -0.00258229f, -0.00264474f, -0.00205384f, -0.000816211f, 0.000481913f};

// mag_coeffs[1..4] (FIR.cpp:582-680): the display zoom's 4-stage elliptic low-passes for 2x .. 16x
// (MAX_ZOOM_ENTRIES = 5, ButtonProc.h:6), a fixed table of the reference
const float kZoomIirCoeffs[4][20] = {
    {0.228454526413293696f, 0.077639329099949764f, 0.228454526413293696f, 0.635534925142242080f, -0.170083307068779194f, 0.436788292542003964f, 0.232307972937606161f, 0.436788292542003964f, 0.365885230717786780f, -0.471769788739400842f, 0.535974654742658707f, 0.557035600464780845f, 0.535974654742658707f, 0.125740787233286133f, -0.754725697183384336f, 0.501116342273565607f, 0.914877831284765408f, 0.501116342273565607f, 0.013862536615004284f, -0.930973052446900984f},
    {0.182208761527446556f, -0.222492493114674145f, 0.182208761527446556f, 1.326111070880959810f, -0.468036100821178802f, 0.337123762652097259f, -0.366352718812586853f, 0.337123762652097259f, 1.337053579516321200f, -0.644948386007929031f, 0.336163175380826074f, -0.199246162162897811f, 0.336163175380826074f, 1.354952684569386670f, -0.828032873168141115f, 0.178588201750411041f, 0.207271695028067304f, 0.178588201750411041f, 1.386486967455699220f, -0.950935065984588657f},
    {0.185643392652478922f, -0.332064345389014803f, 0.185643392652478922f, 1.654637402827731090f, -0.693859842743674182f, 0.327519300813245984f, -0.571358085216950418f, 0.327519300813245984f, 1.715375037176782860f, -0.799055553586324407f, 0.283656142708241688f, -0.441088976843048652f, 0.283656142708241688f, 1.778230635987093860f, -0.904453944560528522f, 0.079685368654848945f, -0.011231810140649204f, 0.079685368654848945f, 1.825046003243238070f, -0.973184930412286708f},
    {0.194769868656866380f, -0.379098413160710079f, 0.194769868656866380f, 1.824436402073870810f, -0.834877726226893380f, 0.333973874901496770f, -0.646106479315673776f, 0.333973874901496770f, 1.871892825636887640f, -0.893734096124207178f, 0.272903880596429671f, -0.513507745397738469f, 0.272903880596429671f, 1.918161772571113750f, -0.950461788366234739f, 0.053535383722369843f, -0.069683422367188122f, 0.053535383722369843f, 1.948900719896301760f, -0.986288064973853129f},
};

// ZoomFFTPrep(), FFT.cpp:38-39: CalcFIRCoeffs(Fir_Zoom_FFT_Decimate_coeffs, 4, 0.5 * SampleRate / 2^zoom, 60, 0, 0.0, SampleRate)
void design_zoom_fir(int spectrumZoom, float (&coeffs)[4]) {
  const float Fstop_Zoom = 0.5 * (float)kSampleRate / (1 << spectrumZoom);
  kaiser_lowpass(coeffs, 4, Fstop_Zoom, 60, (float)kSampleRate);
}

// the bins the filter passes, as Kim1_NR() / SpectralNoiseReduction() derive them from bands[].FLoCut / FHiCut
// (Noise.cpp:134-168, 421-433, 514-531): 93.75 Hz per bin of the 256-point transforms at 24 kS/s
void nr_vad_range(int FLoCut, int FHiCut, int *lo, int *hi) {
  float lf_freq, uf_freq;
  if (FLoCut <= 0 && FHiCut >= 0) {
    lf_freq = 0.0;
    uf_freq = std::fmax(-(float)FLoCut, (float)FHiCut);
  } else if (FLoCut > 0) {
    lf_freq = (float)FLoCut;
    uf_freq = (float)FHiCut;
  } else {
    uf_freq = -(float)FLoCut;
    lf_freq = -(float)FHiCut;
  }
  lf_freq /= (((float)kSampleRate / kDF) / 256);
  uf_freq /= (((float)kSampleRate / kDF) / 256);
  int VAD_low = (uint8_t)(int)lf_freq, VAD_high = (uint8_t)(int)uf_freq;
  if (VAD_low == VAD_high) VAD_high++;
  if (VAD_low < 1) VAD_low = 1;
  else if (VAD_low > 126) VAD_low = 126;
  if (VAD_high < 1) VAD_high = 1;
  else if (VAD_high > 128) VAD_high = 128;
  *lo = VAD_low;
  *hi = VAD_high;
}

bool params_valid(const t41rx_params &p, const char **why) {
  auto fail = [&](const char *m) {
    if (why) *why = m;
    return false;
  };
  if (!(p.fft_length == 512 || p.fft_length == 1024 || p.fft_length == 2048 || p.fft_length == 4096))
    return fail("fft_length must be 512, 1024, 2048 or 4096");
  if ((p.mode < T41RX_DEMOD_USB || p.mode > T41RX_DEMOD_NFM) && p.mode != T41RX_DEMOD_SAM)
    return fail("mode must be USB, LSB, AM, NFM or SAM");
  if (p.mode == T41RX_DEMOD_SAM && p.fft_length != 512) return fail("SAM is built for fft_length 512");
  if (p.FHiCut <= p.FLoCut) return fail("FHiCut must be greater than FLoCut");
  if (p.FHiCut > 12000 || p.FLoCut < -12000) return fail("filter cut-offs beyond +-12 kHz (24 kS/s Nyquist)");
  if (p.mode == T41RX_DEMOD_LSB && p.FLoCut >= 0) return fail("LSB needs FLoCut < 0 (level adjust uses pow(-FLoCut))");
  if (p.mode != T41RX_DEMOD_LSB && p.mode != T41RX_DEMOD_NFM && p.FHiCut <= 0)
    return fail("FHiCut must be > 0 (level adjust uses pow(FHiCut))");
  if (p.audioVolume < 0 || p.audioVolume > 100) return fail("audioVolume out of 0..100");
  if (p.nfmFilterBW <= 0 || p.nfmFilterBW > 96000) return fail("nfmFilterBW out of range");
  if (p.xmtMode < T41RX_SSB_MODE || p.xmtMode > T41RX_DATA_MODE) return fail("bad xmtMode");
  if (p.am_lpf_f0 <= 0) return fail("am_lpf_f0 must be > 0");
  if (p.AGCMode < 0 || p.AGCMode > 4) return fail("AGCMode must be 0 (off) .. 4 (fast)");
  if (p.AGC_thresh < -40 || p.AGC_thresh > 120) return fail("AGC_thresh out of -40..120 dB");
  if (p.nfm_demod < 0 || p.nfm_demod > 1) return fail("nfm_demod must be 0 (quadri-correlator) or 1 (atan2 + de-emphasis)");
  if (p.mode == T41RX_DEMOD_NFM && p.nfm_demod == 1 && p.fft_length != 512) return fail("nfm_demod = 1 is built for fft_length 512");
  if (p.nrOptionSelect < 0 || p.nrOptionSelect > 3) return fail("nrOptionSelect must be 0 (off), 1 (Kim), 2 (spectral) or 3 (LMS)");
  if (p.ANR_notchOn < 0 || p.ANR_notchOn > 1) return fail("ANR_notchOn must be 0 or 1");
  if ((p.nrOptionSelect != 0 || p.ANR_notchOn != 0) && p.fft_length != 512)
    return fail("noise reduction / notch are written for blocks of 256 audio samples: fft_length 512 only");
  if (!(p.NR_alpha >= 0.0f && p.NR_alpha <= 1.0f) || !(p.NR_beta >= 0.0f && p.NR_beta <= 1.0f) || !(p.NR_PSI >= 0.0f && p.NR_PSI < 1e30f))
    return fail("NR_alpha / NR_beta must be in [0, 1], NR_PSI >= 0");
  if (p.nrOptionSelect == 2) {
    // SpectralNoiseReduction()'s musical-noise smoothing reaches 2 NN - 1 = 17 bins below VAD_high and 11 above VAD_low
    // (Noise.cpp:556-574 with NN up to 9): for narrower pass bands the reference indexes outside NR_Nest / NR_G
    int lo, hi;
    nr_vad_range(p.FLoCut, p.FHiCut, &lo, &hi);
    if (hi < 17 || lo + 11 > 127) return fail("nrOptionSelect = 2: the reference indexes outside its arrays for this pass band");
  }
  return true;
}

int design_blob(const t41rx_params &p, void *blob, size_t blob_bytes) {
  const char *why = nullptr;
  if (!blob) return T41RX_ERR_ARG;
  if (!params_valid(p, &why)) return T41RX_ERR_ARG;
  const int N = p.fft_length;
  if (blob_bytes < blob_floats(N) * sizeof(float)) return T41RX_ERR_ARG;
  std::memset(blob, 0, blob_floats(N) * sizeof(float));
  BlobView v = blob_view(blob);
  v.header[0] = (int32_t)kBlobMagic;
  v.header[1] = T41RX_ABI_VERSION;
  v.header[2] = N;
  v.header[3] = p.mode;
  v.header[4] = (int32_t)sizeof(t41rx_params);
  std::memcpy(v.header + 8, &p, sizeof(t41rx_params));

  // --- filter mask: (N/2+1)-tap complex band-pass at 24 kS/s, zero-padded, FFT'd ---
  const int ntaps = N / 2 + 1;  // m_NumTaps, Filter.cpp:18
  std::vector<float> ci, cq;
  complex_bandpass(ci, cq, ntaps, (float)p.FLoCut, (float)p.FHiCut, (float)kSampleRate / kDF);
  std::vector<float> re(N, 0.0f), im(N, 0.0f);
  for (int i = 0; i < ntaps; ++i) {
    re[i] = ci[i];
    im[i] = cq[i];
  }
  // Filter.cpp:276-278 zero-fills interleaved floats N+1 .. 2N-1: that is im[N/2] and every
  // re/im beyond it, so the last tap keeps its I part only.
  im[N / 2] = 0.0f;
  fft_forward_f32(re, im);
  for (int k = 0; k < N; ++k) {
    v.mask[2 * k] = re[k];
    v.mask[2 * k + 1] = im[k];
  }

  // --- AM low-pass (boot-time design, never refreshed: SURVEY App. C #10) ---
  biquad_lowpass(v.lp1, (float)p.am_lpf_f0, 1.3, (float)kSampleRate / kDF);

  // --- resampler FIRs (Filter.cpp:396-417) ---
  int bw = p.FHiCut;
  if (bw < -p.FLoCut) bw = -p.FLoCut;
  if (bw > 10000) bw = 10000;
  kaiser_lowpass(v.dec1, kDec1Taps, (float)bw, kAtt, (float)kSampleRate);
  kaiser_lowpass(v.dec2, kDec2Taps, (float)bw, kAtt, (float)(kSampleRate / kDF1));
  kaiser_lowpass(v.int1, kInt1Taps, (float)bw, kAtt, (float)(kSampleRate / kDF1));
  kaiser_lowpass(v.int2, kInt2Taps, (float)bw, kAtt, (float)kSampleRate);
  if (p.mode == T41RX_DEMOD_NFM) {  // Process.cpp:259 -> Filter.cpp:429-438 (no 10 kHz cap)
    kaiser_lowpass(v.dec1, kDec1Taps, (float)p.nfmFilterBW, kAtt, (float)kSampleRate);
    kaiser_lowpass(v.dec2, kDec2Taps, (float)p.nfmFilterBW, kAtt, (float)(kSampleRate / kDF1));
  }

  agc_constants(p, v.agc);

  // --- per-call scalars ---
  float *s = v.scalars;
  s[kScRfGain] = (float)std::pow(10, (float)p.rfGainAllBands / 20);  // Process.cpp:117
  s[kScBandGain] = (float)p.RFgain;                                   // Process.cpp:133
  s[kScNegIqAmp] = -p.IQAmpCorrectionFactor;                          // Process.cpp:166
  s[kScIqPhase] = p.IQPhaseCorrectionFactor;
  {
    float fk;  // Process.cpp:482-490
    if (p.mode == T41RX_DEMOD_LSB)
      fk = -(float)p.FLoCut * 0.001;
    else
      fk = (float)p.FHiCut * 0.001;
    s[kScLevel] = (p.mode == T41RX_DEMOD_NFM) ? 1.0f : (float)(7.0874 * std::pow(fk, -1.232));
  }
  s[kScFixedGain] = 20.0f;  // DSP_Fn.cpp:453
  {
    const float x = p.audioVolume / 100.0f;  // Process.cpp:955-967
    const float ampl = 5 * x * x * x * x * x;
    s[kScOutScale] = kDF * ampl;  // Process.cpp:929
  }
  s[kScIqCorrOn] = (p.mode == T41RX_DEMOD_USB || p.mode == T41RX_DEMOD_LSB || p.mode == T41RX_DEMOD_AM || p.mode == T41RX_DEMOD_SAM) ? 1.0f : 0.0f;
  {
    int side = 0;  // Freq_Shift.cpp:108-120
    if (p.xmtMode == T41RX_CW_MODE) {
      if (p.mode == 1) side = p.CWFreqShift;
      else if (p.mode == 0) side = -p.CWFreqShift;
    }
    s[kScSideTone] = (float)side;
  }
  s[kScNfmDemod] = (float)p.nfm_demod;
  {
    // The synchronous detector's PLL constants, Demod.cpp:13-18, with the types of that file's
    // expressions (float globals, int and double literals) and gwv.cpp:64-65's omegaN = 200,
    // pll_fmax = 4000, which are constant-initialised and therefore in place when these run.
    const float TPI = 6.283185307179586476925286766559f;  // FIR.h:12-13
    const float omegaN = 200.0f, pll_fmax = +4000.0f;
    const int zeta_help = 65;
    const float zeta = (float)zeta_help / 100.0;
    s[kScSamWmin] = TPI * -pll_fmax * 1 / 24000;
    s[kScSamWmax] = TPI * pll_fmax * 1 / 24000;
    const float g1 = 1.0 - std::exp(-2.0 * omegaN * zeta * 1 / 24000);
    s[kScSamG1] = g1;
    // (exp() of the float argument evaluates in double, like cos(NCO_INC) in FreqShift2: DESIGN.md section 2)
    s[kScSamG2] = -g1 + 2.0 * (1 - std::exp((double)(-omegaN * zeta * 1 / 24000)) * cosf(omegaN * 1 / 24000 * sqrtf(1.0 - zeta * zeta)));
  }
  return T41RX_OK;
}

}  // namespace t41

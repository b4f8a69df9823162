/*
 * oracle/t41_oracle.c -- CPU restatement of ProcessIQData() and the CMSIS-DSP f32 primitives it
 * calls.  TEST INFRASTRUCTURE ONLY (see t41_oracle.h): never linked into the product.
 * PARITY UNPINNED (no reference tests/golden vectors exist; CMSIS-DSP is absent).
 *
 * Citations are relative to /root/reference/software/T41_SDR/.
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 *
 * Numeric conventions restated here:
 *  - every expression keeps the reference's C++ operand types, so float*double promotes to
 *    double exactly where the reference's does (unsuffixed literals are double: Teensy 4.x
 *    builds do not use -fsingle-precision-constant);
 *  - FIR/biquad accumulations are separate multiply and add roundings in tap order (the
 *    documented scalar CMSIS loops);
 *  - arm_cfft_f32 is restated as "unscaled forward DFT, natural order; inverse = conjugate,
 *    forward, conjugate, times 1/N" with f32 twiddles.  CMSIS uses a radix-8 decomposition
 *    for N=512/4096; this file uses radix-2, so only the rounding pattern (~1e-7 relative)
 *    can differ from a real CMSIS build.
 */
#define _GNU_SOURCE
#include "t41_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* FIR.h:10-16: the reference redefines these as *float* constants */
#define PI_F 3.1415926535897932384626433832795f
#define HALF_PI_F 1.5707963267948966192313216916398f
#define TWO_PI_F 6.283185307179586476925286766559f
#define TPI_F TWO_PI_F
#define PIH_F HALF_PI_F
#define FOURPI_F (2.0f * TPI_F)
#define SIXPI_F (3.0f * TPI_F)

#define SAMPLE_RATE 192000 /* T41_SDR.ino:129 */
static const float DF1 = 4.0f;  /* T41_SDR.ino:333 */
static const float DF2 = 2.0f;  /* T41_SDR.ino:334 */
static const float DF = 8.0f;   /* T41_SDR.ino:335 (DF1*DF2) */
static const float N_ATT = 90.0f; /* T41_SDR.ino:336 */

void t41o_default_params(t41o_params *p) {
  memset(p, 0, sizeof(*p));
  p->fft_length = 512;            /* SDT.h:39 */
  p->mode = T41O_DEMOD_USB;       /* bands[] 20M row, T41_SDR.ino:163 */
  p->FLoCut = 200;
  p->FHiCut = 3000;
  p->rfGainAllBands = 1;          /* gwv.cpp:17 */
  p->RFgain = 1;                  /* bands[].RFgain */
  p->IQAmpCorrectionFactor = 1.0f;   /* gwv.cpp:71 */
  p->IQPhaseCorrectionFactor = 0.0f; /* gwv.cpp:72 */
  p->AGCMode = 0;                 /* fixed gain */
  p->audioVolume = 30;            /* gwv.cpp:16 */
  p->nfmFilterBW = 12000;         /* Filter.cpp:16 */
  p->xmtMode = T41O_SSB_MODE;     /* gwv.cpp:22 */
  p->CWFreqShift = 750;
  p->am_lpf_f0 = 3000;            /* boot band 40M LSB -200/-3000, T41_SDR.ino:560-563 */
  p->AGC_thresh = 20;             /* bands[] "AGC" column, T41_SDR.ino:145-168 */
  p->nfm_demod = 0;               /* the live code path, Process.cpp:716 */
  p->nrOptionSelect = 0;          /* gwv.cpp:23 */
  p->ANR_notchOn = 0;             /* Process.cpp:45 */
  p->NR_PSI = 0.0;                /* gwv.cpp:61-63 */
  p->NR_alpha = 0.95;
  p->NR_beta = 0.85;
}

/* ------------------------------------------------------------------------------------------
 * Coefficient design
 * ---------------------------------------------------------------------------------------- */

/* Utility.cpp:211-229 */
float t41o_Izero(float x) {
  float x2 = x / 2.0;
  float summe = 1.0;
  float ds = 1.0;
  float di = 1.0;
  float errorlimit = 1e-9;
  float tmp;
  do {
    tmp = x2 / di;
    tmp *= tmp;
    ds *= tmp;
    summe += ds;
    di += 1.0;
  } while (ds >= errorlimit * summe);
  return summe;
}

/* Utility.cpp:197-203 */
float t41o_MSinc(int m, float fc) {
  float x = m * PIH_F;
  if (m == 0) return 1.0f;
  return sinf(x * fc) / (fc * x);
}

/* FIR.cpp:908-980.  Types 0 (LP), 1 (HP), 2/3 (BP/notch); the path only uses type 0. */
void t41o_CalcFIRCoeffs(float *coeffs, int numCoeffs, float fc, float Astop, int type, float dfc,
                        float Fsamprate) {
  int nc = numCoeffs;
  float Beta;
  float izb;
  float fcf = fc;
  float x, w;
  fc = fc / Fsamprate;
  dfc = dfc / Fsamprate;

  if (Astop < 20.96) {
    Beta = 0.0;
  } else if (Astop >= 50.0) {
    Beta = 0.1102 * (Astop - 8.71);
  } else {
    Beta = 0.5842 * powf((Astop - 20.96), 0.4) + 0.07886 * (Astop - 20.96);
  }
  izb = t41o_Izero(Beta);
  if (type == 0) {
    fcf = fc * 2.0;
    nc = numCoeffs;
  } else if (type == 1) {
    fcf = -fc;
    nc = 2 * (numCoeffs / 2);
  } else {
    fcf = dfc;
    nc = 2 * (numCoeffs / 2);
  }
  /* FIR.cpp:963-967: ii = -nc, -nc+2, ..., nc-2  ->  nc taps of a (nc+1)-long symmetric
   * design whose last tap is never written */
  for (int ii = -nc, jj = 0; ii < nc; ii += 2, jj++) {
    x = (float)ii / (float)nc;
    w = t41o_Izero(Beta * sqrtf(1.0f - x * x)) / izb;
    coeffs[jj] = fcf * t41o_MSinc(ii, fcf) * w;
  }
  if (type == 1) {
    coeffs[nc / 2] += 1;
  } else if (type == 2) {
    for (int jj = 0; jj < nc + 1; jj++) coeffs[jj] *= 2.0f * cosf(PIH_F * (2 * jj - nc) * fc);
  } else if (type == 3) {
    for (int jj = 0; jj < nc + 1; jj++) coeffs[jj] *= -2.0f * cosf(PIH_F * (2 * jj - nc) * fc);
    coeffs[nc / 2] += 1;
  }
}

/* FIR.cpp:1008-1065 with FIR_filter_window == 1 (FIR.cpp:10): 4-term Blackman-Harris */
void t41o_CalcCplxFIRCoeffs(float *cI, float *cQ, int numCoeffs, float FLoCut, float FHiCut,
                            float SampleRate) {
  float nFL = FLoCut / SampleRate;
  float nFH = FHiCut / SampleRate;
  float nFc = (nFH - nFL) / 2.0;
  float nFs = PI_F * (nFH + nFL);
  float fCenter = 0.5 * (float)(numCoeffs - 1);
  float x, z;
  for (int i = 0; i < numCoeffs; i++) {
    x = (float)i - fCenter;
    float ax = (float)i - fCenter;
    if (ax < 0) ax = -ax;
    if (ax < 0.01) {
      z = 2.0 * nFc;
    } else {
      z = (float)sinf(TWO_PI_F * x * nFc) / (PI_F * x) *
          (0.35875 - 0.48829 * cosf((TWO_PI_F * i) / (numCoeffs - 1)) +
           0.14128 * cosf((FOURPI_F * i) / (numCoeffs - 1)) -
           0.01168 * cosf((SIXPI_F * i) / (numCoeffs - 1)));
    }
    cI[i] = z * cosf(nFs * x);
    cQ[i] = z * sinf(nFs * x);
  }
}

/* FIR.cpp:1076-1116, filter_type 0 (low-pass) and 3 (notch) */
void t41o_SetIIRCoeffs(float cs[5], float f0, float Q, float sample_rate, int filter_type) {
  if (f0 > sample_rate / 2.0) f0 = sample_rate / 2.0;
  float w0 = f0 * (TPI_F / sample_rate);
  float sinW0 = sinf(w0);
  float alpha = sinW0 / (Q * 2.0);
  float cosW0 = cosf(w0);
  float scale = 1.0 / (1.0 + alpha);
  if (filter_type == 0) {
    cs[0] = ((1.0 - cosW0) / 2.0) * scale;
    cs[1] = (1.0 - cosW0) * scale;
    cs[2] = cs[0];
    cs[3] = (2.0 * cosW0) * scale;
    cs[4] = (-1.0 + alpha) * scale;
  } else if (filter_type == 3) {
    cs[0] = 1.0;
    cs[1] = -2.0 * cosW0;
    cs[2] = 1.0;
    cs[3] = 2.0 * cosW0 * scale;
    cs[4] = alpha - 1.0;
  }
}

static int valid_fft_length(int n) { return n == 512 || n == 1024 || n == 2048 || n == 4096; }

/* CalcFilters() Filter.cpp:235-249 -> CalcCplxFIRCoeffs + InitFilterMask (Filter.cpp:260-284)
 * + SetDecIntFilters (Filter.cpp:396-417); NFM re-designs dec1/dec2 every block
 * (Process.cpp:259 -> Filter.cpp:429-438). */
/* AGCPrep() (DSP_Fn.cpp:444-468) followed by AGCLoadValues() (DSP_Fn.cpp:368-435) for one
 * AGCMode, as at boot (T41_SDR.ino:791) with AGCMode already read from EEPROM.  The firmware's
 * globals are history dependent (hang_thresh = 1.0 written by modes 3/4 survives a later switch
 * to mode 1/2 because only AGCLoadValues() is re-run, MenuProc.cpp:279); this restates the
 * boot sequence.  Every variable is a float32_t global in the reference. */
static void agc_load_values(const t41o_params *p, float *agc) {
  memset(agc, 0, sizeof(float) * T41O_AGC_NCONST);
  if (p->AGCMode == 0) return;
  /* AGCPrep */
  float tau_attack = 0.001;
  float tau_decay = 0.250;
  int n_tau = 4;
  float max_gain = 10000.0;
  float max_input = 1.0;
  float out_targ = 1.0;
  float var_gain = 1.5;
  float tau_fast_backaverage = 0.250;
  float tau_fast_decay = 0.005;
  float pop_ratio = 5.0;
  float tau_hang_backmult = 0.500;
  float hangtime = 0.250;
  float hang_thresh = 0.250;
  float tau_hang_decay = 0.100;
  /* AGCLoadValues */
  float tmp;
  float sample_rate = (float)SAMPLE_RATE / DF;
  switch (p->AGCMode) {
    case 1: hangtime = 2.000; tau_decay = 2.000; break;                     /* agcLONG */
    case 2: hangtime = 1.000; tau_decay = 0.5; break;                       /* agcSLOW */
    case 3: hang_thresh = 1.0; hangtime = 0.000; tau_decay = 0.250; break;  /* agcMED */
    case 4: hang_thresh = 1.0; hangtime = 0.0; tau_decay = 0.050; break;    /* agcFAST */
    default: break;
  }
  max_gain = powf(10.0, (float)p->AGC_thresh / 20.0);
  int attack_buffsize = (int)ceil(sample_rate * n_tau * tau_attack);
  float attack_mult = 1.0 - expf(-1.0 / (sample_rate * tau_attack));
  float decay_mult = 1.0 - expf(-1.0 / (sample_rate * tau_decay));
  float fast_decay_mult = 1.0 - expf(-1.0 / (sample_rate * tau_fast_decay));
  float fast_backmult = 1.0 - expf(-1.0 / (sample_rate * tau_fast_backaverage));
  float onemfast_backmult = 1.0 - fast_backmult;
  float out_target = out_targ * (1.0 - expf(-(float)n_tau)) * 0.9999;
  float min_volts = out_target / (var_gain * max_gain);
  tmp = log10f(out_target / (max_input * var_gain * max_gain));
  if (tmp == 0.0) tmp = 1e-16;
  float slope_constant = (out_target * (1.0 - 1.0 / var_gain)) / tmp;
  float inv_max_input = 1.0 / max_input;
  tmp = powf(10.0, (hang_thresh - 1.0) / 0.125);
  float hang_level = (max_input * tmp + (out_target / (var_gain * max_gain)) * (1.0 - tmp)) * 0.637;
  float hang_backmult = 1.0 - expf(-1.0 / (sample_rate * tau_hang_backmult));
  float onemhang_backmult = 1.0 - hang_backmult;
  float hang_decay_mult = 1.0 - expf(-1.0 / (sample_rate * tau_hang_decay));

  agc[T41O_AGC_ATTACK_MULT] = attack_mult;
  agc[T41O_AGC_DECAY_MULT] = decay_mult;
  agc[T41O_AGC_FAST_DECAY_MULT] = fast_decay_mult;
  agc[T41O_AGC_FAST_BACKMULT] = fast_backmult;
  agc[T41O_AGC_ONEMFAST_BACKMULT] = onemfast_backmult;
  agc[T41O_AGC_HANG_BACKMULT] = hang_backmult;
  agc[T41O_AGC_ONEMHANG_BACKMULT] = onemhang_backmult;
  agc[T41O_AGC_HANG_DECAY_MULT] = hang_decay_mult;
  agc[T41O_AGC_OUT_TARGET] = out_target;
  agc[T41O_AGC_MIN_VOLTS] = min_volts;
  agc[T41O_AGC_SLOPE_CONSTANT] = slope_constant;
  agc[T41O_AGC_INV_MAX_INPUT] = inv_max_input;
  agc[T41O_AGC_HANG_LEVEL] = hang_level;
  agc[T41O_AGC_POP_RATIO] = pop_ratio;
  agc[T41O_AGC_HANG_COUNT] = (float)(int)(hangtime * SAMPLE_RATE / DF); /* DSP_Fn.cpp:550 */
  agc[T41O_AGC_ATTACK_BUFFSIZE] = (float)attack_buffsize;
}

int t41o_design(const t41o_params *p, t41o_coeffs *c) {
  const int N = p->fft_length;
  if (!valid_fft_length(N)) return -1;
  if (p->AGCMode < 0 || p->AGCMode > 4) return -2;
  memset(c, 0, sizeof(*c));
  agc_load_values(p, c->agc);
  const int m_NumTaps = N / 2 + 1; /* Filter.cpp:18 */
  float *cI = (float *)calloc((size_t)m_NumTaps, sizeof(float));
  float *cQ = (float *)calloc((size_t)m_NumTaps, sizeof(float));
  t41o_CalcCplxFIRCoeffs(cI, cQ, m_NumTaps, (float)p->FLoCut, (float)p->FHiCut,
                         (float)SAMPLE_RATE / DF);
  for (int i = 0; i < m_NumTaps; i++) { /* Filter.cpp:269-274 */
    c->mask[i * 2] = cI[i];
    c->mask[i * 2 + 1] = cQ[i];
  }
  for (int i = N + 1; i < N * 2; i++) c->mask[i] = 0.0; /* Filter.cpp:276-278: wipes cQ[N/2] */
  t41o_cfft_f32(c->mask, N, 0);                          /* Filter.cpp:282 */
  free(cI);
  free(cQ);

  /* AM low-pass: designed once at boot for the boot band, never redesigned
   * (T41_SDR.ino:560-566, Filter.cpp:242-244) */
  t41o_SetIIRCoeffs(c->biquad_lowpass1, (float)p->am_lpf_f0, 1.3, (float)SAMPLE_RATE / DF, 0);

  int filter_BW_highest = p->FHiCut; /* Filter.cpp:400-410 */
  if (filter_BW_highest < -p->FLoCut) filter_BW_highest = -p->FLoCut;
  int LP_F_help = filter_BW_highest;
  if (LP_F_help > 10000) LP_F_help = 10000;
  t41o_CalcFIRCoeffs(c->dec1, T41O_N_DEC1_TAPS, (float)LP_F_help, N_ATT, 0, 0.0,
                     (float)SAMPLE_RATE);
  t41o_CalcFIRCoeffs(c->dec2, T41O_N_DEC2_TAPS, (float)LP_F_help, N_ATT, 0, 0.0,
                     (float)(SAMPLE_RATE / DF1));
  t41o_CalcFIRCoeffs(c->int1, T41O_N_INT1_TAPS, (float)LP_F_help, N_ATT, 0, 0.0,
                     (float)(SAMPLE_RATE / DF1));
  t41o_CalcFIRCoeffs(c->int2, T41O_N_INT2_TAPS, (float)LP_F_help, N_ATT, 0, 0.0,
                     (float)SAMPLE_RATE);
  if (p->mode == T41O_DEMOD_NFM) { /* Filter.cpp:429-438: no 10 kHz cap here */
    int bw = p->nfmFilterBW;
    t41o_CalcFIRCoeffs(c->dec1, T41O_N_DEC1_TAPS, (float)bw, N_ATT, 0, 0.0, (float)SAMPLE_RATE);
    t41o_CalcFIRCoeffs(c->dec2, T41O_N_DEC2_TAPS, (float)bw, N_ATT, 0, 0.0,
                       (float)(SAMPLE_RATE / DF1));
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * CMSIS-DSP f32 primitives
 * ---------------------------------------------------------------------------------------- */

/* f32 twiddle tables (cos, sin of 2*pi*k/n for k < n/2), the analogue of CMSIS' twiddleCoef_N
 * constant tables: computed in double, rounded once to f32.  Built once per size. */
static float *g_tw[13];
static pthread_mutex_t g_tw_lock = PTHREAD_MUTEX_INITIALIZER;

static const float *twiddles_for(int n, int logn) {
  float *tw = __atomic_load_n(&g_tw[logn], __ATOMIC_ACQUIRE);
  if (tw) return tw;
  pthread_mutex_lock(&g_tw_lock);
  tw = g_tw[logn];
  if (!tw) {
    tw = (float *)malloc(sizeof(float) * (size_t)n);
    for (int k = 0; k < n / 2; k++) {
      double a = 2.0 * 3.14159265358979323846 * (double)k / (double)n;
      tw[2 * k] = (float)cos(a);
      tw[2 * k + 1] = (float)sin(a);
    }
    __atomic_store_n(&g_tw[logn], tw, __ATOMIC_RELEASE);
  }
  pthread_mutex_unlock(&g_tw_lock);
  return tw;
}

/* arm_cfft_f32(S, buf, ifftFlag, bitReverseFlag=1): in-place, interleaved re/im. */
void t41o_cfft_f32(float *buf, int n, int ifft) {
  int logn = 0;
  while ((1 << logn) < n) logn++;
  if (logn > 12 || (1 << logn) != n) return;
  const float *tw = twiddles_for(n, logn);
  if (ifft) /* CMSIS: conjugate input */
    for (int i = 0; i < n; i++) buf[2 * i + 1] = -buf[2 * i + 1];
  /* bit reversal, then radix-2 decimation-in-time */
  for (int i = 0, j = 0; i < n; i++) {
    if (i < j) {
      float tr = buf[2 * i], ti = buf[2 * i + 1];
      buf[2 * i] = buf[2 * j];
      buf[2 * i + 1] = buf[2 * j + 1];
      buf[2 * j] = tr;
      buf[2 * j + 1] = ti;
    }
    int bit = n >> 1;
    while (bit && (j & bit)) {
      j ^= bit;
      bit >>= 1;
    }
    j |= bit;
  }
  for (int len = 2; len <= n; len <<= 1) {
    int half = len >> 1;
    int step = n / len;
    for (int base = 0; base < n; base += len) {
      for (int k = 0; k < half; k++) {
        float wr = tw[2 * (k * step)];
        float wi = -tw[2 * (k * step) + 1]; /* e^{-j a} */
        int a = base + k, b = base + k + half;
        float xr = buf[2 * b], xi = buf[2 * b + 1];
        float tr = xr * wr - xi * wi;
        float ti = xr * wi + xi * wr;
        float ur = buf[2 * a], ui = buf[2 * a + 1];
        buf[2 * a] = ur + tr;
        buf[2 * a + 1] = ui + ti;
        buf[2 * b] = ur - tr;
        buf[2 * b + 1] = ui - ti;
      }
    }
  }
  if (ifft) { /* CMSIS: conjugate and scale by 1/fftLen */
    float inv = 1.0f / (float)n;
    for (int i = 0; i < n; i++) {
      buf[2 * i] = buf[2 * i] * inv;
      buf[2 * i + 1] = -buf[2 * i + 1] * inv;
    }
  }
}

/* arm_fir_decimate_f32: state = [numTaps-1 history | blockSize new]; output n =
 * sum_i pCoeffs[i]*state[n*M+i] accumulated in tap order from 0; history = last numTaps-1. */
void t41o_fir_decimate_f32(const float *coeffs, int ntaps, int M, float *state, const float *src,
                           float *dst, int blockSize) {
  const int nout = blockSize / M;
  float *cur = state + (ntaps - 1);
  for (int i = 0; i < blockSize; i++) cur[i] = src[i]; /* src may alias dst: copied first */
  for (int n = 0; n < nout; n++) {
    const float *px = state + n * M;
    float acc = 0.0f;
    for (int i = 0; i < ntaps; i++) acc += px[i] * coeffs[i];
    dst[n] = acc;
  }
  memmove(state, state + nout * M, sizeof(float) * (size_t)(ntaps - 1));
}

/* arm_fir_interpolate_f32: phaseLength P = numTaps/L; state = [P-1 history | new]; for input
 * n and j = 1..L: out[n*L+j-1] = sum_t state[n+t]*pCoeffs[(L-j)+t*L]. */
void t41o_fir_interpolate_f32(const float *coeffs, int ntaps, int L, float *state,
                              const float *src, float *dst, int blockSize) {
  const int P = ntaps / L;
  float *cur = state + (P - 1);
  for (int i = 0; i < blockSize; i++) cur[i] = src[i];
  for (int n = 0; n < blockSize; n++) {
    for (int j = 1; j <= L; j++) {
      float acc = 0.0f;
      const float *px = state + n;
      const float *pc = coeffs + (L - j);
      for (int t = 0; t < P; t++) {
        acc += px[t] * pc[t * L];
      }
      dst[n * L + (j - 1)] = acc;
    }
  }
  memmove(state, state + blockSize, sizeof(float) * (size_t)(P - 1));
}

/* arm_biquad_cascade_df2T_f32, one stage; coeffs {b0,b1,b2,a1,a2} with a's pre-negated */
void t41o_biquad_df2T_f32(const float c[5], float st[2], const float *src, float *dst, int n) {
  float d1 = st[0], d2 = st[1];
  for (int i = 0; i < n; i++) {
    float x = src[i];
    float acc = c[0] * x + d1;
    d1 = c[1] * x + d2;
    d1 += c[3] * acc;
    d2 = c[2] * x;
    d2 += c[4] * acc;
    dst[i] = acc;
  }
  st[0] = d1;
  st[1] = d2;
}

/* arm_biquad_cascade_df1_f32, one stage; state {x[n-1],x[n-2],y[n-1],y[n-2]} */
void t41o_biquad_df1_f32(const float c[5], float st[4], const float *src, float *dst, int n) {
  float x1 = st[0], x2 = st[1], y1 = st[2], y2 = st[3];
  for (int i = 0; i < n; i++) {
    float x = src[i];
    float acc = (c[0] * x) + (c[1] * x1) + (c[2] * x2) + (c[3] * y1) + (c[4] * y2);
    x2 = x1;
    x1 = x;
    y2 = y1;
    y1 = acc;
    dst[i] = acc;
  }
  st[0] = x1;
  st[1] = x2;
  st[2] = y1;
  st[3] = y2;
}

/* ------------------------------------------------------------------------------------------
 * Channel state
 * ---------------------------------------------------------------------------------------- */
struct t41o_channel {
  int N, D, L;
  float dc_state[2];              /* HP_DC_Butter_state2, Process.cpp:42 (shared by I and Q) */
  double Osc_Vect_Q, Osc_Vect_I;  /* Freq_Shift.cpp:13-14 */
  float *dec1_I_state, *dec1_Q_state; /* T41_SDR.ino:388,393 */
  float *dec2_I_state, *dec2_Q_state; /* T41_SDR.ino:389-390 */
  float *int1_state, *int2_state;     /* T41_SDR.ino:394,391 */
  float *last_L, *last_R;             /* last_sample_buffer_L/R, T41_SDR.ino:403-404 */
  int first_block;                    /* Process.cpp:47 */
  float wold;                         /* Process.cpp:73 */
  float lp1_state[4];                 /* biquad_lowpass1_state, T41_SDR.ino:373 */
  float nfm_last_i, nfm_last_q;       /* Demod.cpp:221-222 */
  float nfm_last_phase;               /* fmdemod_atan_cf's static, Demod.cpp:373 */
  float sam_phzerror, sam_fil_out, sam_omega2; /* AMDecodeSAM's PLL statics, Demod.cpp:19-23 */
  /* AGC() statics and globals, DSP_Fn.cpp:28-36, 481-492 */
  uint8_t agc_decay_type, agc_state;
  float *agc_abs_ring, *agc_ring;     /* [RB_SIZE], [2*RB_SIZE] */
  float agc_fast_backaverage, agc_hang_backaverage, agc_ring_max, agc_save_volts, agc_volts;
  int agc_hang_counter, agc_out_index;
  uint32_t agc_in_index;
  int agc_in_index_set;               /* AGCLoadValues(): in_index = attack_buffsize + out_index */
  float agc_edges[25];                /* test aid: how often state a was followed by state b, [5*a + b] */
  /* working buffers (the reference's globals) */
  float *float_buffer_L, *float_buffer_R, *float_buffer_L_EX, *float_buffer_R_EX;
  float *FFT_buffer, *iFFT_buffer;
  /* taps */
  float *tap_ncoI, *tap_ncoQ, *tap_decI, *tap_decQ, *tap_ifft, *tap_demod, *tap_volts;
  /* Process.cpp:550-570: audioSpectBuffer, audioMaxSquared, AudioMaxIndex, audioMaxSquaredAve */
  float audioSpectBuffer[1024];
  float audioMaxSquared, audioMaxSquaredAve;
  uint32_t AudioMaxIndex;
  /* display FFT (FFT.cpp): spectrumZoom (-1 = not computed), the zoom filters' state, the ring, the low-pass memory */
  int display_zoom;
  float zoom_iir_I_state[16], zoom_iir_Q_state[16];   /* IIR_biquad_Zoom_FFT_[IQ]_state, T41_SDR.ino:378-379 */
  float zoom_fir_coeffs[4];                           /* Fir_Zoom_FFT_Decimate_coeffs, FFT.cpp:20 */
  float *zoom_fir_I_state, *zoom_fir_Q_state;         /* 3 + frame_len */
  float FFT_ring_buffer_x[512], FFT_ring_buffer_y[512];
  int zoom_sample_ptr;
  float FFT_spec[512], FFT_spec_old[512];
  t41o_nr *nr;                                        /* Noise.cpp state (t41_nr_oracle.c) */
};

static float *fzalloc(size_t n) { return (float *)calloc(n, sizeof(float)); }

/* DSP_Fn.cpp:18-20, 470: RB_SIZE = (int)(24000.0 * 8 * 0.01 + 1) */
#define AGC_RB_SIZE ((int)(24000.0 * 8 * 0.01 + 1))

t41o_channel *t41o_channel_create(int fft_length) {
  if (!valid_fft_length(fft_length)) return NULL;
  t41o_channel *ch = (t41o_channel *)calloc(1, sizeof(*ch));
  const int N = fft_length, D = N / 2, L = 4 * N; /* N_BLOCKS*BUFFER_SIZE, T41_SDR.ino:368 */
  ch->N = N;
  ch->D = D;
  ch->L = L;
  ch->dec1_I_state = fzalloc((size_t)(T41O_N_DEC1_TAPS - 1 + L));
  ch->dec1_Q_state = fzalloc((size_t)(T41O_N_DEC1_TAPS - 1 + L));
  ch->dec2_I_state = fzalloc((size_t)(T41O_N_DEC2_TAPS - 1 + L / 4));
  ch->dec2_Q_state = fzalloc((size_t)(T41O_N_DEC2_TAPS - 1 + L / 4));
  ch->int1_state = fzalloc((size_t)(24 - 1 + D));
  ch->int2_state = fzalloc((size_t)(8 - 1 + 2 * D));
  ch->last_L = fzalloc((size_t)D);
  ch->last_R = fzalloc((size_t)D);
  ch->float_buffer_L = fzalloc((size_t)L);
  ch->float_buffer_R = fzalloc((size_t)L);
  ch->float_buffer_L_EX = fzalloc((size_t)L);
  ch->float_buffer_R_EX = fzalloc((size_t)L);
  ch->FFT_buffer = fzalloc((size_t)(2 * N));
  ch->iFFT_buffer = fzalloc((size_t)(2 * N + 1));
  ch->zoom_fir_I_state = fzalloc((size_t)(3 + L));
  ch->zoom_fir_Q_state = fzalloc((size_t)(3 + L));
  ch->display_zoom = -1;
  ch->tap_ncoI = fzalloc((size_t)L);
  ch->tap_ncoQ = fzalloc((size_t)L);
  ch->tap_decI = fzalloc((size_t)D);
  ch->tap_decQ = fzalloc((size_t)D);
  ch->tap_ifft = fzalloc((size_t)(2 * N));
  ch->tap_demod = fzalloc((size_t)D);
  ch->tap_volts = fzalloc((size_t)D);
  ch->agc_abs_ring = fzalloc(AGC_RB_SIZE);
  ch->agc_ring = fzalloc(2 * AGC_RB_SIZE);
  ch->nr = t41o_nr_create();
  t41o_channel_reset(ch);
  return ch;
}

void t41o_channel_destroy(t41o_channel *ch) {
  if (!ch) return;
  free(ch->dec1_I_state);
  free(ch->dec1_Q_state);
  free(ch->dec2_I_state);
  free(ch->dec2_Q_state);
  free(ch->int1_state);
  free(ch->int2_state);
  free(ch->last_L);
  free(ch->last_R);
  free(ch->float_buffer_L);
  free(ch->float_buffer_R);
  free(ch->float_buffer_L_EX);
  free(ch->float_buffer_R_EX);
  free(ch->FFT_buffer);
  free(ch->iFFT_buffer);
  free(ch->tap_ncoI);
  free(ch->tap_ncoQ);
  free(ch->tap_decI);
  free(ch->tap_decQ);
  free(ch->tap_ifft);
  free(ch->tap_demod);
  free(ch->tap_volts);
  free(ch->agc_abs_ring);
  free(ch->agc_ring);
  t41o_nr_destroy(ch->nr);
  free(ch);
}

t41o_nr *t41o_channel_nr(t41o_channel *ch) { return ch ? ch->nr : NULL; }

void t41o_channel_reset(t41o_channel *ch) {
  const int L = ch->L, D = ch->D;
  ch->dc_state[0] = ch->dc_state[1] = 0.0f;
  ch->Osc_Vect_Q = 1.0;
  ch->Osc_Vect_I = 0.0;
  memset(ch->dec1_I_state, 0, sizeof(float) * (size_t)(T41O_N_DEC1_TAPS - 1 + L));
  memset(ch->dec1_Q_state, 0, sizeof(float) * (size_t)(T41O_N_DEC1_TAPS - 1 + L));
  memset(ch->dec2_I_state, 0, sizeof(float) * (size_t)(T41O_N_DEC2_TAPS - 1 + L / 4));
  memset(ch->dec2_Q_state, 0, sizeof(float) * (size_t)(T41O_N_DEC2_TAPS - 1 + L / 4));
  memset(ch->int1_state, 0, sizeof(float) * (size_t)(23 + D));
  memset(ch->int2_state, 0, sizeof(float) * (size_t)(7 + 2 * D));
  memset(ch->last_L, 0, sizeof(float) * (size_t)D);
  memset(ch->last_R, 0, sizeof(float) * (size_t)D);
  ch->first_block = 1;
  ch->wold = 0.0f;
  memset(ch->lp1_state, 0, sizeof(ch->lp1_state));
  ch->nfm_last_i = ch->nfm_last_q = 0.0f;
  ch->nfm_last_phase = 0.0f;
  ch->sam_phzerror = ch->sam_fil_out = ch->sam_omega2 = 0.0f;
  /* DSP_Fn.cpp:32-36, 481-492 */
  ch->agc_decay_type = 0;
  ch->agc_state = 0;
  memset(ch->agc_abs_ring, 0, sizeof(float) * AGC_RB_SIZE);
  memset(ch->agc_ring, 0, sizeof(float) * 2 * AGC_RB_SIZE);
  ch->agc_fast_backaverage = ch->agc_hang_backaverage = 0;
  ch->agc_ring_max = 0.0;
  ch->agc_save_volts = 0.0;
  ch->agc_volts = 0.0;
  ch->agc_hang_counter = 0;
  ch->agc_out_index = -1;
  ch->agc_in_index = 0;
  ch->agc_in_index_set = 0;
  memset(ch->agc_edges, 0, sizeof(ch->agc_edges));
  memset(ch->audioSpectBuffer, 0, sizeof(ch->audioSpectBuffer));
  ch->audioMaxSquared = ch->audioMaxSquaredAve = 0.0f;
  ch->AudioMaxIndex = 0;
  t41o_nr_reset(ch->nr);
}

int t41o_channel_tap(const t41o_channel *ch, int which, float *dst, int maxlen) {
  const float *src = NULL;
  int n = 0;
  switch (which) {
    case T41O_TAP_POST_NCO_I: src = ch->tap_ncoI; n = ch->L; break;
    case T41O_TAP_POST_NCO_Q: src = ch->tap_ncoQ; n = ch->L; break;
    case T41O_TAP_DEC_I: src = ch->tap_decI; n = ch->D; break;
    case T41O_TAP_DEC_Q: src = ch->tap_decQ; n = ch->D; break;
    case T41O_TAP_IFFT: src = ch->tap_ifft; n = 2 * ch->N; break;
    case T41O_TAP_DEMOD: src = ch->tap_demod; n = ch->D; break;
    case T41O_TAP_AGC_VOLTS: src = ch->tap_volts; n = ch->D; break;
    case T41O_TAP_AUDIO_SPECT: src = ch->audioSpectBuffer; n = 1024; break;
    case T41O_TAP_AGC_EDGES: src = ch->agc_edges; n = 25; break;
    case T41O_TAP_FFT_SPEC: src = ch->FFT_spec; n = 512; break;
    case T41O_TAP_FFT_SPEC_OLD: src = ch->FFT_spec_old; n = 512; break;
    case T41O_TAP_AUDIO_MAX: {
      float t[3] = {ch->audioMaxSquared, (float)ch->AudioMaxIndex, ch->audioMaxSquaredAve};
      int m = maxlen < 3 ? maxlen : 3;
      memcpy(dst, t, sizeof(float) * (size_t)m);
      return m;
    }
    default: return -1;
  }
  if (n > maxlen) n = maxlen;
  memcpy(dst, src, sizeof(float) * (size_t)n);
  return n;
}

/* ------------------------------------------------------------------------------------------
 * The block function
 * ---------------------------------------------------------------------------------------- */

/* ---- display FFT (FFT.cpp:28-251), up to FFT_spec / FFT_spec_old; the pixel mapping that follows
 * (log10f_fast, display scale, pixel offsets) is display code and is not restated ---- */
#define SPECTRUM_RES 512 /* Display.h:12 */
/* mag_coeffs[1..4] (FIR.cpp:582-680): 4-stage elliptic low-passes for 2x .. 16x (MAX_ZOOM_ENTRIES = 5, ButtonProc.h:6) */
static const float mag_coeffs_1_4[4][20] = {
    {0.228454526413293696f, 0.077639329099949764f, 0.228454526413293696f, 0.635534925142242080f, -0.170083307068779194f, 0.436788292542003964f, 0.232307972937606161f, 0.436788292542003964f, 0.365885230717786780f, -0.471769788739400842f, 0.535974654742658707f, 0.557035600464780845f, 0.535974654742658707f, 0.125740787233286133f, -0.754725697183384336f, 0.501116342273565607f, 0.914877831284765408f, 0.501116342273565607f, 0.013862536615004284f, -0.930973052446900984f},
    {0.182208761527446556f, -0.222492493114674145f, 0.182208761527446556f, 1.326111070880959810f, -0.468036100821178802f, 0.337123762652097259f, -0.366352718812586853f, 0.337123762652097259f, 1.337053579516321200f, -0.644948386007929031f, 0.336163175380826074f, -0.199246162162897811f, 0.336163175380826074f, 1.354952684569386670f, -0.828032873168141115f, 0.178588201750411041f, 0.207271695028067304f, 0.178588201750411041f, 1.386486967455699220f, -0.950935065984588657f},
    {0.185643392652478922f, -0.332064345389014803f, 0.185643392652478922f, 1.654637402827731090f, -0.693859842743674182f, 0.327519300813245984f, -0.571358085216950418f, 0.327519300813245984f, 1.715375037176782860f, -0.799055553586324407f, 0.283656142708241688f, -0.441088976843048652f, 0.283656142708241688f, 1.778230635987093860f, -0.904453944560528522f, 0.079685368654848945f, -0.011231810140649204f, 0.079685368654848945f, 1.825046003243238070f, -0.973184930412286708f},
    {0.194769868656866380f, -0.379098413160710079f, 0.194769868656866380f, 1.824436402073870810f, -0.834877726226893380f, 0.333973874901496770f, -0.646106479315673776f, 0.333973874901496770f, 1.871892825636887640f, -0.893734096124207178f, 0.272903880596429671f, -0.513507745397738469f, 0.272903880596429671f, 1.918161772571113750f, -0.950461788366234739f, 0.053535383722369843f, -0.069683422367188122f, 0.053535383722369843f, 1.948900719896301760f, -0.986288064973853129f},
};

/* ZoomFFTPrep(), FFT.cpp:35-56 */
int t41o_channel_set_display(t41o_channel *ch, int spectrumZoom) {
  if (spectrumZoom < -1 || spectrumZoom > 4) return -1;
  if (ch->N != 512) return -2; /* the display code is written for the 2048-sample frame */
  ch->display_zoom = spectrumZoom;
  memset(ch->zoom_iir_I_state, 0, sizeof(ch->zoom_iir_I_state));
  memset(ch->zoom_iir_Q_state, 0, sizeof(ch->zoom_iir_Q_state));
  memset(ch->zoom_fir_I_state, 0, sizeof(float) * (size_t)(3 + ch->L));
  memset(ch->zoom_fir_Q_state, 0, sizeof(float) * (size_t)(3 + ch->L));
  memset(ch->FFT_ring_buffer_x, 0, sizeof(ch->FFT_ring_buffer_x));
  memset(ch->FFT_ring_buffer_y, 0, sizeof(ch->FFT_ring_buffer_y));
  memset(ch->FFT_spec, 0, sizeof(ch->FFT_spec));
  memset(ch->FFT_spec_old, 0, sizeof(ch->FFT_spec_old));
  ch->zoom_sample_ptr = 0;
  if (spectrumZoom > 0) {
    float Fstop_Zoom = 0.5 * (float)192000 / (1 << spectrumZoom);
    t41o_CalcFIRCoeffs(ch->zoom_fir_coeffs, 4, Fstop_Zoom, 60, 0, 0.0, (float)192000);
  }
  return 0;
}

/* CalcZoom1Magn(), FFT.cpp:208-251 (spectrumZoom == 0), on float_buffer_L/R before FreqShift1 */
static void CalcZoom1Magn(t41o_channel *ch, const float *float_buffer_L, const float *float_buffer_R) {
  float buffer_spec_FFT[2 * SPECTRUM_RES];
  float spec_help = 0.0;
  float LPFcoeff = 0.7;
  for (int i = 0; i < SPECTRUM_RES; i++) {
    buffer_spec_FFT[i * 2] = float_buffer_L[i] * (0.5 - 0.5 * cos(6.28 * i / SPECTRUM_RES));
    buffer_spec_FFT[i * 2 + 1] = float_buffer_R[i] * (0.5 - 0.5 * cos(6.28 * i / SPECTRUM_RES));
  }
  t41o_cfft_f32(buffer_spec_FFT, SPECTRUM_RES, 0);
  for (int i = 0; i < SPECTRUM_RES / 2; i++) {
    ch->FFT_spec[i + SPECTRUM_RES / 2] = (buffer_spec_FFT[i * 2] * buffer_spec_FFT[i * 2] + buffer_spec_FFT[i * 2 + 1] * buffer_spec_FFT[i * 2 + 1]);
    ch->FFT_spec[i] = (buffer_spec_FFT[(i + SPECTRUM_RES / 2) * 2] * buffer_spec_FFT[(i + SPECTRUM_RES / 2) * 2] +
                       buffer_spec_FFT[(i + SPECTRUM_RES / 2) * 2 + 1] * buffer_spec_FFT[(i + SPECTRUM_RES / 2) * 2 + 1]);
  }
  for (int x = 0; x < SPECTRUM_RES; x++) {
    spec_help = LPFcoeff * ch->FFT_spec[x] + (1.0 - LPFcoeff) * ch->FFT_spec_old[x];
    ch->FFT_spec_old[x] = spec_help;
  }
}

/* ZoomFFTExe(BUFFER_SIZE * N_BLOCKS), FFT.cpp:67-152 (spectrumZoom != 0), on float_buffer_L/R after FreqShift1 */
static void ZoomFFTExe(t41o_channel *ch, const float *float_buffer_L, const float *float_buffer_R, int blockSize) {
  const int spectrumZoom = ch->display_zoom;
  float LPFcoeff;
  float onem_LPFcoeff;
  float *x_buffer = ch->float_buffer_L_EX, *y_buffer = ch->float_buffer_R_EX; /* free at this point (filled later) */
  float buffer_spec_FFT[2 * SPECTRUM_RES];
  int sample_no = SPECTRUM_RES;
  float multiplier;
  const int M = 1 << spectrumZoom; /* spectrumZoom < 7 */
  const float *coeffs = mag_coeffs_1_4[spectrumZoom - 1];

  sample_no = blockSize / (1 << spectrumZoom);
  if (sample_no > SPECTRUM_RES) sample_no = SPECTRUM_RES;

  /* arm_biquad_cascade_df1_f32 with 4 stages */
  {
    const float *src = float_buffer_L;
    for (int st = 0; st < 4; st++) {
      t41o_biquad_df1_f32(coeffs + 5 * st, ch->zoom_iir_I_state + 4 * st, src, x_buffer, blockSize);
      src = x_buffer;
    }
    src = float_buffer_R;
    for (int st = 0; st < 4; st++) {
      t41o_biquad_df1_f32(coeffs + 5 * st, ch->zoom_iir_Q_state + 4 * st, src, y_buffer, blockSize);
      src = y_buffer;
    }
  }
  t41o_fir_decimate_f32(ch->zoom_fir_coeffs, 4, M, ch->zoom_fir_I_state, x_buffer, x_buffer, blockSize);
  t41o_fir_decimate_f32(ch->zoom_fir_coeffs, 4, M, ch->zoom_fir_Q_state, y_buffer, y_buffer, blockSize);

  for (int i = 0; i < sample_no; i++) {
    ch->FFT_ring_buffer_x[ch->zoom_sample_ptr] = x_buffer[i];
    ch->FFT_ring_buffer_y[ch->zoom_sample_ptr] = y_buffer[i];
    ch->zoom_sample_ptr++;
    if (ch->zoom_sample_ptr >= SPECTRUM_RES) ch->zoom_sample_ptr = 0;
  }
  multiplier = (float)spectrumZoom;
  if (spectrumZoom > 3) multiplier = (float)(1 << spectrumZoom);
  for (int idx = 0; idx < SPECTRUM_RES; idx++) {
    buffer_spec_FFT[idx * 2 + 0] = multiplier * ch->FFT_ring_buffer_x[ch->zoom_sample_ptr] * (0.5 - 0.5 * cos(6.28 * idx / SPECTRUM_RES));
    buffer_spec_FFT[idx * 2 + 1] = multiplier * ch->FFT_ring_buffer_y[ch->zoom_sample_ptr] * (0.5 - 0.5 * cos(6.28 * idx / SPECTRUM_RES));
    ch->zoom_sample_ptr++;
    if (ch->zoom_sample_ptr >= SPECTRUM_RES) ch->zoom_sample_ptr = 0;
  }
  LPFcoeff = 0.7;
  onem_LPFcoeff = 1.0 - LPFcoeff;
  t41o_cfft_f32(buffer_spec_FFT, SPECTRUM_RES, 0);
  for (int i = 0; i < SPECTRUM_RES / 2; i++) {
    ch->FFT_spec[i + SPECTRUM_RES / 2] = (buffer_spec_FFT[i * 2] * buffer_spec_FFT[i * 2] + buffer_spec_FFT[i * 2 + 1] * buffer_spec_FFT[i * 2 + 1]);
    ch->FFT_spec[i] = (buffer_spec_FFT[(i + SPECTRUM_RES / 2) * 2] * buffer_spec_FFT[(i + SPECTRUM_RES / 2) * 2] +
                       buffer_spec_FFT[(i + SPECTRUM_RES / 2) * 2 + 1] * buffer_spec_FFT[(i + SPECTRUM_RES / 2) * 2 + 1]);
  }
  for (int i = 0; i < SPECTRUM_RES; i++) {
    ch->FFT_spec[i] = LPFcoeff * ch->FFT_spec[i] + onem_LPFcoeff * ch->FFT_spec_old[i];
    ch->FFT_spec_old[i] = ch->FFT_spec[i];
  }
}

/* HP_DC_Filter_Coeffs2, FIR.cpp:87-89 */
static const float HP_DC_Filter_Coeffs2[5] = {
    0.927176191943378969, -0.927176191943378969, 0.000000000000000000, 0.854352383886757938,
    0.000000000000000000};

/* Utility.cpp:269-285 */
static float AlphaBetaMag(float inphase, float quadrature) {
  const float alpha = 0.960433870103;
  const float beta = 0.397824734759;
  float abs_inphase = fabs(inphase);
  float abs_quadrature = fabs(quadrature);
  if (abs_inphase > abs_quadrature) return alpha * abs_inphase + beta * abs_quadrature;
  return alpha * abs_quadrature + beta * abs_inphase;
}

/* Process.cpp:955-967 */
static float VolumeToAmplification(int volume) {
  float x = volume / 100.0f;
  float ampl = 5 * x * x * x * x * x;
  return ampl;
}

#define fmdemod_quadri_K 0.340447550238101026565118445432744920253753662109375 /* Demod.h:7 */

/* Demod.cpp:220-235.  Note the reference saves input[input_size-2], input[input_size-1] as the
 * "last sample": that is complex sample input_size/2-1, not the last one (faithfully kept). */
static void nfmdemod(t41o_channel *ch, const float *input, float *output, int input_size) {
  output[0] = fmdemod_quadri_K *
              (input[0] * (input[1] - ch->nfm_last_q) - input[1] * (input[0] - ch->nfm_last_i)) /
              (input[0] * input[0] + input[1] * input[1]);
  for (int i = 1; i < input_size; i++) {
    float qnow = input[i * 2 + 1];
    float qlast = input[(i - 1) * 2 + 1];
    float inow = input[i * 2];
    float ilast = input[(i - 1) * 2];
    output[i] = fmdemod_quadri_K * (qnow * ilast - inow * qlast) / (inow * inow + qnow * qnow);
  }
  ch->nfm_last_i = input[input_size - 2];
  ch->nfm_last_q = input[input_size - 1];
}

/* ---- the NFM variant the reference keeps commented out (nfm_demod = 1): parity for it is pinned by
 * this restatement alone -- the firmware never ran it as written here ---- */
#define T41O_PI 3.1415926535897932384626433832795f  /* FIR.h:10 */
#define T41O_TPI 6.283185307179586476925286766559f  /* FIR.h:12-13: TPI = TWO_PI */

/* Utility.cpp:298-302 */
static float ApproxAtan(float z) {
  const float n1 = 0.97239411f;
  const float n2 = -0.19194795f;
  return (n1 + n2 * z * z) * z;
}

/* Demod.cpp:148-197.  Where pi/2 is meant (|y| >= |x|, and x == 0) the source adds TPI = 2 pi
 * (SURVEY App. C #8): kept as written. */
static float ApproxAtan2(float y, float x) {
  if (x != 0.0f) {
    if (fabsf(x) > fabsf(y)) {
      const float z = y / x;
      if (x > 0.0f) return ApproxAtan(z);
      else if (y >= 0.0f) return ApproxAtan(z) + T41O_PI;
      else return ApproxAtan(z) - T41O_PI;
    } else {
      const float z = x / y;
      if (y > 0.0f) return -ApproxAtan(z) + T41O_TPI;
      else return -ApproxAtan(z) - T41O_TPI;
    }
  } else {
    if (y > 0.0f) return T41O_TPI;
    else if (y < 0.0f) return -T41O_TPI;
  }
  return 0.0f;
}

/* ---- arm_sin_f32 / arm_cos_f32 (CMSIS-DSP FastMathFunctions, the 512-entry table with linear
 * interpolation that CMSIS-DSP has shipped since 1.4.5): in = x / (2 pi) [+ 0.25 for the cosine],
 * reduced to [0, 1) by truncation, index = (uint16_t)(512 in), result = (1 - fract) T[index] +
 * fract T[index + 1].  The library's table is a list of 8-decimal literals that is not in this
 * container: T[k] here = sin(2 pi k / 512) evaluated in double and rounded to f32 (a CHOICE, like
 * the other CMSIS conventions listed in DESIGN.md section 2: parity unpinned). */
#define FAST_MATH_TABLE_SIZE 512
static float sinTable_f32[FAST_MATH_TABLE_SIZE + 1];
static int sinTable_ready = 0;
static void sinTable_init(void) {
  if (sinTable_ready) return;
  for (int k = 0; k <= FAST_MATH_TABLE_SIZE; k++)
    sinTable_f32[k] = (float)sin(6.283185307179586476925286766559 * (double)k / (double)FAST_MATH_TABLE_SIZE);
  sinTable_f32[0] = 0.0f;
  sinTable_f32[FAST_MATH_TABLE_SIZE / 2] = 0.0f; /* the library's table has exact zeros there (and -0.0f at the end) */
  sinTable_f32[FAST_MATH_TABLE_SIZE] = -0.0f;
  sinTable_ready = 1;
}
const float *t41o_sin_table(void) { sinTable_init(); return sinTable_f32; }
static float fast_sincos_f32(float in) { /* the part arm_sin_f32 and arm_cos_f32 share */
  int32_t n = (int32_t)in;
  if (in < 0.0f) n--;
  in = in - (float)n;
  float findex = (float)FAST_MATH_TABLE_SIZE * in;
  uint16_t index = (uint16_t)findex;
  if (index >= FAST_MATH_TABLE_SIZE) {
    index = 0;
    findex -= (float)FAST_MATH_TABLE_SIZE;
  }
  float fract = findex - (float)index;
  float a = sinTable_f32[index], b = sinTable_f32[index + 1];
  return (1.0f - fract) * a + fract * b;
}
float t41o_arm_sin_f32(float x) { sinTable_init(); return fast_sincos_f32(x * 0.159154943092f); }
float t41o_arm_cos_f32(float x) { sinTable_init(); return fast_sincos_f32(x * 0.159154943092f + 0.25f); }

/* AMDecodeSAM(), Demod.cpp:40-139, with the file-scope PLL constants of Demod.cpp:13-23 evaluated as
 * the C++ there evaluates them (gwv.cpp:64-65: omegaN = 200, pll_fmax = 4000, constant-initialised,
 * so they hold those values when Demod.cpp's dynamic initialisers run).  As written:
 *  - `exp(-1 / 24000 * tauR)`: -1 / 24000 is an integer division = 0, so mtauR = mtauI = 1 and the
 *    "fade leveler" adds dc_insert - dc = 0 - 0 to every sample;
 *  - ApproxAtan2 returns +-2 pi where +-pi/2 is meant (App. C #8): the phase detector jumps whenever
 *    |corr[1]| >= |corr[0]|;
 *  - float_buffer_R receives audiou = 0 (it is not played); the carrier display arithmetic behind the
 *    loop is display code. */
static void AMDecodeSAM(t41o_channel *ch, const float *iFFT_buffer, float *float_buffer_L, float *float_buffer_R, int FFT_length) {
  const float omegaN = 200.0f, pll_fmax = +4000.0f; /* gwv.cpp:64-65 */
  int zeta_help = 65;
  float zeta = (float)zeta_help / 100.0;
  float omega_min = T41O_TPI * -pll_fmax * 1 / 24000;
  float omega_max = T41O_TPI * pll_fmax * 1 / 24000;
  float g1 = 1.0 - exp(-2.0 * omegaN * zeta * 1 / 24000);
  float g2 = -g1 + 2.0 * (1 - exp(-omegaN * zeta * 1 / 24000) * cosf(omegaN * 1 / 24000 * sqrtf(1.0 - zeta * zeta)));
  float tauR = 0.02;
  float tauI = 1.4;
  float dc = 0.0, dc_insert = 0.0, dcu = 0.0, dc_insertu = 0.0;
  float mtauR = exp(-1 / 24000 * tauR);
  float onem_mtauR = 1.0 - mtauR;
  float mtauI = exp(-1 / 24000 * tauI);
  float onem_mtauI = 1.0 - mtauI;
  float phzerror = ch->sam_phzerror, fil_out = ch->sam_fil_out, omega2 = ch->sam_omega2;
  float det, del_out;
  sinTable_init();
  for (int i = 0; i < FFT_length / 2; i++) {
    float Sin, Cos, ai, bi, aq, bq, audio, audiou = 0, corr[2];
    Sin = t41o_arm_sin_f32(phzerror);
    Cos = t41o_arm_cos_f32(phzerror);
    ai = Cos * iFFT_buffer[FFT_length + i * 2];
    bi = Sin * iFFT_buffer[FFT_length + i * 2];
    aq = Cos * iFFT_buffer[FFT_length + i * 2 + 1];
    bq = Sin * iFFT_buffer[FFT_length + i * 2 + 1];
    corr[0] = +ai + bq;
    corr[1] = -bi + aq;
    audio = (ai - bi) + (aq + bq);
    /* fade_leveler = 1 */
    dc = mtauR * dc + onem_mtauR * audio;
    dc_insert = mtauI * dc_insert + onem_mtauI * corr[0];
    audio = audio + dc_insert - dc;
    float_buffer_L[i] = audio;
    dcu = mtauR * dcu + onem_mtauR * audiou;
    dc_insertu = mtauI * dc_insertu + onem_mtauI * corr[0];
    audiou = audiou + dc_insertu - dcu;
    float_buffer_R[i] = audiou;
    det = ApproxAtan2(corr[1], corr[0]);
    del_out = fil_out;
    omega2 = omega2 + g2 * det;
    if (omega2 < omega_min) omega2 = omega_min;
    else if (omega2 > omega_max) omega2 = omega_max;
    fil_out = g1 * det + omega2;
    phzerror = phzerror + del_out;
    while (phzerror >= T41O_TPI) phzerror -= T41O_TPI;
    while (phzerror < 0.0) phzerror += T41O_TPI;
  }
  ch->sam_phzerror = phzerror;
  ch->sam_fil_out = fil_out;
  ch->sam_omega2 = omega2;
}
/* the PLL constants as the product's designer must reproduce them: {omega_min, omega_max, g1, g2} */
void t41o_sam_constants(float out[4]) {
  const float omegaN = 200.0f, pll_fmax = +4000.0f;
  int zeta_help = 65;
  float zeta = (float)zeta_help / 100.0;
  out[0] = T41O_TPI * -pll_fmax * 1 / 24000;
  out[1] = T41O_TPI * pll_fmax * 1 / 24000;
  float g1 = 1.0 - exp(-2.0 * omegaN * zeta * 1 / 24000);
  out[2] = g1;
  out[3] = -g1 + 2.0 * (1 - exp(-omegaN * zeta * 1 / 24000) * cosf(omegaN * 1 / 24000 * sqrtf(1.0 - zeta * zeta)));
}

/* Demod.cpp:368-392 */
static void fmdemod_atan_cf(t41o_channel *ch, const float *input, float *output, int input_size) {
  float phase, dphase;
  for (int i = 0; i < input_size; i++) {
    phase = ApproxAtan2(input[i * 2 + 1], input[i * 2]);
    dphase = phase - ch->nfm_last_phase;
    if (dphase < -T41O_PI) dphase += 2 * T41O_PI;
    if (dphase > T41O_PI) dphase -= 2 * T41O_PI;
    output[i] = dphase / T41O_PI;
    ch->nfm_last_phase = phase;
  }
}

/* Demod.cpp:324-344: deemphasis_nfm_predefined_fir_24000 and the block-wise FIR as written: only
 * input_size - taps_length outputs are produced, the rest of `output` is left as it was */
static const float deemphasis_nfm_fir_24000[81] = {
  0.000481913, -0.000816211, -0.00205384, -0.00264474, -0.00258229, -0.00247939, -0.00305299, -0.00448116, -0.00620366, -0.00737591, -0.00761292, -0.00737176, -0.0075984, -0.00890065, -0.0109592, -0.0127338, -0.0133493, -0.0129165, -0.0125289, -0.013351, -0.0155348, -0.0179452, -0.0190498, -0.0183068, -0.016827, -0.0165808, -0.0186455, -0.0219659, -0.0238965, -0.0223995, -0.0182146, -0.0149414, -0.0163342, -0.0223751, -0.0271497, -0.020849, 0.00446391, 0.0485999, 0.100768, 0.143223, 0.159583, 0.143223, 0.100768, 0.0485999, 0.00446391, -0.020849, -0.0271497, -0.0223751, -0.0163342, -0.0149414, -0.0182146, -0.0223995, -0.0238965, -0.0219659, -0.0186455, -0.0165808, -0.016827, -0.0183068, -0.0190498, -0.0179452, -0.0155348, -0.013351, -0.0125289, -0.0129165, -0.0133493, -0.0127338, -0.0109592, -0.00890065, -0.0075984, -0.00737176, -0.00761292, -0.00737591, -0.00620366, -0.00448116, -0.00305299, -0.00247939, -0.00258229, -0.00264474, -0.00205384, -0.000816211, 0.000481913
};
static void deemphasis_nfm_ff(const float *input, float *output, int input_size) {
  const int taps_length = 81;
  for (int i = 0; i < input_size - taps_length; i++) {
    float acc = 0;
    for (int ti = 0; ti < taps_length; ti++) acc += deemphasis_nfm_fir_24000[ti] * input[i + ti];
    output[i] = acc;
  }
}

/* DSP_Fn.cpp:494-502 (AGCMode == 0): fixed_gain = 20 (DSP_Fn.cpp:453) on the upper half */
static void AGC_off(float *iFFT_buffer, int N) {
  const float fixed_gain = 20.0;
  for (int i = 0; i < N / 2; i++) {
    iFFT_buffer[N + 2 * i + 0] = fixed_gain * iFFT_buffer[N + 2 * i + 0];
    iFFT_buffer[N + 2 * i + 1] = fixed_gain * iFFT_buffer[N + 2 * i + 1];
  }
}

/* Utility.cpp:245-258 */
static float log10f_fast(float X) {
  float Y, F;
  int E;
  F = frexpf(fabsf(X), &E);
  Y = 1.23149591368684f;
  Y *= F;
  Y += -4.11852516267426f;
  Y *= F;
  Y += 6.02197014179219f;
  Y *= F;
  Y += -3.13396450166353f;
  Y += E;
  return (Y * 0.3010299956639812f);
}

/* wiring.h min(): the conditional's type is the common type of both operands */
#define MIN_D(a, b) ((a) < (b) ? (a) : (b))

/* DSP_Fn.cpp:479-632, AGCMode != 0.  Variable names and statement order follow the reference;
 * its file-scope/static variables live in the channel, its AGCLoadValues() outputs in g[].
 * pmode = 1 (DSP_Fn.cpp:35): magnitude by sqrtf.  hang_enable = 1 (DSP_Fn.cpp:458). */
static void AGC_on(t41o_channel *ch, const float *g, float *iFFT_buffer, int FFT_length) {
  int k;
  float mult;
  const unsigned ring_buffsize = AGC_RB_SIZE;
  float *abs_ring = ch->agc_abs_ring, *ring = ch->agc_ring;
  float abs_out_sample;
  float out_sample[2];
  const int attack_buffsize = (int)g[T41O_AGC_ATTACK_BUFFSIZE];
  const float attack_mult = g[T41O_AGC_ATTACK_MULT], decay_mult = g[T41O_AGC_DECAY_MULT];
  const float fast_decay_mult = g[T41O_AGC_FAST_DECAY_MULT];
  const float fast_backmult = g[T41O_AGC_FAST_BACKMULT];
  const float onemfast_backmult = g[T41O_AGC_ONEMFAST_BACKMULT];
  const float hang_backmult = g[T41O_AGC_HANG_BACKMULT];
  const float onemhang_backmult = g[T41O_AGC_ONEMHANG_BACKMULT];
  const float hang_decay_mult = g[T41O_AGC_HANG_DECAY_MULT];
  const float out_target = g[T41O_AGC_OUT_TARGET], min_volts = g[T41O_AGC_MIN_VOLTS];
  const float slope_constant = g[T41O_AGC_SLOPE_CONSTANT];
  const float inv_max_input = g[T41O_AGC_INV_MAX_INPUT];
  const float hang_level = g[T41O_AGC_HANG_LEVEL], pop_ratio = g[T41O_AGC_POP_RATIO];
  const int hang_enable = 1;
  uint8_t decay_type = ch->agc_decay_type, state = ch->agc_state;
  float fast_backaverage = ch->agc_fast_backaverage, hang_backaverage = ch->agc_hang_backaverage;
  float ring_max = ch->agc_ring_max, save_volts = ch->agc_save_volts, volts = ch->agc_volts;
  int hang_counter = ch->agc_hang_counter, out_index = ch->agc_out_index;
  uint32_t in_index = ch->agc_in_index;
  if (!ch->agc_in_index_set) { /* AGCLoadValues(), DSP_Fn.cpp:410 */
    in_index = attack_buffsize + out_index;
    ch->agc_in_index_set = 1;
  }

  for (unsigned i = 0; i < (unsigned)FFT_length / 2; i++) {
    if (++out_index >= (int)ring_buffsize) out_index -= ring_buffsize;
    if (++in_index >= ring_buffsize) in_index -= ring_buffsize;

    out_sample[0] = ring[2 * out_index + 0];
    out_sample[1] = ring[2 * out_index + 1];
    abs_out_sample = abs_ring[out_index];
    ring[2 * in_index + 0] = iFFT_buffer[FFT_length + 2 * i + 0];
    ring[2 * in_index + 1] = iFFT_buffer[FFT_length + 2 * i + 1];
    abs_ring[in_index] = sqrtf(ring[2 * in_index + 0] * ring[2 * in_index + 0] +
                               ring[2 * in_index + 1] * ring[2 * in_index + 1]);

    fast_backaverage = fast_backmult * abs_out_sample + onemfast_backmult * fast_backaverage;
    hang_backaverage = hang_backmult * abs_out_sample + onemhang_backmult * hang_backaverage;

    if ((abs_out_sample >= ring_max) && (abs_out_sample > 0.0)) {
      ring_max = 0.0;
      k = out_index;
      for (int j = 0; j < attack_buffsize; j++) {
        if (++k == (int)ring_buffsize) k = 0;
        if (abs_ring[k] > ring_max) ring_max = abs_ring[k];
      }
    }
    if (abs_ring[in_index] > ring_max) ring_max = abs_ring[in_index];

    if (hang_counter > 0) --hang_counter;

    const uint8_t state_before = state;
    switch (state) {
      case 0:
        if (ring_max >= volts) {
          volts += (ring_max - volts) * attack_mult;
        } else {
          if (volts > pop_ratio * fast_backaverage) {
            state = 1;
            volts += (ring_max - volts) * fast_decay_mult;
          } else {
            if (hang_enable && (hang_backaverage > hang_level)) {
              state = 2;
              hang_counter = (int)g[T41O_AGC_HANG_COUNT];
              decay_type = 1;
            } else {
              state = 3;
              volts += (ring_max - volts) * decay_mult;
              decay_type = 0;
            }
          }
        }
        break;
      case 1:
        if (ring_max >= volts) {
          state = 0;
          volts += (ring_max - volts) * attack_mult;
        } else {
          if (volts > save_volts) {
            volts += (ring_max - volts) * fast_decay_mult;
          } else {
            if (hang_counter > 0) {
              state = 2;
            } else {
              if (decay_type == 0) {
                state = 3;
                volts += (ring_max - volts) * decay_mult;
              } else {
                state = 4;
                volts += (ring_max - volts) * hang_decay_mult;
              }
            }
          }
        }
        break;
      case 2:
        if (ring_max >= volts) {
          state = 0;
          save_volts = volts;
          volts += (ring_max - volts) * attack_mult;
        } else {
          if (hang_counter == 0) {
            state = 4;
            volts += (ring_max - volts) * hang_decay_mult;
          }
        }
        break;
      case 3:
        if (ring_max >= volts) {
          state = 0;
          save_volts = volts;
          volts += (ring_max - volts) * attack_mult;
        } else {
          volts += (ring_max - volts) * decay_mult * .05;
        }
        break;
      case 4:
        if (ring_max >= volts) {
          state = 0;
          save_volts = volts;
          volts += (ring_max - volts) * attack_mult;
        } else {
          volts += (ring_max - volts) * hang_decay_mult;
        }
        break;
    }
    if (volts < min_volts) volts = min_volts; /* no AGC action is taking place */
    ch->tap_volts[i] = volts;
    ch->agc_edges[5 * state_before + state] += 1.0f;

    mult = (out_target - slope_constant * MIN_D(0.0, log10f_fast(inv_max_input * volts))) / volts;
    iFFT_buffer[FFT_length + 2 * i + 0] = out_sample[0] * mult;
    iFFT_buffer[FFT_length + 2 * i + 1] = out_sample[1] * mult;
  }
  ch->agc_decay_type = decay_type;
  ch->agc_state = state;
  ch->agc_fast_backaverage = fast_backaverage;
  ch->agc_hang_backaverage = hang_backaverage;
  ch->agc_ring_max = ring_max;
  ch->agc_save_volts = save_volts;
  ch->agc_volts = volts;
  ch->agc_hang_counter = hang_counter;
  ch->agc_out_index = out_index;
  ch->agc_in_index = in_index;
}

static void AGC(t41o_channel *ch, const t41o_params *p, const t41o_coeffs *c, float *iFFT_buffer,
                int N) {
  if (p->AGCMode == 0)
    AGC_off(iFFT_buffer, N);
  else
    AGC_on(ch, c->agc, iFFT_buffer, N);
}

/* arm_max_f32 (CMSIS-DSP, scalar): maximum and the index of its FIRST occurrence */
static void arm_max(const float *src, uint32_t n, float *result, uint32_t *index) {
  float out = src[0];
  uint32_t outIndex = 0;
  for (uint32_t i = 1; i < n; i++) {
    if (out < src[i]) {
      out = src[i];
      outIndex = i;
    }
  }
  *result = out;
  *index = outIndex;
}

/* Process.cpp:550-570 (and :790-805 in the NFM pass) with updateDisplayFlag == 1: the audio
 * spectrum the display and the S-meter (Display.cpp:980-985) are fed from.  The pixel mapping
 * (audioYPixel, map(), log10f) is display code and not restated.  The loop bound 1024 is the
 * firmware's literal (2 * FFT_LENGTH for FFT_LENGTH 512): the squares are of the individual
 * re / im floats, in reversed order. */
static void audio_spectrum(t41o_channel *ch, const float *iFFT_buffer) {
  for (int k = 0; k < 1024; k++) ch->audioSpectBuffer[1023 - k] = (iFFT_buffer[k] * iFFT_buffer[k]);
  arm_max(ch->audioSpectBuffer, 1024, &ch->audioMaxSquared, &ch->AudioMaxIndex);
  ch->audioMaxSquaredAve = .5 * ch->audioMaxSquared + .5 * ch->audioMaxSquaredAve;
}

static void cmplx_mult_cmplx(const float *a, const float *b, float *d, int n) {
  for (int i = 0; i < n; i++) { /* arm_cmplx_mult_cmplx_f32 */
    float ar = a[2 * i], ai = a[2 * i + 1], br = b[2 * i], bi = b[2 * i + 1];
    d[2 * i] = ar * br - ai * bi;
    d[2 * i + 1] = ar * bi + ai * br;
  }
}

int t41o_process_frame(t41o_channel *ch, const t41o_params *p, const t41o_coeffs *c, long NCOFreq,
                       const float *I, const float *Q, float *audio) {
  const int N = ch->N, D = ch->D, L = ch->L;
  if (p->fft_length != N) return -1;
  if (p->AGCMode < 0 || p->AGCMode > 4) return -2;
  const int mode = p->mode;
  if ((mode < T41O_DEMOD_USB || mode > T41O_DEMOD_NFM) && mode != T41O_DEMOD_SAM) return -3;
  float *fL = ch->float_buffer_L, *fR = ch->float_buffer_R;
  float *exL = ch->float_buffer_L_EX, *exR = ch->float_buffer_R_EX;
  float *FFT_buffer = ch->FFT_buffer, *iFFT_buffer = ch->iFFT_buffer;

  /* the f32 I/Q boundary: float_buffer_L = I, float_buffer_R = Q (after Process.cpp:107-108) */
  memcpy(fL, I, sizeof(float) * (size_t)L);
  memcpy(fR, Q, sizeof(float) * (size_t)L);

  /* Process.cpp:117-119 */
  float rfGainValue = pow(10, (float)p->rfGainAllBands / 20);
  for (int i = 0; i < L; i++) fL[i] = fL[i] * rfGainValue;
  for (int i = 0; i < L; i++) fR[i] = fR[i] * rfGainValue;

  /* Process.cpp:127-128: one shared instance, L then R (reference hard-codes 2048 = frame_len) */
  t41o_biquad_df2T_f32(HP_DC_Filter_Coeffs2, ch->dc_state, fL, fL, L);
  t41o_biquad_df2T_f32(HP_DC_Filter_Coeffs2, ch->dc_state, fR, fR, L);

  /* Process.cpp:133-134 */
  {
    float g = (float)p->RFgain;
    for (int i = 0; i < L; i++) fL[i] = fL[i] * g;
    for (int i = 0; i < L; i++) fR[i] = fR[i] * g;
  }

  /* Process.cpp:165-173: USB, LSB, AM (and SAM) only */
  if (mode == T41O_DEMOD_LSB || mode == T41O_DEMOD_AM || mode == T41O_DEMOD_USB || mode == T41O_DEMOD_SAM) {
    float s = -p->IQAmpCorrectionFactor;
    for (int i = 0; i < L; i++) fL[i] = fL[i] * s;
    float factor = p->IQPhaseCorrectionFactor; /* Utility.cpp:178-187 */
    if (factor < 0.0) {
      for (int i = 0; i < L; i++) fR[i] = fR[i] + fL[i] * factor;
    } else {
      for (int i = 0; i < L; i++) fL[i] = fL[i] + fR[i] * factor;
    }
  }

  if (ch->display_zoom == 0) CalcZoom1Magn(ch, fL, fR); /* Process.cpp:185-187 */

  /* FreqShift1, Freq_Shift.cpp:42-65 */
  for (int i = 0; i < L; i += 4) {
    float hh1, hh2;
    hh1 = -fR[i + 1];
    hh2 = fL[i + 1];
    fL[i + 1] = hh1;
    fR[i + 1] = hh2;
    hh1 = -fL[i + 2];
    hh2 = -fR[i + 2];
    fL[i + 2] = hh1;
    fR[i + 2] = hh2;
    hh1 = fR[i + 3];
    hh2 = -fL[i + 3];
    fL[i + 3] = hh1;
    fR[i + 3] = hh2;
  }
  if (ch->display_zoom > 0) ZoomFFTExe(ch, fL, fR, L); /* Process.cpp:211-215 (updateDisplayFlag == 1) */
  for (int i = 0; i < L; i++) {
    exL[i] = fL[i];
    exR[i] = fR[i];
  }

  /* FreqShift2, Freq_Shift.cpp:94-141 */
  {
    int sideToneShift = 0;
    if (p->xmtMode == T41O_CW_MODE) {
      if (mode == 1) sideToneShift = p->CWFreqShift;
      else if (mode == 0) sideToneShift = -p->CWFreqShift;
    }
    float NCO_INC = 2.0 * PI_F * (NCOFreq + sideToneShift) / 192000.0;
    double OSC_COS = cos(NCO_INC);
    double OSC_SIN = sin(NCO_INC);
    double Osc_Vect_Q = ch->Osc_Vect_Q, Osc_Vect_I = ch->Osc_Vect_I;
    for (int i = 0; i < L; i++) {
      double Osc_Q = (Osc_Vect_Q * OSC_COS) - (Osc_Vect_I * OSC_SIN);
      double Osc_I = (Osc_Vect_I * OSC_COS) + (Osc_Vect_Q * OSC_SIN);
      double Osc_Gain = 1.95 - ((Osc_Vect_Q * Osc_Vect_Q) + (Osc_Vect_I * Osc_Vect_I));
      Osc_Vect_Q = Osc_Gain * Osc_Q;
      Osc_Vect_I = Osc_Gain * Osc_I;
      float freqAdjFactor = 1.1;
      fL[i] = (exL[i] * freqAdjFactor * Osc_Q) + (exR[i] * freqAdjFactor * Osc_I);
      fR[i] = (exR[i] * freqAdjFactor * Osc_Q) - (exL[i] * freqAdjFactor * Osc_I);
    }
    ch->Osc_Vect_Q = Osc_Vect_Q;
    ch->Osc_Vect_I = Osc_Vect_I;
  }
  memcpy(ch->tap_ncoI, fL, sizeof(float) * (size_t)L);
  memcpy(ch->tap_ncoQ, fR, sizeof(float) * (size_t)L);

  /* decimation: Process.cpp:262-267 (NFM) / 474-479 (default) */
  t41o_fir_decimate_f32(c->dec1, T41O_N_DEC1_TAPS, 4, ch->dec1_I_state, fL, fL, L);
  t41o_fir_decimate_f32(c->dec1, T41O_N_DEC1_TAPS, 4, ch->dec1_Q_state, fR, fR, L);
  t41o_fir_decimate_f32(c->dec2, T41O_N_DEC2_TAPS, 2, ch->dec2_I_state, fL, fL, L / 4);
  t41o_fir_decimate_f32(c->dec2, T41O_N_DEC2_TAPS, 2, ch->dec2_Q_state, fR, fR, L / 4);

  if (mode == T41O_DEMOD_NFM) {
    /* Process.cpp:272-275 */
    for (int i = 0; i < D; i++) {
      FFT_buffer[N + i * 2] = fL[i];
      FFT_buffer[N + i * 2 + 1] = fR[i];
    }
    memcpy(ch->tap_decI, fL, sizeof(float) * (size_t)D);
    memcpy(ch->tap_decQ, fR, sizeof(float) * (size_t)D);
  } else {
    /* Process.cpp:481-492 level adjust */
    float freqKHzFcut;
    float volScaleFactor;
    if (mode == T41O_DEMOD_LSB) freqKHzFcut = -(float)p->FLoCut * 0.001;
    else freqKHzFcut = (float)p->FHiCut * 0.001;
    volScaleFactor = 7.0874 * pow(freqKHzFcut, -1.232);
    for (int i = 0; i < D; i++) fL[i] = fL[i] * volScaleFactor;
    for (int i = 0; i < D; i++) fR[i] = fR[i] * volScaleFactor;
    memcpy(ch->tap_decI, fL, sizeof(float) * (size_t)D);
    memcpy(ch->tap_decQ, fR, sizeof(float) * (size_t)D);

    /* Process.cpp:498-522 overlap-save assemble */
    if (ch->first_block) {
      for (int i = 0; i < N; i++) FFT_buffer[i] = 0.0;
      ch->first_block = 0;
    } else {
      for (int i = 0; i < D; i++) {
        FFT_buffer[i * 2] = ch->last_L[i];
        FFT_buffer[i * 2 + 1] = ch->last_R[i];
      }
    }
    for (int i = 0; i < D; i++) {
      ch->last_L[i] = fL[i];
      ch->last_R[i] = fR[i];
      FFT_buffer[N + i * 2] = fL[i];
      FFT_buffer[N + i * 2 + 1] = fR[i];
    }
    t41o_cfft_f32(FFT_buffer, N, 0);                      /* Process.cpp:535 */
    cmplx_mult_cmplx(FFT_buffer, c->mask, iFFT_buffer, N); /* Process.cpp:547 */
    audio_spectrum(ch, iFFT_buffer);                       /* Process.cpp:550-570 */
    t41o_cfft_f32(iFFT_buffer, N, 1);                     /* Process.cpp:595 */
    AGC(ch, p, c, iFFT_buffer, N);                        /* Process.cpp:605 */
  }

  /* demodulation, Process.cpp:615-761 */
  switch (mode) {
    case T41O_DEMOD_USB:
    case T41O_DEMOD_LSB:
      for (int i = 0; i < D; i++) {
        fL[i] = iFFT_buffer[N + (i * 2)];
        fR[i] = fL[i];
      }
      break;
    case T41O_DEMOD_AM:
      for (int i = 0; i < D; i++) { /* Process.cpp:698-704 */
        float audiotmp = AlphaBetaMag(iFFT_buffer[N + (i * 2)], iFFT_buffer[N + (i * 2) + 1]);
        float w = audiotmp + ch->wold * 0.99f;
        fL[i] = w - ch->wold;
        ch->wold = w;
      }
      t41o_biquad_df1_f32(c->biquad_lowpass1, ch->lp1_state, fL, fR, D); /* Process.cpp:705 */
      memcpy(fL, fR, sizeof(float) * (size_t)D);
      break;
    case T41O_DEMOD_SAM:
      AMDecodeSAM(ch, iFFT_buffer, fL, fR, N); /* Process.cpp:754-755 */
      break;
    case T41O_DEMOD_NFM:
      if (p->nfm_demod == 1) fmdemod_atan_cf(ch, &FFT_buffer[N], fL, D); /* the commented-out alternative */
      else nfmdemod(ch, &FFT_buffer[N], fL, D); /* Process.cpp:716 */
      for (int i = 1; i < D; i++) {        /* Process.cpp:719-727 (starts at 1) */
        float tmp = fL[i];
        tmp = (1 < tmp) ? 1 : tmp;
        tmp = (-1 > tmp) ? -1 : tmp;
        fL[i] = tmp;
      }
      if (p->nfm_demod == 1) {
        /* Process.cpp:734-735 uncommented: float_buffer_R still holds the decimated Q samples, so the
         * last 81 "de-emphasised" samples of every block are those (the buzz the comment mentions) */
        deemphasis_nfm_ff(fL, fR, D);
        memcpy(fL, fR, sizeof(float) * (size_t)D);
      }
      break;
  }

  if (mode == T41O_DEMOD_NFM) { /* Process.cpp:765-816: real overlap-save through the mask */
    for (int i = 0; i < D; i++) {
      FFT_buffer[i * 2] = ch->last_L[i];
      FFT_buffer[i * 2 + 1] = 0;
      ch->last_L[i] = fL[i];
      FFT_buffer[N + i * 2] = fL[i];
      FFT_buffer[N + i * 2 + 1] = 0;
    }
    t41o_cfft_f32(FFT_buffer, N, 0);
    cmplx_mult_cmplx(FFT_buffer, c->mask, iFFT_buffer, N);
    audio_spectrum(ch, iFFT_buffer);                      /* Process.cpp:790-805 */
    t41o_cfft_f32(iFFT_buffer, N, 1);
    AGC(ch, p, c, iFFT_buffer, N);                        /* Process.cpp:810 */
    for (int i = 0; i < D; i++) fL[i] = iFFT_buffer[N + (i * 2)];
  }
  memcpy(ch->tap_ifft, iFFT_buffer, sizeof(float) * (size_t)(2 * N));
  memcpy(ch->tap_demod, fL, sizeof(float) * (size_t)D);

  /* noise reduction and automatic notch, Process.cpp:841-866 (written for blocks of 256 samples) */
  if (p->nrOptionSelect != 0 || p->ANR_notchOn == 1) {
    if (N != 512 || !t41o_nr_supported(p)) return -4;
    t41o_nr_block(ch->nr, p, fL, fR);
  }

  /* interpolation, Process.cpp:917-920 (iFFT_buffer is scratch for the x2 stage) */
  t41o_fir_interpolate_f32(c->int1, T41O_N_INT1_TAPS, 2, ch->int1_state, fL, iFFT_buffer, D);
  t41o_fir_interpolate_f32(c->int2, T41O_N_INT2_TAPS, 4, ch->int2_state, iFFT_buffer, fL, 2 * D);

  /* Process.cpp:929 */
  {
    float s = DF * VolumeToAmplification(p->audioVolume);
    for (int i = 0; i < L; i++) audio[i] = fL[i] * s;
  }
  (void)DF2;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Batch driver (tests / CPU baseline)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  t41o_channel **chs;
  int c0, c1, nframes;
  const t41o_params *p;
  const t41o_coeffs *c;
  const int32_t *nco;
  const float *I, *Q;
  float *audio;
  int rc;
} batch_job;

static void *batch_worker(void *arg) {
  batch_job *j = (batch_job *)arg;
  j->rc = 0;
  for (int ch = j->c0; ch < j->c1; ch++) {
    const int L = j->chs[ch]->L;
    const size_t stride = (size_t)j->nframes * (size_t)L;
    for (int f = 0; f < j->nframes; f++) {
      size_t off = (size_t)ch * stride + (size_t)f * (size_t)L;
      int rc = t41o_process_frame(j->chs[ch], j->p, j->c, (long)j->nco[ch], j->I + off, j->Q + off,
                                  j->audio + off);
      if (rc) j->rc = rc;
    }
  }
  return NULL;
}

int t41o_process_batch(t41o_channel **chs, int nchan, int nframes, const t41o_params *p,
                       const t41o_coeffs *c, const int32_t *NCOFreq, const float *I,
                       const float *Q, float *audio, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > nchan) nthreads = nchan;
  if (nthreads > 256) nthreads = 256;
  batch_job jobs[256];
  pthread_t th[256];
  int per = (nchan + nthreads - 1) / nthreads;
  int used = 0;
  for (int t = 0; t < nthreads; t++) {
    int c0 = t * per, c1 = c0 + per;
    if (c0 >= nchan) break;
    if (c1 > nchan) c1 = nchan;
    jobs[t] = (batch_job){chs, c0, c1, nframes, p, c, NCOFreq, I, Q, audio, 0};
    used++;
  }
  if (used == 1) {
    batch_worker(&jobs[0]);
    return jobs[0].rc;
  }
  for (int t = 0; t < used; t++) pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
  int rc = 0;
  for (int t = 0; t < used; t++) {
    pthread_join(th[t], NULL);
    if (jobs[t].rc) rc = jobs[t].rc;
  }
  return rc;
}

/* ------------------------------------------------------------------------------------------
 * The q15 boundary of ProcessIQData(): Process.cpp:102-111 (in) and :936-937 (out)
 * ---------------------------------------------------------------------------------------- */

/* arm_q15_to_float (CMSIS-DSP, scalar): pDst[i] = (float32_t)pSrc[i] / 32768.0f */
void t41o_q15_to_float(const int16_t *src, float *dst, int n) {
  for (int i = 0; i < n; i++) dst[i] = ((float)src[i] / 32768.0f);
}

/* arm_float_to_q15 (CMSIS-DSP, scalar, ARM_MATH_ROUNDING not defined -- the library default):
 * pDst[i] = (q15_t)__SSAT((q31_t)(pSrc[i] * 32768.0f), 16): truncation toward zero, saturation */
void t41o_float_to_q15(const float *src, int16_t *dst, int n) {
  for (int i = 0; i < n; i++) {
    float in = src[i] * 32768.0f;
    int32_t v; /* (q31_t)(float): the Cortex-M VCVT saturates out-of-range values */
    if (in >= 2147483648.0f) v = INT32_MAX;
    else if (in <= -2147483648.0f) v = INT32_MIN;
    else v = (int32_t)in;
    if (v > 32767) v = 32767; /* __SSAT(v, 16) */
    if (v < -32768) v = -32768;
    dst[i] = (int16_t)v;
  }
}

/* One ProcessIQData() including its sample-format boundary.  Q_in_L / Q_in_R: the N_BLOCKS = 16
 * blocks of BUFFER_SIZE = 128 q15 samples read from the two record queues, back to back
 * (generalised to frame_len samples).  Process.cpp:107-108 converts the R queue into float_buffer_L
 * (I) and the L queue into float_buffer_R (Q); :936-937 plays arm_float_to_q15(float_buffer_L). */
int t41o_process_frame_q15(t41o_channel *ch, const t41o_params *p, const t41o_coeffs *c,
                           long NCOFreq, const int16_t *Q_in_L, const int16_t *Q_in_R,
                           int16_t *Q_out_L) {
  const int L = ch->L;
  float *I = (float *)malloc(sizeof(float) * (size_t)L * 3);
  float *Q = I + L, *audio = Q + L;
  const int BUFFER_SIZE = 128; /* SDT.h:70 */
  for (int i = 0; i < L / BUFFER_SIZE; i++) {
    t41o_q15_to_float(Q_in_R + BUFFER_SIZE * i, &I[BUFFER_SIZE * i], BUFFER_SIZE);
    t41o_q15_to_float(Q_in_L + BUFFER_SIZE * i, &Q[BUFFER_SIZE * i], BUFFER_SIZE);
  }
  int rc = t41o_process_frame(ch, p, c, NCOFreq, I, Q, audio);
  if (rc == 0) t41o_float_to_q15(audio, Q_out_L, L);
  free(I);
  return rc;
}

int t41o_process_batch_q15(t41o_channel **chs, int nchan, int nframes, const t41o_params *p,
                           const t41o_coeffs *c, const int32_t *NCOFreq, const int16_t *Q_in_L,
                           const int16_t *Q_in_R, int16_t *Q_out_L) {
  for (int ch = 0; ch < nchan; ch++) {
    const int L = chs[ch]->L;
    for (int f = 0; f < nframes; f++) {
      const size_t o = ((size_t)ch * (size_t)nframes + (size_t)f) * (size_t)L;
      int rc = t41o_process_frame_q15(chs[ch], p, c, NCOFreq[ch], Q_in_L + o, Q_in_R + o, Q_out_L + o);
      if (rc) return rc;
    }
  }
  return 0;
}

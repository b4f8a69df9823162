/*
 * oracle/t41_nr_oracle.c -- CPU restatement of the T41 receive path's optional noise reduction and
 * automatic notch: Kim1_NR() (Noise.cpp:108-313), Xanr() (Noise.cpp:322-370),
 * SpectralNoiseReduction() (Noise.cpp:379-655) and their call sites Process.cpp:841-866.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see t41_oracle.h).  PARITY UNPINNED like the rest:
 * the reference holds no fixtures for these functions either.
 *
 * Citations are relative to /root/reference/software/T41_SDR/.  Every expression keeps the
 * reference's float / double promotions (unsuffixed literals are double, float32_t variables are
 * float; compile with -ffp-contract=off).  The functions work on blocks of FFT_length / 2 = 256
 * audio samples at 24 kS/s and are written for that size (NR_FFT_L = 256, Noise.h:8).
 *
 * Things the reference does that a reader might not expect, restated as written:
 *  - Xanr() writes its result to float_buffer_R.  The notch call site copies R to L afterwards
 *    (Process.cpp:862-866); the "LMS NR" call site (Process.cpp:852-857) does NOT -- it scales
 *    float_buffer_L, Xanr()'s unmodified INPUT, by 1.5 -- so option 3 changes the audio by that
 *    gain only, and advances the adaptive filter's state (which the notch shares).
 *  - In Xanr() the `else if` of the leak adaptation binds to the INNER `if` (Noise.cpp:351-356):
 *    whenever nev < nel the index goes up by ANR_lincr and, not exceeding the maximum, straight
 *    down by ANR_ldecr, i.e. it is pinned at ANR_lidx_min.
 *  - The "conjugate symmetric" partner both spectral functions weight together with bin i is
 *    complex bin 255 - i (float index 512 - 2i - 2, Noise.cpp:263-264, 593-594), not 256 - i.
 *  - Kim1_NR() never touches bins outside [VAD_low, VAD_high) of NR_Gts, so they keep their
 *    power-on values; SpectralNoiseReduction() runs its musical-noise smoothing INSIDE the loop
 *    over the bins (Noise.cpp:529-588, the closing brace commented `end of "if ..."` closes that
 *    loop), and passes the audio through untouched for its first 20 half-blocks while it averages
 *    the noise estimate (NR_first_time_2 < 3: the inverse FFT and overlap-add sit inside `== 3`).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "t41_oracle.h"

#define NR_FFT_L 256         /* Noise.h:8 */
#define ANR_DLINE_SIZE 512   /* Noise.h:5 */
#define NR_L_FRAMES 3        /* Noise.cpp:16 */
#define NR_N_FRAMES 15       /* Noise.cpp:17 */
#define PI_F 3.1415926535897932384626433832795f /* FIR.h:10 */

/* sqrtHann[256], Noise.cpp:49-83: a fixed table of the reference (= sin(pi i / 255) to 8-9 digits) */
static const float sqrtHann[256] = {
    0.0f, 0.01231966f, 0.024637449f, 0.036951499f, 0.049259941f, 0.061560906f, 0.073852527f, 0.086132939f,
    0.098400278f, 0.110652682f, 0.122888291f, 0.135105247f, 0.147301698f, 0.159475791f, 0.171625679f, 0.183749518f,
    0.195845467f, 0.207911691f, 0.219946358f, 0.231947641f, 0.24391372f, 0.255842778f, 0.267733003f, 0.279582593f,
    0.291389747f, 0.303152674f, 0.314869589f, 0.326538713f, 0.338158275f, 0.349726511f, 0.361241666f, 0.372701992f,
    0.384105749f, 0.395451207f, 0.406736643f, 0.417960345f, 0.429120609f, 0.440215741f, 0.451244057f, 0.462203884f,
    0.473093557f, 0.483911424f, 0.494655843f, 0.505325184f, 0.515917826f, 0.526432163f, 0.536866598f, 0.547219547f,
    0.557489439f, 0.567674716f, 0.577773831f, 0.587785252f, 0.597707459f, 0.607538946f, 0.617278221f, 0.626923806f,
    0.636474236f, 0.645928062f, 0.65528385f, 0.664540179f, 0.673695644f, 0.682748855f, 0.691698439f, 0.700543038f,
    0.709281308f, 0.717911923f, 0.726433574f, 0.734844967f, 0.743144825f, 0.75133189f, 0.759404917f, 0.767362681f,
    0.775203976f, 0.78292761f, 0.790532412f, 0.798017227f, 0.805380919f, 0.812622371f, 0.819740483f, 0.826734175f,
    0.833602385f, 0.840344072f, 0.846958211f, 0.853443799f, 0.859799851f, 0.866025404f, 0.872119511f, 0.878081248f,
    0.88390971f, 0.889604013f, 0.895163291f, 0.900586702f, 0.905873422f, 0.911022649f, 0.916033601f, 0.920905518f,
    0.92563766f, 0.930229309f, 0.934679767f, 0.938988361f, 0.943154434f, 0.947177357f, 0.951056516f, 0.954791325f,
    0.958381215f, 0.961825643f, 0.965124085f, 0.968276041f, 0.971281032f, 0.974138602f, 0.976848318f, 0.979409768f,
    0.981822563f, 0.984086337f, 0.986200747f, 0.988165472f, 0.989980213f, 0.991644696f, 0.993158666f, 0.994521895f,
    0.995734176f, 0.996795325f, 0.99770518f, 0.998463604f, 0.999070481f, 0.99952572f, 0.99982925f, 0.999981027f,
    0.999981027f, 0.99982925f, 0.99952572f, 0.999070481f, 0.998463604f, 0.99770518f, 0.996795325f, 0.995734176f,
    0.994521895f, 0.993158666f, 0.991644696f, 0.989980213f, 0.988165472f, 0.986200747f, 0.984086337f, 0.981822563f,
    0.979409768f, 0.976848318f, 0.974138602f, 0.971281032f, 0.968276041f, 0.965124085f, 0.961825643f, 0.958381215f,
    0.954791325f, 0.951056516f, 0.947177357f, 0.943154434f, 0.938988361f, 0.934679767f, 0.930229309f, 0.92563766f,
    0.920905518f, 0.916033601f, 0.911022649f, 0.905873422f, 0.900586702f, 0.895163291f, 0.889604013f, 0.88390971f,
    0.878081248f, 0.872119511f, 0.866025404f, 0.859799851f, 0.853443799f, 0.846958211f, 0.840344072f, 0.833602385f,
    0.826734175f, 0.819740483f, 0.812622371f, 0.805380919f, 0.798017227f, 0.790532412f, 0.78292761f, 0.775203976f,
    0.767362681f, 0.759404917f, 0.75133189f, 0.743144825f, 0.734844967f, 0.726433574f, 0.717911923f, 0.709281308f,
    0.700543038f, 0.691698439f, 0.682748855f, 0.673695644f, 0.664540179f, 0.65528385f, 0.645928062f, 0.636474236f,
    0.626923806f, 0.617278221f, 0.607538946f, 0.597707459f, 0.587785252f, 0.577773831f, 0.567674716f, 0.557489439f,
    0.547219547f, 0.536866598f, 0.526432163f, 0.515917826f, 0.505325184f, 0.494655843f, 0.483911424f, 0.473093557f,
    0.462203884f, 0.451244057f, 0.440215741f, 0.429120609f, 0.417960345f, 0.406736643f, 0.395451207f, 0.384105749f,
    0.372701992f, 0.361241666f, 0.349726511f, 0.338158275f, 0.326538713f, 0.314869589f, 0.303152674f, 0.291389747f,
    0.279582593f, 0.267733003f, 0.255842778f, 0.24391372f, 0.231947641f, 0.219946358f, 0.207911691f, 0.195845467f,
    0.183749518f, 0.171625679f, 0.159475791f, 0.147301698f, 0.135105247f, 0.122888291f, 0.110652682f, 0.098400278f,
    0.086132939f, 0.073852527f, 0.061560906f, 0.049259941f, 0.036951499f, 0.024637449f, 0.01231966f, 0.0f,
};

struct t41o_nr {
  /* Noise.cpp:19-38 */
  float NR_FFT_buffer[512];
  float NR_output_audio_buffer[NR_FFT_L];
  float NR_last_iFFT_result[NR_FFT_L / 2];
  float NR_last_sample_buffer_L[NR_FFT_L / 2];
  float NR_X[NR_FFT_L / 2][3];
  float NR_E[NR_FFT_L / 2][15];
  float NR_M[NR_FFT_L / 2];
  float NR_Nest[NR_FFT_L / 2][2];
  float NR_lambda[NR_FFT_L / 2];
  float NR_Gts[NR_FFT_L / 2][2];
  float NR_G[NR_FFT_L / 2];
  float NR_SNR_prio[NR_FFT_L / 2];
  float NR_SNR_post[NR_FFT_L / 2];
  float NR_Hk_old[NR_FFT_L / 2];
  float NR_long_tone_gain[NR_FFT_L / 2];
  float ANR_d[ANR_DLINE_SIZE];
  float ANR_w[ANR_DLINE_SIZE];
  /* Noise.cpp:40-56 (the ones that change) */
  int ANR_in_idx;
  float ANR_lidx, ANR_ngamma;
  /* Kim1_NR()'s statics, Noise.cpp:109-110 */
  uint32_t NR_X_pointer, NR_E_pointer;
  /* SpectralNoiseReduction()'s statics, Noise.cpp:390-419 */
  uint8_t NR_init_counter;
  int NR_first_time_2;
  float pslp[NR_FFT_L / 2], xt[NR_FFT_L / 2];
  float xih1r, pfac;
  int spectral_statics_set;
};

t41o_nr *t41o_nr_create(void) {
  t41o_nr *s = (t41o_nr *)calloc(1, sizeof(*s));
  if (s) t41o_nr_reset(s);
  return s;
}
void t41o_nr_destroy(t41o_nr *s) { free(s); }

/* InitializeDataArrays()'s CLEAR_VAR block (T41_SDR.ino:479-504), then SpectralNoiseReductionInit()
 * (Noise.cpp:692-707, called at T41_SDR.ino:657), the initialisers of Noise.cpp:40-56 and the
 * function-local statics */
void t41o_nr_reset(t41o_nr *s) {
  memset(s, 0, sizeof(*s));
  for (int i = 0; i < NR_FFT_L / 2; i++) {
    s->NR_last_sample_buffer_L[i] = 0.1;
    s->NR_Hk_old[i] = 0.1;
    s->NR_Nest[i][0] = 0.01;
    s->NR_Nest[i][1] = 0.015;
    s->NR_Gts[i][1] = 0.1;
    s->NR_M[i] = 500.0;
    s->NR_E[i][0] = 0.1;
    s->NR_X[i][1] = 0.5;
    s->NR_SNR_post[i] = 2.0;
    s->NR_SNR_prio[i] = 1.0;
    s->NR_long_tone_gain[i] = 1.0;
  }
  s->ANR_in_idx = 0;
  s->ANR_lidx = 120.0;
  s->ANR_ngamma = 0.001;
  s->NR_first_time_2 = 1;
}

/* Noise.cpp:134-168 / 421-433, 514-531: the bins the filter passes, from bands[].FLoCut / FHiCut */
static void vad_range(const t41o_params *p, uint8_t *lo, uint8_t *hi) {
  const int SampleRate = 192000;  /* T41_SDR.ino:129 */
  const float DF = 8.0;           /* T41_SDR.ino:335 */
  uint8_t VAD_low = 0, VAD_high = 127;
  float lf_freq, uf_freq;
  if (p->FLoCut <= 0 && p->FHiCut >= 0) {
    lf_freq = 0.0;
    uf_freq = fmax(-(float)p->FLoCut, (float)p->FHiCut);
  } else {
    if (p->FLoCut > 0) {
      lf_freq = (float)p->FLoCut;
      uf_freq = (float)p->FHiCut;
    } else {
      uf_freq = -(float)p->FLoCut;
      lf_freq = -(float)p->FHiCut;
    }
  }
  lf_freq /= ((SampleRate / DF) / NR_FFT_L);
  uf_freq /= ((SampleRate / DF) / NR_FFT_L);
  VAD_low = (int)lf_freq;
  VAD_high = (int)uf_freq;
  if (VAD_low == VAD_high) {
    VAD_high++;
  }
  if (VAD_low < 1) {
    VAD_low = 1;
  } else if (VAD_low > NR_FFT_L / 2 - 2) {
    VAD_low = NR_FFT_L / 2 - 2;
  }
  if (VAD_high < 1) {
    VAD_high = 1;
  } else if (VAD_high > NR_FFT_L / 2) {
    VAD_high = NR_FFT_L / 2;
  }
  *lo = VAD_low;
  *hi = VAD_high;
}

/* Kim1_NR(), Noise.cpp:108-313 */
static void Kim1_NR(t41o_nr *s, const t41o_params *p, float *float_buffer_L, float *float_buffer_R) {
  float *NR_FFT_buffer = s->NR_FFT_buffer;
  const float NR_alpha = p->NR_alpha, NR_beta = p->NR_beta, NR_PSI = p->NR_PSI;
  const uint8_t NR_use_X = 0;  /* Noise.cpp:15 */
  uint8_t VAD_low, VAD_high;
  float NR_sum;
  float NR_KIM_K = 1.0;
  float NR_onemalpha = (1.0 - NR_alpha);
  float NR_onemtwobeta = (1.0 - (2.0 * NR_beta));
  float NR_T;
  vad_range(p, &VAD_low, &VAD_high);

  for (int k = 0; k < 2; k++) {
    for (int i = 0; i < NR_FFT_L / 2; i++) {
      NR_FFT_buffer[i * 2] = s->NR_last_sample_buffer_L[i];
      NR_FFT_buffer[i * 2 + 1] = 0.0;
    }
    for (int i = 0; i < NR_FFT_L / 2; i++) {
      s->NR_last_sample_buffer_L[i] = float_buffer_L[i + k * (NR_FFT_L / 2)];
    }
    for (int i = 0; i < NR_FFT_L / 2; i++) {
      NR_FFT_buffer[NR_FFT_L + i * 2] = float_buffer_L[i + k * (NR_FFT_L / 2)];
      NR_FFT_buffer[NR_FFT_L + i * 2 + 1] = 0.0;
    }
    for (int idx = 0; idx < NR_FFT_L; idx++) { /* Hann window, Noise.cpp:188-191 */
      float temp_sample = 0.5 * (float)(1.0 - (cosf(PI_F * 2.0 * (float)idx / (float)((NR_FFT_L)-1))));
      NR_FFT_buffer[idx * 2] *= temp_sample;
    }
    t41o_cfft_f32(NR_FFT_buffer, NR_FFT_L, 0);
    for (int i = 0; i < NR_FFT_L / 2; i++) {
      s->NR_X[i][s->NR_X_pointer] = (NR_FFT_buffer[i * 2] * NR_FFT_buffer[i * 2] + NR_FFT_buffer[i * 2 + 1] * NR_FFT_buffer[i * 2 + 1]);
    }
    for (int i = VAD_low; i < VAD_high; i++) {
      NR_sum = 0.0;
      for (int j = 0; j < NR_L_FRAMES; j++) {
        NR_sum = NR_sum + s->NR_X[i][j];
      }
      s->NR_E[i][s->NR_E_pointer] = NR_sum / (float)NR_L_FRAMES;
    }
    for (int i = VAD_low; i < VAD_high; i++) {
      s->NR_M[i] = s->NR_E[i][0];
      for (uint8_t j = 1; j < NR_N_FRAMES; j++) {
        if (s->NR_E[i][j] < s->NR_M[i]) {
          s->NR_M[i] = s->NR_E[i][j];
        }
      }
    }
    for (int i = VAD_low; i < VAD_high; i++) {
      NR_T = s->NR_X[i][s->NR_X_pointer] / s->NR_M[i];
      if (NR_T > NR_PSI) {
        s->NR_lambda[i] = s->NR_M[i];
      } else {
        s->NR_lambda[i] = s->NR_E[i][s->NR_E_pointer];
      }
    }
    for (int i = VAD_low; i < VAD_high; i++) {
      if (NR_use_X) {
        s->NR_G[i] = 1.0 - (s->NR_lambda[i] * NR_KIM_K / s->NR_X[i][s->NR_X_pointer]);
        if (s->NR_G[i] < 0.0) s->NR_G[i] = 0.0;
      } else {
        s->NR_G[i] = 1.0 - (s->NR_lambda[i] * NR_KIM_K / s->NR_E[i][s->NR_E_pointer]);
        if (s->NR_G[i] < 0.0) s->NR_G[i] = 0.0;
      }
      s->NR_Gts[i][0] = NR_alpha * s->NR_Gts[i][1] + (NR_onemalpha)*s->NR_G[i];
      s->NR_Gts[i][1] = s->NR_Gts[i][0];
    }
    for (int i = 1; i < ((NR_FFT_L / 2) - 1); i++) {
      s->NR_G[i] = NR_beta * s->NR_Gts[i - 1][0] + NR_onemtwobeta * s->NR_Gts[i][0] + NR_beta * s->NR_Gts[i + 1][0];
    }
    s->NR_G[0] = (NR_onemtwobeta + NR_beta) * s->NR_Gts[0][0] + NR_beta * s->NR_Gts[1][0];
    s->NR_G[(NR_FFT_L / 2) - 1] = NR_beta * s->NR_Gts[(NR_FFT_L / 2) - 2][0] + (NR_onemtwobeta + NR_beta) * s->NR_Gts[(NR_FFT_L / 2) - 1][0];
    for (int i = 0; i < NR_FFT_L / 2; i++) {
      NR_FFT_buffer[i * 2] = NR_FFT_buffer[i * 2] * s->NR_G[i];
      NR_FFT_buffer[i * 2 + 1] = NR_FFT_buffer[i * 2 + 1] * s->NR_G[i];
      NR_FFT_buffer[NR_FFT_L * 2 - i * 2 - 2] = NR_FFT_buffer[NR_FFT_L * 2 - i * 2 - 2] * s->NR_G[i];
      NR_FFT_buffer[NR_FFT_L * 2 - i * 2 - 1] = NR_FFT_buffer[NR_FFT_L * 2 - i * 2 - 1] * s->NR_G[i];
    }
    s->NR_X_pointer = s->NR_X_pointer + 1;
    if (s->NR_X_pointer >= NR_L_FRAMES) {
      s->NR_X_pointer = 0;
    }
    s->NR_E_pointer = s->NR_E_pointer + 1;
    if (s->NR_E_pointer >= NR_N_FRAMES) {
      s->NR_E_pointer = 0;
    }
    t41o_cfft_f32(NR_FFT_buffer, NR_FFT_L, 1);
    for (int i = 0; i < NR_FFT_L / 2; i++) {
      s->NR_output_audio_buffer[i + k * (NR_FFT_L / 2)] = NR_FFT_buffer[i * 2] + s->NR_last_iFFT_result[i];
    }
    for (int i = 0; i < NR_FFT_L / 2; i++) {
      s->NR_last_iFFT_result[i] = NR_FFT_buffer[NR_FFT_L + i * 2];
    }
  }
  for (int i = 0; i < NR_FFT_L; i++) {
    float_buffer_L[i] = s->NR_output_audio_buffer[i];
    float_buffer_R[i] = float_buffer_L[i];
  }
}

/* Xanr(), Noise.cpp:322-370: variable-leak LMS, automatic notch (ANR_notch = 1) or noise reduction (0) */
static void Xanr(t41o_nr *s, int ANR_notch, const float *float_buffer_L, float *float_buffer_R) {
  const int ANR_buff_size = 256; /* FFT_length / 2, Noise.cpp:40 */
  const int ANR_delay = 16, ANR_mask = ANR_DLINE_SIZE - 1, ANR_taps = 64;
  const float ANR_den_mult = 6.25e-10, ANR_gamma = 0.1, ANR_lidx_min = 120.0, ANR_lidx_max = 200.0;
  const float ANR_lincr = 1.0, ANR_ldecr = 3.0, ANR_two_mu = 0.0001;
  int idx;
  float c0, c1;
  float y, error, sigma, inv_sigp;
  float nel, nev;
  float *ANR_d = s->ANR_d, *ANR_w = s->ANR_w;

  for (int i = 0; i < ANR_buff_size; i++) {
    ANR_d[s->ANR_in_idx] = float_buffer_L[i];
    y = 0;
    sigma = 0;
    for (int j = 0; j < ANR_taps; j++) {
      idx = (s->ANR_in_idx + j + ANR_delay) & ANR_mask;
      y += ANR_w[j] * ANR_d[idx];
      sigma += ANR_d[idx] * ANR_d[idx];
    }
    inv_sigp = 1.0 / (sigma + 1e-10);
    error = ANR_d[s->ANR_in_idx] - y;
    if (ANR_notch)
      float_buffer_R[i] = error;
    else
      float_buffer_R[i] = y;
    if ((nel = error * (1.0 - ANR_two_mu * sigma * inv_sigp)) < 0.0) nel = -nel;
    if ((nev = ANR_d[s->ANR_in_idx] - (1.0 - ANR_two_mu * s->ANR_ngamma) * y - ANR_two_mu * error * sigma * inv_sigp) < 0.0) nev = -nev;
    if (nev < nel) {
      if ((s->ANR_lidx += ANR_lincr) > ANR_lidx_max)
        s->ANR_lidx = ANR_lidx_max;
      else if ((s->ANR_lidx -= ANR_ldecr) < ANR_lidx_min)
        s->ANR_lidx = ANR_lidx_min;
    }
    s->ANR_ngamma = ANR_gamma * (s->ANR_lidx * s->ANR_lidx) * (s->ANR_lidx * s->ANR_lidx) * ANR_den_mult;
    c0 = 1.0 - ANR_two_mu * s->ANR_ngamma;
    c1 = ANR_two_mu * error * inv_sigp;
    for (int j = 0; j < ANR_taps; j++) {
      idx = (s->ANR_in_idx + j + ANR_delay) & ANR_mask;
      ANR_w[j] = c0 * ANR_w[j] + c1 * ANR_d[idx];
    }
    s->ANR_in_idx = (s->ANR_in_idx + ANR_mask) & ANR_mask;
  }
}

/* SpectralNoiseReduction(), Noise.cpp:379-655 */
static void SpectralNoiseReduction(t41o_nr *s, const t41o_params *p, float *float_buffer_L, float *float_buffer_R) {
  float *NR_FFT_buffer = s->NR_FFT_buffer;
  const float NR_alpha = p->NR_alpha;
  uint8_t VAD_low = 0, VAD_high = 127;
  const float tinc = 0.00533333;
  const float tax = 0.0239;
  const float tap = 0.05062;
  const float psthr = 0.99;
  const float pnsaf = 0.01;
  const float asnr = 20;
  const float psini = 0.5;
  const float pspri = 0.5;
  float ax, ap, xih1;
  ax = expf(-tinc / tax);
  ap = expf(-tinc / tap);
  xih1 = powf(10, (float)asnr / 10.0);
  if (!s->spectral_statics_set) { /* function-local statics with dynamic initialisers: set on the first call */
    s->xih1r = 1.0 / (1.0 + xih1) - 1.0;
    s->pfac = (1.0 / pspri - 1.0) * (1.0 + xih1);
    s->spectral_statics_set = 1;
  }
  const float xih1r = s->xih1r, pfac = s->pfac;
  float snr_prio_min = powf(10, -(float)20 / 20.0);
  float *pslp = s->pslp, *xt = s->xt;
  float xtr, pre_power, post_power, power_ratio;
  int16_t NN;
  const int16_t NR_width = 4;
  const float power_threshold = 0.4;
  float ph1y[NR_FFT_L / 2];

  if (s->NR_first_time_2 == 1) {
    for (int i = 0; i < NR_FFT_L / 2; i++) {
      s->NR_last_sample_buffer_L[i] = 0.0;
      s->NR_G[i] = 1.0;
      s->NR_Hk_old[i] = 1.0;
      s->NR_Nest[i][0] = 0.0;
      s->NR_Nest[i][1] = 1.0;
      pslp[i] = 0.5;
    }
    s->NR_first_time_2 = 2;
  }

  for (int k = 0; k < 2; k++) {
    for (int i = 0; i < NR_FFT_L / 2; i++) {
      NR_FFT_buffer[i * 2] = s->NR_last_sample_buffer_L[i];
      NR_FFT_buffer[i * 2 + 1] = 0.0;
    }
    for (int i = 0; i < NR_FFT_L / 2; i++) {
      s->NR_last_sample_buffer_L[i] = float_buffer_L[i + k * (NR_FFT_L / 2)];
    }
    for (int i = 0; i < NR_FFT_L / 2; i++) {
      NR_FFT_buffer[NR_FFT_L + i * 2] = float_buffer_L[i + k * (NR_FFT_L / 2)];
      NR_FFT_buffer[NR_FFT_L + i * 2 + 1] = 0.0;
    }
    for (int idx = 0; idx < NR_FFT_L; idx++) {
      NR_FFT_buffer[idx * 2] *= sqrtHann[idx];
    }
    t41o_cfft_f32(NR_FFT_buffer, NR_FFT_L, 0);
    for (int i = 0; i < NR_FFT_L / 2; i++) {
      s->NR_X[i][0] = (NR_FFT_buffer[i * 2] * NR_FFT_buffer[i * 2] + NR_FFT_buffer[i * 2 + 1] * NR_FFT_buffer[i * 2 + 1]);
    }
    if (s->NR_first_time_2 == 2) {
      for (int i = 0; i < NR_FFT_L / 2; i++) {
        s->NR_Nest[i][0] = s->NR_Nest[i][0] + 0.05 * s->NR_X[i][0];
        xt[i] = psini * s->NR_Nest[i][0];
      }
      s->NR_init_counter++;
      if (s->NR_init_counter > 19) {
        s->NR_init_counter = 0;
        s->NR_first_time_2 = 3;
      }
    }
    if (s->NR_first_time_2 == 3) {
      for (int i = 0; i < NR_FFT_L / 2; i++) {
        ph1y[i] = 1.0 / (1.0 + pfac * expf(xih1r * s->NR_X[i][0] / xt[i]));
        pslp[i] = ap * pslp[i] + (1.0 - ap) * ph1y[i];
        if (pslp[i] > psthr) {
          ph1y[i] = 1.0 - pnsaf;
        } else {
          ph1y[i] = fmin(ph1y[i], 1.0);
        }
        xtr = (1.0 - ph1y[i]) * s->NR_X[i][0] + ph1y[i] * xt[i];
        xt[i] = ax * xt[i] + (1.0 - ax) * xtr;
      }
      for (int i = 0; i < NR_FFT_L / 2; i++) {
        s->NR_SNR_post[i] = fmax(fmin(s->NR_X[i][0] / xt[i], 1000.0), snr_prio_min);
        s->NR_SNR_prio[i] = fmax(NR_alpha * s->NR_Hk_old[i] + (1.0 - NR_alpha) * fmax(s->NR_SNR_post[i] - 1.0, 0.0), 0.0);
      }
      vad_range(p, &VAD_low, &VAD_high);
      float v;
      for (int i = VAD_low; i < VAD_high; i++) {
        {
          v = s->NR_SNR_prio[i] * s->NR_SNR_post[i] / (1.0 + s->NR_SNR_prio[i]);
          s->NR_G[i] = 1.0 / s->NR_SNR_post[i] * sqrtf((0.7212 * v + v * v));
          s->NR_Hk_old[i] = s->NR_SNR_post[i] * s->NR_G[i] * s->NR_G[i];
        }
        /* musical noise treatment, inside the loop over the bins (inner loops shadow i) */
        pre_power = 0.0;
        post_power = 0.0;
        for (int i = VAD_low; i < VAD_high; i++) {
          pre_power += s->NR_X[i][0];
          post_power += s->NR_G[i] * s->NR_G[i] * s->NR_X[i][0];
        }
        power_ratio = post_power / pre_power;
        if (power_ratio > power_threshold) {
          power_ratio = 1.0;
          NN = 1;
        } else {
          NN = 1 + 2 * (int)(0.5 + NR_width * (1.0 - power_ratio / power_threshold));
        }
        for (int i = VAD_low + NN / 2; i < VAD_high - NN / 2; i++) {
          s->NR_Nest[i][0] = 0.0;
          for (int m = i - NN / 2; m <= i + NN / 2; m++) {
            s->NR_Nest[i][0] += s->NR_G[m];
          }
          s->NR_Nest[i][0] /= (float)NN;
        }
        for (int i = VAD_low; i < VAD_low + NN / 2; i++) {
          s->NR_Nest[i][0] = 0.0;
          for (int m = i; m < (i + NN); m++) {
            s->NR_Nest[i][0] += s->NR_G[m];
          }
          s->NR_Nest[i][0] /= (float)NN;
        }
        for (int i = VAD_high - NN; i < VAD_high; i++) {
          s->NR_Nest[i][0] = 0.0;
          for (int m = i; m > (i - NN); m--) {
            s->NR_Nest[i][0] += s->NR_G[m];
          }
          s->NR_Nest[i][0] /= (float)NN;
        }
        for (int i = VAD_low + NN / 2; i < VAD_high - NN / 2; i++) {
          s->NR_G[i] = s->NR_Nest[i][0];
        }
      }
      for (int i = 0; i < NR_FFT_L / 2; i++) {
        NR_FFT_buffer[i * 2] = NR_FFT_buffer[i * 2] * s->NR_G[i] * s->NR_long_tone_gain[i];
        NR_FFT_buffer[i * 2 + 1] = NR_FFT_buffer[i * 2 + 1] * s->NR_G[i] * s->NR_long_tone_gain[i];
        NR_FFT_buffer[NR_FFT_L * 2 - i * 2 - 2] = NR_FFT_buffer[NR_FFT_L * 2 - i * 2 - 2] * s->NR_G[i] * s->NR_long_tone_gain[i];
        NR_FFT_buffer[NR_FFT_L * 2 - i * 2 - 1] = NR_FFT_buffer[NR_FFT_L * 2 - i * 2 - 1] * s->NR_G[i] * s->NR_long_tone_gain[i];
      }
      t41o_cfft_f32(NR_FFT_buffer, NR_FFT_L, 1);
      for (int idx = 0; idx < NR_FFT_L; idx++) {
        NR_FFT_buffer[idx * 2] *= sqrtHann[idx];
      }
      for (int i = 0; i < NR_FFT_L / 2; i++) {
        float_buffer_L[i + k * (NR_FFT_L / 2)] = NR_FFT_buffer[i * 2] + s->NR_last_iFFT_result[i];
        float_buffer_R[i + k * (NR_FFT_L / 2)] = float_buffer_L[i + k * (NR_FFT_L / 2)];
      }
      for (int i = 0; i < NR_FFT_L / 2; i++) {
        s->NR_last_iFFT_result[i] = NR_FFT_buffer[NR_FFT_L + i * 2];
      }
    }
  }
}

int t41o_nr_supported(const t41o_params *p) {
  if (p->nrOptionSelect < 0 || p->nrOptionSelect > 3) return 0;
  if (p->nrOptionSelect == 2) {
    /* the musical-noise smoothing reaches 2 NN - 1 = 17 bins below VAD_high and NN + NN / 2 - 2 = 11 above
     * VAD_low (Noise.cpp:556-574 with NN up to 9): outside NR_Nest / NR_G for narrower pass bands */
    uint8_t lo, hi;
    vad_range(p, &lo, &hi);
    if (hi < 17 || lo + 11 > NR_FFT_L / 2 - 1) return 0;
  }
  return 1;
}

/* Process.cpp:841-866 on one block of FFT_length / 2 = 256 samples */
void t41o_nr_block(t41o_nr *s, const t41o_params *p, float *float_buffer_L, float *float_buffer_R) {
  switch (p->nrOptionSelect) {
    case 0:
      break;
    case 1:
      Kim1_NR(s, p, float_buffer_L, float_buffer_R);
      for (int i = 0; i < 256; i++) float_buffer_L[i] = float_buffer_L[i] * 30; /* arm_scale_f32(float_buffer_L, 30, ...) */
      break;
    case 2:
      SpectralNoiseReduction(s, p, float_buffer_L, float_buffer_R);
      break;
    case 3:
      Xanr(s, 0, float_buffer_L, float_buffer_R);
      for (int i = 0; i < 256; i++) float_buffer_L[i] = float_buffer_L[i] * 1.5f;
      break;
  }
  if (p->ANR_notchOn == 1) {
    Xanr(s, 1, float_buffer_L, float_buffer_R);
    memcpy(float_buffer_L, float_buffer_R, sizeof(float) * 256); /* arm_copy_f32 */
  }
}

/* test access: a copy of selected state arrays.  which: 0 ANR_w[64], 1 NR_G[128], 2 NR_Gts[.][1] [128], 3 xt[128], 4 NR_Hk_old[128] */
int t41o_nr_peek(const t41o_nr *s, int which, float *dst, int maxlen) {
  int n = 0;
  float tmp[128];
  const float *src = NULL;
  switch (which) {
    case 0: src = s->ANR_w; n = 64; break;
    case 1: src = s->NR_G; n = 128; break;
    case 2: for (int i = 0; i < 128; i++) tmp[i] = s->NR_Gts[i][1]; src = tmp; n = 128; break;
    case 3: src = s->xt; n = 128; break;
    case 4: src = s->NR_Hk_old; n = 128; break;
    default: return 0;
  }
  if (n > maxlen) n = maxlen;
  memcpy(dst, src, sizeof(float) * (size_t)n);
  return n;
}

/*
 * oracle/t41_oracle.h -- CPU restatement of the T41 receive DSP block function.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg may load it.  The product
 * (t41_sdr_amd/, include/t41rx.h) never links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (tmr4/T41_SDR, /root/reference/software/T41_SDR) ships no
 * tests, golden vectors or known-answer fixtures for this path, cannot be compiled here
 * (Arduino/Teensy only) and its arithmetic lives in ARM CMSIS-DSP (arm_math.h, SDT.h:22-23;
 * version not pinned by the reference), which is neither vendored nor installed.  This file
 * restates (a) the reference's own C++ for the path and (b) the published scalar algorithms of
 * the CMSIS-DSP f32 primitives it calls.  It is cross-checked against an independent float64
 * numpy/scipy model in tests/test_oracle_vs_f64.py.
 *
 * All "file:line" citations are relative to /root/reference/software/T41_SDR/.
 * Plain C (C11), scalar f32 with the reference's float/double promotions; compile with
 * -ffp-contract=off so that a*b+c is two roundings as in the documented CMSIS scalar loops.
 */
#ifndef T41_ORACLE_H
#define T41_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* demodulation modes, SDT.h:58-68 */
enum {
  T41O_DEMOD_USB = 0,
  T41O_DEMOD_LSB = 1,
  T41O_DEMOD_AM  = 2,
  T41O_DEMOD_NFM = 3,
  T41O_DEMOD_SAM = 8 /* SDT.h:67 */
};

/* xmtMode, SDT.h:48-50 */
enum { T41O_SSB_MODE = 0, T41O_CW_MODE = 1, T41O_DATA_MODE = 2 };

#define T41O_N_DEC1_TAPS 28 /* n_dec1_taps, T41_SDR.ino:344 (evaluates to 28) */
#define T41O_N_DEC2_TAPS 46 /* n_dec2_taps, T41_SDR.ino:345 (evaluates to 46) */
#define T41O_N_INT1_TAPS 48 /* T41_SDR.ino:595-603 */
#define T41O_N_INT2_TAPS 32 /* T41_SDR.ino:608-616 */

/* The globals ProcessIQData() reads, gathered in one POD. */
typedef struct {
  int32_t fft_length;       /* FFT_LENGTH SDT.h:39 (512); 1024/2048/4096 = synthetic generalisation */
  int32_t mode;             /* bands[currentBand].mode */
  int32_t FLoCut;           /* bands[currentBand].FLoCut, Hz */
  int32_t FHiCut;           /* bands[currentBand].FHiCut, Hz */
  int32_t rfGainAllBands;   /* gwv.cpp:17 */
  int32_t RFgain;           /* bands[currentBand].RFgain (int) */
  float   IQAmpCorrectionFactor;   /* gwv.cpp:71 */
  float   IQPhaseCorrectionFactor; /* gwv.cpp:72 */
  int32_t AGCMode;          /* gwv.cpp:15; 0 = fixed gain, 1..4 = long/slow/med/fast (DSP_Fn.cpp:373-402) */
  int32_t audioVolume;      /* gwv.cpp:16 */
  int32_t nfmFilterBW;      /* Filter.cpp:16 */
  int32_t xmtMode;          /* gwv.cpp:22 */
  int32_t CWFreqShift;      /* Freq_Shift.cpp:113-116 */
  int32_t am_lpf_f0;        /* cutoff the AM biquad was designed for at boot, T41_SDR.ino:560-566 (3000) */
  int32_t AGC_thresh;       /* bands[currentBand].AGC_thresh, SDT.h:190 (20 in every bands[] row) */
  int32_t nfm_demod;        /* 0 = nfmdemod() + limiter, what the firmware runs (Process.cpp:716-727); 1 = the variant
                               the source keeps commented out: fmdemod_atan_cf (Demod.cpp:368-392, ApproxAtan2
                               Demod.cpp:148-197) + limiter + deemphasis_nfm_ff (Demod.cpp:328-344, Process.cpp:734-735) */
  /* optional post-demodulation stages, Process.cpp:841-866 (oracle/t41_nr_oracle.c); fft_length 512 only */
  int32_t nrOptionSelect;   /* gwv.cpp:23: 0 off, 1 Kim1_NR() (x30), 2 SpectralNoiseReduction(), 3 Xanr() LMS NR (x1.5) */
  int32_t ANR_notchOn;      /* Process.cpp:45: 1 = Xanr() as automatic notch behind the noise reduction */
  float   NR_PSI;           /* gwv.cpp:61 (0.0) */
  float   NR_alpha;         /* gwv.cpp:62 (0.95) */
  float   NR_beta;          /* gwv.cpp:63 (0.85) */
} t41o_params;

/* what AGCPrep() + AGCLoadValues() (DSP_Fn.cpp:368-468) leave in the AGC globals, as f32 */
enum {
  T41O_AGC_ATTACK_MULT = 0, T41O_AGC_DECAY_MULT, T41O_AGC_FAST_DECAY_MULT, T41O_AGC_FAST_BACKMULT,
  T41O_AGC_ONEMFAST_BACKMULT, T41O_AGC_HANG_BACKMULT, T41O_AGC_ONEMHANG_BACKMULT,
  T41O_AGC_HANG_DECAY_MULT, T41O_AGC_OUT_TARGET, T41O_AGC_MIN_VOLTS, T41O_AGC_SLOPE_CONSTANT,
  T41O_AGC_INV_MAX_INPUT, T41O_AGC_HANG_LEVEL, T41O_AGC_POP_RATIO,
  T41O_AGC_HANG_COUNT,      /* (int)(hangtime * SampleRate / DF), DSP_Fn.cpp:550 */
  T41O_AGC_ATTACK_BUFFSIZE, /* (int)ceil(sample_rate * n_tau * tau_attack) = 97 (f32 product) */
  T41O_AGC_NCONST
};

/* Coefficients CalcFilters()/SetDecIntFilters()/InitFilterMask() produce. */
typedef struct {
  float dec1[T41O_N_DEC1_TAPS];
  float dec2[T41O_N_DEC2_TAPS];
  float int1[T41O_N_INT1_TAPS];
  float int2[T41O_N_INT2_TAPS];
  float biquad_lowpass1[5];
  float agc[T41O_AGC_NCONST]; /* zeros when AGCMode == 0 */
  float mask[2 * 4096];     /* FIR_filter_mask, 2*fft_length floats used */
} t41o_coeffs;

typedef struct t41o_channel t41o_channel; /* per-channel persistent state, opaque */

void t41o_default_params(t41o_params *p);

/* ---- coefficient design (FIR.cpp / Filter.cpp / Utility.cpp restatements) ---- */
float t41o_Izero(float x);
float t41o_MSinc(int m, float fc);
void  t41o_CalcFIRCoeffs(float *coeffs, int numCoeffs, float fc, float Astop, int type, float dfc,
                         float Fsamprate);
void  t41o_CalcCplxFIRCoeffs(float *cI, float *cQ, int numCoeffs, float FLoCut, float FHiCut,
                             float SampleRate);
void  t41o_SetIIRCoeffs(float coefficient_set[5], float f0, float Q, float sample_rate,
                        int filter_type);
/* full designer: what CalcFilters() (+ per-block SetDecIntFilters(nfmFilterBW) in NFM) leaves in
 * the coefficient arrays for these params */
int   t41o_design(const t41o_params *p, t41o_coeffs *c);

/* ---- CMSIS-DSP f32 primitive restatements ---- */
void t41o_cfft_f32(float *buf, int n, int ifft);                       /* arm_cfft_f32(S,buf,ifft,1) */
void t41o_fir_decimate_f32(const float *coeffs, int ntaps, int M, float *state,
                           const float *src, float *dst, int blockSize);
void t41o_fir_interpolate_f32(const float *coeffs, int ntaps, int L, float *state,
                              const float *src, float *dst, int blockSize);
void t41o_biquad_df2T_f32(const float coeffs[5], float state[2], const float *src, float *dst,
                          int n);
void t41o_biquad_df1_f32(const float coeffs[5], float state[4], const float *src, float *dst,
                         int n);

/* arm_sin_f32 / arm_cos_f32 as restated for AMDecodeSAM (Demod.cpp:75-76), their 513-entry table, and the PLL
 * constants of Demod.cpp:13-18 {omega_min, omega_max, g1, g2} */
float t41o_arm_sin_f32(float x);
float t41o_arm_cos_f32(float x);
const float *t41o_sin_table(void);
void t41o_sam_constants(float out[4]);

/* ---- noise reduction / automatic notch, Noise.cpp (oracle/t41_nr_oracle.c) ---- */
typedef struct t41o_nr t41o_nr;
t41o_nr *t41o_nr_create(void);
void t41o_nr_destroy(t41o_nr *s);
void t41o_nr_reset(t41o_nr *s); /* InitializeDataArrays() + SpectralNoiseReductionInit() + static initialisers */
/* Process.cpp:841-866 on one block of 256 audio samples @24 kS/s (float_buffer_L in and out, _R scratch) */
void t41o_nr_block(t41o_nr *s, const t41o_params *p, float *float_buffer_L, float *float_buffer_R);
int t41o_nr_peek(const t41o_nr *s, int which, float *dst, int maxlen);
/* 0 when SpectralNoiseReduction() would index outside its arrays for these cut-offs (Noise.cpp:556-574 with NN = 9) */
int t41o_nr_supported(const t41o_params *p);

/* ---- channel state + the block function ---- */
t41o_channel *t41o_channel_create(int fft_length);
void t41o_channel_destroy(t41o_channel *ch);
void t41o_channel_reset(t41o_channel *ch); /* power-on values (Osc_Vect_Q = 1, first_block = 1) */

/* One ProcessIQData() call (Process.cpp:70-944) for one channel:
 * I,Q: frame_len = 4*fft_length planar f32 samples in, audio: frame_len f32 samples out. */
int t41o_process_frame(t41o_channel *ch, const t41o_params *p, const t41o_coeffs *c,
                       long NCOFreq, const float *I, const float *Q, float *audio);

/* Batched convenience for tests/bench: channels are independent; runs nframes consecutive
 * frames of each channel.  Layout: I,Q,audio are [nchan][nframes*frame_len].
 * nthreads <= 1 runs serially. */
int t41o_process_batch(t41o_channel **chs, int nchan, int nframes, const t41o_params *p,
                       const t41o_coeffs *c, const int32_t *NCOFreq, const float *I,
                       const float *Q, float *audio, int nthreads);

/* The q15 boundary of ProcessIQData() (Process.cpp:102-111, 936-937): arm_q15_to_float /
 * arm_float_to_q15 restated, and the block function on the record/play queues' sample format
 * (I is read from the R queue, Q from the L queue).  Layouts as above with int16 samples. */
void t41o_q15_to_float(const int16_t *src, float *dst, int n);
void t41o_float_to_q15(const float *src, int16_t *dst, int n);
int t41o_process_frame_q15(t41o_channel *ch, const t41o_params *p, const t41o_coeffs *c,
                           long NCOFreq, const int16_t *Q_in_L, const int16_t *Q_in_R,
                           int16_t *Q_out_L);
int t41o_process_batch_q15(t41o_channel **chs, int nchan, int nframes, const t41o_params *p,
                           const t41o_coeffs *c, const int32_t *NCOFreq, const int16_t *Q_in_L,
                           const int16_t *Q_in_R, int16_t *Q_out_L);

/* debugging taps for stage-level parity tests: copies of intermediate buffers of the
 * last processed frame (lengths in floats); returns number of floats written */
enum {
  T41O_TAP_POST_NCO_I = 0, /* frame_len */
  T41O_TAP_POST_NCO_Q = 1,
  T41O_TAP_DEC_I = 2,      /* fft_length/2, after dec2 (+ level adjust in SSB/AM) */
  T41O_TAP_DEC_Q = 3,
  T41O_TAP_IFFT = 4,       /* 2*fft_length interleaved, after AGC */
  T41O_TAP_DEMOD = 5,      /* fft_length/2 audio before interpolation */
  T41O_TAP_AGC_VOLTS = 6,  /* fft_length/2: `volts` after every sample of the last AGC() call */
  T41O_TAP_AUDIO_SPECT = 7, /* 1024: audioSpectBuffer of the last frame (Process.cpp:550-553) */
  T41O_TAP_AUDIO_MAX = 8,  /* 3: audioMaxSquared, AudioMaxIndex, audioMaxSquaredAve (Process.cpp:569-570) */
  T41O_TAP_AGC_EDGES = 9,  /* 25: since reset, how often AGC state a was followed by b, [5*a + b] (test aid) */
  T41O_TAP_FFT_SPEC = 10,     /* 512: FFT_spec after CalcZoom1Magn() / ZoomFFTExe() of the last frame (FFT.cpp:67-251) */
  T41O_TAP_FFT_SPEC_OLD = 11  /* 512: FFT_spec_old (the display's low-pass memory) */
};
int t41o_channel_tap(const t41o_channel *ch, int which, float *dst, int maxlen);
t41o_nr *t41o_channel_nr(t41o_channel *ch); /* the channel's noise-reduction state (tests) */
/* the display FFT side output (FFT.cpp:67-251): spectrumZoom 0..4 = on (as with updateDisplayFlag == 1
 * in every frame), -1 = off (the default).  Like ZoomFFTPrep(): zoom filters re-initialised, zoom_sample_ptr = 0. */
int t41o_channel_set_display(t41o_channel *ch, int spectrumZoom);


/* ---- transmit exciter, ExciterIQData() (Exciter.cpp:46-169): oracle/t41_tx_oracle.c ---- */
typedef struct t41o_tx_channel t41o_tx_channel;
t41o_tx_channel *t41o_tx_create(void);
void t41o_tx_destroy(t41o_tx_channel *ch);
void t41o_tx_reset(t41o_tx_channel *ch);
/* one frame: 2048 q15 samples per input queue in, 2048 per output queue out; mode = T41O_DEMOD_* */
int t41o_tx_process_frame(t41o_tx_channel *ch, int mode, float IQXAmpCorrectionFactor, float IQXPhaseCorrectionFactor,
                          const int16_t *Q_in_L_Ex, const int16_t *Q_in_R_Ex, int16_t *Q_out_L_Ex, int16_t *Q_out_R_Ex);
/* the fixed tables FIR.cpp:177-276, 373-579: 0 coeffs192K_10K_LPF_FIR, 1 coeffs48K_8K_LPF_FIR, 2 / 3 FIR_Hilbert_coeffs_45 / _neg45 */
const float *t41o_tx_table(int which, int *n);

#ifdef __cplusplus
}
#endif
#endif

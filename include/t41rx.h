/*
 * include/t41rx.h -- C ABI of the MI355X-native T41 receive DSP hot path.
 *
 * Drop-in boundary for tmr4/T41_SDR's `void ProcessIQData();` (Process.h:15, body
 * Process.cpp:70-944).  The reference function takes no arguments: it reads/writes firmware
 * globals.  Each entry point below names the reference global(s)/function it stands in for
 * (file:line relative to /root/reference/software/T41_SDR/).  Plain pointers and sizes only;
 * no C++/torch types cross this boundary.
 *
 * Data model: a context owns `n_channels` independent receive channels (the reference has
 * exactly one: its static globals).  One `t41rx_process_*` call = one ProcessIQData() call on
 * every channel (or `n_frames` consecutive calls), with all per-channel persistent state
 * (FIR delay lines, NCO phase, overlap-save block, DC-block state ...) kept in device memory.
 *
 * Buffers (the reference's float_buffer_L / float_buffer_R, T41_SDR.ino:375-376, after the
 * arm_q15_to_float conversion of Process.cpp:107-108, i.e. float_buffer_L = I, _R = Q):
 *   I, Q   : const float [n_channels][n_frames * frame_len]   planar f32, read-only
 *   audio  : float       [n_channels][n_frames * frame_len]   mono f32 @192 kS/s
 *            (= float_buffer_L just before arm_float_to_q15, Process.cpp:929-936)
 *   frame_len = BUFFER_SIZE * N_BLOCKS = 4 * fft_length  (2048 for FFT_LENGTH 512;
 *   SDT.h:39,70, T41_SDR.ino:368).
 *
 * Errors: the reference returns nothing and hangs in while(1) on init failure
 * (T41_SDR.ino:574-616).  Every function here returns an int status (0 = OK, <0 = error) and
 * never aborts.  There is NO CPU fallback: if the HIP device or kernels are unavailable the
 * call fails with T41RX_ERR_HIP.
 */
#ifndef T41RX_H
#define T41RX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T41RX_ABI_VERSION 5

/* The library is built with -fvisibility=hidden and an export list (t41_sdr_amd/csrc/exports.map): the entry points
 * declared here and in t41tx.h are its only dynamic symbols. */
#ifndef T41RX_API
#if defined(__GNUC__) || defined(__clang__)
#define T41RX_API __attribute__((visibility("default")))
#else
#define T41RX_API
#endif
#endif

/* status codes */
#define T41RX_OK 0
#define T41RX_ERR_ARG (-1)        /* bad argument (null pointer, size, range) */
#define T41RX_ERR_UNSUPPORTED (-2)/* valid in the reference but not built here */
#define T41RX_ERR_HIP (-3)        /* HIP runtime / device error; see t41rx_last_error() */
#define T41RX_ERR_NOMEM (-4)
#define T41RX_ERR_STATE (-5)      /* blob / state size or version mismatch */

/* demodulation modes: bands[currentBand].mode, SDT.h:58-68 */
#define T41RX_DEMOD_USB 0
#define T41RX_DEMOD_LSB 1
#define T41RX_DEMOD_AM 2
#define T41RX_DEMOD_NFM 3
#define T41RX_DEMOD_SAM 8 /* synchronous AM, AMDecodeSAM() Demod.cpp:40-139 (SDT.h:67); fft_length 512 */

/* xmtMode, SDT.h:48-50 (only affects the CW side-tone offset of the NCO, Freq_Shift.cpp:108-120) */
#define T41RX_SSB_MODE 0
#define T41RX_CW_MODE 1
#define T41RX_DATA_MODE 2

/* The firmware globals ProcessIQData() reads, gathered into one POD (SURVEY 8b "Parameters"). */
typedef struct t41rx_params {
  int32_t fft_length;              /* FFT_LENGTH, SDT.h:39. 512 = reference; see t41rx_supported_fft_length */
  int32_t mode;                    /* bands[currentBand].mode: T41RX_DEMOD_USB / LSB / AM / NFM / SAM (SDT.h:58-68) */
  int32_t FLoCut;                  /* bands[currentBand].FLoCut [Hz], SDT.h:186 */
  int32_t FHiCut;                  /* bands[currentBand].FHiCut [Hz], SDT.h:185 */
  int32_t rfGainAllBands;          /* gwv.cpp:17, Process.cpp:117 */
  int32_t RFgain;                  /* bands[currentBand].RFgain, Process.cpp:133 */
  float   IQAmpCorrectionFactor;   /* IQAmpCorrectionFactor[currentBand], gwv.cpp:71 */
  float   IQPhaseCorrectionFactor; /* IQPhaseCorrectionFactor[currentBand], gwv.cpp:72 */
  int32_t AGCMode;                 /* gwv.cpp:15; 0 = off (fixed_gain 20, DSP_Fn.cpp:494-502), 1..4 = long/slow/
                                      med/fast look-ahead AGC (DSP_Fn.cpp:373-402, 504-631) */
  int32_t audioVolume;             /* gwv.cpp:16, Process.cpp:929 */
  int32_t nfmFilterBW;             /* Filter.cpp:16, Process.cpp:259 */
  int32_t xmtMode;                 /* gwv.cpp:22 */
  int32_t CWFreqShift;             /* Freq_Shift.cpp:113-116 */
  int32_t am_lpf_f0;               /* cutoff biquad_lowpass1 was designed for at boot, T41_SDR.ino:560-566 */
  int32_t AGC_thresh;              /* bands[currentBand].AGC_thresh [dB], SDT.h:190, DSP_Fn.cpp:408 */
  int32_t nfm_demod;               /* NFM discriminator: 0 = nfmdemod() (quadri-correlator) + limiter, what the
                                      firmware runs (Process.cpp:716-727); 1 = the alternative its source keeps
                                      commented out: fmdemod_atan_cf (Demod.cpp:368-392, ApproxAtan2 :148-197 with its
                                      2 pi for pi / 2 as written) + limiter + deemphasis_nfm_ff applied block-wise
                                      (Demod.cpp:324-344, Process.cpp:734-735).  fft_length 512 only. */
  /* The optional stages between the demodulator and the interpolators, Process.cpp:841-866 (all off in the firmware's
   * defaults).  fft_length 512; the functions are written for blocks of 256 audio samples. */
  int32_t nrOptionSelect;          /* gwv.cpp:23: 0 off; 1 Kim1_NR() (Noise.cpp:108-313) then x30; 2 SpectralNoiseReduction()
                                      (Noise.cpp:379-655); 3 Xanr() as LMS noise reduction (Noise.cpp:322-370) then x1.5 --
                                      as written the call site scales Xanr()'s INPUT, so 3 only changes the gain (and advances
                                      the adaptive filter the notch shares) */
  int32_t ANR_notchOn;             /* Process.cpp:45, :862-866: 1 = Xanr() as automatic notch behind the noise reduction */
  float   NR_PSI;                  /* gwv.cpp:61 (0.0): Kim1_NR()'s noise-floor switch */
  float   NR_alpha;                /* gwv.cpp:62 (0.95): time smoothing of the gains (Kim, spectral) */
  float   NR_beta;                 /* gwv.cpp:63 (0.85): Kim1_NR()'s smoothing over neighbouring bins */
} t41rx_params;

typedef struct t41rx_ctx t41rx_ctx; /* opaque: coefficient arrays + per-channel state + device buffers */

/* ---- library ---- */
T41RX_API int         t41rx_abi_version(void);
T41RX_API const char *t41rx_strerror(int status);
T41RX_API const char *t41rx_last_error(void);          /* thread-local detail of the last failing call */
T41RX_API int         t41rx_supported_fft_length(int fft_length); /* 1 for 512, 1024, 2048, 4096 */

/* Defaults of gwv.cpp:14-96 / bands[] T41_SDR.ino:145-168 (20 m row: USB, 200..3000 Hz) with
 * AGCMode forced to 0. */
T41RX_API void t41rx_default_params(t41rx_params *p);

/* ---- coefficient design: host-side, no GPU needed ----
 * The arrays CalcFilters() (Filter.cpp:235-249), InitFilterMask() (Filter.cpp:260-284),
 * SetDecIntFilters() (Filter.cpp:396-438) and InitializeDataArrays() (T41_SDR.ino:560-566) leave
 * behind, serialised as one blob:
 *   header (32 x int32: magic, abi, fft_length, mode, sizeof(t41rx_params), 3 reserved, then the
 *   t41rx_params the blob was designed for, padded to 24 words) |
 *   FIR_dec1_coeffs[28] | FIR_dec2_coeffs[46] | FIR_int1_coeffs[48] | FIR_int2_coeffs[32] |
 *   biquad_lowpass1_coeffs[5] | scalars[16] (gains, level adjust, volume, the SAM PLL constants) | AGC constants[16] (what AGCPrep() +
 *   AGCLoadValues() leave behind, DSP_Fn.cpp:368-468; zeros for AGCMode 0) |
 *   FIR_filter_mask[2*fft_length]                                                (all f32)
 * This blob is what rank 0 broadcasts over RCCL after a filter change. */
T41RX_API size_t t41rx_coeff_blob_bytes(int fft_length);
T41RX_API int    t41rx_design_coeffs(const t41rx_params *p, void *blob, size_t blob_bytes);

/* ---- context ---- */
/* InitializeDataArrays() (T41_SDR.ino:473-667): allocate state for n_channels channels on HIP
 * device `device_id`, power-on state, design + upload coefficients for *p. */
T41RX_API int t41rx_create(t41rx_ctx **out, int device_id, int n_channels, const t41rx_params *p);
T41RX_API int t41rx_destroy(t41rx_ctx *ctx);

/* SetupMode()/CalcFilters() (Filter.cpp:235-249, 341-385): parameters changed between two
 * ProcessIQData() calls.  Coefficients are re-designed and uploaded; like the reference, the
 * streaming state (FIR delay lines etc.) is NOT reset.  fft_length cannot change. */
T41RX_API int t41rx_set_params(t41rx_ctx *ctx, const t41rx_params *p);
T41RX_API int t41rx_get_params(const t41rx_ctx *ctx, t41rx_params *p);

/* Coefficient blob of the context (see t41rx_design_coeffs).  set = install a blob designed
 * elsewhere (e.g. received by broadcast); it must match the context's fft_length.  The context
 * takes over the parameters stored in the blob (mode, AGCMode, cut-offs, gains ...), so afterwards
 * t41rx_get_params() returns the designer's and a later t41rx_set_params() starts from them. */
T41RX_API int t41rx_get_coeffs(const t41rx_ctx *ctx, void *blob, size_t blob_bytes);
T41RX_API int t41rx_set_coeffs(t41rx_ctx *ctx, const void *blob, size_t blob_bytes);

/* NCOFreq (T41_SDR.ino:131, Tune.cpp:141-196), one value per channel, host array of n_channels.
 * Phase-continuous: the oscillator state (Osc_Vect_Q/I, Freq_Shift.cpp:13-14) is kept. */
T41RX_API int t41rx_set_nco_freq(t41rx_ctx *ctx, const int32_t *nco_freq_hz, int n);

/* Power-on state: Osc_Vect_Q = 1, Osc_Vect_I = 0, all delay lines zero, first_block = 1
 * (Freq_Shift.cpp:13-14, Process.cpp:42,47). */
T41RX_API int t41rx_reset(t41rx_ctx *ctx);

T41RX_API int t41rx_n_channels(const t41rx_ctx *ctx);
T41RX_API int t41rx_frame_len(const t41rx_ctx *ctx);

/* How the frames of one process call lie in I / Q / audio (f32 and q15 entry points alike).  The reference has one
 * channel and one frame per call (float_buffer_L/R[2048], T41_SDR.ino:375-376), so a batch of channels over several
 * frames has two natural shapes:
 *   T41RX_LAYOUT_CHANNEL_MAJOR (default)  [n_channels][n_frames * frame_len]: every channel's samples of the call
 *                                         contiguous in time (one long float_buffer per channel);
 *   T41RX_LAYOUT_TIME_MAJOR               [n_frames][n_channels][frame_len]: the [n_channels][frame_len] buffers of
 *                                         n_frames consecutive single-frame calls stacked as they arrive, one batch of
 *                                         frames every 10.67 ms (ABI 5).
 * Same arithmetic, same results, same state either way: only the addresses of (channel, frame) differ.  The side
 * outputs and stage taps keep their documented [n_channels][n_frames][...] shapes.  fft_length 512 only for the
 * time-major layout (T41RX_ERR_UNSUPPORTED otherwise; a later set_params to a long fft_length is refused likewise). */
#define T41RX_LAYOUT_CHANNEL_MAJOR 0
#define T41RX_LAYOUT_TIME_MAJOR 1
T41RX_API int t41rx_set_buffer_layout(t41rx_ctx *ctx, int layout);
T41RX_API int t41rx_get_buffer_layout(const t41rx_ctx *ctx);

/* ---- the hot path: ProcessIQData() on every channel ----
 * Device-pointer form: dI, dQ, dAudio are device pointers ([n_channels][n_frames*frame_len], or time-major:
 * t41rx_set_buffer_layout);
 * the kernel is enqueued on `hip_stream` (a hipStream_t, may be NULL = default stream) and the
 * call returns without synchronising. */
T41RX_API int t41rx_process_device(t41rx_ctx *ctx, const float *dI, const float *dQ, float *dAudio,
                         int n_frames, void *hip_stream);
/* Host-pointer form (the reference's calling convention: caller-owned host arrays): copies
 * in, runs the same kernel, copies out, synchronises. */
T41RX_API int t41rx_process_host(t41rx_ctx *ctx, const float *I, const float *Q, float *audio,
                       int n_frames);

/* The same call on the firmware's own sample format either side of the path: q15 samples as the
 * AudioRecordQueues deliver them and as the AudioPlayQueue takes them.  Restates
 *   arm_q15_to_float(Q_in_R.readBuffer(), &float_buffer_L[...]); arm_q15_to_float(Q_in_L.readBuffer(),
 *   &float_buffer_R[...])                     Process.cpp:107-108 (note the swap: I comes from the R queue)
 *   arm_float_to_q15(float_buffer_L, q15_buffer_LTemp, 2048); Q_out_L.play(...)    Process.cpp:936-937
 * with CMSIS-DSP's conversions (x / 32768; truncating, saturating (q15_t)__SSAT((q31_t)(x * 32768), 16)).
 * Layout [n_channels][n_frames*frame_len] int16, the 16 blocks of 128 of a frame back to back (or time-major,
 * t41rx_set_buffer_layout).  The side outputs and stage taps work here as on the f32 entry points (ABI 5): the
 * reference computes its display FFT and audio spectrum inside every ProcessIQData() call on exactly these q15-fed
 * buffers (Process.cpp:107-108 -> :184-186, :211-215, :550-570). */
T41RX_API int t41rx_process_device_q15(t41rx_ctx *ctx, const int16_t *dQ_in_L, const int16_t *dQ_in_R,
                             int16_t *dQ_out_L, int n_frames, void *hip_stream);
T41RX_API int t41rx_process_host_q15(t41rx_ctx *ctx, const int16_t *Q_in_L, const int16_t *Q_in_R,
                           int16_t *Q_out_L, int n_frames);

/* ---- checkpoint of the streaming state (the reference never persists it) ----
 * Layout (ABI 5): a 32-byte header of eight int32 words
 *     [0] magic "T41S"  [1] T41RX_ABI_VERSION  [2] fft_length  [3] n_channels  [4] floats per channel record
 *     [5] section mask (below)  [6] spectrumZoom of the display section (0 without one)  [7] reserved, 0
 * then the path's per-channel records (FIR delay lines, oscillator, overlap block, AGC, demodulator words), then the
 * sections header word 5 names, in this order:
 *     bit 0  noise reduction / notch: Xanr()'s taps, delay line and leak words, then the Kim / spectral per-bin memories
 *            (Noise.cpp:19-56) -- present once one of those stages has run in this context;
 *     bit 1  display FFT: zoom filters, ring, FFT_spec_old (FFT.cpp:14-26) -- present while
 *            t41rx_set_display_spectrum() is on.
 * t41rx_state_bytes() therefore GROWS when a noise-reduction stage first runs or the display spectrum is switched on:
 * query it right before every t41rx_get_state() (a buffer sized at creation gets T41RX_ERR_STATE "state buffer too
 * small").  fft_length cannot change on a live context, so the path records' size never does.
 * t41rx_set_state() refuses (T41RX_ERR_STATE) a checkpoint of another ABI, FFT length or channel count, one with an
 * unknown section bit or a size that does not follow from its header, a display section while the display spectrum is
 * off here or taken at another spectrumZoom, and any word the kernels use as an index, a divisor or a state number that
 * is out of range or not integral: AGC state words, oscillator amplitude, synchronous-detector PLL words (phase in
 * [0, 2 pi], frequency within +-pll_fmax), the notch's leak index, the noise reduction's ring pointers, the zoom ring's
 * pointer.  What it does to the side stages: a section the checkpoint carries is restored; a memory this context has
 * allocated but the checkpoint does not carry goes back to its power-on values (InitializeDataArrays() /
 * SpectralNoiseReductionInit() / ZoomFFTPrep()) -- never the values of the stream being replaced.
 * T41RX_ERR_STATE is also what t41rx_get_state(), t41rx_process_host() and t41rx_process_host_q15() -- the calls that
 * synchronise -- return if a wait inside the pipelined AGC / SAM kernels has run out since the last t41rx_reset() or
 * restored checkpoint (their waits are bounded so that a broken hand-over cannot hang the GPU; it cannot happen unless
 * the kernel is wrong, and then the samples are not to be trusted). */
T41RX_API size_t t41rx_state_bytes(const t41rx_ctx *ctx);
T41RX_API int    t41rx_get_state(t41rx_ctx *ctx, void *host_buf, size_t bytes);
T41RX_API int    t41rx_set_state(t41rx_ctx *ctx, const void *host_buf, size_t bytes);

/* ---- stage taps for parity debugging (device pointers, may each be NULL) ----
 * When set, the next process calls also write, per channel and frame:
 *   post_nco : [n_channels][n_frames*frame_len*2]  I/Q after FreqShift2 (planar: I then Q per frame)
 *   dec      : [n_channels][n_frames*fft_length]   I/Q after decimate-by-8 (+ level adjust)
 *   demod    : [n_channels][n_frames*fft_length/2] audio @24 kS/s before interpolation
 * max_frames = the n_frames the buffers are sized for: a process call with more frames is refused
 * (T41RX_ERR_ARG) instead of writing past them.  fft_length 512 only (T41RX_ERR_UNSUPPORTED). */
T41RX_API int t41rx_set_debug_taps(t41rx_ctx *ctx, float *d_post_nco, float *d_dec, float *d_demod, int max_frames);

/* ---- the path's display by-product: the audio spectrum and the S-meter's input ----
 * What ProcessIQData() leaves behind when updateDisplayFlag == 1 (Process.cpp:550-570; NFM:
 * :790-805): audioSpectBuffer[1023 - k] = iFFT_buffer[k]^2 over the 1024 floats of the masked
 * spectrum (squares of the individual re / im values, reversed), arm_max_f32 of it, and the
 * running average the S-meter reads (Display.cpp:980-985).  Device pointers, both NULL = off
 * (the default).  While set, every processed frame writes
 *   d_spect : [n_channels][n_frames][1024]  audioSpectBuffer
 *   d_max   : [n_channels][n_frames][3]     audioMaxSquared, (float)AudioMaxIndex, audioMaxSquaredAve
 * and updates the per-channel audioMaxSquaredAve.  The pixel mapping (audioYPixel) is display
 * code and stays with the caller.  fft_length 512; f32 and q15 entry points.  max_frames as above. */
T41RX_API int t41rx_set_audio_spectrum(t41rx_ctx *ctx, float *d_spect, float *d_max, int max_frames);

/* ---- the display FFT: what ShowSpectrum() draws from (FFT.cpp:67-251) ----
 * CalcZoom1Magn() (spectrumZoom = 0: Hann-windowed 512-point FFT of the frame's first 512 I/Q samples
 * after gains, DC high-pass and IQ correction, Process.cpp:185-187) or ZoomFFTExe() (spectrumZoom
 * 1..4 = 2x..16x: Fs/4 shift, 4-stage elliptic IIR mag_coeffs[zoom], 4-tap decimating FIR by 2^zoom,
 * 512-sample ring, window, FFT; Process.cpp:211-215), both up to FFT_spec[512] (squared magnitudes,
 * halves swapped so that DC sits at index 256) and the display's low-pass memory FFT_spec_old[512].
 * The pixel mapping behind them (log10f_fast, display scale, pixel offsets) stays with the caller.
 * Device pointers, both NULL = off (the default).  While set, every processed frame is treated as
 * one with updateDisplayFlag == 1 and writes
 *   d_spec     : [n_channels][n_frames][512]  FFT_spec
 *   d_spec_old : [n_channels][n_frames][512]  FFT_spec_old
 * Setting it (or changing spectrumZoom) starts from cleared zoom filters, ring and low-pass memory.
 * fft_length 512; f32 and q15 entry points; max_frames as for the stage taps. */
T41RX_API int t41rx_set_display_spectrum(t41rx_ctx *ctx, float *d_spec, float *d_spec_old, int spectrumZoom, int max_frames);

#ifdef __cplusplus
}
#endif
#endif /* T41RX_H */

/* include/t41tx.h -- C ABI of the MI355X-native T41 transmit exciter (the TX mirror of t41rx.h).
 *
 * Drop-in boundary for the reference's
 *     void ExciterIQData();        (software/T41_SDR/Exciter.cpp:46-169)
 * which turns one frame of microphone samples (16 blocks of 128 q15 samples per AudioRecordQueue,
 * Q_in_L_Ex / Q_in_R_Ex, 192 kS/s) into the I and Q drive of the SSB exciter (Q_out_L_Ex /
 * Q_out_R_Ex, q15, 192 kS/s): decimate by 4 (48 taps, coeffs192K_10K_LPF_FIR) and by 2 (24 taps,
 * coeffs48K_8K_LPF_FIR), two 100-tap Hilbert FIRs (+45 / -45 degrees), the TX IQ amplitude / phase
 * correction, interpolate by 2 (48 taps) and by 4 (32 taps) per channel, x 20, arm_float_to_q15.
 * The globals it reads become t41tx_params; its static CMSIS instance states (T41_SDR.ino:278-299)
 * become per-channel state owned by the context.  Citations: software/T41_SDR/ of the reference.
 * Not restated: the transmit equaliser (DoExciterEQ, xmitEQFlag = OFF) and the CW / data exciters.
 */
#ifndef T41TX_H
#define T41TX_H
#include <stddef.h>
#include <stdint.h>

#ifndef T41RX_API /* exported entry point (see t41rx.h) */
#if defined(__GNUC__) || defined(__clang__)
#define T41RX_API __attribute__((visibility("default")))
#else
#define T41RX_API
#endif
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* status codes and mode numbers are t41rx.h's (T41RX_OK, T41RX_ERR_*, T41RX_DEMOD_*) */

typedef struct t41tx_params {
  int32_t mode;                     /* bands[currentBand].mode: LSB / USB select the sign of the I scaling (Exciter.cpp:117-126) */
  float   IQXAmpCorrectionFactor;   /* IQXAmpCorrectionFactor[currentBandA], gwv.cpp:73 */
  float   IQXPhaseCorrectionFactor; /* IQXPhaseCorrectionFactor[currentBandA], gwv.cpp:74 */
} t41tx_params;

typedef struct t41tx_ctx t41tx_ctx;

T41RX_API void t41tx_default_params(t41tx_params *p);             /* USB, 1, 0 (gwv.cpp:73-74) */
/* SetupMode() for transmit: allocate n_channels exciters on HIP device device_id, filter states cleared */
T41RX_API int t41tx_create(t41tx_ctx **out, int device_id, int n_channels, const t41tx_params *p);
T41RX_API int t41tx_destroy(t41tx_ctx *ctx);
T41RX_API int t41tx_set_params(t41tx_ctx *ctx, const t41tx_params *p);  /* states are kept, like the firmware's */
T41RX_API int t41tx_reset(t41tx_ctx *ctx);                               /* all CMSIS instance states to zero */
T41RX_API int t41tx_n_channels(const t41tx_ctx *ctx);

/* ExciterIQData() on every channel, n_frames consecutive frames of 2048 samples per queue.
 * Device pointers, [n_channels][n_frames * 2048] int16 each; dQ_in_R_Ex may be NULL: the firmware
 * decimates that queue and then overwrites the result with a copy of the L channel
 * (Exciter.cpp:98), so its samples never reach the output.  Enqueued on hip_stream, no sync. */
T41RX_API int t41tx_process_device_q15(t41tx_ctx *ctx, const int16_t *dQ_in_L_Ex, const int16_t *dQ_in_R_Ex,
                             int16_t *dQ_out_L_Ex, int16_t *dQ_out_R_Ex, int n_frames, void *hip_stream);
/* host-pointer form: copies in, runs the same kernel, copies out, synchronises */
T41RX_API int t41tx_process_host_q15(t41tx_ctx *ctx, const int16_t *Q_in_L_Ex, const int16_t *Q_in_R_Ex,
                           int16_t *Q_out_L_Ex, int16_t *Q_out_R_Ex, int n_frames);

#ifdef __cplusplus
}
#endif
#endif /* T41TX_H */

// t41_sdr_amd/csrc/rx_kernels.hpp -- kernel argument block + launcher declaration (what rx_host.cpp sees of the kernels).
#pragma once
#include <hip/hip_runtime.h>

#include "rx_internal.hpp"

namespace t41 {

// constant table (float2 units), one per context, FFT_LENGTH = 512:
//   mask  [8][64] : FIR_filter_mask[lane + 64 r] / N
//   tw1   [7][64] : W512^(lane * q),        q = 1..7   (forward sign)
//   tw2   [7][64] : W64^((lane & 7) * q),   q = 1..7
//   sincos[256]   : (cos, sin)(2 pi i / 256)
//   hp8[64], hp4[64]: DC high-pass scan multipliers (a1^(n((l&15)+1)), a1^(n((l&31)+1))), n = 8, 4
constexpr int kTabMask = 0;
constexpr int kTabTw1 = 512;
constexpr int kTabTw2 = kTabTw1 + 7 * 64;
constexpr int kTabSinCos = kTabTw2 + 7 * 64;
constexpr int kTabHp8 = kTabSinCos + 256;
constexpr int kTabHp4 = kTabHp8 + 64;
//   am[64][6]     : per-lane constants of the AM demodulator's two wave scans (chunks of 4 samples
//                   per lane): the DC blocker's 0.99^(4((l&15)+1)) and 0.99^(4((l&31)+1)) as doubles
//                   (2 float2), and the biquad's transition-matrix powers (M^4)^((l&15)+1),
//                   (M^4)^((l&31)+1) as 2 x 2 float2 each
constexpr int kTabAm = kTabHp4 + 64;
//   sam[260]      : arm_sin_f32's table, 513 floats sin(2 pi k / 512) (+ padding), for the synchronous detector
constexpr int kTabSam = kTabAm + 64 * 6;
constexpr int kTabEntries512 = kTabSam + 260;

struct RxArgs {
  const float *__restrict__ I;
  const float *__restrict__ Q;
  float *__restrict__ out;
  float *__restrict__ state;
  const DevCoef *__restrict__ coef;
  const float2 *__restrict__ tab;
  const ChanNco *__restrict__ nco;
  int nchan;
  int nframes;
  // where (channel c, frame f) of I / Q / out starts, in samples: c * chan_stride + f * frame_stride (t41rx_set_buffer_layout)
  long long chan_stride;
  long long frame_stride;
  float *dbg_nco;
  float *dbg_dec;
  float *dbg_demod;
  float *dbg_pre;          // [nchan][nframes][2][2048]: I / Q after the IQ correction, before the Fs/4 shift (display FFT input)
  float *spect;            // audioSpectBuffer side output [nchan][nframes][1024] (or null)
  float *spect_max;        // [nchan][nframes][3]: audioMaxSquared, AudioMaxIndex, audioMaxSquaredAve
  // FFT_LENGTH 4096 pipeline scratch (device, owned by the context)
  float *mid;              // [nchan][nseg * 256] complex: /8-decimated, level-adjusted I/Q
  float *aud24;            // [nchan][nseg * 256] real: filtered audio @24 kS/s
  const float2 *tab4k;     // tw4096[7][512] | mask4096[8][512] (see kTab4k*)
  int nframes4k;           // number of long frames (= nframes / seg for the part kernels)
  int seg;                 // 2048-sample segments per frame = fft_length / 512 (1 for the fused 512 kernel)
  int plain;               // 1: unit band/IQ gains and zero IQ phase correction -> specialised kernel
  int agc;                 // 1: AGCMode != 0 (look-ahead AGC, DSP_Fn.cpp:504-631)
  // the gains of the first stage, by value: a kernel argument is one scalar-load round trip away
  // at kernel start, a field behind `coef` is two
  float g_rf;              // sc[kScRfGain]
  float g_band;            // sc[kScBandGain]
  float neg_iq_amp;        // sc[kScNegIqAmp]
  float iq_phase;          // sc[kScIqPhase]
  int iq_corr_on;          // sc[kScIqCorrOn] != 0
  int q15;                 // 1: I, Q, out point at int16 (q15) samples instead of f32 (Process.cpp:102-111, 936)
  int nfm_atan;            // NFM: 1 = atan2 discriminator + block-wise de-emphasis (t41rx_params::nfm_demod)
  int seg_run;             // long FFT, segment-parallel kernels: consecutive segments one wave runs
  int nco_rd;              // long FFT: which of the two NcoState copies holds the call's start state (the other is written)
  // noise reduction / notch pipeline (FFT_LENGTH 512, tap kernels): when set, the fused kernel stops behind the
  // demodulator and leaves the frame's 256 audio samples @24 kS/s here, [nchan][nframes * 256] in time order;
  // nr_kernels.hip then runs Process.cpp:841-866 on them in place and launch_back512() interpolates (aud24 = this)
  float *aud_out;
  // AGC on, pipelined kernel (rx_kernels.hip: agc_prep_pipe): per channel three slots of 1024 floats -- the serial
  // chain's operands and results and the samples the gain is applied to, for the frames in flight (or null: barrier form)
  float *agc_pipe;
};

// constant table of the N = 512 R point fast convolution (float2 units):
//   tw  [R-1][512] : W_N^(k' q), q = 1..R-1, k' < 512 (forward sign)
//   mask[R][512]   : FIR_filter_mask[q + R m] / N at [q][m]
constexpr int tab_long_entries(int R) { return (R - 1) * 512 + R * 512; }

hipError_t launch_rx(const RxArgs &a, int fft_length, int mode, hipStream_t s);
// what the kernel translation units of this library were built as (rx_experiments.hpp; rx_dispatch.hip): 0 = the product,
// bit 0 = a timing experiment with WRONG RESULTS by construction, bit 1 = a diagnostic build (stamps / counters)
int kernel_build_flags();
// FFT_LENGTH 512, behind launch_rx() with aud_out set: interpolators, volume and stores from a.aud24 (f32 or q15 samples out)
hipError_t launch_back512(const RxArgs &a, hipStream_t s);

// display FFT (CalcZoom1Magn / ZoomFFTExe, FFT.cpp:67-251) on the dbg_pre tap of the frames just processed
struct DispArgs {
  const float *pre;        // RxArgs::dbg_pre
  float *disp;             // [nchan][kDispFloats] display state
  float *spec;             // [nchan][nframes][512] FFT_spec
  float *spec_old;         // [nchan][nframes][512] FFT_spec_old
  const float2 *tab;       // the context's constant table (FFT twiddles)
  const double *win;       // [512]: 0.5 - 0.5 cos(6.28 i / 512), as the reference's double expression
  float iir[20];           // mag_coeffs[spectrumZoom] (unused for zoom 0)
  float fir[4];            // Fir_Zoom_FFT_Decimate_coeffs
  int nchan, nframes, zoom;
};
hipError_t launch_display(const DispArgs &a, hipStream_t s);

}  // namespace t41

// t41_sdr_amd/csrc/nr_kernels.hpp -- argument block, state layouts and launcher of the noise-reduction / notch kernels
// (nr_kernels.hip; Process.cpp:841-866, Noise.cpp).  Product code: nothing from oracle/.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace t41 {

// ---- Xanr() state, CHANNEL-MINOR ([row][channel]: one lane per channel reads and writes coalesced) ----
constexpr int kAnrTaps = 64;                      // ANR_taps, Noise.cpp:46
constexpr int kAnrDelay = 16;                     // ANR_delay, Noise.cpp:41
constexpr int kAnrHist = kAnrTaps + kAnrDelay - 1;  // 79: the live part of the 512-entry delay line ANR_d
constexpr int kAnrStW = 0;                        // ANR_w[0..63]
constexpr int kAnrStHist = kAnrStW + kAnrTaps;    // the last 79 samples written to ANR_d, oldest first
constexpr int kAnrStLidx = kAnrStHist + kAnrHist; // ANR_lidx (120.0 at power-on)
constexpr int kAnrStNgamma = kAnrStLidx + 1;      // ANR_ngamma (0.001 at power-on)
constexpr int kAnrStRows = kAnrStNgamma + 1;

// ---- Kim1_NR() / SpectralNoiseReduction() state, one contiguous record per channel (floats) ----
// bins i = 0 .. 127 (NR_FFT_L / 2).  Power-on values: InitializeDataArrays() + SpectralNoiseReductionInit()
// (T41_SDR.ino:479-504, Noise.cpp:692-707) -- nr_reset_record().
constexpr int kNrX = 0;                 // NR_X[i][j] as [j][i], j < 3
constexpr int kNrE = kNrX + 3 * 128;    // NR_E[i][j] as [j][i], j < 15
constexpr int kNrGts1 = kNrE + 15 * 128;  // NR_Gts[i][1]
constexpr int kNrGts0 = kNrGts1 + 128;  // NR_Gts[i][0] (bins outside the pass band keep what they had)
constexpr int kNrG = kNrGts0 + 128;     // NR_G
constexpr int kNrLastIn = kNrG + 128;   // NR_last_sample_buffer_L
constexpr int kNrLastOut = kNrLastIn + 128;  // NR_last_iFFT_result
constexpr int kNrNest = kNrLastOut + 128;    // NR_Nest[i][0]
constexpr int kNrPslp = kNrNest + 128;  // pslp (SpectralNoiseReduction()'s static)
constexpr int kNrXt = kNrPslp + 128;    // xt
constexpr int kNrHk = kNrXt + 128;      // NR_Hk_old
constexpr int kNrScal = kNrHk + 128;    // NR_X_pointer, NR_E_pointer, NR_first_time_2, NR_init_counter (as floats) + pad
constexpr int kNrSpecFloats = kNrScal + 8;

// window tables (host-made: the Hann expression of Noise.cpp:188-191 with the host's cosf, and the reference's sqrtHann[] literals)
constexpr int kNrTabHann = 0, kNrTabSqrtHann = 256, kNrTabFloats = 512;
extern const float kSqrtHann[256];  // Noise.cpp:49-83, a fixed table of the reference (nr_tables.cpp)
void nr_make_tables(float (&tab)[kNrTabFloats]);
void nr_reset_record(float *rec);                          // one channel's kNrSpecFloats
void nr_reset_anr(float *anr, size_t nchan);               // kAnrStRows x nchan

struct NrArgs {
  float *aud;            // [nchan][nframes * 256] demodulated audio @24 kS/s, processed in place
  float *anr;            // Xanr() state
  float *spec;           // Kim / spectral state
  const float *tab_nr;   // window tables
  const float2 *tab;     // the context's constant table (FFT twiddles, rx_kernels.hpp)
  int nchan, nframes;
  int nr_option;         // nrOptionSelect
  int notch;             // ANR_notchOn
  float alpha, beta, psi;  // NR_alpha, NR_beta, NR_PSI
  int vad_lo, vad_hi;    // design.cpp: nr_vad_range()
};
hipError_t launch_nr(const NrArgs &a, hipStream_t s);

}  // namespace t41

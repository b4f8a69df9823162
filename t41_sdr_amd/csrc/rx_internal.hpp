// t41_sdr_amd/csrc/rx_internal.hpp -- layouts shared by the host side and the HIP kernels.
// Product code: must not include or link anything from oracle/.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/t41rx.h"

namespace t41 {

constexpr int kSampleRate = 192000;  // T41_SDR.ino:129
constexpr int kDec1Taps = 28;        // n_dec1_taps, T41_SDR.ino:344
constexpr int kDec2Taps = 46;        // n_dec2_taps, T41_SDR.ino:345
constexpr int kInt1Taps = 48;        // T41_SDR.ino:595-603 (L=2, phase length 24)
constexpr int kInt2Taps = 32;        // T41_SDR.ino:608-616 (L=4, phase length 8)

// ---- coefficient blob (host, canonical reference layout; what RCCL broadcasts) ----
constexpr uint32_t kBlobMagic = 0x54343152u;  // "T41R"
// header: magic, abi, fft_length, mode, sizeof(t41rx_params), 3 reserved | the t41rx_params the blob
// was designed for, padded to 24 words.  A context that installs the blob (t41rx_set_coeffs: the
// broadcast path) takes its parameters from here, so every rank ends up with the designer's.
constexpr int kBlobParamInts = 24;
constexpr int kBlobHeaderInts = 8 + kBlobParamInts;
static_assert(sizeof(t41rx_params) <= kBlobParamInts * sizeof(int32_t), "params section of the blob header");
constexpr int kNumScalars = 16;
enum Scalar {
  kScRfGain = 0,      // (float)pow(10, rfGainAllBands/20), Process.cpp:117
  kScBandGain = 1,    // (float)bands[].RFgain, Process.cpp:133
  kScNegIqAmp = 2,    // -IQAmpCorrectionFactor, Process.cpp:166
  kScIqPhase = 3,     // IQPhaseCorrectionFactor, Utility.cpp:178
  kScLevel = 4,       // volScaleFactor, Process.cpp:490
  kScFixedGain = 5,   // fixed_gain, DSP_Fn.cpp:453
  kScOutScale = 6,    // DF * VolumeToAmplification(audioVolume), Process.cpp:929
  kScIqCorrOn = 7,    // 1 when mode in {USB, LSB, AM}, Process.cpp:165-173
  kScSideTone = 8,    // sideToneShift [Hz], Freq_Shift.cpp:108-120
  kScNfmDemod = 9,    // t41rx_params::nfm_demod (0 quadri-correlator, 1 atan2 + de-emphasis)
  kScSamWmin = 10,    // SAM PLL: omega_min, omega_max, g1, g2 (Demod.cpp:15-18)
  kScSamWmax = 11,
  kScSamG1 = 12,
  kScSamG2 = 13,
};
// deemphasis_nfm_predefined_fir_24000 (Demod.cpp:324-325), a fixed table of the reference
constexpr int kDeemphTaps = 81;
extern const float kDeemphFir24000[kDeemphTaps];
// what AGCPrep() + AGCLoadValues() (DSP_Fn.cpp:368-468) leave in the AGC globals
constexpr int kNumAgc = 16;
enum AgcConst {
  kAgcAttackMult = 0, kAgcDecayMult, kAgcFastDecayMult, kAgcFastBackmult, kAgcOnemFastBackmult,
  kAgcHangBackmult, kAgcOnemHangBackmult, kAgcHangDecayMult, kAgcOutTarget, kAgcMinVolts,
  kAgcSlopeConstant, kAgcInvMaxInput, kAgcHangLevel, kAgcPopRatio,
  kAgcHangCount,       // (int)(hangtime * SampleRate / DF), DSP_Fn.cpp:550
  kAgcAttackBuffsize,  // (int)ceil(sample_rate * n_tau * tau_attack), DSP_Fn.cpp:409 (= 97)
};
struct BlobView {
  int32_t *header;
  float *dec1, *dec2, *int1, *int2, *lp1, *scalars, *agc, *mask;
};
constexpr size_t blob_floats(int fft_length) {
  return kBlobHeaderInts + kDec1Taps + kDec2Taps + kInt1Taps + kInt2Taps + 5 + kNumScalars + kNumAgc +
         2 * (size_t)fft_length;
}
inline BlobView blob_view(void *blob) {
  BlobView v;
  v.header = reinterpret_cast<int32_t *>(blob);
  float *f = reinterpret_cast<float *>(blob) + kBlobHeaderInts;
  v.dec1 = f;
  v.dec2 = v.dec1 + kDec1Taps;
  v.int1 = v.dec2 + kDec2Taps;
  v.int2 = v.int1 + kInt1Taps;
  v.lp1 = v.int2 + kInt2Taps;
  v.scalars = v.lp1 + 5;
  v.agc = v.scalars + kNumScalars;
  v.mask = v.agc + kNumAgc;
  return v;
}

// ---- display FFT side output (FFT.cpp:28-251) ----
extern const float kZoomIirCoeffs[4][20];
void design_zoom_fir(int spectrumZoom, float (&coeffs)[4]);
// per-channel display state (floats): the zoom filters' memories, the 512-sample ring and its
// pointer, FFT_spec_old
constexpr int kDispIir = 0;       // [2][16]: IIR_biquad_Zoom_FFT_{I,Q}_state (x1, x2, y1, y2 per stage)
constexpr int kDispFir = 32;      // [2][4]: the decimating FIR's last 3 inputs (+ pad)
constexpr int kDispPtr = 40;      // zoom_sample_ptr (int32)
constexpr int kDispRing = 64;     // [2][512]: FFT_ring_buffer_x / _y
constexpr int kDispOld = 64 + 1024;  // [512]: FFT_spec_old
constexpr int kDispFloats = kDispOld + 512;

// host designer (design.cpp)
int design_blob(const t41rx_params &p, void *blob, size_t blob_bytes);
bool params_valid(const t41rx_params &p, const char **why);
void nr_vad_range(int FLoCut, int FHiCut, int *lo, int *hi);  // Kim / spectral NR: bins [lo, hi) the filter passes

// ---- device-side constant block (one per context) ----
// scalar-loadable coefficient struct, uniform for every wave
struct DevCoef {
  float dec1[kDec1Taps];
  float dec2[48];  // 46 used; FIR_dec2_coeffs x the level adjust sc[kScLevel] (rx_host.cpp: upload_coeffs)
  float int1[kInt1Taps];
  float int2[kInt2Taps];  // FIR_int2_coeffs x the volume factor sc[kScOutScale] (rx_host.cpp: upload_coeffs)
  float lp1[8];    // 5 used
  float sc[16];    // Scalar enum
  float agc[16];   // AgcConst enum
  float deemph[96];  // kDeemphFir24000, zero padded (scalar loads come in chunks of 16)
};

// per-channel NCO constants, recomputed on the host when NCOFreq changes (set_nco_freq)
struct ChanNco {
  uint64_t phase_inc;  // rotation per sample in turns, 0.64 fixed point
  float wk[8][2];      // 1.1 * A * e^{j 2 pi k dphi}, k = 0..7 (A = steady |Osc|)
  double r_star_sq;    // steady-state |Osc_Vect|^2 (fixed point of the amplitude loop)
  double w_abs;        // |cos + j sin| of the rotation constant
};
static_assert(sizeof(ChanNco) == 8 + 64 + 16, "ChanNco layout");

// ---- per-channel streaming state in HBM (floats), one contiguous record per channel ----
// Offsets in floats; every section 16-byte aligned.  "pad" entries keep the delay lines
// float4-aligned with the newest history sample adjacent to the first new sample.
constexpr int kStDec1 = 0;      // 28 complex (I,Q interleaved): [0] pad, [1..27] = last 27 post-NCO samples
constexpr int kStDec2 = 56;     // 48 complex: [0..2] pad, [3..47] = last 45 /4-decimator outputs
constexpr int kStInt1 = 152;    // 24: [0] pad, [1..23] = last 23 demodulated samples
constexpr int kStInt2 = 176;    // 8:  [0] pad, [1..7]  = last 7 int1 outputs
constexpr int kStMisc = 184;    // 16 floats: see below
constexpr int kMiscDc = 0;      // DC-HP d1 (HP_DC_Butter_state2[0]); [1] = d2 (always 0)
constexpr int kMiscWold = 2;    // AM DC-block wold, Process.cpp:73
constexpr int kMiscNfmI = 4;    // nfmdemod last_sample_i/q, Demod.cpp:221-222
constexpr int kMiscNfmQ = 5;
constexpr int kMiscNfmPhase = 6; // fmdemod_atan_cf's last_phase (nfm_demod = 1), Demod.cpp:373
constexpr int kMiscLp1 = 8;     // biquad_lowpass1_state[4]
constexpr int kMiscMaxSqAve = 12; // audioMaxSquaredAve, Process.cpp:570
constexpr int kMiscSamPhz = 13;   // AMDecodeSAM's PLL statics phzerror, fil_out, omega2, Demod.cpp:19-23
constexpr int kMiscSamFil = 14;
constexpr int kMiscSamOmega = 15;
constexpr int kStNco = 200;     // 8 floats = two NcoState (16 B each); FFT_LENGTH 512 uses the first, the long
                                // FFT lengths alternate between them from call to call (rx_kernels.hip)
constexpr int kStOverlap = 256; // fft_length floats: last_sample_buffer_L/R as [k][lane] (re,im)
struct NcoState {
  uint64_t phase;  // arg(Osc_Vect_Q + j Osc_Vect_I) in turns, 0.64 fixed point
  double r;        // |Osc_Vect|
};
// AGC (DSP_Fn.cpp:504-631), after the overlap block: the look-ahead delay line and the statics.
// The reference's ring holds 1921 entries of which attack_buffsize = 97 are live; 100 are kept
// (16-byte alignment) as interleaved (re, im), oldest first.
constexpr int kAgcDelay = 97;   // attack_buffsize the kernel is built for
constexpr int kAgcHist = 100;   // complex samples kept
constexpr int kAgcHistFloats = 2 * kAgcHist;
// 8 words behind the history: fast_backaverage, hang_backaverage, volts, save_volts (f32),
// state, decay_type, hang_counter (int32), pad
constexpr int kAgcScalars = 8;
constexpr int kAgcStFba = 0, kAgcStHba = 1, kAgcStVolts = 2, kAgcStSave = 3, kAgcStState = 4, kAgcStDecayType = 5,
              kAgcStHangCounter = 6;
constexpr size_t st_agc(int fft_length) { return (size_t)kStOverlap + (size_t)fft_length; }
constexpr size_t state_floats(int fft_length) { return st_agc(fft_length) + kAgcHistFloats + kAgcScalars; }

}  // namespace t41

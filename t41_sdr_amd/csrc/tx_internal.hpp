// t41_sdr_amd/csrc/tx_internal.hpp -- layouts shared by the host side and the HIP kernel of the
// transmit exciter.  Product code: must not include or link anything from oracle/.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/t41rx.h"
#include "../../include/t41tx.h"

namespace t41 {

extern const float kTx192k10k[48], kTx48k8k[48], kTxHilbert45[100], kTxHilbertNeg45[100];

// device coefficient block (scalar-loadable)
struct TxCoef {
  float c192[48];   // /4 decimator (48 taps) and x4 interpolator (first 32)
  float c48[48];    // /2 decimator (first 24) and x2 interpolator (48)
  float h45[100];   // FIR_Hilbert_L
  float hn45[100];  // FIR_Hilbert_R
};

// per-channel state (floats): the CMSIS instance states' history parts, T41_SDR.ino:278-299
constexpr int kTxStDec1 = 0;     // 47 (+1 pad): last 47 input samples @192 kS/s
constexpr int kTxStDec2 = 48;    // 23 (+1): last 23 /4 outputs
constexpr int kTxStHilL = 72;    // 99 (+1): last 99 samples @24 kS/s
constexpr int kTxStHilR = 172;   // 99 (+1)
constexpr int kTxStInt1I = 272;  // 23 (+1)
constexpr int kTxStInt1Q = 296;  // 23 (+1)
constexpr int kTxStInt2I = 320;  // 7 (+1)
constexpr int kTxStInt2Q = 328;  // 7 (+1)
constexpr int kTxStateFloats = 336;

struct TxArgs {
  const int16_t *__restrict__ inL;
  int16_t *__restrict__ outL;
  int16_t *__restrict__ outR;
  float *__restrict__ state;
  const TxCoef *__restrict__ coef;
  int nchan, nframes;
  float i_scale;   // +IQXAmp (LSB) / -IQXAmp (USB), Exciter.cpp:117-126
  float iq_phase;  // IQXPhaseCorrectionFactor
  int corr_on;     // LSB or USB
};

hipError_t launch_tx(const TxArgs &a, hipStream_t s);

}  // namespace t41

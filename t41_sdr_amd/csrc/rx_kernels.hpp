// t41_sdr_amd/csrc/rx_kernels.hpp -- kernel argument block + launcher declaration.
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
constexpr int kTabEntries512 = kTabHp4 + 64;

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
  float *dbg_nco;
  float *dbg_dec;
  float *dbg_demod;
};

hipError_t launch_rx(const RxArgs &a, int fft_length, int mode, hipStream_t s);

}  // namespace t41

// t41_sdr_amd/csrc/rx512_launch.hpp -- picks the rx512_kernel instantiation for a call (FFT_LENGTH 512); included by the
// one translation unit per demodulator family that instantiates them.
#pragma once
#include "rx512_kernel.hpp"
#include "rx_launch.hpp"

namespace t41 {

// (T41RX_AGC_PIPE=0 in the environment: the barrier form for calls of any length -- the tests that compare the two forms)
static inline bool agc_pipe_env() {
  static const bool on = [] {
    const char *e = std::getenv("T41RX_AGC_PIPE");
    return !e || std::atoi(e) != 0;
  }();
  return on;
}

template <int MODE>
static hipError_t launch512(const RxArgs &a, hipStream_t s, bool debug) {
  // One 16-wave workgroup per CU (all 160 KiB of LDS, declared statically by the kernel): a
  // 4096-channel batch is one full, balanced wave of work on 256 CUs, and every wave keeps its
  // channel for all the frames of the launch.
  // (AGC on: 4-wave workgroups, see Geo)
#define T41RX_GO(DBG, PLN, AGCv, Q15v)                                                                   \
  hipLaunchKernelGGL((rx512_kernel<MODE, DBG, 0, PLN, AGCv, Q15v>),                                      \
                     dim3((a.nchan + Geo<0, AGCv>::kWaves - 1) / Geo<0, AGCv>::kWaves), dim3(Geo<0, AGCv>::kWaves * 64), 0, s, a)
  // AGC on, calls of four frames or more without taps: the pipelined variant (agc_prep_pipe); shorter calls have
  // nothing to overlap and take the barrier form, which computes the same values (T41RX_AGC_PIPE=0: experiments, tests)
  const bool pipe_env = agc_pipe_env();
  if constexpr (MODE != kModeSam) {
    if (a.agc && a.agc_pipe && !debug && a.nframes >= 4 && pipe_env) {
#define T41RX_GOP(PLN, Q15v)                                                                             \
  hipLaunchKernelGGL((rx512_kernel<MODE, false, 0, PLN, true, Q15v, false, true>), dim3((a.nchan + Geo<0>::kWaves - 1) / Geo<0>::kWaves), \
                     dim3(Geo<0>::kWaves * 64), 0, s, a)
      if (a.q15) {
        if (a.plain) T41RX_GOP(true, true); else T41RX_GOP(false, true);
      } else {
        if (a.plain) T41RX_GOP(true, false); else T41RX_GOP(false, false);
      }
#undef T41RX_GOP
      return hipGetLastError();
    }
  }
  if (a.q15 && debug) {  // q15 samples either side with the side outputs / stage taps (round 4; general front end)
    if (a.agc)
      T41RX_GO(true, false, true, true);
    else
      T41RX_GO(true, false, false, true);
  } else if (a.q15) {  // the firmware's q15 sample format either side
    if (a.agc) {
      if (a.plain)
        T41RX_GO(false, true, true, true);
      else
        T41RX_GO(false, false, true, true);
    } else if (a.plain)
      T41RX_GO(false, true, false, true);
    else
      T41RX_GO(false, false, false, true);
  } else if (a.agc) {
    if (debug)
      T41RX_GO(true, false, true, false);
    else if (a.plain)
      T41RX_GO(false, true, true, false);
    else
      T41RX_GO(false, false, true, false);
  } else if (debug)
    T41RX_GO(true, false, false, false);
  else if (a.plain)
    T41RX_GO(false, true, false, false);
  else
    T41RX_GO(false, false, false, false);
#undef T41RX_GO
  return hipGetLastError();
}

}  // namespace t41

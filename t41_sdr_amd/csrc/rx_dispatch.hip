// t41_sdr_amd/csrc/rx_dispatch.hip -- launch_rx(): which kernel family runs a call.  No kernel is instantiated here.
#include <cstdlib>

#include "rx_experiments.hpp"
#include "rx_launch.hpp"
#include "t41rx.h"

namespace t41 {

// what this library's kernel translation units were built as (rx_experiments.hpp): 0 = the product
int kernel_build_flags() { return kKernelBuildFlags; }

// FFT_LENGTH 512 R (R = 2, 4, 8): front half (R segments per frame) -> N-point fast convolution -> back half
static hipError_t launch_long(const RxArgs &a, int mode, hipStream_t s) {
  // FFT_LENGTH 4096, SSB audio with the fixed gain, f32 samples: the whole chain in one kernel (T41RX_FUSE_FRONT=0
  // or an explicit T41RX_SEG_RUN select the two-kernel pipeline: experiments, and the tests that compare the two)
  static const bool fuse_front_env = [] {
    const char *e = std::getenv("T41RX_FUSE_FRONT");
    return !e || std::atoi(e) != 0;
  }();
  if (a.seg == 8 && mode != T41RX_DEMOD_NFM && mode != T41RX_DEMOD_AM && !a.agc && !a.q15 && fuse_front_env && !std::getenv("T41RX_SEG_RUN")) {
    return launch_fastconv_fused(a, s);
  }
  hipError_t e = launch_long_front(a, mode, s);
  if (e != hipSuccess) return e;
  const bool cplx = a.agc || mode == T41RX_DEMOD_AM;
  // real audio with the fixed gain, f32 samples out: the interpolators run behind pass 3 of the
  // fast convolution (no `aud24` round trip, no third kernel)
  static const bool fuse_env = [] { const char *e = std::getenv("T41RX_FUSE_BACK"); return !e || std::atoi(e) != 0; }();  // experiments
  const bool fused = !cplx && !a.q15 && fuse_env;
  e = launch_fastconv(a, cplx, fused, s);
  if (e != hipSuccess || fused) return e;
  return launch_long_back(a, mode, s);
}

hipError_t launch_rx(const RxArgs &a, int fft_length, int mode, hipStream_t s) {
  const bool debug = a.dbg_nco || a.dbg_dec || a.dbg_demod || a.spect || a.dbg_pre || a.aud_out;  // side outputs (and the NR hand-over) ride on the tap kernels
  if (fft_length == 1024 || fft_length == 2048 || fft_length == 4096) {
    if (a.seg * 512 != fft_length) return hipErrorInvalidValue;
    return launch_long(a, mode, s);
  }
  if (fft_length != 512) return hipErrorInvalidValue;
  switch (mode) {
    case T41RX_DEMOD_USB:
    case T41RX_DEMOD_LSB:
      return launch512_ssb(a, s, debug);
    case T41RX_DEMOD_AM:
      return launch512_am(a, s, debug);
    case T41RX_DEMOD_NFM:
      return launch512_nfm(a, s, debug);
    case T41RX_DEMOD_SAM:
      return launch512_sam(a, s, debug);
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace t41

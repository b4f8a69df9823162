// t41_sdr_amd/csrc/rx_long.hip -- the two ends of the long-FFT pipeline (FFT_LENGTH 1024 / 2048 / 4096): rx512_kernel<..., PART = 1>
// (loads .. /8 decimation -> `mid`) and <..., PART = 2> (`aud24` -> interpolators -> stores), a 2048-sample segment at a time;
// the N-point fast convolution between them is fastconv.hip.  launch_back512: the PART 2 kernel behind the noise-reduction
// stages at FFT_LENGTH 512.
#include "rx512_kernel.hpp"
#include "rx_launch.hpp"

namespace t41 {

hipError_t launch_long_front(const RxArgs &a, int mode, hipStream_t s) {
  const int grid = (a.nchan + 3) / 4;
  // one wave per (channel, segment) wherever the segments can run independently (SEGPAR)
  const int grid_par = (int)(((size_t)a.nchan * (size_t)((a.nframes + a.seg_run - 1) / a.seg_run) + 3) / 4);
  if (mode == T41RX_DEMOD_NFM) {  // nfmdemod()'s "last sample" chains the frames: sequential
    if (a.q15)
      hipLaunchKernelGGL((rx512_kernel<kModeNfm, false, 1, false, false, true>), dim3(grid), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((rx512_kernel<kModeNfm, false, 1, false>), dim3(grid), dim3(256), 0, s, a);
  } else if (a.q15) {
    hipLaunchKernelGGL((rx512_kernel<kModeSsb, false, 1, false, false, true, true>), dim3(grid_par), dim3(256), 0, s, a);
  } else if (a.plain) {  // unit band / IQ gains (the firmware defaults): the correction stage drops out
    hipLaunchKernelGGL((rx512_kernel<kModeSsb, false, 1, true, false, false, true>), dim3(grid_par), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((rx512_kernel<kModeSsb, false, 1, false, false, false, true>), dim3(grid_par), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

hipError_t launch_long_back(const RxArgs &a, int mode, hipStream_t s) {
  const int grid = (a.nchan + 3) / 4;
  const int grid_par = (int)(((size_t)a.nchan * (size_t)((a.nframes + a.seg_run - 1) / a.seg_run) + 3) / 4);
#define T41RX_BACK(MODEv, AGCv)                                                                                   \
  do {                                                                                                            \
    if (a.q15)                                                                                                    \
      hipLaunchKernelGGL((rx512_kernel<MODEv, false, 2, false, AGCv, true>), dim3(grid), dim3(256), 0, s, a); \
    else                                                                                                          \
      hipLaunchKernelGGL((rx512_kernel<MODEv, false, 2, false, AGCv, false>), dim3(grid), dim3(256), 0, s, a); \
  } while (0)
#define T41RX_BACK_PAR()                                                                                            \
  do {                                                                                                              \
    if (a.q15)                                                                                                      \
      hipLaunchKernelGGL((rx512_kernel<kModeSsb, false, 2, false, false, true, true>), dim3(grid_par), dim3(256), 0, s, a); \
    else                                                                                                            \
      hipLaunchKernelGGL((rx512_kernel<kModeSsb, false, 2, false, false, false, true>), dim3(grid_par), dim3(256), 0, s, a); \
  } while (0)
  if (mode == T41RX_DEMOD_AM) {
    if (a.agc)
      T41RX_BACK(kModeAm, true);
    else
      T41RX_BACK(kModeAm, false);
  } else if (a.agc) {
    T41RX_BACK(kModeSsb, true);
  } else {  // SSB / NFM audio with the fixed gain: the gain law and the demodulators keep no state here
    T41RX_BACK_PAR();
  }
#undef T41RX_BACK
#undef T41RX_BACK_PAR
  return hipGetLastError();
}

hipError_t launch_back512(const RxArgs &a, hipStream_t s) {
  // the long-FFT pipeline's segment-parallel back kernel with one segment per frame: it takes its interpolator
  // memories from the channel's record and leaves the call's last ones there
  if (a.seg != 1 || !a.aud24) return hipErrorInvalidValue;
  const int grid_par = (int)(((size_t)a.nchan * (size_t)((a.nframes + a.seg_run - 1) / a.seg_run) + 3) / 4);
  if (a.q15)  // arm_float_to_q15 behind the volume (Process.cpp:936)
    hipLaunchKernelGGL((rx512_kernel<kModeSsb, false, 2, false, false, true, true>), dim3(grid_par), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((rx512_kernel<kModeSsb, false, 2, false, false, false, true>), dim3(grid_par), dim3(256), 0, s, a);
  return hipGetLastError();
}

T41RX_CLK_READER(t41rx_debug_read_clk_long)

}  // namespace t41

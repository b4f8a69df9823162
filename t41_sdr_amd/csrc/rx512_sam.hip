// t41_sdr_amd/csrc/rx512_sam.hip -- the rx512_kernel<kModeSam, ...> instantiations: the synchronous detector (Demod.cpp:40-139)
// on the general front end; barrier form, pipelined PLL (AGC off) and the two-stage pipeline behind the AGC (PSA).
#include "rx512_launch.hpp"

namespace t41 {

hipError_t launch512_sam(const RxArgs &a, hipStream_t s, bool debug) {
  const bool pipe_env = agc_pipe_env();
  if (!a.agc && a.agc_pipe && !debug && a.nframes >= 4 && pipe_env) {  // the PLL pipelined against the neighbouring frames (sam_chain_pipe)
    const dim3 g16((a.nchan + Geo<0>::kWaves - 1) / Geo<0>::kWaves), b16(Geo<0>::kWaves * 64);
    if (a.q15) hipLaunchKernelGGL((rx512_kernel<kModeSam, false, 0, false, false, true, false, true>), g16, b16, 0, s, a);
    else hipLaunchKernelGGL((rx512_kernel<kModeSam, false, 0, false, false, false, false, true>), g16, b16, 0, s, a);
    return hipGetLastError();
  }
  if (a.agc && a.agc_pipe && !debug && a.nframes >= 4 && pipe_env) {  // round 4: AGC chain and PLL, each on a duty wave of its own (PSA)
    const dim3 g16((a.nchan + Geo<0>::kWaves - 1) / Geo<0>::kWaves), b16(Geo<0>::kWaves * 64);
    if (a.q15) hipLaunchKernelGGL((rx512_kernel<kModeSam, false, 0, false, true, true, false, true>), g16, b16, 0, s, a);
    else hipLaunchKernelGGL((rx512_kernel<kModeSam, false, 0, false, true, false, false, true>), g16, b16, 0, s, a);
    return hipGetLastError();
  }
  const int grid = (a.nchan + 3) / 4;
  if (a.q15 && debug) {  // ... with the side outputs / stage taps
    if (a.agc) hipLaunchKernelGGL((rx512_kernel<kModeSam, true, 0, false, true, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((rx512_kernel<kModeSam, true, 0, false, false, true>), dim3(grid), dim3(256), 0, s, a);
  } else if (a.q15) {  // the firmware's sample format either side
    if (a.agc) hipLaunchKernelGGL((rx512_kernel<kModeSam, false, 0, false, true, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((rx512_kernel<kModeSam, false, 0, false, false, true>), dim3(grid), dim3(256), 0, s, a);
  } else if (a.agc) {
    if (debug) hipLaunchKernelGGL((rx512_kernel<kModeSam, true, 0, false, true, false>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((rx512_kernel<kModeSam, false, 0, false, true, false>), dim3(grid), dim3(256), 0, s, a);
  } else if (debug) {
    hipLaunchKernelGGL((rx512_kernel<kModeSam, true, 0, false, false, false>), dim3(grid), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((rx512_kernel<kModeSam, false, 0, false, false, false>), dim3(grid), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

T41RX_CLK_READER(t41rx_debug_read_clk_sam)

}  // namespace t41

// t41_sdr_amd/csrc/rx512_nfm.hip -- the rx512_kernel<kModeNfm, ...> instantiations (FFT_LENGTH 512, the whole chain fused).
#include "rx512_launch.hpp"

namespace t41 {

hipError_t launch512_nfm(const RxArgs &a, hipStream_t s, bool debug) { return launch512<kModeNfm>(a, s, debug); }

T41RX_CLK_READER(t41rx_debug_read_clk_nfm)

}  // namespace t41

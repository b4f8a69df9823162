// t41_sdr_amd/csrc/rx_launch.hpp -- the launchers of the kernel families, one translation unit each (rx_dispatch.hip picks).
#pragma once
#include "rx_kernels.hpp"

namespace t41 {

// FFT_LENGTH 512, the whole chain in rx512_kernel<MODE, ...> (rx512_ssb.hip / rx512_am.hip / rx512_nfm.hip / rx512_sam.hip)
hipError_t launch512_ssb(const RxArgs &a, hipStream_t s, bool debug);
hipError_t launch512_am(const RxArgs &a, hipStream_t s, bool debug);
hipError_t launch512_nfm(const RxArgs &a, hipStream_t s, bool debug);
hipError_t launch512_sam(const RxArgs &a, hipStream_t s, bool debug);
// FFT_LENGTH 1024 / 2048 / 4096: the two ends of the pipeline (rx_long.hip: rx512_kernel<..., PART = 1 / 2>) ...
hipError_t launch_long_front(const RxArgs &a, int mode, hipStream_t s);
hipError_t launch_long_back(const RxArgs &a, int mode, hipStream_t s);
// ... and the N-point fast convolution between them, or the one-kernel form (fastconv.hip)
hipError_t launch_fastconv(const RxArgs &a, bool cplx, bool fused_back, hipStream_t s);
hipError_t launch_fastconv_fused(const RxArgs &a, hipStream_t s);

}  // namespace t41

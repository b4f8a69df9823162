// t41_sdr_amd/csrc/fastconv.hip -- the long-FFT fast convolution kernels' instantiations (fastconv_kernels.hpp).
#include "fastconv_kernels.hpp"
#include "rx_launch.hpp"

namespace t41 {

// real audio with the fixed gain, f32 samples out (`fused_back`): the interpolators run behind pass 3 of the fast
// convolution (no `aud24` round trip, no third kernel); cplx: AM / the AGC take the complex valid half
hipError_t launch_fastconv(const RxArgs &a, bool cplx, bool fused, hipStream_t s) {
#define T41RX_FC(Rv)                                                                                             \
  do {                                                                                                           \
    if (cplx)                                                                                                    \
      hipLaunchKernelGGL((fastconv_kernel<Rv, true>), dim3(a.nchan), dim3(64 * kFcWaves), fc_lds_floats(Rv) * sizeof(float), s, a); \
    else if (fused) {                                                                                            \
      static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&fastconv_kernel<Rv, false, true>), \
          hipFuncAttributeMaxDynamicSharedMemorySize, fcb_lds_floats(Rv) * sizeof(float)); \
      if (attr != hipSuccess) return attr;                                                                       \
      hipLaunchKernelGGL((fastconv_kernel<Rv, false, true>), dim3(a.nchan), dim3(64 * kFcWaves), fcb_lds_floats(Rv) * sizeof(float), s, a); \
    }                                                                                                            \
    else                                                                                                         \
      hipLaunchKernelGGL((fastconv_kernel<Rv, false>), dim3(a.nchan), dim3(64 * kFcWaves), fc_lds_floats(Rv) * sizeof(float), s, a); \
  } while (0)
  if (a.seg == 8)
    T41RX_FC(8);
  else if (a.seg == 4)
    T41RX_FC(4);
  else
    T41RX_FC(2);
#undef T41RX_FC
  return hipGetLastError();
}

// FFT_LENGTH 4096, SSB audio with the fixed gain, f32 samples: the whole chain in one kernel
hipError_t launch_fastconv_fused(const RxArgs &a, hipStream_t s) {
  static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&fastconv_fused_kernel<false>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, fcb_lds_floats(8) * sizeof(float));
  static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&fastconv_fused_kernel<true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, fcb_lds_floats(8) * sizeof(float));
  if (attr0 != hipSuccess) return attr0;
  if (attr1 != hipSuccess) return attr1;
  if (a.plain)
    hipLaunchKernelGGL((fastconv_fused_kernel<true>), dim3(a.nchan), dim3(64 * kFcWaves), fcb_lds_floats(8) * sizeof(float), s, a);
  else
    hipLaunchKernelGGL((fastconv_fused_kernel<false>), dim3(a.nchan), dim3(64 * kFcWaves), fcb_lds_floats(8) * sizeof(float), s, a);
  return hipGetLastError();
}

T41RX_CLK_READER(t41rx_debug_read_clk_fc)

}  // namespace t41

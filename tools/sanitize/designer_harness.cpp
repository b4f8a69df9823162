#include "../../t41_sdr_amd/csrc/rx_internal.hpp"
#include <cstdio>
#include <vector>
using namespace t41;
int main() {
  int n = 0;
  for (int agc = 0; agc <= 4; agc++) for (int mode = 0; mode < 4; mode++) for (int N : {512, 1024, 2048, 4096}) {
    t41rx_params p{}; p.fft_length = N; p.mode = mode; p.FLoCut = 200; p.FHiCut = 3000; p.rfGainAllBands = 1; p.RFgain = 1;
    p.IQAmpCorrectionFactor = 1; p.AGCMode = agc; p.audioVolume = 30; p.nfmFilterBW = 12000; p.CWFreqShift = 750; p.am_lpf_f0 = 3000; p.AGC_thresh = 20;
    if (mode == 1) { p.FLoCut = -3000; p.FHiCut = -200; } if (mode == 2) { p.FLoCut = -3000; }
    std::vector<float> blob(blob_floats(N));
    const char *why = nullptr;
    if (!params_valid(p, &why)) { std::printf("invalid: %s\n", why); return 1; }
    if (design_blob(p, blob.data(), blob.size() * sizeof(float)) != T41RX_OK) return 2;
    if (design_blob(p, blob.data(), blob.size() * sizeof(float) - 4) == T41RX_OK) return 3;  // too small must fail
    ++n;
  }
  std::printf("designer sanitizer harness ok (%d designs)\n", n);
}

// t41_sdr_amd/csrc/display_kernel.hip -- the display FFT side output (t41rx_set_display_spectrum).
#include "rx_device.hpp"

namespace t41 {

// ------------------------------------------------------------------------------------------
// Display FFT side output: CalcZoom1Magn() (spectrumZoom 0, FFT.cpp:208-251) and ZoomFFTExe()
// (spectrumZoom 1..4, FFT.cpp:67-152) up to FFT_spec / FFT_spec_old; the pixel mapping behind them
// is display code.  One wave per channel on the dbg_pre tap of the frames just processed (the
// firmware runs it once per display refresh, not per frame: nothing here is tuned).  Zoom: the
// 4-stage IIR and the decimating FIR are serial in the sample index -- every lane runs the I (even
// lanes) or the Q (odd lanes) chain redundantly, sample by sample, with the reference's order of
// operations; windowing, the 512-point FFT, magnitudes and the low-pass are wave-parallel.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void display_kernel(const DispArgs a) {
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) float xbuf[8 * kFftRow * 2];  // FFT exchange
  __shared__ float stage[2][2048];                                     // zoom: I / Q after the Fs/4 shift
  __shared__ float ring[2][512];
  __shared__ float dec[2][512];                                        // zoom: decimated samples of this frame
  const int lane = threadIdx.x;
  const int ch = blockIdx.x;
  if (ch >= a.nchan) return;
  constexpr int L = 2048, R = 512;
  float *ds = a.disp + (size_t)ch * kDispFloats;
  const cf *tab = reinterpret_cast<const cf *>(a.tab);
  cf tw1[7], tw2[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    tw1[q] = tab[kTabTw1 + 64 * q + lane];
    tw2[q] = tab[kTabTw2 + 64 * q + lane];
  }
  const int zoom = a.zoom;
  const int chain = lane & 1;
  // zoom filter memories of my chain, the ring, the low-pass memory
  float st[16], fh[3];
  int ptr = 0;
  if (zoom > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = ds[kDispIir + 16 * chain + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) fh[i] = ds[kDispFir + 4 * chain + i];
    ptr = reinterpret_cast<const int *>(ds)[kDispPtr];
    for (int i = lane; i < 2 * R; i += 64) (&ring[0][0])[i] = ds[kDispRing + i];
  }
  float old[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) old[r] = ds[kDispOld + ((lane + 64 * r + 256) & 511)];  // index of bin lane + 64 r
  double win[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) win[r] = a.win[lane + 64 * r];
  const float LPFcoeff = 0.7f;
  for (int f = 0; f < a.nframes; ++f) {
    const float *pI = a.pre + ((size_t)ch * a.nframes + f) * (2 * L), *pQ = pI + L;
    cf v[8];
    if (zoom == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {  // float * double -> double -> float, FFT.cpp:222-223
        const int i = lane + 64 * r;
        v[r] = cf{(float)((double)pI[i] * win[r]), (float)((double)pQ[i] * win[r])};
      }
    } else {
      __syncthreads();
      for (int n = lane; n < L; n += 64) {  // FreqShift1 (Freq_Shift.cpp:42-65): x j^n
        const float xi = pI[n], xq = pQ[n];
        const int m = n & 3;
        stage[0][n] = (m == 0) ? xi : (m == 1) ? -xq : (m == 2) ? -xi : xq;
        stage[1][n] = (m == 0) ? xq : (m == 1) ? xi : (m == 2) ? -xq : -xi;
      }
      __syncthreads();
      const int M = 1 << zoom;
      const int sample_no = (L / M > R) ? R : L / M;
      float h0 = fh[0], h1 = fh[1], h2 = fh[2];
      for (int n = 0; n < L; ++n) {
        float x = stage[chain][n];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {  // arm_biquad_cascade_df1_f32: acc = b0 x + b1 x1 + b2 x2 + a1 y1 + a2 y2
          const float *c = a.iir + 5 * s4;
          float acc = c[0] * x;
          acc += c[1] * st[4 * s4 + 0];
          acc += c[2] * st[4 * s4 + 1];
          acc += c[3] * st[4 * s4 + 2];
          acc += c[4] * st[4 * s4 + 3];
          st[4 * s4 + 1] = st[4 * s4 + 0];
          st[4 * s4 + 0] = x;
          st[4 * s4 + 3] = st[4 * s4 + 2];
          st[4 * s4 + 2] = acc;
          x = acc;
        }
        if ((n & (M - 1)) == 0) {  // arm_fir_decimate_f32, 4 taps: y[k] = sum_i c[i] state[k M + i], newest sample = x
          const int k = n >> zoom;
          float acc = a.fir[0] * h0;
          acc += a.fir[1] * h1;
          acc += a.fir[2] * h2;
          acc += a.fir[3] * x;
          if (k < sample_no && lane < 2) dec[chain][k] = acc;
        }
        h0 = h1;
        h1 = h2;
        h2 = x;
      }
      fh[0] = h0;
      fh[1] = h1;
      fh[2] = h2;
      __syncthreads();
      for (int k = lane; k < sample_no; k += 64) {  // FFT.cpp:98-104
        ring[0][(ptr + k) & 511] = dec[0][k];
        ring[1][(ptr + k) & 511] = dec[1][k];
      }
      ptr = (ptr + sample_no) & 511;
      __syncthreads();
      const float multiplier = (zoom > 3) ? (float)(1 << zoom) : (float)zoom;  // FFT.cpp:106-109
#pragma unroll
      for (int r = 0; r < 8; ++r) {  // float * float -> float, * double -> double -> float, FFT.cpp:110-111
        const int idx = lane + 64 * r;
        const float mx = multiplier * ring[0][(ptr + idx) & 511], my = multiplier * ring[1][(ptr + idx) & 511];
        v[r] = cf{(float)((double)mx * win[r]), (float)((double)my * win[r])};
      }
      // (zoom_sample_ptr ends where it started after the 512 reads)
    }
    fft512<false>(v, tw1, tw2, xbuf, lane);
    float *so = a.spec + ((size_t)ch * a.nframes + f) * R, *oo = a.spec_old + ((size_t)ch * a.nframes + f) * R;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int x = (lane + 64 * r + 256) & 511;  // bins 0..255 -> upper half, 256..511 -> lower half
      const float m = v[r].x * v[r].x + v[r].y * v[r].y;
      float spec, nold;
      if (zoom == 0) {  // FFT.cpp:241-243: float + double
        nold = (float)((double)(LPFcoeff * m) + (1.0 - (double)LPFcoeff) * (double)old[r]);
        spec = m;
      } else {  // FFT.cpp:136-137: all float
        const float onem = (float)(1.0 - (double)LPFcoeff);
        spec = LPFcoeff * m + onem * old[r];
        nold = spec;
      }
      old[r] = nold;
      so[x] = spec;
      oo[x] = nold;
    }
  }
  // state back
#pragma unroll
  for (int r = 0; r < 8; ++r) ds[kDispOld + ((lane + 64 * r + 256) & 511)] = old[r];
  if (zoom > 0) {
    if (lane < 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) ds[kDispIir + 16 * chain + i] = st[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) ds[kDispFir + 4 * chain + i] = fh[i];
    }
    if (lane == 0) reinterpret_cast<int *>(ds)[kDispPtr] = ptr;
    __syncthreads();
    for (int i = lane; i < 2 * R; i += 64) ds[kDispRing + i] = (&ring[0][0])[i];
  }
}

hipError_t launch_display(const DispArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(display_kernel, dim3(a.nchan), dim3(64), 0, s, a);
  return hipGetLastError();
}

}  // namespace t41

// t41_sdr_amd/csrc/tx_kernels.hip -- gfx950 kernel of the T41 transmit exciter, ExciterIQData()
// (Exciter.cpp:46-169): the receive path's resamplers run backwards with fixed tables.
//
// One 64-lane wave = one channel, all the frames of a call, every stage fused; the delay lines live
// in the wave's LDS for the whole call (HBM state read once, written once).  Per frame a channel
// reads 4 KiB (2048 q15 microphone samples) and writes 8 KiB (2048 q15 I + 2048 q15 Q).
//   arm_q15_to_float                               Exciter.cpp:63-64
//   /4, 48 taps (coeffs192K_10K_LPF_FIR)           :84      dec1: 8 outputs per lane
//   /2, 24 taps (coeffs48K_8K_LPF_FIR[0..23])      :88      dec2: 4 outputs per lane
//   copy L -> R                                    :98
//   Hilbert +45 / -45 degrees, 100 taps each       :110-111 4 consecutive outputs per lane, one window
//   TX IQ amplitude / phase correction             :117-127
//   x2, 48 taps; x4, 32 taps, per channel          :141-148 polyphase (arm_fir_interpolate_f32)
//   x 20, arm_float_to_q15                         :151-162
// Every FIR sums its taps in CMSIS-DSP's order with separate multiplies and adds (no FMA
// contraction): the q15 outputs then equal the reference arithmetic's except where a float sits
// within rounding of a truncation boundary.
// Not tuned (scalar f32 MACs, LDS windows): a side path of this library, SURVEY 8f rank 3.
#include <hip/hip_runtime.h>

#include "tx_internal.hpp"

namespace t41 {

namespace {

__device__ __forceinline__ void wave_sync() {
  // one wave per workgroup: LDS executes its instructions in order; only the compiler must not reorder
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ float4 lds4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
typedef const __attribute__((address_space(4))) TxCoef *CoefPtr;

// (q15_t)__SSAT((q31_t)(x * 32768.0f), 16): CMSIS-DSP's arm_float_to_q15 without ARM_MATH_ROUNDING
__device__ __forceinline__ unsigned q15_pack2(float x0, float x1) {
  int a = (int)(x0 * 32768.0f), b = (int)(x1 * 32768.0f);
  a = a < -32768 ? -32768 : (a > 32767 ? 32767 : a);
  b = b < -32768 ? -32768 : (b > 32767 ? 32767 : b);
  return ((unsigned)a & 0xffffu) | ((unsigned)b << 16);
}

// LDS layout of the wave (floats): every delay line as [history | new], history first
constexpr int kXs = 0;                    // 47 + 2048 (+1)
constexpr int kY1 = kXs + 2096;           // 23 + 512 (+1)
constexpr int kHl = kY1 + 536;            // 99 + 256 (+1): both Hilbert filters see the same samples (Exciter.cpp:98)
constexpr int kI1 = kHl + 356;            // [2][23 + 256 (+1)]
constexpr int kI2 = kI1 + 2 * 280;        // [2][7 + 512 (+1)]
constexpr int kLdsFloats = kI2 + 2 * 520;

}  // namespace

__global__ __launch_bounds__(64) void tx_kernel(const TxArgs a) {
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  const int lane = threadIdx.x;
  const int ch = blockIdx.x;
  if (ch >= a.nchan) return;
  float *st = a.state + (size_t)ch * kTxStateFloats;
  const CoefPtr cf = (CoefPtr)a.coef;

  // ---- delay lines: HBM -> LDS, once per call
  if (lane < 47) lds[kXs + lane] = st[kTxStDec1 + lane];
  if (lane < 23) lds[kY1 + lane] = st[kTxStDec2 + lane];
  for (int i = lane; i < 99; i += 64) lds[kHl + i] = st[kTxStHilL + i];  // (FIR_Hilbert_state_R holds the same samples)
  if (lane < 23) {
    lds[kI1 + lane] = st[kTxStInt1I + lane];
    lds[kI1 + 280 + lane] = st[kTxStInt1Q + lane];
  }
  if (lane < 7) {
    lds[kI2 + lane] = st[kTxStInt2I + lane];
    lds[kI2 + 520 + lane] = st[kTxStInt2Q + lane];
  }
  wave_sync();

  for (int f = 0; f < a.nframes; ++f) {
    const size_t base = ((size_t)ch * a.nframes + f) * 2048;
    // ---- arm_q15_to_float (x / 32768, exact)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const uint4 w = *reinterpret_cast<const uint4 *>(a.inL + base + 512 * it + 8 * lane);
      const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        lds[kXs + 47 + 512 * it + 8 * lane + 2 * k] = (float)(short)(ww[k] & 0xffffu) * (1.0f / 32768.0f);
        lds[kXs + 47 + 512 * it + 8 * lane + 2 * k + 1] = (float)((int)ww[k] >> 16) * (1.0f / 32768.0f);
      }
    }
    wave_sync();
    // ---- /4: y[m] = sum_i c[i] state[4 m + i], m = lane + 64 j
#pragma unroll 2
    for (int j = 0; j < 8; ++j) {
      const int m = lane + 64 * j;
      float acc = 0.0f;
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const float4 t = lds4(lds + kXs + 4 * m + 4 * q);
        acc += t.x * cf->c192[4 * q];
        acc += t.y * cf->c192[4 * q + 1];
        acc += t.z * cf->c192[4 * q + 2];
        acc += t.w * cf->c192[4 * q + 3];
      }
      lds[kY1 + 23 + m] = acc;
    }
    wave_sync();
    // ---- /2: y[m] = sum_i c[i] state[2 m + i], m = lane + 64 j
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = lane + 64 * j;
      float acc = 0.0f;
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const float2 t = *reinterpret_cast<const float2 *>(lds + kY1 + 2 * m + 2 * q);
        acc += t.x * cf->c48[2 * q];
        acc += t.y * cf->c48[2 * q + 1];
      }
      lds[kHl + 99 + m] = acc;
    }
    wave_sync();
    // ---- Hilbert pair: y[n] = sum_i c[i] state[n + i], n = 4 lane + j, one window for both filters
    float I4[4], Q4[4];
    {
      float w[104];
#pragma unroll
      for (int q = 0; q < 26; ++q) {
        const float4 t = lds4(lds + kHl + 4 * lane + 4 * q);
        w[4 * q] = t.x;
        w[4 * q + 1] = t.y;
        w[4 * q + 2] = t.z;
        w[4 * q + 3] = t.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float ai = 0.0f, aq = 0.0f;
#pragma unroll
        for (int i = 0; i < 100; ++i) {
          ai += w[j + i] * cf->h45[i];
          aq += w[j + i] * cf->hn45[i];
        }
        I4[j] = ai;
        Q4[j] = aq;
      }
    }
    // ---- TX IQ correction (Exciter.cpp:117-127; IQPhaseCorrection Utility.cpp:178-187)
    if (a.corr_on) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        I4[j] = I4[j] * a.i_scale;
        if (a.iq_phase < 0.0f) Q4[j] = Q4[j] + I4[j] * a.iq_phase;
        else I4[j] = I4[j] + Q4[j] * a.iq_phase;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) Q4[j] = Q4[j] * 1.00f;
    // ---- x2 (48 taps, 24 per phase) then x4 (32 taps, 8 per phase), I then Q
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float *s1 = lds + kI1 + 280 * c, *s2 = lds + kI2 + 520 * c;
      wave_sync();
      *reinterpret_cast<float2 *>(s1 + 23 + 4 * lane + 1) = make_float2(c ? Q4[1] : I4[1], c ? Q4[2] : I4[2]);  // (23 + 4 lane + 1 is even)
      s1[23 + 4 * lane] = c ? Q4[0] : I4[0];
      s1[23 + 4 * lane + 3] = c ? Q4[3] : I4[3];
      wave_sync();
      {
        // out[2 n + j - 1] = sum_t state[n + t] c[(2 - j) + 2 t], n = 4 lane + u
        float w[28];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          const float4 t = lds4(s1 + 4 * lane + 4 * q);
          w[4 * q] = t.x;
          w[4 * q + 1] = t.y;
          w[4 * q + 2] = t.z;
          w[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float o0 = 0.0f, o1 = 0.0f;
#pragma unroll
          for (int t = 0; t < 24; ++t) {
            o0 += w[u + t] * cf->c48[1 + 2 * t];
            o1 += w[u + t] * cf->c48[2 * t];
          }
          s2[7 + 8 * lane + 2 * u] = o0;
          s2[7 + 8 * lane + 2 * u + 1] = o1;
        }
      }
      wave_sync();
      // out[4 n + j - 1] = sum_t state[n + t] c[(4 - j) + 4 t], n = lane + 64 u: 8 bytes per lane and u
      int16_t *out = (c ? a.outR : a.outL) + base;
#pragma unroll 2
      for (int u = 0; u < 8; ++u) {
        const int n = lane + 64 * u;
        float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float x = s2[n + t];
          o[0] += x * cf->c192[3 + 4 * t];
          o[1] += x * cf->c192[2 + 4 * t];
          o[2] += x * cf->c192[1 + 4 * t];
          o[3] += x * cf->c192[4 * t];
        }
        // x 20 (Exciter.cpp:151-152), arm_float_to_q15 (:161-162)
        *reinterpret_cast<uint2 *>(out + 4 * n) = make_uint2(q15_pack2(o[0] * 20.0f, o[1] * 20.0f), q15_pack2(o[2] * 20.0f, o[3] * 20.0f));
      }
    }
    // ---- roll the delay lines: the last numTaps - 1 samples move to the front
    wave_sync();
    {
      const float x1 = (lane < 47) ? lds[kXs + 2048 + lane] : 0.0f;
      const float y1 = (lane < 23) ? lds[kY1 + 512 + lane] : 0.0f;
      const float h0 = lds[kHl + 256 + lane], h1 = (lane < 35) ? lds[kHl + 256 + 64 + lane] : 0.0f;
      const float i1a = (lane < 23) ? lds[kI1 + 256 + lane] : 0.0f, i1b = (lane < 23) ? lds[kI1 + 280 + 256 + lane] : 0.0f;
      const float i2a = (lane < 7) ? lds[kI2 + 512 + lane] : 0.0f, i2b = (lane < 7) ? lds[kI2 + 520 + 512 + lane] : 0.0f;
      wave_sync();
      if (lane < 47) lds[kXs + lane] = x1;
      if (lane < 23) lds[kY1 + lane] = y1;
      lds[kHl + lane] = h0;
      if (lane < 35) lds[kHl + 64 + lane] = h1;
      if (lane < 23) {
        lds[kI1 + lane] = i1a;
        lds[kI1 + 280 + lane] = i1b;
      }
      if (lane < 7) {
        lds[kI2 + lane] = i2a;
        lds[kI2 + 520 + lane] = i2b;
      }
    }
    wave_sync();
  }
  // ---- delay lines back to HBM
  if (lane < 47) st[kTxStDec1 + lane] = lds[kXs + lane];
  if (lane < 23) st[kTxStDec2 + lane] = lds[kY1 + lane];
  for (int i = lane; i < 99; i += 64) {
    st[kTxStHilL + i] = lds[kHl + i];
    st[kTxStHilR + i] = lds[kHl + i];
  }
  if (lane < 23) {
    st[kTxStInt1I + lane] = lds[kI1 + lane];
    st[kTxStInt1Q + lane] = lds[kI1 + 280 + lane];
  }
  if (lane < 7) {
    st[kTxStInt2I + lane] = lds[kI2 + lane];
    st[kTxStInt2Q + lane] = lds[kI2 + 520 + lane];
  }
}

hipError_t launch_tx(const TxArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(tx_kernel, dim3(a.nchan), dim3(64), 0, s, a);
  return hipGetLastError();
}

}  // namespace t41

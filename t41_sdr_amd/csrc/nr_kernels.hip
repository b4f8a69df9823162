// t41_sdr_amd/csrc/nr_kernels.hip -- gfx950 kernels of the receive path's optional stages between the
// demodulator and the interpolators (Process.cpp:841-866): noise reduction (Kim1_NR() Noise.cpp:108-313,
// SpectralNoiseReduction() Noise.cpp:379-655, Xanr() as LMS noise reduction Noise.cpp:322-370) and the
// automatic notch (Xanr() again).  All off in the firmware's defaults; FFT_LENGTH 512.
//
// They run on the call's demodulated audio @24 kS/s, which the fused kernel leaves in a scratch
// ([channel][frame * 256], rx_kernels.hip: RxArgs::aud_out), in place, frame by frame; the long-FFT
// pipeline's back kernel then interpolates (launch_back512).
//
//  * anr_kernel: Xanr() is a 64-tap adaptive FIR whose every output feeds back into its taps, sample by
//    sample, and whose sums the reference accumulates in tap order.  ONE LANE PER CHANNEL: a wave runs 64
//    channels' filters, every lane the reference's loop as written (same operations, same order, no
//    contraction -- bit-identical to the scalar code; the step is shared by the TWO waves of the workgroup, see
//    anr_pass_y), taps in 64 registers, the delay line's live window
//    in an LDS tile laid out [time][channel] (row pitch 65 floats: conflict-free both for the transposing
//    stage-in / stage-out and for the per-lane window reads).  The tile is filled and drained with
//    coalesced row accesses of the [channel][time] scratch.
//  * nrspec_kernel<KIND>: the two spectral-weighting functions, one wave per channel.  Both transform
//    half-overlapped 256-sample frames, two per block, with real input; here the block's two frames ride
//    as real and imaginary part through ONE 512-point transform of the path's register-resident FFT
//    (zero-interleaved, so that bins 0..255 hold the 256-point transform), are separated by conjugate
//    symmetry, weighted, and go back through ONE inverse transform the same way.  The reference weights bin
//    i together with bin 255 - i (not 256 - i) and keeps the real part of the inverse transform: the real
//    part sees the conjugate-symmetric half of the weighted spectrum, i.e. bin k weighted by
//    (G[k] + G[k-1]) / 2 -- which is what is applied here.  Per-bin statistics: two bins per lane.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "nr_kernels.hpp"
#include "rx_kernels.hpp"
#include "wave_fft.hpp"

namespace t41 {

// T41RX_ANR_PREFETCH=0: the notch's window requested at the top of its own sample step, as in round 4 (A/B builds)
#ifndef T41RX_ANR_PREFETCH
#define T41RX_ANR_PREFETCH 1
#endif
// T41RX_NRSPEC_LDS_LOOP=1: the spectral function's bin loop as round 4 ran it, gains in LDS (A/B builds)
#ifndef T41RX_NRSPEC_LDS_LOOP
#define T41RX_NRSPEC_LDS_LOOP 0
#endif

// ------------------------------------------------------------------------------------------
// Xanr(): 16 channels per wave, a channel's 64 taps on the four lanes (c, c + 16, c + 32, c + 48)
// ------------------------------------------------------------------------------------------
// Round 3 ran one lane per channel (64 taps in 64 registers): a sample's step was ~430 instructions, 300 of them the
// products, squares and the taps' update -- independent work that one wave had to issue between the 64 additions of the
// filter's output, which the reference accumulates in tap order and which therefore stay a chain.  Round 4 spreads that
// work over the wave: lane (g, c) = 16 g + c holds 16 taps of channel c, so products / update are 8 packed
// instructions per lane instead of 32, and the chain walks through the four lane groups in tap order: every lane adds
// its 16 products to whatever partial sum it holds, the partial sum that matters sits in the "active" group, and after
// 16 additions it moves on to the next group by ONE lane-swap instruction (v_permlane16_swap / v_permlane32_swap; the
// groups are visited in the order 0, 1, 3, 2 because those are the moves the two instructions make).  Same additions,
// same order, same roundings as the scalar loop; a sample costs the chain's 64 dependent additions and little else.
constexpr int kAnrCw = 16;                     // channels per wave / workgroup
constexpr int kAnrRow = kAnrCw + 1;            // LDS row pitch in floats (odd: the transposing stage-in / stage-out and the window reads are conflict-free)
constexpr int kAnrTile = kAnrHist + 256;       // rows of the input tile: 79 of history + the frame
constexpr int kAnrSigOff = ((kAnrTile + 256) * kAnrRow + 3) & ~3;
constexpr size_t kAnrLdsBytes = ((size_t)kAnrSigOff + 3 * kAnrCw * 4) * sizeof(float);  // input tile + output tile + three (round 4: two) slots of 16 x (sigma, 1 / sigma', 1 - 2 mu sigma / sigma' as a double)

// first tap of lane group g: the chain visits the groups in the order 0, 1, 3, 2
__device__ __forceinline__ int anr_tap_base(int g) { return 16 * ((g == 2) ? 3 : (g == 3) ? 2 : g); }
// the partial sum moves from the group that has just finished to the one that continues (every other lane: don't care)
template <int PHASE>
__device__ __forceinline__ float anr_hand_on(float y) {
  float a = y, b = y;
  if (PHASE == 0) {        // group 0 -> group 1
    lane_swap16(a, b);
    return a;
  } else if (PHASE == 1) { // group 1 -> group 3
    lane_swap32(a, b);
    return a;
  } else {                 // group 3 -> group 2
    lane_swap16(a, b);
    return b;
  }
}
// group 2's value to every group
__device__ __forceinline__ float anr_from_group2(float y) {
  float a = y, b = y;
  lane_swap32(a, b);  // b = {y2, y3, y2, y3}
  float c = b, d = b;
  lane_swap16(c, d);  // c = {y2, y2, y2, y2}
  return c;
}
// sum of p[0 .. 15] in tap order on top of the running sum, through the four groups: the reference's 64 additions
__device__ __forceinline__ float anr_chain(const f2 (&p)[8]) {
#pragma clang fp contract(off)
  float y = 0;
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      y += p[t].y;
      y += p[t].x;
    }
    if (ph == 0) y = anr_hand_on<0>(y);
    else if (ph == 1) y = anr_hand_on<1>(y);
    else if (ph == 2) y = anr_hand_on<2>(y);
  }
  return y;  // valid in group 2
}

// One pass of Noise.cpp:331-369 over the 256 samples in T[kAnrHist ..], split over the workgroup's TWO waves (same
// lane -> (group, channel) map in both): wave 1 takes the window's sum of squares (ANR_sigma) and what depends on it
// alone (an IEEE double division), wave 0 the filter output, the error, the leak logic and the taps' update; sigma
// crosses through LDS (two slots: wave 1 may be a sample ahead) behind ONE workgroup barrier per sample.  Every value
// by the same operations in the same order as the scalar loop.
// O (may be null): output tile; SIG: [2][16] float4.
template <bool NOTCH>
__device__ __forceinline__ void anr_pass_y(const float *T, float *O, const float *SIG, f2 (&w)[8], float &lidx, float &ngamma, int lane) {
#pragma clang fp contract(off)
  const float ANR_den_mult = 6.25e-10, ANR_gamma = 0.1, ANR_lidx_min = 120.0, ANR_lidx_max = 200.0;
  const float ANR_lincr = 1.0, ANR_ldecr = 3.0, ANR_two_mu = 0.0001;
  const int c = lane & 15, g = lane >> 4, tb = anr_tap_base(g);
  // The taps' update of sample i is fused into the products of sample i + 1 (it needs sample i's window, which dp still
  // holds): tap pair t is updated, then used.  Same values as updating all taps first.
  float c0 = 1.0f, c1 = 0.0f;  // pending update of the previous sample (none yet: w * 1 + 0 * d would not be exact for
  bool pending = false;        // -0 / NaN taps, so it is skipped rather than applied)
#if T41RX_ANR_PREFETCH
  // Round 5: a sample's window is requested at the top of the PREVIOUS sample's step, into a third register set -- the
  // delay line is the input signal, so every window of the frame is in the tile before the pass starts.  Requested at
  // the top of its own step, as in round 4, the window is waited for twice per sample (two batches of eight reads) at the
  // head of the dependent sequence.  (Requested behind the products into the previous window's registers -- two sets --
  // the chain's first sixteen additions no longer interleave with the products: 148 against 140 us per frame.)
  // (ONE lane-dependent base, opaque to the compiler, and the sixteen rows at immediate offsets from it: eight
  // ds_read2_b32 -- the pairs are 17 dwords apart, 15 x 17 = 255 just fits the offset field.  Left to itself hipcc keeps
  // eight address registers, adds the sample's offset to each and issues sixteen single reads: 25 instructions for 9.)
  typedef const __attribute__((address_space(3))) float *LdsRow;
  LdsRow wbase = (LdsRow)(T + c + (kAnrTaps - 16 - tb) * kAnrRow);
  asm volatile("" : "+v"(wbase));
  auto window = [&](int i, f2 (&dj)[8], float &d_in) {
    d_in = T[i * kAnrRow + c + kAnrHist * kAnrRow];  // ANR_d[ANR_in_idx]
    LdsRow row = wbase + i * kAnrRow;
#pragma unroll
    for (int t = 0; t < 16; t += 2)
      dj[t / 2] = f2{row[(14 - t) * kAnrRow], row[(15 - t) * kAnrRow]};
  };
  // dj: this sample's window (in registers); dp: the previous sample's, for the pending update; dx: takes the next one's
  auto step = [&](int i, int slot, const f2 (&dj)[8], const f2 (&dp)[8], f2 (&dx)[8], const float d_in, float &d_next) {
    if (i + 1 < 256) window(i + 1, dx, d_next);
    // (wave 1 runs a sample ahead: this sample's sigma, with what depends on sigma alone, has been in its slot since the
    // previous sample's barrier)
    const float4 sg = *reinterpret_cast<const float4 *>(SIG + 4 * (kAnrCw * slot + c));  // (slot = i % 3: static per unrolled step)
    __builtin_amdgcn_sched_barrier(0);  // (the requests go out HERE: left alone, the scheduler sinks them behind the chain)
    f2 p[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (pending) w[t] = splat(c0) * w[t] + splat(c1) * dp[t];
      p[t] = w[t] * dj[t];
    }
    const float y = anr_from_group2(anr_chain(p));
    __syncthreads();  // keeps wave 1 exactly one sample ahead (it must not overwrite a slot this wave has yet to read)
    const float sigma = sg.x, inv_sigp = sg.y;  // inv_sigp = (float)(1.0 / ((double)sigma + 1e-10))
    const double one_m = __hiloint2double(__float_as_int(sg.w), __float_as_int(sg.z));  // 1.0 - (double)(ANR_two_mu * sigma * inv_sigp)
    const float error = d_in - y;
    if (O && g == 0) O[i * kAnrRow + c] = NOTCH ? error : y;
    // |nel|, |nev|: the reference's `if (x < 0) x = -x` and fabsf() differ for x = -0 only, and the two are only ever
    // compared with each other; as operand modifiers of that comparison they cost nothing (as written hipcc compares the
    // DOUBLES with zero and selects: two half-rate compares and two selects per sample)
    const float nel = fabsf((float)((double)error * one_m));
    const float nev = fabsf((float)((double)d_in - (1.0 - (double)(ANR_two_mu * ngamma)) * (double)y - (double)(ANR_two_mu * error * sigma * inv_sigp)));
    if (nev < nel) {  // as written (Noise.cpp:351-356): the else-if belongs to the inner if.  (Branches on purpose: most samples
      // skip the block; the same logic as selects is seven instructions on every sample and measured 2 us per frame slower)
      lidx += ANR_lincr;
      if (lidx > ANR_lidx_max) {
        lidx = ANR_lidx_max;
      } else {
        lidx -= ANR_ldecr;
        if (lidx < ANR_lidx_min) lidx = ANR_lidx_min;
      }
    }
    ngamma = ANR_gamma * (lidx * lidx) * (lidx * lidx) * ANR_den_mult;
    c0 = (float)(1.0 - (double)(ANR_two_mu * ngamma));
    c1 = ANR_two_mu * error * inv_sigp;
    pending = true;
  };
  f2 da[8], db[8], dc[8];
  float din_a, din_b = 0.0f, din_c = 0.0f;
#pragma unroll
  for (int t = 0; t < 8; ++t) dc[t] = splat(0.0f);
  window(0, da, din_a);
  __syncthreads();  // wave 1's sigma of sample 0
  for (int i = 0; i < 255; i += 3) {  // 85 x 3 samples
    step(i, 0, da, dc, db, din_a, din_b);      // window i in da, i - 1 in dc; i + 1 -> db
    step(i + 1, 1, db, da, dc, din_b, din_c);  // i + 2 -> dc
    step(i + 2, 2, dc, db, da, din_c, din_a);  // i + 3 -> da
  }
  step(255, 0, da, dc, db, din_a, din_b);  // (255 = 3 x 85)
#pragma unroll
  for (int t = 0; t < 8; ++t) db[t] = da[t];  // the last sample's update below takes its window from db
#else
  auto step = [&](int i, f2 (&dj)[8], const f2 (&dp)[8]) {
    const float *row = T + i * kAnrRow + c;
    const float d_in = row[kAnrHist * kAnrRow];  // ANR_d[ANR_in_idx]
    // A register pair holds taps (j + 1, j) in (.x, .y): the window's rows ascend in time, i.e. descend in j, so one
    // ds_read2_b32 fills a pair without a move.  (idx = in_idx + j + ANR_delay: the sample written j + 16 steps ago)
#pragma unroll
    for (int t = 0; t < 16; t += 2)
      dj[t / 2] = f2{row[(kAnrTaps - 2 - tb - t) * kAnrRow], row[(kAnrTaps - 1 - tb - t) * kAnrRow]};
    // (two taps per multiply instruction -- v_pk_mul_f32 rounds each product exactly like the scalar multiply)
    f2 p[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (pending) w[t] = splat(c0) * w[t] + splat(c1) * dp[t];
      p[t] = w[t] * dj[t];
    }
    const float y = anr_from_group2(anr_chain(p));
    __syncthreads();  // wave 1's sigma of this sample is in its slot, with what depends on sigma alone
    const float4 sg = *reinterpret_cast<const float4 *>(SIG + 4 * (kAnrCw * (i & 1) + c));
    const float sigma = sg.x, inv_sigp = sg.y;  // inv_sigp = (float)(1.0 / ((double)sigma + 1e-10))
    const double one_m = __hiloint2double(__float_as_int(sg.w), __float_as_int(sg.z));  // 1.0 - (double)(ANR_two_mu * sigma * inv_sigp)
    const float error = d_in - y;
    if (O && g == 0) O[i * kAnrRow + c] = NOTCH ? error : y;
    float nel = (float)((double)error * one_m);
    if (nel < 0.0f) nel = -nel;
    float nev = (float)((double)d_in - (1.0 - (double)(ANR_two_mu * ngamma)) * (double)y - (double)(ANR_two_mu * error * sigma * inv_sigp));
    if (nev < 0.0f) nev = -nev;
    if (nev < nel) {  // as written (Noise.cpp:351-356): the else-if belongs to the inner if
      lidx += ANR_lincr;
      if (lidx > ANR_lidx_max) {
        lidx = ANR_lidx_max;
      } else {
        lidx -= ANR_ldecr;
        if (lidx < ANR_lidx_min) lidx = ANR_lidx_min;
      }
    }
    ngamma = ANR_gamma * (lidx * lidx) * (lidx * lidx) * ANR_den_mult;
    c0 = (float)(1.0 - (double)(ANR_two_mu * ngamma));
    c1 = ANR_two_mu * error * inv_sigp;
    pending = true;
  };
  f2 da[8], db[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) db[t] = splat(0.0f);
  for (int i = 0; i < 256; i += 2) {
    step(i, da, db);
    step(i + 1, db, da);
  }
#endif
  // the last sample's update (its window is in db)
#pragma unroll
  for (int t = 0; t < 8; ++t) w[t] = splat(c0) * w[t] + splat(c1) * db[t];
}
__device__ __forceinline__ void anr_pass_sigma(const float *T, float *SIG, int lane) {
#pragma clang fp contract(off)
  const int c = lane & 15, g = lane >> 4, tb = anr_tap_base(g);
  auto sigma_of = [&](int i) {
    const float *row = T + i * kAnrRow + c;
    f2 q[8];
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
      const f2 d = f2{row[(kAnrTaps - 2 - tb - t) * kAnrRow], row[(kAnrTaps - 1 - tb - t) * kAnrRow]};
      q[t / 2] = d * d;
    }
    const float sigma = anr_chain(q);  // (group 2 holds it)
    const float ANR_two_mu = 0.0001;
    const float inv_sigp = (float)(1.0 / ((double)sigma + 1e-10));
    const double one_m = 1.0 - (double)(ANR_two_mu * sigma * inv_sigp);
    if (g == 2)
      *reinterpret_cast<float4 *>(SIG + 4 * (kAnrCw * (T41RX_ANR_PREFETCH ? i % 3 : (i & 1)) + c)) =
          make_float4(sigma, inv_sigp, __int_as_float(__double2loint(one_m)), __int_as_float(__double2hiint(one_m)));
  };
#if T41RX_ANR_PREFETCH
  // one sample AHEAD of the filter wave: sigma of sample i is in its slot before the barrier of sample i - 1, so the filter
  // wave reads it at the top of its step, not behind its chain (the slot it overwrites, i + 1's = i - 1's, was read at the
  // top of step i - 1, a barrier ago)
  sigma_of(0);
  __syncthreads();
  for (int i = 0; i < 256; ++i) {
    if (i + 1 < 256) sigma_of(i + 1);
    __syncthreads();
  }
#else
  for (int i = 0; i < 256; ++i) {
    sigma_of(i);
    __syncthreads();
  }
#endif
}

__global__ __launch_bounds__(128, 2) void anr_kernel(const NrArgs a) {
  // (The copy / scale loops below are kept rolled: fully unrolled, with every address hoisted out of the frame loop, they
  // took the kernel to 353 VGPRs; 161 now.  Measured at the end of round 4, all bit-identical to this form
  // (tools/anr_ab_check.py) and none faster than 1 %: no workgroup barrier per sample -- sigma through two slots guarded
  // by LDS counters, wave 1 up to two samples ahead --, the next sample's window and this sample's sigma requested a
  // chain ahead of their use, both outcomes of the leak logic formed off the path and selected, the window read through
  // one lane-dependent base with immediate offsets (8 ds_read2_b32 instead of 16 address computations and 16 reads), the
  // first sample's "no pending update" as a compile-time case (16 selects per sample fewer): 187 -> 165 instructions per
  // sample, 138 us per frame either way.  Timing experiments: 16 instead of 64 additions per chain -25 us (4.5 cycles
  // per addition), no leak decision -5, no sigma wave -6.  A sample is ONE dependent sequence -- update, product, 64
  // additions, 5 lane swaps, error, two double-precision expressions, select, update -- on one wave; what it costs is that
  // sequence's latency, ~1000 cycles, and neither the instruction count nor the synchronisation.)
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *T = sm, *O = sm + kAnrTile * kAnrRow, *SIG = sm + kAnrSigOff;  // (16-byte aligned)
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0: filter output and update, 1: sum of squares
  const int c = lane & 15, g = lane >> 4, tb = anr_tap_base(g);
  const int ch0 = blockIdx.x * kAnrCw;
  const int nlive = (a.nchan - ch0 < kAnrCw) ? a.nchan - ch0 : kAnrCw;
  const int ch = ch0 + (c < nlive ? c : nlive - 1);  // dead lanes shadow the last live channel (their stores are skipped)
  const size_t nch = (size_t)a.nchan;
  f2 w[8];  // ANR_w (wave 0): this lane's 16 taps, two per register pair: (.x, .y) = taps (tb + 2 t + 1, tb + 2 t)
  float lidx = 0, ngamma = 0;
  if (wv == 0) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
      w[t] = f2{a.anr[(size_t)(kAnrStW + tb + 2 * t + 1) * nch + ch], a.anr[(size_t)(kAnrStW + tb + 2 * t) * nch + ch]};
    _Pragma("unroll 4") for (int r = g; r < kAnrHist; r += 4) T[r * kAnrRow + c] = a.anr[(size_t)(kAnrStHist + r) * nch + ch];
    lidx = a.anr[(size_t)kAnrStLidx * nch + ch], ngamma = a.anr[(size_t)kAnrStNgamma * nch + ch];
  }
  for (int f = 0; f < a.nframes; ++f) {
    // stage the frame in: column cc of the tile = channel ch0 + cc, 256 consecutive samples, 4 x 256 B per channel (the
    // two waves take alternate channels)
    __syncthreads();
_Pragma("nounroll")
    for (int cc = wv; cc < nlive; cc += 2) {
      const float *src = a.aud + ((size_t)(ch0 + cc) * a.nframes + f) * 256;
#pragma unroll
      for (int q = 0; q < 4; ++q) T[(kAnrHist + lane + 64 * q) * kAnrRow + cc] = src[lane + 64 * q];
    }
_Pragma("nounroll")
    for (int cc = nlive + ((nlive ^ wv) & 1); cc < kAnrCw; cc += 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) T[(kAnrHist + lane + 64 * q) * kAnrRow + cc] = 0.0f;
    }
    __syncthreads();
    if (a.nr_option == 3) {  // Process.cpp:852-857: Xanr() as noise reduction; its result stays in float_buffer_R, float_buffer_L is scaled by 1.5
      if (wv == 0) anr_pass_y<false>(T, nullptr, SIG, w, lidx, ngamma, lane);
      else anr_pass_sigma(T, SIG, lane);
      __syncthreads();
      if (wv == 0) {
        if (a.notch) {
          // the notch pass sees the scaled block behind the unscaled one: its delay line = the last 79 unscaled samples
          _Pragma("unroll 4") for (int r = g; r < kAnrHist; r += 4) T[r * kAnrRow + c] = T[(256 + r) * kAnrRow + c];
          __builtin_amdgcn_wave_barrier();
          _Pragma("unroll 4") for (int i = g; i < 256; i += 4) T[(kAnrHist + i) * kAnrRow + c] = T[(kAnrHist + i) * kAnrRow + c] * 1.5f;
        } else {
          _Pragma("unroll 4") for (int i = g; i < 256; i += 4) O[i * kAnrRow + c] = T[(kAnrHist + i) * kAnrRow + c] * 1.5f;
        }
      }
      __syncthreads();
    }
    if (a.notch) {  // Process.cpp:862-866
      if (wv == 0) anr_pass_y<true>(T, O, SIG, w, lidx, ngamma, lane);
      else anr_pass_sigma(T, SIG, lane);
      __syncthreads();
    }
    // the delay line's live part for the next block
    if (wv == 0)
      _Pragma("unroll 4") for (int r = g; r < kAnrHist; r += 4) T[r * kAnrRow + c] = T[(256 + r) * kAnrRow + c];
    __syncthreads();
_Pragma("nounroll")
    for (int cc = wv; cc < nlive; cc += 2) {
      float *dst = a.aud + ((size_t)(ch0 + cc) * a.nframes + f) * 256;
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[lane + 64 * q] = O[(lane + 64 * q) * kAnrRow + cc];
    }
  }
  if (wv == 0 && c < nlive) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      a.anr[(size_t)(kAnrStW + tb + 2 * t + 1) * nch + ch] = w[t].x;
      a.anr[(size_t)(kAnrStW + tb + 2 * t) * nch + ch] = w[t].y;
    }
    _Pragma("unroll 4") for (int r = g; r < kAnrHist; r += 4) a.anr[(size_t)(kAnrStHist + r) * nch + ch] = T[r * kAnrRow + c];
    if (g == 0) {
      a.anr[(size_t)kAnrStLidx * nch + ch] = lidx;
      a.anr[(size_t)kAnrStNgamma * nch + ch] = ngamma;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Kim1_NR() / SpectralNoiseReduction(), one wave per channel
// ------------------------------------------------------------------------------------------
// sum over the wave, the same value in every lane: an inclusive scan inside the VALU (four row shifts, two row
// broadcasts: DPP, no LDS round trips -- the spectral function's gain loop calls this once per bin) and lane 63's total
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<kDppRowShr1, 0xf, true>(0.0f, v);
  v += dpp_f<kDppRowShr2, 0xf, true>(0.0f, v);
  v += dpp_f<kDppRowShr4, 0xf, true>(0.0f, v);
  v += dpp_f<kDppRowShr8, 0xf, true>(0.0f, v);
  v += dpp_f<kDppRowBcast15, 0xa, false>(0.0f, v);
  v += dpp_f<kDppRowBcast31, 0xc, false>(0.0f, v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// the same sum, for its total only: six fused DPP additions (what wave_sum() compiles to spends three instructions on each
// of the two row-stitching steps); same operands, same order
__device__ __forceinline__ float wave_sum_fused(float v) {
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0"
      : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ void lds_fence() { asm volatile("" ::: "memory"); }
constexpr unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
// is the (non-negative) power ratio with bit pattern rb within 5e-5 of one of the values NN changes at?  (Non-negative
// floats order like their bit patterns; NaN patterns sit above every bound: "not near", and NN = 1 like the formula's.)
__device__ __forceinline__ bool nr_ratio_near_edge(unsigned rb) {
  bool near = false;
  for (float b : {0.4f, 0.35f, 0.25f, 0.15f, 0.05f}) near = near || (rb >= fbits(b - 5e-5f) && rb <= fbits(b + 5e-5f));
  return near;
}

template <int KIND>
__global__ __launch_bounds__(64) void nrspec_kernel(const NrArgs a) {
#pragma clang fp contract(off)
  // One region serves, one after the other within a frame, as the FFT's exchange buffer, the spectrum for the partner
  // access, the two weighted spectra and the two output frames: a workgroup is ONE wave, whose LDS operations execute
  // in program order, and each tenant is dead (in registers) before the next one is written.  With the per-bin
  // memories that leaves the spectral function under 10 KiB, i.e. 16 workgroups per CU: the whole 4096-channel batch
  // resident in one round instead of two.
  __shared__ __attribute__((aligned(16))) float U[8 * kFftRow * 2];
  static_assert(8 * kFftRow * 2 >= 520 && 8 * kFftRow * 2 >= 512, "tenants of the shared region");
  float *xbuf = U;                               // FFT exchange
  cf *Zs = reinterpret_cast<cf *>(U);            // [257] the 256-point spectrum of frame 0 + j frame 1 (partner access), [256] = [0]
  cf *W0 = reinterpret_cast<cf *>(U), *W1 = reinterpret_cast<cf *>(U + 260);  // [129] each: weighted, conjugate-symmetrised spectra, bins 0..128
  float *Y0 = U, *Y1 = U + 256;                  // [256] each
  __shared__ float S[384];          // NR_last_sample_buffer_L | the block's 256 samples
  __shared__ float Gb[130];         // gains with one pad either side: Gb[1 + i]
  constexpr int KW = (KIND == 1) ? 128 : 1;  // (Kim1_NR()'s memories only)
  __shared__ float Xs[KIND == 1 ? 3 : 1][KW], Es[KIND == 1 ? 15 : 1][KW];  // Kim1_NR()'s frame histories
  __shared__ float Gts1[KW], Gts0[KW];
  __shared__ float GstP[8 + 128 + 8];  // (padded: the smoothing windows of the edge bins reach 8 below / 4 above)
  float *Gst = GstP + 8;
  __shared__ float Lout[128], Nest[128], Pslp[128], Xt[128], Hk[128];
  const int lane = threadIdx.x;
  const int ch = blockIdx.x;
  if (ch >= a.nchan) return;
  float *st = a.spec + (size_t)ch * kNrSpecFloats;
  const float *win = a.tab_nr + (KIND == 1 ? kNrTabHann : kNrTabSqrtHann);
  const cf *tab = reinterpret_cast<const cf *>(a.tab);
  cf tw1[7], tw2[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    tw1[q] = tab[kTabTw1 + 64 * q + lane];
    tw2[q] = tab[kTabTw2 + 64 * q + lane];
  }
  // ---- the channel's record -> LDS
  if (KIND == 1) {
    for (int i = lane; i < 3 * 128; i += 64) (&Xs[0][0])[i] = st[kNrX + i];
    for (int i = lane; i < 15 * 128; i += 64) (&Es[0][0])[i] = st[kNrE + i];
  }
  if (lane < 8) {
    GstP[lane] = 0.0f;
    GstP[8 + 128 + lane] = 0.0f;
  }
  for (int i = lane; i < 128; i += 64) {
    if (KIND == 1) {
      Gts1[i] = st[kNrGts1 + i];
      Gts0[i] = st[kNrGts0 + i];
    }
    Gst[i] = st[kNrG + i];
    S[i] = st[kNrLastIn + i];
    Lout[i] = st[kNrLastOut + i];
    Nest[i] = st[kNrNest + i];
    Pslp[i] = st[kNrPslp + i];
    Xt[i] = st[kNrXt + i];
    Hk[i] = st[kNrHk + i];
  }
  float lastX[2] = {0.0f, 0.0f};
  int xp = (int)st[kNrScal + 0], ep = (int)st[kNrScal + 1], first_time_2 = (int)st[kNrScal + 2], init_counter = (int)st[kNrScal + 3];
  const int lo = a.vad_lo, hi = a.vad_hi;
  const float NR_alpha = a.alpha, NR_beta = a.beta, NR_PSI = a.psi;
  __syncthreads();

  // SpectralNoiseReduction()'s per-call constants and statics (Noise.cpp:394-419)
  const float tinc = 0.00533333, tax = 0.0239, tap = 0.05062, psthr = 0.99, pnsaf = 0.01, asnr = 20, psini = 0.5, pspri = 0.5;
  const float ax = expf(-tinc / tax), ap = expf(-tinc / tap);
  const float xih1 = powf(10, (float)asnr / 10.0);
  const float xih1r = 1.0 / (1.0 + xih1) - 1.0;
  const float pfac = (1.0 / pspri - 1.0) * (1.0 + xih1);
  const float snr_prio_min = powf(10, -(float)20 / 20.0);
  const float power_threshold = 0.4;
  const int NR_width = 4;

  if (KIND == 2 && first_time_2 == 1) {  // Noise.cpp:439-450
    for (int i = lane; i < 128; i += 64) {
      S[i] = 0.0;
      Gst[i] = 1.0;
      Hk[i] = 1.0;
      Nest[i] = 0.0;
      Pslp[i] = 0.5;
    }
    first_time_2 = 2;
    __syncthreads();
  }

  float *gio = a.aud + (size_t)ch * a.nframes * 256;
  for (int f = 0; f < a.nframes; ++f) {
    // ---- the block, behind the previous block's second half
#pragma unroll
    for (int q = 0; q < 4; ++q) S[128 + lane + 64 * q] = gio[(size_t)f * 256 + lane + 64 * q];
    __syncthreads();
    // ---- both frames through one transform: z[n] = w[n] (S[n] + j S[128 + n]), zero-interleaved
    cf v[8];
    {
      const int m = lane >> 1;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int n = m + 32 * r;
        const float wn = win[n];
        const cf z = cf{S[n] * wn, S[128 + n] * wn};
        v[r] = (lane & 1) ? cf{0.0f, 0.0f} : z;
      }
    }
    __syncthreads();
    if (lane < 64) {  // the second half becomes NR_last_sample_buffer_L
      S[lane] = S[256 + lane];
      S[64 + lane] = S[320 + lane];
    }
    fft512<false>(v, tw1, tw2, xbuf, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) Zs[lane + 64 * r] = v[r];
    if (lane == 0) Zs[256] = v[0];
    __syncthreads();
    // ---- separate: F0[k] = (Z[k] + conj Z[256-k]) / 2, F1[k] = (Z[k] - conj Z[256-k]) / 2j; bins lane, lane + 64
    cf F[2][2];     // [frame][r]
    float X[2][2];  // squared magnitudes
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int k = lane + 64 * r;
      const cf za = Zs[k], zp = Zs[256 - k];
      F[0][r] = cf{0.5f * (za.x + zp.x), 0.5f * (za.y - zp.y)};
      F[1][r] = cf{0.5f * (za.y + zp.y), -0.5f * (za.x - zp.x)};
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) X[k2][r] = F[k2][r].x * F[k2][r].x + F[k2][r].y * F[k2][r].y;
    }
    const cf z128 = Zs[128];  // F0[128] = Re, F1[128] = Im
    if (KIND == 2) {  // NR_X[bindx][0] as the call's last half-block leaves it (Noise.cpp:478), for the record
      lastX[0] = X[1][0];
      lastX[1] = X[1][1];
    }
    bool proc[2] = {true, true};
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      float G[2] = {0.0f, 0.0f};  // this frame's NR_G for my two bins
      if (KIND == 1) {
        // ---- Kim1_NR(), Noise.cpp:197-257
        const float NR_KIM_K = 1.0;
        const float NR_onemalpha = (1.0 - NR_alpha);
        const float NR_onemtwobeta = (1.0 - (2.0 * NR_beta));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int i = lane + 64 * r;
          Xs[xp][i] = X[k2][r];
          if (i >= lo && i < hi) {
            float NR_sum = 0.0;
            NR_sum = NR_sum + Xs[0][i];
            NR_sum = NR_sum + Xs[1][i];
            NR_sum = NR_sum + Xs[2][i];
            const float E = NR_sum / (float)3;
            Es[ep][i] = E;
            float M = Es[0][i];
            for (int j = 1; j < 15; ++j)
              if (Es[j][i] < M) M = Es[j][i];
            const float NR_T = X[k2][r] / M;
            const float lambda = (NR_T > NR_PSI) ? M : E;
            float g = (float)(1.0 - (double)(lambda * NR_KIM_K / E));
            if (g < 0.0f) g = 0.0f;
            const float gts = NR_alpha * Gts1[i] + NR_onemalpha * g;
            Gts0[i] = gts;
            Gts1[i] = gts;
          }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int i = lane + 64 * r;
          if (i == 0) G[r] = (NR_onemtwobeta + NR_beta) * Gts0[0] + NR_beta * Gts0[1];
          else if (i == 127) G[r] = NR_beta * Gts0[126] + (NR_onemtwobeta + NR_beta) * Gts0[127];
          else G[r] = NR_beta * Gts0[i - 1] + NR_onemtwobeta * Gts0[i] + NR_beta * Gts0[i + 1];
        }
        xp = (xp + 1 >= 3) ? 0 : xp + 1;
        ep = (ep + 1 >= 15) ? 0 : ep + 1;
      } else {
        // ---- SpectralNoiseReduction(), Noise.cpp:476-601
        if (first_time_2 == 2) {
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int i = lane + 64 * r;
            const float ne = (float)((double)Nest[i] + 0.05 * (double)X[k2][r]);
            Nest[i] = ne;
            Xt[i] = psini * ne;
          }
          init_counter++;
          if (init_counter > 19) {
            init_counter = 0;
            first_time_2 = 3;
          }
        }
        proc[k2] = (first_time_2 == 3);
        if (first_time_2 == 3) {
          float post[2], prio[2];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int i = lane + 64 * r;
            const float x = X[k2][r];
            float xt = Xt[i];
            float ph1y = (float)(1.0 / (1.0 + (double)(pfac * expf(xih1r * x / xt))));
            const float ps = (float)((double)(ap * Pslp[i]) + (1.0 - (double)ap) * (double)ph1y);
            Pslp[i] = ps;
            if (ps > psthr) ph1y = (float)(1.0 - (double)pnsaf);
            else ph1y = (float)fmin((double)ph1y, 1.0);
            const float xtr = (float)((1.0 - (double)ph1y) * (double)x + (double)(ph1y * xt));
            xt = (float)((double)(ax * xt) + (1.0 - (double)ax) * (double)xtr);
            Xt[i] = xt;
            post[r] = (float)fmax(fmin((double)(x / xt), 1000.0), (double)snr_prio_min);
            prio[r] = (float)fmax((double)(NR_alpha * Hk[i]) + (1.0 - (double)NR_alpha) * fmax((double)post[r] - 1.0, 0.0), 0.0);
          }
          // the gain each bin receives in its own pass of the loop below depends on its two SNRs alone (Noise.cpp:531-535):
          // computed here for both of the lane's bins, so that the bin-by-bin loop only moves values
          float gnew[2], hknew[2];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const float vv = (float)((double)(prio[r] * post[r]) / (1.0 + (double)prio[r]));
            gnew[r] = (float)(1.0 / (double)post[r] * (double)sqrtf((float)(0.7212 * (double)vv + (double)(vv * vv))));
            hknew[r] = post[r] * gnew[r] * gnew[r];
          }
          __syncthreads();
          // pre_power: the same sum in every pass of the loop below
          float pre_part = 0.0f;
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int i = lane + 64 * r;
            if (i >= lo && i < hi) pre_part += X[k2][r];
          }
          const float pre_power = wave_sum(pre_part);
#if T41RX_NRSPEC_LDS_LOOP
          for (int i = lo; i < hi; ++i) {  // Noise.cpp:529-588: the musical-noise treatment runs inside this loop
            if (lane == (i & 63)) {
              Gst[i] = (i >> 6) ? gnew[1] : gnew[0];
              Hk[i] = (i >> 6) ? hknew[1] : hknew[0];
            }
            __syncthreads();
            float post_part = 0.0f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const int j = lane + 64 * r;
              if (j >= lo && j < hi) post_part += Gst[j] * Gst[j] * X[k2][r];
            }
            const float post_power = wave_sum(post_part);
            float power_ratio = post_power / pre_power;
            {
              bool near_edge = false;
#pragma unroll
              for (float b : {0.4f, 0.35f, 0.25f, 0.15f, 0.05f}) near_edge = near_edge || fabsf(power_ratio - b) < 4e-5f;
              if (near_edge) {
                float *Pw = U + 640;  // (free between the transforms: the spectra occupy U[0 .. 520))
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                  const int j = lane + 64 * r;
                  Pw[j] = X[k2][r];
                  Pw[128 + j] = Gst[j] * Gst[j] * X[k2][r];
                }
                __syncthreads();
                float pre_seq = 0.0f, post_seq = 0.0f;
                for (int j = lo; j < hi; ++j) {
                  pre_seq += Pw[j];
                  post_seq += Pw[128 + j];
                }
                power_ratio = post_seq / pre_seq;
                __syncthreads();
              }
            }
            int NN;
            if (power_ratio > power_threshold) {
              power_ratio = 1.0;
              NN = 1;
            } else {
              NN = 1 + 2 * (int)(0.5 + (double)NR_width * (1.0 - (double)(power_ratio / power_threshold)));
            }
            if (NN > 1) {
              const int h = NN / 2;
              float nv[2] = {0.0f, 0.0f};
              bool mine[2] = {false, false};
#pragma unroll
              for (int r = 0; r < 2; ++r) {
                const int j = lane + 64 * r;
                if (j >= lo + h && j < hi - h) {
                  mine[r] = true;
                  float s = 0.0;
                  if (j >= hi - NN) {
                    for (int m = j; m > j - NN; --m) s += Gst[m];
                  } else {
                    for (int m = j - h; m <= j + h; ++m) s += Gst[m];
                  }
                  nv[r] = s / (float)NN;
                }
              }
              __syncthreads();
#pragma unroll
              for (int r = 0; r < 2; ++r)
                if (mine[r]) Gst[lane + 64 * r] = nv[r];
            }
            __syncthreads();
          }
#else
          // Round 5.  The loop is serial by construction (every pass sees the gains the previous pass's smoothing left)
          // and the four waves of a SIMD keep its VALU busy: what a pass costs is its VALU instructions.  Round 4 kept
          // the gains in LDS and spent ~140 of them per pass; this form spends ~40:
          //  * the lane's two gains live in registers (Gr); bins 64..127 are left out of the loop when the pass band
          //    ends below bin 64 (wave-uniform: it does for every filter up to 6 kHz);
          //  * a pass that smooths publishes the gains once and requests its whole window in one go from a padded array
          //    (no address clamps), the centred and the upper-edge average are summed from registers in the reference's
          //    order, and s / NN is the correctly rounded quotient from a multiplication and two FMAs (NN is 3, 5, 7 or 9
          //    and 1 / NN is correctly rounded: Markstein's theorem; checked on 64 M quotients per divisor);
          //  * the power ratio only ever decides NN, which changes at 0.4, 0.35, 0.25, 0.15 and 0.05: away from those
          //    (5e-5, bit patterns compared on the scalar unit) post_power x (1 / pre_power) decides, next to one of
          //    them round 4's decision code runs as it was (true quotient, the reference's summation order), so NN is
          //    decided exactly as before;
          //  * the wave sum's six steps are six fused DPP additions.
          // The same additions in the same order as the reference's loops: outputs and records are bit-identical to
          // round 4's (tools/anr_ab_check.py).
          float Gr[2] = {Gst[lane], Gst[64 + lane]};
          const bool r1_live = hi > 64;
          const bool in0 = lane >= lo && lane < hi, in1 = lane + 64 >= lo && lane + 64 < hi;
          const float inv_pre = 1.0f / pre_power;
          for (int i = lo; i < hi; ++i) {  // Noise.cpp:529-588: the musical-noise treatment runs inside this loop
            {
              const bool me = lane == (i & 63);
              if (i >> 6) Gr[1] = me ? gnew[1] : Gr[1];
              else Gr[0] = me ? gnew[0] : Gr[0];
            }
            float post_part = in0 ? Gr[0] * Gr[0] * X[k2][0] : 0.0f;
            if (r1_live) post_part = in1 ? post_part + Gr[1] * Gr[1] * X[k2][1] : post_part;
            const float post_power = wave_sum_fused(post_part);
            int NN;
            // (read into a scalar register by hand: through the builtin the compiler keeps the value in a VGPR and does
            // the ten comparisons below on the VALU)
            unsigned rb;
            asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1" : "=s"(rb) : "v"(post_power * inv_pre));
            if (!nr_ratio_near_edge(rb)) {
              NN = rb > fbits(0.35f) ? 1 : rb > fbits(0.25f) ? 3 : rb > fbits(0.15f) ? 5 : rb > fbits(0.05f) ? 7 : 9;
            } else {
              float power_ratio = post_power / pre_power;
              // The reference adds both sums up bin by bin in float (Noise.cpp:542-546); the tree sums above differ from
              // that by rounding only.  Next to an edge the sums are redone in the reference's order (every lane, the same
              // values through LDS), so NN is decided exactly as the scalar code decides it.
              bool near_edge = false;
#pragma unroll
              for (float b : {0.4f, 0.35f, 0.25f, 0.15f, 0.05f}) near_edge = near_edge || fabsf(power_ratio - b) < 4e-5f;
              if (near_edge) {
                float *Pw = U + 640;  // (free between the transforms: the spectra occupy U[0 .. 520))
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                  const int j = lane + 64 * r;
                  Pw[j] = X[k2][r];
                  Pw[128 + j] = Gr[r] * Gr[r] * X[k2][r];
                }
                __syncthreads();
                float pre_seq = 0.0f, post_seq = 0.0f;
                for (int j = lo; j < hi; ++j) {
                  pre_seq += Pw[j];
                  post_seq += Pw[128 + j];
                }
                power_ratio = post_seq / pre_seq;
                __syncthreads();
              }
              if (power_ratio > power_threshold) {
                power_ratio = 1.0;
                NN = 1;
              } else {
                NN = 1 + 2 * (int)(0.5 + (double)NR_width * (1.0 - (double)(power_ratio / power_threshold)));
              }
            }
            if (NN > 1) {
              // centred average over NN bins; the reference's upper-edge pass (a backward average) overwrites the
              // centred value of the last NN - NN/2 inner bins before the copy back (Noise.cpp:556-584)
              Gst[lane] = Gr[0];
              if (r1_live) Gst[64 + lane] = Gr[1];
              lds_fence();  // (one wave: its LDS operations execute in program order)
              if (NN <= 9) {
                auto smooth = [&](auto nn_tag) {
                  constexpr int N = decltype(nn_tag)::value, H = N / 2;
                  constexpr float rN = 1.0f / (float)N;
#pragma unroll
                  for (int r = 0; r < 2; ++r) {
                    if (r == 1 && !r1_live) continue;
                    const int j = lane + 64 * r;
                    float g[N + H];  // g[t] = NR_G[j - N + 1 + t]
#pragma unroll
                    for (int t = 0; t < N + H; ++t) g[t] = Gst[j - N + 1 + t];
                    float sc = 0.0, sb = 0.0;
#pragma unroll
                    for (int m = -H; m <= H; ++m) sc += g[N - 1 + m];     // m = j - h .. j + h
#pragma unroll
                    for (int m = 0; m > -N; --m) sb += g[N - 1 + m];      // m = j .. j - NN + 1
                    const float sm = (j >= hi - N) ? sb : sc;
                    const float q0 = sm * rN;
                    const float nv = fmaf(fmaf(-q0, (float)N, sm), rN, q0);  // = sm / (float)N, correctly rounded
                    if (j >= lo + H && j < hi - H) Gr[r] = nv;
                  }
                };
                switch (NN) {
                  case 3: smooth(std::integral_constant<int, 3>{}); break;
                  case 5: smooth(std::integral_constant<int, 5>{}); break;
                  case 7: smooth(std::integral_constant<int, 7>{}); break;
                  default: smooth(std::integral_constant<int, 9>{}); break;
                }
              } else {  // (not reachable with NR_width = 4 and a ratio >= 0; kept as the reference's loops)
                Gst[64 + lane] = Gr[1];
                lds_fence();
                const int h = NN / 2;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                  const int j = lane + 64 * r;
                  if (j >= lo + h && j < hi - h) {
                    float sm = 0.0;
                    if (j >= hi - NN) {
                      for (int m = j; m > j - NN; --m) sm += Gst[m];
                    } else {
                      for (int m = j - h; m <= j + h; ++m) sm += Gst[m];
                    }
                    Gr[r] = sm / (float)NN;
                  }
                }
              }
              lds_fence();
            }
          }
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int i = lane + 64 * r;
            Gst[i] = Gr[r];
            if (i >= lo && i < hi) Hk[i] = hknew[r];
          }
          __syncthreads();
#endif
#pragma unroll
          for (int r = 0; r < 2; ++r) G[r] = Gst[lane + 64 * r] * 1.0f;  // x NR_long_tone_gain (1.0, Noise.cpp:706)
        }
      }
      // ---- this frame's weighted spectrum, conjugate-symmetrised: bin k by (G[k] + G[k-1]) / 2 (bin 0: G[0]; 128: G[127])
      __syncthreads();
      Gb[1 + lane] = G[0];
      Gb[65 + lane] = G[1];
      __syncthreads();
      cf *Wk = k2 ? W1 : W0;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int k = lane + 64 * r;
        const float g = (k == 0) ? Gb[1] : 0.5f * (Gb[1 + k] + Gb[k]);
        Wk[k] = (k == 0) ? cf{F[k2][r].x * g, 0.0f} : cf{F[k2][r].x * g, F[k2][r].y * g};
      }
      if (lane == 0) Wk[128] = cf{(k2 ? z128.y : z128.x) * Gb[128], 0.0f};
      __syncthreads();
    }
    // ---- one inverse transform for both frames: V[k] = W0[k] + j W1[k], V[256 - k] = conj W0[k] + j conj W1[k]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = lane + 64 * r;
      const int kk = (k <= 128) ? k : 256 - k;
      const cf a0 = W0[kk], a1 = W1[kk];
      const float sg = (k <= 128) ? 1.0f : -1.0f;
      // a0 (+-conj) + j a1 (+-conj): re = a0.x - sg a1.y, im = sg a0.y + a1.x
      v[r] = cf{a0.x - sg * a1.y, sg * a0.y + a1.x};
      v[r + 4] = v[r];
    }
    fft512<true>(v, tw1, tw2, xbuf, lane);
    if (!(lane & 1)) {
      const int m = lane >> 1;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int n = m + 32 * r;
        const float sw = (KIND == 2) ? win[n] : 1.0f;  // the spectral function windows again after the inverse transform
        Y0[n] = v[r].x * (1.0f / 512.0f) * sw;
        Y1[n] = v[r].y * (1.0f / 512.0f) * sw;
      }
    }
    __syncthreads();
    // ---- overlap-add (Noise.cpp:289-296 / 640-647), Kim: x 30 (Process.cpp:846)
    {
      float o[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int i = lane + 64 * q;
        const float l0 = Lout[i];
        float first, l1;
        if (proc[0]) {
          first = Y0[i] + l0;
          l1 = Y0[128 + i];
        } else {
          first = gio[(size_t)f * 256 + i];
          l1 = l0;
        }
        float second, l2;
        if (proc[1]) {
          second = Y1[i] + l1;
          l2 = Y1[128 + i];
        } else {
          second = gio[(size_t)f * 256 + 128 + i];
          l2 = l1;
        }
        Lout[i] = l2;
        o[q] = (KIND == 1) ? first * 30.0f : first;
        o[2 + q] = (KIND == 1) ? second * 30.0f : second;
      }
      gio[(size_t)f * 256 + lane] = o[0];
      gio[(size_t)f * 256 + 64 + lane] = o[1];
      gio[(size_t)f * 256 + 128 + lane] = o[2];
      gio[(size_t)f * 256 + 192 + lane] = o[3];
    }
    __syncthreads();
  }
  // ---- the record back
  if (KIND == 1) {
    for (int i = lane; i < 3 * 128; i += 64) st[kNrX + i] = (&Xs[0][0])[i];
    for (int i = lane; i < 15 * 128; i += 64) st[kNrE + i] = (&Es[0][0])[i];
  } else if (a.nframes > 0) {
    // the spectral function stores NR_X[.][0] every frame and never touches NR_X[.][1..2] or NR_E: what Kim1_NR()'s
    // three-frame average starts from after a live switch of nrOptionSelect from 2 to 1, as in the firmware
    st[kNrX + lane] = lastX[0];
    st[kNrX + 64 + lane] = lastX[1];
  }
  for (int i = lane; i < 128; i += 64) {
    if (KIND == 1) {
      st[kNrGts1 + i] = Gts1[i];
      st[kNrGts0 + i] = Gts0[i];
    }
    st[kNrG + i] = Gst[i];
    st[kNrLastIn + i] = S[i];
    st[kNrLastOut + i] = Lout[i];
    st[kNrNest + i] = Nest[i];
    st[kNrPslp + i] = Pslp[i];
    st[kNrXt + i] = Xt[i];
    st[kNrHk + i] = Hk[i];
  }
  if (lane == 0) {
    st[kNrScal + 0] = (float)xp;
    st[kNrScal + 1] = (float)ep;
    st[kNrScal + 2] = (float)first_time_2;
    st[kNrScal + 3] = (float)init_counter;
  }
}

hipError_t launch_nr(const NrArgs &a, hipStream_t s) {
  if (a.nr_option == 1)
    hipLaunchKernelGGL((nrspec_kernel<1>), dim3(a.nchan), dim3(64), 0, s, a);
  else if (a.nr_option == 2)
    hipLaunchKernelGGL((nrspec_kernel<2>), dim3(a.nchan), dim3(64), 0, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (a.nr_option == 3 || a.notch) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&anr_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAnrLdsBytes);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(anr_kernel, dim3((a.nchan + kAnrCw - 1) / kAnrCw), dim3(128), kAnrLdsBytes, s, a);
    e = hipGetLastError();
  }
  return e;
}

}  // namespace t41

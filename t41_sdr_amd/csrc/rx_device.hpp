// t41_sdr_amd/csrc/rx_device.hpp -- gfx950 (MI355X / CDNA4) kernels of the T41 RX hot path: the device helpers every
// kernel family shares (this file), the serial chains (rx_chains.hpp: AGC gain law, SAM PLL, their pipelined forms), the
// fused FFT_LENGTH 512 kernel (rx512_kernel.hpp), the long-FFT fast convolution (fastconv_kernels.hpp).  One translation
// unit per kernel family instantiates them (rx512_ssb / _am / _nfm / _sam.hip, rx_long.hip, fastconv.hip, display_kernel.hip;
// rx_dispatch.hip picks the launcher), so a change to one family rebuilds that family only.
//
// One 64-lane wavefront runs ProcessIQData() (Process.cpp:70-944) for one channel, every stage
// fused, for ALL the frames of a launch: 2048 complex f32 samples in -> 2048 real f32 samples out
// per frame, so HBM sees each frame once (16 KiB in + 8 KiB out).  FFT_LENGTH 512, AGC off: one
// 16-wave workgroup per CU owns all 160 KiB of LDS and the channel's ~3.8 KiB streaming state
// stays on chip (LDS + registers) between the first and the last frame of the launch (Geo<0>);
// one workgroup barrier, for the shared twiddle staging; each wave otherwise synchronises with
// itself only (LDS is in-order per wave).  AGC on / SAM / the long-FFT part kernels: 4-wave
// workgroups, four per CU, state through HBM per frame, two more barriers per frame around the
// serial gain law / PLL.  16 waves per CU x 256 CUs = 4096 channels in flight = BASELINE
// config 2's batch.
//
// Stage map (reference file:line -> code below):
//   gains, DC high-pass           Process.cpp:117-134        front_end()   (parallel affine scan)
//   IQ amp/phase correction       Process.cpp:165-173        front_end()
//   Fs/4 shift (x j^n)            Freq_Shift.cpp:42-65       front_end()   (register renaming)
//   quadrature NCO mix            Freq_Shift.cpp:94-141      front_end()   (fixed-point phase)
//   decimate /4 (28 taps)         Process.cpp:474-475        dec1 section  (polyphase via LDS)
//   decimate /2 (46 taps)         Process.cpp:478-479        dec2 section
//   level adjust                  Process.cpp:481-492
//   overlap-save + 512-pt FFT     Process.cpp:498-535        fft512<false> (radix-8 x3, in regs)
//   x FIR_filter_mask             Process.cpp:547
//   inverse FFT                   Process.cpp:595            fft512<true>
//   AGC off (fixed gain) / on     DSP_Fn.cpp:494-502 / 504-631  agc_apply(), agc_chain()
//   SSB / AM / NFM demod          Process.cpp:616-624,688-694 / 697-707 / 716-727,765-816
//   interpolate x2 (48 taps)      Process.cpp:917            int1 section
//   interpolate x4 (32 taps)      Process.cpp:920            int2 section (lane shuffles)
//   volume                        Process.cpp:929
//   (q15 samples either side)     Process.cpp:102-111, 936-937  WQ15 kernels
// FFT_LENGTH 1024 / 2048 / 4096: the same kernel split in two (PART 1 / 2) around fastconv_kernel<R>.
//
// No MFMA: FIR taps and FFT butterflies are not dense contractions (BASELINE north_star).
//
// Arithmetic is packed FP32 throughout: measured on MI355X (tools/ubench/valu_rate.hip) a
// wave64 v_fma_f32 and a v_pk_fma_f32 both issue once per ~4 cycles per SIMD, so the packed
// form does twice the work per issue slot (72 vs 140 TFLOP/s).  Every stage therefore works on
// (I, Q) / (re, im) / (even, odd) register pairs: one v_pk_fma_f32 per complex FIR tap with the
// tap broadcast from an SGPR via op_sel, two packed instructions per complex multiply
// (op_sel / neg modifiers, inline asm because hipcc does not fold the swizzles), and
// one per complex add or +-j rotation.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "rx_experiments.hpp"
#include "rx_internal.hpp"
#include "rx_kernels.hpp"
#include "wave_fft.hpp"

// T41RX_AGC_GROUPMAX=1: the look-ahead window's 23 group maxima requested in two batches instead of pair by pair
// (round 5: twelve LDS round trips fewer per frame and no faster, profiles/r05_ab_saddr4_modes.txt; off)
#ifndef T41RX_AGC_GROUPMAX
#define T41RX_AGC_GROUPMAX 0
#endif
// T41RX_KEEP_PL=0: the lane's part of the oscillator phases worked out per frame (A/B builds)
#ifndef T41RX_KEEP_PL
#define T41RX_KEEP_PL 1
#endif
// T41RX_DEC1_REGTAIL=0: the /4 decimator reads its whole window from LDS, the lane's own eight samples included (A/B builds)
#ifndef T41RX_DEC1_REGTAIL
#define T41RX_DEC1_REGTAIL 1
#endif
// T41RX_DEC1_REGTAIL_AGC=1: the same in the kernels with the AGC (measured: neutral to 1 % slower)
#ifndef T41RX_DEC1_REGTAIL_AGC
#define T41RX_DEC1_REGTAIL_AGC 0
#endif
// T41RX_SAM_SCAN_FUSED / T41RX_SAM_KEEP_PL = 1: the fused scan steps / the kept phase product in the SAM kernel (without the AGC) too
// (measured: 56.1 -> 56.9 -> 58.7 us per frame with one, with both: the registers cost more than the instructions save)
#ifndef T41RX_SAM_SCAN_FUSED
#define T41RX_SAM_SCAN_FUSED 0
#endif
#ifndef T41RX_SAM_KEEP_PL
#define T41RX_SAM_KEEP_PL 0
#endif
// T41RX_WRITELANE=0: a scalar goes into one lane of a register by move + compare + select (A/B builds)
#ifndef T41RX_WRITELANE
#define T41RX_WRITELANE 1
#endif
// T41RX_SCAN_DPP=0: the DC high-pass scan's steps inside the rows as moves + packed multiply-adds (A/B builds)
#ifndef T41RX_SCAN_DPP
#define T41RX_SCAN_DPP 1
#endif

namespace t41 {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
// (T41RX_ABLATE / T41RX_LOO / T41RX_CUT: timing experiments with wrong results, rx_experiments.hpp)
// Fused kernel: how many of the next frame's sub-blocks are requested during the current frame
// (0: none, 1: sub-block 0 + the I tail, 2: sub-blocks 0 and 1 + the I tail).  Each one costs 16
// registers that stay live through the back end.
#ifndef T41RX_PIPE_PF
#define T41RX_PIPE_PF 0  // AGC: 2 is 2.4 % slower (212 bytes of scratch per lane, spilled and reloaded every frame), 1 (108 bytes)
#endif                   // and 0 (28 bytes) run alike, and 0 moves 1.31 instead of 1.55 x the algorithmic bytes
#ifndef T41RX_PIPE_PF_SAM
#define T41RX_PIPE_PF_SAM 1  // (the synchronous detector: 0 is 2 % slower)
#endif
#ifndef T41RX_PF
#define T41RX_PF 2
#endif

// Issue priority falls as a wave advances through its frame (3: loads, mixer, decimators; 2:
// FFTs and demodulator; 0: interpolators and stores), so the waves sharing a SIMD progress evenly
// instead of oldest-first, which left each SIMD with one or two latency-bound waves for the last
// third of the launch (per-wave end times from the -DT41RX_STAMP build: 24 .. 35 us within every
// CU).  Measured: 33.8 -> 31.6 us; eight other schedules tried, rising priorities lose 0.2 us.
// (Multi-frame launches, measured: no priorities at all +1.5 %, a priority per (wave, frame)
// rotating over the waves of a SIMD +0.5 %, start offsets between the waves of a CU up to a whole
// frame period +-0.5 %: the waves spread over the frame by themselves within a few frames.)
#define PRIO(n) __builtin_amdgcn_s_setprio(n)
// (round 5, measured and left off: a static bias by wave age.  The arbiter favours the oldest wave of a SIMD; stamps give the
//  four generations of a 16-wave workgroup lifetimes of 627 / 645 / 664 / 673 us in a 693 us launch, profiles/r05_wave_spread.txt.
//  1: the younger half one level up in the last two thirds of the frame; 2: the older half one level down in the first two.)
#ifndef T41RX_PRIO_AGE
#define T41RX_PRIO_AGE 0
#endif
#ifndef T41RX_FRESH
#define T41RX_FRESH 1
#endif
// experiment: a 1024-float output transposition in two halves (64-byte store segments)
#ifndef T41RX_X_HALFTR
#define T41RX_X_HALFTR 0
#endif
#if T41RX_FRESH
#define FRESH_LANE() asm volatile("" : "+v"(lane))
#else
#define FRESH_LANE() do {} while (0)
#endif
// The same for the wave index (round 5, the long-FFT one-kernel form): hipcc hoists every `row base + k * 0x100` derived from it out
// of the frame loop as a loop invariant, runs out of SGPRs, spills the lot to VGPR lanes and reads each back with a
// v_readlane_b32 -- a VALU instruction for what one s_add_i32 recomputes for free.
#ifndef T41RX_FRESH_WV
#define T41RX_FRESH_WV 1
#endif
#if T41RX_FRESH_WV
#define FRESH_WV(w) asm volatile("" : "+s"(w))
#else
#define FRESH_WV(w) do {} while (0)
#endif

// Diagnostic build only (-DT41RX_STAMP): s_memtime stamps at phase boundaries; lane p of each wave
// accumulates the cycles of phase p and writes them behind the demod debug tap at the end.
// The stamps drain lgkmcnt, so read the SHARES, not the total.
#ifdef T41RX_STAMP
#define STAMP(p)                                                                     \
  do {                                                                               \
    unsigned long long t_;                                                           \
    __builtin_amdgcn_sched_barrier(0);                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                               \
    if (lane == (p)) stamp_acc += t_ - stamp_last;                                   \
    stamp_last = t_;                                                                 \
  } while (0)
#define STAMP_PARAMS , unsigned long long &stamp_acc, unsigned long long &stamp_last
#define STAMP_ARGS , stamp_acc, stamp_last
#else
#define STAMP(p) do {} while (0)
#define STAMP_PARAMS
#define STAMP_ARGS
#endif

// The same with the twiddles read from the workgroup's LDS tables right where they are used
// (tw1l: this lane's column of tw1[7][64]; tw2l: its column of the compacted tw2[7][8]) instead of
// held in 28 registers across both transforms: the fused kernel carries the next frame's input
// prefetch through its back end and has no registers to spare.  `mid()` runs between the first and
// the second stage (the fused kernel requests the filter mask there).
template <bool INV, typename MID>
__device__ __forceinline__ void fft512_ldstw(cf (&v)[8], const cf *tw1l, const cf *tw2l, float *__restrict__ xbuf,
                                             int lane, MID mid) {
  cf *xb = reinterpret_cast<cf *>(xbuf);
  dft8<INV>(v);
#pragma unroll
  for (int q = 1; q < 8; ++q) {
    const cf w = tw1l[64 * (q - 1)];
    v[q] = INV ? cmulc(v[q], w) : cmul(v[q], w);
  }
#if T41RX_FFT_X1_PERM
  fft_exchange1_perm(v);
  mid();
#else
  wave_sync();
#pragma unroll
  for (int q = 0; q < 8; ++q) xb[q * kFftRow + lane] = v[q];
  wave_sync();
  mid();
  {
    const int l1 = lane & 7, q = lane >> 3;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) v[k2] = xb[q * kFftRow + l1 + 8 * k2];
  }
#endif
  dft8<INV>(v);
#pragma unroll
  for (int q = 1; q < 8; ++q) {
    const cf w = tw2l[8 * (q - 1)];
    v[q] = INV ? cmulc(v[q], w) : cmul(v[q], w);
  }
  wave_sync();
  {
    const int l1 = lane & 7, q = lane >> 3;
#pragma unroll
    for (int q2 = 0; q2 < 8; ++q2) xb[q2 * kFftRow + q + fft_x2(l1)] = v[q2];
  }
  wave_sync();
  {
    const int q = lane & 7, q2 = lane >> 3;
#pragma unroll
    for (int l1 = 0; l1 < 8; ++l1) v[l1] = xb[q2 * kFftRow + q + fft_x2(l1)];
  }
  dft8<INV>(v);
}

// Two independent 512-point transforms in lockstep (the fast convolution's two rows per wave): the
// second one's arithmetic fills the first one's LDS round trips -- a wave alone on its SIMD issues
// nothing while it waits for an exchange, and that kernel is bound by exactly those waits.
template <bool INV>
__device__ __forceinline__ void fft512_ldstw_x2(cf (&v)[8], cf (&u)[8], const cf *tw1l, const cf *tw2l,
                                                float *__restrict__ xbuf_v, float *__restrict__ xbuf_u, int lane) {
  cf *xv = reinterpret_cast<cf *>(xbuf_v), *xu = reinterpret_cast<cf *>(xbuf_u);
  const int l1 = lane & 7, q3 = lane >> 3;
  dft8<INV>(v);
  dft8<INV>(u);
#pragma unroll
  for (int q = 1; q < 8; ++q) {
    const cf w = tw1l[64 * (q - 1)];
    v[q] = INV ? cmulc(v[q], w) : cmul(v[q], w);
    u[q] = INV ? cmulc(u[q], w) : cmul(u[q], w);
  }
#if T41RX_FFT_X1_PERM
  fft_exchange1_perm(v);
  fft_exchange1_perm(u);
#else
  wave_sync();
#pragma unroll
  for (int q = 0; q < 8; ++q) xv[q * kFftRow + lane] = v[q];
#pragma unroll
  for (int q = 0; q < 8; ++q) xu[q * kFftRow + lane] = u[q];
  wave_sync();
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) v[k2] = xv[q3 * kFftRow + l1 + 8 * k2];
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) u[k2] = xu[q3 * kFftRow + l1 + 8 * k2];
#endif
  dft8<INV>(v);
  dft8<INV>(u);
#pragma unroll
  for (int q = 1; q < 8; ++q) {
    const cf w = tw2l[8 * (q - 1)];
    v[q] = INV ? cmulc(v[q], w) : cmul(v[q], w);
    u[q] = INV ? cmulc(u[q], w) : cmul(u[q], w);
  }
  wave_sync();
#pragma unroll
  for (int q2 = 0; q2 < 8; ++q2) xv[q2 * kFftRow + q3 + fft_x2(l1)] = v[q2];
#pragma unroll
  for (int q2 = 0; q2 < 8; ++q2) xu[q2 * kFftRow + q3 + fft_x2(l1)] = u[q2];
  wave_sync();
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = xv[q3 * kFftRow + l1 + fft_x2(j)];
#pragma unroll
  for (int j = 0; j < 8; ++j) u[j] = xu[q3 * kFftRow + l1 + fft_x2(j)];
  dft8<INV>(v);
  dft8<INV>(u);
}

// ------------------------------------------------------------------------------------------
// DC high-pass (HP_DC_Filter_Coeffs2, FIR.cpp:87-89): y = b0 x + d; d' = b1 x + a1 y  (b2=a2=0)
// ------------------------------------------------------------------------------------------
constexpr double kHpB0 = 0.927176191943378969;
constexpr double kHpB1 = -0.927176191943378969;
constexpr double kHpA1 = 0.854352383886757938;
constexpr double cpow(double b, int e) {
  double r = 1.0;
  for (int i = 0; i < e; ++i) r *= b;
  return r;
}
template <int n>
struct HpTab {
  float scanA[4];  // a1^(n * 2^s), s = 0..3: carry multiplier across 1, 2, 4, 8 lanes
  float pw[n];     // a1^k
  constexpr HpTab() : scanA{}, pw{} {
    for (int s = 0; s < 4; ++s) scanA[s] = (float)cpow(kHpA1, n << s);
    for (int k = 0; k < n; ++k) pw[k] = (float)cpow(kHpA1, k);
  }
};

// Inclusive wave scan of the affine carry map d_out = A d_in + B with the same A = a1^n on
// every lane: 4 row_shr steps inside each 16-lane row, then row_bcast:15 / row_bcast:31 to
// stitch the rows.  m15 = A^((lane&15)+1), m31 = A^((lane&31)+1) (per-lane constants).
// Works on an (I, Q) pair of chains at once.
template <int n, bool FUSED = true>
__device__ __forceinline__ f2 hp_scan(f2 B, float m15, float m31) {
  constexpr HpTab<n> T{};
  // (FUSED: four more VGPRs hold the steps' multipliers -- the AGC / SAM kernels, which sit at the 128-register limit,
  // spill for it and keep the moves)
  if (FUSED && T41RX_SCAN_DPP) {
    // (round 5: the four steps inside the rows the same way -- two v_fmac_f32_dpp instead of two v_mov_b32_dpp and a
    // packed multiply-add; the same fused operations on the same operands, so the same bits)
    float sx = B.x, sy = B.y;
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\tv_fmac_f32_dpp %0, %0, %3 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %1, %3 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\tv_fmac_f32_dpp %0, %0, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %1, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\tv_fmac_f32_dpp %0, %0, %5 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %1, %5 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(sx), "+v"(sy)
        : "v"(T.scanA[0]), "v"(T.scanA[1]), "v"(T.scanA[2]), "v"(T.scanA[3]));
    B = f2{sx, sy};
  } else {
    B = pk_fma(splat(T.scanA[0]), dpp_f2<kDppRowShr1, 0xf, true>(B), B);
    B = pk_fma(splat(T.scanA[1]), dpp_f2<kDppRowShr2, 0xf, true>(B), B);
    B = pk_fma(splat(T.scanA[2]), dpp_f2<kDppRowShr4, 0xf, true>(B), B);
    B = pk_fma(splat(T.scanA[3]), dpp_f2<kDppRowShr8, 0xf, true>(B), B);
  }
  // the two row-stitching steps as v_fmac_f32 with the DPP operand built in (VOP2; the packed
  // form needs the shuffled value in a register first, zeroed for the rows the step leaves alone):
  // rows outside row_mask are simply not written
  // (inline asm: hipcc does not fold a DPP move into the multiply-add; the s_nop are the two wait
  // states a DPP read needs after a VALU write of its source, which the compiler does not insert
  // inside an asm statement)
  float bx = B.x, by = B.y;
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %1, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0\n\tv_fmac_f32_dpp %0, %0, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %1, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0"  // (a DPP read of %1 may follow)
      : "+v"(bx), "+v"(by)
      : "v"(m15), "v"(m31));
  return f2{bx, by};
}

// Runs the recurrence over `n` consecutive (I, Q) samples per lane (lane-major: lane l owns
// samples l*n .. l*n+n-1), all 64 lanes in parallel: local pass with zero carry, wave scan of
// the carries, rank-1 fix-up.  `carry` (wave-uniform pair) is the filter state entering lane 0
// for the I chain and the Q chain and is replaced by the state leaving lane 63.
// x arrives PRE-SCALED by b0 (the caller folds it into the RF-gain multiply it does anyway): with
// b1 = -b0 the step is then y = x' + d, d' = a1 y - x' -- two packed operations per (I, Q) sample
// instead of three (b0 x and b1 x each round once here; the reference rounds b0 x inside the sum:
// a difference of one ulp of x, far inside the path's tolerance).
static_assert(kHpB1 == -kHpB0, "the pre-scaled form of the DC high-pass needs b1 = -b0");
template <int n, bool FUSED = true>
__device__ __forceinline__ void dc_highpass(f2 (&x)[n], f2 &carry, int lane, float m15, float m31) {
  constexpr HpTab<n> T{};
  const float a1 = (float)kHpA1;
  f2 d = (lane == 0) ? carry : splat(0.0f);
#pragma unroll
  for (int k = 0; k < n; ++k) {
    const f2 y = x[k] + d;
    d = pk_fma(splat(a1), y, -x[k]);
    x[k] = y;
  }
  const f2 B = hp_scan<n, FUSED>(d, m15, m31);
  const f2 e = f2{lane_up1(B.x), lane_up1(B.y)};
#pragma unroll
  for (int k = 0; k < n; ++k) x[k] = pk_fma(splat(T.pw[k]), e, x[k]);
  carry = f2{__int_as_float(__builtin_amdgcn_readlane(__float_as_int(B.x), 63)),
             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(B.y), 63))};
}

// filter state after `n` samples per lane when only the end state matters (zero start state)
template <int n, bool FUSED = true>
__device__ __forceinline__ float dc_highpass_end_state(const float (&x)[n], float m15, float m31) {
  const float c = (float)(kHpB1 + kHpA1 * kHpB0), a1 = (float)kHpA1;
  float d = 0.0f;
#pragma unroll
  for (int k = 0; k < n; ++k) d = fmaf(a1, d, c * x[k]);
  const f2 B = hp_scan<n, FUSED>(f2{d, 0.0f}, m15, m31);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(B.x), 63));
}

// `old` with lane K replaced by the wave-uniform `s`: one v_writelane_b32 (this hipcc has no builtin for it and compiles
// `lane == K ? s : old` to a move, a compare and a select).  No wait states needed around it: the lane select is an
// immediate, and a VGPR it writes is read by plain VALU instructions only (no DPP / permlane reader right behind it).
template <int K>
__device__ __forceinline__ float write_lane(float old, float s) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(s), "n"(K));
  return old;
}
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double uniform_f64(double v) {
  return __longlong_as_double((long long)uniform_u64((uint64_t)__double_as_longlong(v)));
}
__device__ __forceinline__ float uniform_f32(float v) {
  return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
}

// ------------------------------------------------------------------------------------------
// LDS layout (floats).  Workgroup: [tables | wave 0 | wave 1 | wave 2 | wave 3]
// ------------------------------------------------------------------------------------------
// tables: mask[8][64], tw1[7][64] as in the constant table, tw2 compacted to [7][8] (float2 each).
// Per wave, everything is interleaved complex (I, Q):
// X  : post-NCO samples of one 512-sample sub-block + 28-entry history.
//      logical complex index j: [0] pad, [1..27] history, [28+k] new sample k.  One 16-byte
//      slot of padding after every 8 complex = after every lane's share, so (a) the lane stride
//      is 5 slots and the ds_read_b128 windows of the /4 decimator are bank-conflict-free
//      (5 is odd: any 16 lanes distinct mod 16 hit 16 distinct slots) and (b) every lane sees
//      the pads at the same offsets of its window, i.e. all LDS offsets are immediates.
//      Doubles as FFT exchange / transposition scratch.
// Y1 : /4 decimator outputs of two sub-blocks (256) + 48-entry history = 304 complex, logical j:
//      [0..2] pad, [3..47] history, [48+m] new; stored as two planes of even / odd slots (y1slot()).
constexpr int kLdsTabMask = 0, kLdsTabTw1 = 512, kLdsTabTw2 = 512 + 448;  // float2 units
constexpr int kLdsTabFloats = 2 * (512 + 448 + 56);                          // 2032 floats
constexpr int kXFloats = 1352;  // 2 * (xpad(539) + 1) = 1348, rounded to 16 B
constexpr int kY1Floats = 608;  // 304 complex
// (offsets of X, Y1 and the back-end scratch: Geo<PART> below)
// 2052 floats = 8208 B: X + Y1 (1960) rounded up so the whole slice can double as the 2048-float
// output transposition buffer; tables + 4 slices = exactly the 40 KiB the launch requests
constexpr int kLdsFloatsPerWave = 2052;
static_assert(kXFloats + kY1Floats <= kLdsFloatsPerWave && kLdsFloatsPerWave >= 2048, "slice too small");
static_assert(kXFloats >= 8 * kFftRow * 2, "FFT exchange buffer must fit in X");
__device__ __forceinline__ constexpr int xpad(int j) { return j + ((j >> 3) << 1); }  // complex units
// Y1 is stored as two planes of 76 16-byte slots: slot s (complex 2s, 2s+1) of the logical array
// lives in plane s & 1 at position s >> 1.  The /2 decimator's window of lane l starts at slot 2 l,
// so with a linear layout the 16 lanes a ds_read_b128 serves together sit 2 slots apart and collide
// pairwise; split by parity, each read walks one plane with a lane stride of one slot.  Offsets stay
// immediates: window slot i of lane l = plane (i & 1), position l + (i >> 1).
constexpr int kY1Plane = 76;  // slots per plane (304 complex)
__device__ __forceinline__ constexpr int y1slot(int s) { return 4 * ((s & 1) * kY1Plane + (s >> 1)); }  // float offset of slot s

typedef const __attribute__((address_space(4))) DevCoef *CoefPtr;
__device__ __forceinline__ CoefPtr fresh_coef(CoefPtr p) {
  asm volatile("" : "+s"(p));
  return p;
}

__device__ __forceinline__ float4 lds4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
// "whatever is in the registers": the start value of a float4 that some lanes load under a condition and the same lanes
// use under the same condition behind a wave_sync().  With zeros as the start value the compiler has to write them (it
// cannot see through the fence that the other lanes' values are never read): 33 v_mov_b32 v, 0 per SSB frame, round 5.
// T41RX_ZERO_INIT=1: zeros (A/B builds).
#ifndef T41RX_ZERO_INIT
#define T41RX_ZERO_INIT 0
#endif
__device__ __forceinline__ float4 any_float4() {
#if T41RX_ZERO_INIT
  return make_float4(0, 0, 0, 0);
#else
  // (an undefined value on purpose; __builtin_nondeterministic_value() would do, but a frozen undef is lowered to 0)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
  float4 t;
  return t;
#pragma clang diagnostic pop
#endif
}
// streaming (read-once / write-once) global accesses: nontemporal, so they do not evict the
// per-channel state and the constant tables from L2 / Infinity Cache
typedef float f4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg_stream(const float *p) {
  const f4n t = __builtin_nontemporal_load(reinterpret_cast<const f4n *>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void stg_stream(float *p, float4 v) {
  __builtin_nontemporal_store(f4n{v.x, v.y, v.z, v.w}, reinterpret_cast<f4n *>(p));
}
// the same at (wave-uniform base) + (per-lane offset): the request takes its address as `v_off, s[base:base+1] offset:imm`
// instead of a 64-bit VGPR pair somebody has to compute (round 5: 13 v_ashrrev_i32 + 22 v_lshl_add_u64 + 5 v_lshlrev_b64
// + 14 v_add(c)_co_u32 per SSB frame).  Three things are needed for the selector to see `sgpr + zext(vgpr32)`:
// the offset widened UNSIGNED (`base + int` sign-extends: lane went through FRESH_LANE, its range is unknown); the
// widening in the request's own basic block (hence fresh_off(): a widened offset is hoisted and shared otherwise, and the
// selector, which works block by block, then sees a 64-bit register, not a zext); and the base an opaque scalar, or
// neighbouring requests' addresses are merged into one 64-bit VGPR sum + immediates again.
// T41RX_SADDR=0: the previous address form (A/B builds).
#ifndef T41RX_SADDR
#define T41RX_SADDR 1
#endif
// T41RX_PHASE_SPLIT=0: the oscillator phases of a frame as round 4 worked them out (A/B builds)
#ifndef T41RX_PHASE_SPLIT
#define T41RX_PHASE_SPLIT 1
#endif
#ifndef T41RX_SADDR_BASE
#define T41RX_SADDR_BASE 1
#endif
struct LaneOff {
  unsigned bytes;
};
__device__ __forceinline__ LaneOff fresh_off(int floats) {  // one per group of requests that share the lane offset
  unsigned o = (unsigned)floats * 4u;
#if T41RX_SADDR
  asm volatile("" : "+v"(o));
#endif
  return LaneOff{o};
}
typedef __attribute__((address_space(1))) char *GlobalBytes;
typedef const __attribute__((address_space(1))) char *CGlobalBytes;
typedef const __attribute__((address_space(1))) f4n *CGlobalF4;
typedef __attribute__((address_space(1))) f4n *GlobalF4;
__device__ __forceinline__ CGlobalBytes global_at(const float *ubase, LaneOff o) {
  CGlobalBytes b = (CGlobalBytes)(ubase);
#if T41RX_SADDR_BASE
  asm("" : "+s"(b));
#endif
  return b + o.bytes;
}
__device__ __forceinline__ GlobalBytes global_at(float *ubase, LaneOff o) {
  GlobalBytes b = (GlobalBytes)(ubase);
#if T41RX_SADDR_BASE
  asm("" : "+s"(b));
#endif
  return b + o.bytes;
}
// (imm: a compile-time number of floats on top, for the request's immediate-offset field: up to 1023)
__device__ __forceinline__ float4 ldg_stream(const float *ubase, LaneOff o, int imm = 0) {
#if T41RX_SADDR
  const f4n t = __builtin_nontemporal_load((CGlobalF4)(global_at(ubase, o) + 4 * imm));
  return make_float4(t.x, t.y, t.z, t.w);
#else
  return ldg_stream(reinterpret_cast<const float *>(reinterpret_cast<const char *>(ubase) + (int)o.bytes) + imm);
#endif
}
__device__ __forceinline__ void stg_stream(float *ubase, LaneOff o, float4 v) {
#if T41RX_SADDR
  __builtin_nontemporal_store(f4n{v.x, v.y, v.z, v.w}, (GlobalF4)global_at(ubase, o));
#else
  stg_stream(reinterpret_cast<float *>(reinterpret_cast<char *>(ubase) + (int)o.bytes), v);
#endif
}
__device__ __forceinline__ float2 ldg2(const float2 *ubase, unsigned idx) {  // ubase[idx] of a wave-uniform table
#if T41RX_SADDR
  typedef float f2n __attribute__((ext_vector_type(2)));
  CGlobalBytes b = (CGlobalBytes)(ubase);
#if T41RX_SADDR_BASE
  asm("" : "+s"(b));
#endif
  unsigned o = idx * 8u;
  asm volatile("" : "+v"(o));  // (or zext(trunc(P >> 56) * 8) becomes a 64-bit and(P >> 53, 0x7f8): no zext left to select)
  const f2n t = *(const __attribute__((address_space(1))) f2n *)(b + o);
  return make_float2(t.x, t.y);
#else
  return ubase[(int)idx];
#endif
}
// the 8-byte entry at ubase + o (+ imm entries: compile-time, for the offset field, up to 511) of a wave-uniform table
__device__ __forceinline__ f2 ldg_cf(const f2 *ubase, LaneOff o, int imm = 0) {
  typedef float f2n __attribute__((ext_vector_type(2)));
#if T41RX_SADDR
  const f2n t = *(const __attribute__((address_space(1))) f2n *)(global_at(reinterpret_cast<const float *>(ubase), o) + 8 * imm);
#else
  const f2n t = *reinterpret_cast<const f2n *>(reinterpret_cast<const char *>(ubase) + (int)o.bytes + 8 * imm);
#endif
  return f2{t.x, t.y};
}
__device__ __forceinline__ float4 ldg4(const float *ubase, LaneOff o) {  // (ordinary, cached load)
#if T41RX_SADDR
  const f4n t = *(CGlobalF4)global_at(ubase, o);
  return make_float4(t.x, t.y, t.z, t.w);
#else
  return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(ubase) + (int)o.bytes);
#endif
}

// the two q15 samples packed in one 32-bit word, as floats (exact)
__device__ __forceinline__ float q15_lo(float w) { return (float)(short)(__float_as_uint(w) & 0xffffu); }
__device__ __forceinline__ float q15_hi(float w) { return (float)(__float_as_int(w) >> 16); }
// arm_float_to_q15 (CMSIS-DSP scalar path without ARM_MATH_ROUNDING): (q15_t)__SSAT((q31_t)(x * 32768.0f), 16);
// two of them packed, first sample in the low half
__device__ __forceinline__ unsigned q15_pack2(float x0, float x1) {
  int a = (int)(x0 * 32768.0f), b = (int)(x1 * 32768.0f);  // v_cvt_i32_f32: toward zero, saturating
  a = a < -32768 ? -32768 : (a > 32767 ? 32767 : a);
  b = b < -32768 ? -32768 : (b > 32767 ? 32767 : b);
  return ((unsigned)a & 0xffffu) | ((unsigned)b << 16);
}

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) float *CFloatPtr;
// the tap arrays of the coefficient block as float offsets from its start
constexpr int kCoDec1 = offsetof(DevCoef, dec1) / 4, kCoDec2 = offsetof(DevCoef, dec2) / 4, kCoInt1 = offsetof(DevCoef, int1) / 4,
              kCoInt2 = offsetof(DevCoef, int2) / 4, kCoDeemph = offsetof(DevCoef, deemph) / 4;
// Scalar (SMEM) loads of N consecutive taps starting at a 16-byte aligned offset `off` (floats) of the
// coefficient block `base`.  The pointer is re-derived through an opaque asm each time so the loads
// stay next to their use (bounded SGPR live ranges).
// (round 5: what is laundered is the BASE of the coefficient block, and the tap's offset -- a constant once the caller's
// loops are unrolled -- is added behind the laundering, so it becomes the s_load's immediate.  Laundering the sum made
// hipcc materialise one 64-bit pointer per chunk, hoist all twelve of them out of the frame loop, spill them to lanes of a
// VGPR and read them back with two v_readlane_b32 each: 50 VALU instructions per frame and 26 SGPRs, ISA of round 4's kernel.)
#ifndef T41RX_TAPS_SUM
#define T41RX_TAPS_SUM 0
#endif
template <int N>
__device__ __forceinline__ void load_taps(float (&dst)[N], CFloatPtr base, int off) {
#if T41RX_TAPS_SUM  // (A/B: round 4's form)
  CFloatPtr p = base + off;
  asm volatile("" : "+s"(p));
#else
  asm volatile("" : "+s"(base));
  const CFloatPtr p = base + off;
#endif
  static_assert(N % 4 == 0, "tap chunks are multiples of 4");
  int i = 0;
#pragma unroll
  for (; i + 16 <= N; i += 16) {
    const f16v t = *reinterpret_cast<const __attribute__((address_space(4))) f16v *>(p + i);
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[i + j] = t[j];
  }
#pragma unroll
  for (; i + 8 <= N; i += 8) {
    const f8v t = *reinterpret_cast<const __attribute__((address_space(4))) f8v *>(p + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[i + j] = t[j];
  }
#pragma unroll
  for (; i + 4 <= N; i += 4) {
    const f4v t = *reinterpret_cast<const __attribute__((address_space(4))) f4v *>(p + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[i + j] = t[j];
  }
}

// Two adjacent complex outputs of a decimating FIR from one lane-contiguous LDS window:
//   acc0 = sum_i c[i] * w[OFF0 + i],  acc1 = sum_i c[i] * w[OFF1 + i],  i = 0..NT-1 in order
// (arm_fir_decimate_f32's tap order; I and Q share the taps, so each MAC is ONE v_pk_fma_f32
// with the tap broadcast from an SGPR).  w[] is NLOAD ds_read_b128 (2 complex each) starting at
// `win`; IDX maps a logical complex offset to its padded LDS offset.  The window is streamed:
// values are consumed right after their load, taps arrive in 8-wide scalar-load chunks just
// before first use, and the accumulators are pinned every GROUP loads so the compiler cannot
// hoist the whole window into registers.
// one tap on a complex sample: ONE v_pk_fma_f32 (tap broadcast by op_sel), or -- T41RX_FIR_PLAIN, an experiment: is the
// packed form the cheaper one for a chip that holds its clock down under this kernel? -- two v_fma_f32 (same roundings)
#ifndef T41RX_FIR_PLAIN
#define T41RX_FIR_PLAIN 0
#endif
__device__ __forceinline__ cf fir_mac(float tap, cf x, cf acc) {
#if T41RX_FIR_PLAIN
  cf r;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r.x) : "s"(tap), "v"(x.x), "v"(acc.x));
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r.y) : "s"(tap), "v"(x.y), "v"(acc.y));
  return r;
#else
  return pk_fma(splat(tap), x, acc);
#endif
}
// (T41RX_LOO 13 / 14, timing experiments: the /2 / the /4 decimator's window taken from registers `regsrc` instead of
// LDS -- the arithmetic kept, the LDS reads gone: what would a decimator that needs no window reads be worth?)
// (TAIL0, `tail`: round 5 -- the window's loads from TAIL0 on are the lane's OWN newest samples, which it still holds in
// registers: taken from there, the same values, and the LDS reads are not issued)
template <int NT, int OFF0, int OFF1, int NLOAD, int GROUP, int TAIL0 = 1 << 20, typename IDX>
__device__ __forceinline__ void fir_pair(const float *win, IDX idx, CoefPtr coef, int taps, cf &acc0, cf &acc1, const cf *regsrc = nullptr,
                                         const cf *tail = nullptr) {
  constexpr int NTP = (NT + 7) & ~7;
  float tc[NTP];
  acc0 = splat(0.0f);
  acc1 = splat(0.0f);
#pragma unroll
  for (int l = 0; l < NLOAD; ++l) {
    if (l > 0 && (l % GROUP) == 0) asm volatile("" : "+v"(acc0), "+v"(acc1)::"memory");
    float4 t;
    if (regsrc) t = make_float4(regsrc[(2 * l) & 7].x, regsrc[(2 * l) & 7].y, regsrc[(2 * l + 1) & 7].x, regsrc[(2 * l + 1) & 7].y);
    else if (tail && l >= TAIL0) t = make_float4(tail[2 * (l - TAIL0)].x, tail[2 * (l - TAIL0)].y, tail[2 * (l - TAIL0) + 1].x, tail[2 * (l - TAIL0) + 1].y);
    else t = lds4(win + 2 * idx(2 * l));
    const cf tv[2] = {cf{t.x, t.y}, cf{t.z, t.w}};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int x = 2 * l + j;
      const int i0 = x - OFF0, i1 = x - OFF1;
      if (i0 >= 0 && i0 < NT) {
        if ((i0 & 7) == 0) {
          float chunk[8];
          load_taps<8>(chunk, (CFloatPtr)coef, taps + i0);
#pragma unroll
          for (int q = 0; q < 8; ++q) tc[i0 + q] = chunk[q];
        }
        acc0 = fir_mac(tc[i0], tv[j], acc0);
      }
      if (i1 >= 0 && i1 < NT) acc1 = fir_mac(tc[i1], tv[j], acc1);
    }
  }
}

typedef const __attribute__((address_space(4))) ChanNco *NcoPtr;
__device__ __forceinline__ NcoPtr fresh_nco(NcoPtr p) {
  asm volatile("" : "+s"(p));
  return p;
}


#ifdef T41RX_CLK
// Diagnostic build only (-DT41RX_CLK, tools/clock_probe.py): every wave leaves the shader-clock and the constant
// 100 MHz counter's ticks between its start and its end here (its clock under this load = their ratio x 100 MHz).
static __device__ unsigned long long g_t41_clk[4 * 8192];  // per wave: shader cycles, 100 MHz ticks, start tick, HW_ID | XCC_ID << 32
#define T41RX_CLK_BEGIN()                                                                                              \
  unsigned long long clk_c0, clk_r0;                                                                                   \
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk_c0), "=s"(clk_r0)::"memory")
#define T41RX_CLK_END(wave_id)                                                                                         \
  do {                                                                                                                 \
    unsigned long long c1_, r1_;                                                                                       \
    unsigned hw_, xcc_;                                                                                                \
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1_), "=s"(r1_)::"memory"); \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                                  \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                                \
    if ((threadIdx.x & 63) == 0 && (wave_id) < 8192) {                                                                 \
      g_t41_clk[4 * (wave_id)] = c1_ - clk_c0;                                                                         \
      g_t41_clk[4 * (wave_id) + 1] = r1_ - clk_r0;                                                                     \
      g_t41_clk[4 * (wave_id) + 2] = clk_r0;                                                                           \
      g_t41_clk[4 * (wave_id) + 3] = hw_ | ((unsigned long long)(xcc_ & 0xf) << 32);                                   \
    }                                                                                                                  \
  } while (0)
// one buffer and one reader per translation unit (the kernels of a TU share it): T41RX_CLK_READER(name) defines the
// extern "C" reader tools/clock_probe.py binds -- t41rx_debug_read_clk for the SSB kernels, _am / _nfm / _sam / _long / _fc
#define T41RX_CLK_READER(name)                                                                                         \
  extern "C" __attribute__((visibility("default"))) int name(unsigned long long *host, int n) {                        \
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_t41_clk), sizeof(unsigned long long) * (size_t)n);              \
  }
#else
#define T41RX_CLK_READER(name)
#define T41RX_CLK_BEGIN() do {} while (0)
#define T41RX_CLK_END(wave_id) do {} while (0)
#endif

}  // namespace t41

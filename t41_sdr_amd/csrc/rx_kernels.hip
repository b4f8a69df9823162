// t41_sdr_amd/csrc/rx_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the T41 RX hot path.
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
#include <hip/hip_runtime.h>

#include "rx_internal.hpp"
#include "rx_kernels.hpp"
#include "wave_fft.hpp"

namespace t41 {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
// Timing experiments (tools/ablation_table.py): T41RX_ABLATE = n cuts stages from the END of the
// chain (1 interpolators, 2 FFTs, 3 /2 decimator, 4 /4 decimator, 5 NCO, 6 DC high-pass, 7 1-KiB
// store instructions, 8 the fused kernel's 16 x 64 B store instructions); 9 keeps all arithmetic but makes every wave use the same 16 channels'
// buffers (cache-resident I/O).  Outputs are WRONG for n > 0; the product builds with 0.
#ifndef T41RX_ABLATE
#define T41RX_ABLATE 0
#endif
// T41RX_LOO = n (leave one out) cuts exactly stage n of that list and keeps every other one.
#ifndef T41RX_LOO
#define T41RX_LOO 0
#endif
#define T41RX_CUT(n) ((T41RX_ABLATE >= (n) && T41RX_ABLATE <= 8) || T41RX_LOO == (n))
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
template <int n>
__device__ __forceinline__ f2 hp_scan(f2 B, float m15, float m31) {
  constexpr HpTab<n> T{};
  B = pk_fma(splat(T.scanA[0]), dpp_f2<kDppRowShr1, 0xf, true>(B), B);
  B = pk_fma(splat(T.scanA[1]), dpp_f2<kDppRowShr2, 0xf, true>(B), B);
  B = pk_fma(splat(T.scanA[2]), dpp_f2<kDppRowShr4, 0xf, true>(B), B);
  B = pk_fma(splat(T.scanA[3]), dpp_f2<kDppRowShr8, 0xf, true>(B), B);
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
template <int n>
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
  const f2 B = hp_scan<n>(d, m15, m31);
  const f2 e = f2{lane_up1(B.x), lane_up1(B.y)};
#pragma unroll
  for (int k = 0; k < n; ++k) x[k] = pk_fma(splat(T.pw[k]), e, x[k]);
  carry = f2{__int_as_float(__builtin_amdgcn_readlane(__float_as_int(B.x), 63)),
             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(B.y), 63))};
}

// filter state after `n` samples per lane when only the end state matters (zero start state)
template <int n>
__device__ __forceinline__ float dc_highpass_end_state(const float (&x)[n], float m15, float m31) {
  const float c = (float)(kHpB1 + kHpA1 * kHpB0), a1 = (float)kHpA1;
  float d = 0.0f;
#pragma unroll
  for (int k = 0; k < n; ++k) d = fmaf(a1, d, c * x[k]);
  const f2 B = hp_scan<n>(f2{d, 0.0f}, m15, m31);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(B.x), 63));
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
// Scalar (SMEM) loads of N consecutive taps starting at a 16-byte aligned offset of the
// coefficient block.  The pointer is re-derived through an opaque asm each time so the loads
// stay next to their use (bounded SGPR live ranges).
template <int N>
__device__ __forceinline__ void load_taps(float (&dst)[N], CFloatPtr p) {
  asm volatile("" : "+s"(p));
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
template <int NT, int OFF0, int OFF1, int NLOAD, int GROUP, typename IDX>
__device__ __forceinline__ void fir_pair(const float *win, IDX idx, CFloatPtr taps, cf &acc0, cf &acc1, const cf *regsrc = nullptr) {
  constexpr int NTP = (NT + 7) & ~7;
  float tc[NTP];
  acc0 = splat(0.0f);
  acc1 = splat(0.0f);
#pragma unroll
  for (int l = 0; l < NLOAD; ++l) {
    if (l > 0 && (l % GROUP) == 0) asm volatile("" : "+v"(acc0), "+v"(acc1)::"memory");
    float4 t;
    if (regsrc) t = make_float4(regsrc[(2 * l) & 7].x, regsrc[(2 * l) & 7].y, regsrc[(2 * l + 1) & 7].x, regsrc[(2 * l + 1) & 7].y);
    else t = lds4(win + 2 * idx(2 * l));
    const cf tv[2] = {cf{t.x, t.y}, cf{t.z, t.w}};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int x = 2 * l + j;
      const int i0 = x - OFF0, i1 = x - OFF1;
      if (i0 >= 0 && i0 < NT) {
        if ((i0 & 7) == 0) {
          float chunk[8];
          load_taps<8>(chunk, taps + i0);
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

// ------------------------------------------------------------------------------------------
// AGC on (AGCMode 1..4): AGC(), DSP_Fn.cpp:504-631
// ------------------------------------------------------------------------------------------
// What the reference does per 24 kS/s sample: push the new (I, Q) into a ring, pop the one from
// attack_buffsize = 97 samples ago, keep ring_max = max |z| over the 97 newest entries (kept
// incrementally there, rescanned when the maximum leaves: the same value), run two one-pole
// averages of |popped|, step a five-state attack / hang / decay law for `volts` against
// ring_max, and scale the popped sample by a gain computed from volts.
// Here: everything that is a function of the samples alone (magnitudes, the sliding maximum,
// the b*x terms of the averages, the gain from volts, the scaling) is done by the whole wave,
// four consecutive samples per lane.  What is inherently serial -- 256 steps of the volts law
// and the two averages, which must keep the reference's f32 rounding step by step (the slow
// decay moves volts by a few ulps per sample, so any reassociation drifts by percents over a
// second) -- runs as a scalar chain, one LANE per channel: lanes 0..3 of the workgroup's first
// wave take the four channels of the workgroup between two workgroup barriers.
// LDS (floats, in the wave's slice).  Two layouts:
//  * AgcLds<false> (the long-FFT back kernel, 4-wave workgroups): the scratch starts at the slice.
//  * AgcLds<true>  (the fused FFT_LENGTH 512 kernel, whose delay lines, overlap block and x2
//    history stay in LDS across frames, see Geo<0>): the scratch avoids them -- ring_max / volts
//    reuse the |z| array (every read of |z| precedes the first write of ring_max in program order,
//    and LDS executes a wave's instructions in order), the (b x) pairs sit in the free part of Y1.
template <bool RESIDENT>
struct AgcLds {
  static constexpr int Z = RESIDENT ? 68 : 0;        // (re, im)[356]: [0..99] the last 100 inputs, [100 + i] this frame's input i
  static constexpr int A = Z + 712;                  // |z|[356], same indexing
  static constexpr int G = A + 356;                  // max of every aligned group of four |z| (89 used)
  static constexpr int S = RESIDENT ? G + 92 : 1928; // the 8 state words (rx_internal.hpp: kAgcSt*)
  static constexpr int R = RESIDENT ? A : 1160;      // ring_max[256]; the chain replaces it by volts[256]
  // (fast_backmult, hang_backmult) * abs_out_sample, [256] pairs, in two halves of 128 pairs: the
  // resident layout puts them behind the 12 history slots of either Y1 plane
  static constexpr int P0 = RESIDENT ? 1348 + 48 : 1416;
  static constexpr int P1 = RESIDENT ? 1348 + 4 * kY1Plane + 48 : 1416 + 256;
  __device__ static constexpr int pofs(int i) { return i < 256 ? P0 + i : P1 + i - 256; }  // float i of the 512
};
static_assert(AgcLds<false>::S + kAgcScalars <= kLdsFloatsPerWave, "AGC scratch must fit the wave slice");
static_assert(AgcLds<true>::S + kAgcScalars <= 1348 && AgcLds<true>::P0 + 256 <= 1348 + 4 * kY1Plane &&
              AgcLds<true>::P1 + 256 <= 1348 + 608, "AGC scratch vs resident state");
static_assert(kAgcDelay == 97 && kAgcHist == 100, "window arithmetic below is written for 97 / 100");
#ifndef T41RX_AGC_COOP
#define T41RX_AGC_COOP 1
#endif
constexpr bool kAgcCoop = T41RX_AGC_COOP;

// DSP_Fn.cpp:520-523 (pmode = 1): separate roundings, correctly rounded square root
__device__ __forceinline__ float agc_mag(cf z) {
#pragma clang fp contract(off)
  const float a = z.x * z.x, b = z.y * z.y;
  return __builtin_sqrtf(a + b);
}

// DSP_Fn.cpp:627: mult = (out_target - slope_constant * min(0.0, log10f_fast(inv_max_input * volts))) / volts
// (double arithmetic from the min() on: its 0.0 literal promotes), log10f_fast = Utility.cpp:245-258
__device__ __forceinline__ float agc_mult(float volts, float inv_max_input, float out_target, float slope_constant) {
#pragma clang fp contract(off)
  const float t = fabsf(inv_max_input * volts);
  const float F = __builtin_amdgcn_frexp_mantf(t);
  const int E = __builtin_amdgcn_frexp_expf(t);
  float Y = 1.23149591368684f;
  Y *= F;
  Y += -4.11852516267426f;
  Y *= F;
  Y += 6.02197014179219f;
  Y *= F;
  Y += -3.13396450166353f;
  Y += (float)E;
  const float lg = Y * 0.3010299956639812f;
  const double m = (0.0 < (double)lg) ? 0.0 : (double)lg;
  return (float)(((double)out_target - (double)slope_constant * m) / (double)volts);
}

// The serial part for ONE channel per lane: sl = that channel's LDS slice.
//
// A single wave runs it, so what counts is the number of instructions per step (one VALU issue per
// 4 cycles) and the length of the dependent chain volts -> volts.  Steps are taken four at a
// time.  agc_fast_block() assumes the common case -- every lane either attacks (ring_max >=
// volts: same update from every state, then state 0) or stays in its decay state -- with
// predicated straight-line code, and reports whether any lane met something else (a decision of
// state 0, fast decay reaching save_volts, the hang counter expiring).  Only then the block is
// redone by agc_slow_block(), the reference's switch statement as written.  Both produce the same
// f32 values step by step.
struct AgcState {
  float fast_backaverage, hang_backaverage, volts, save_volts;
  int state, decay_type, hang_counter;
};
struct AgcConsts {
  float attack_mult, decay_mult, fast_decay_mult, hang_decay_mult, onemfast_backmult, onemhang_backmult;
  float min_volts, hang_level, pop_ratio;
  int hang_count;
};

// DSP_Fn.cpp:525-629 for four consecutive samples, any state sequence
__device__ __forceinline__ void agc_slow_block(AgcState &st, const AgcConsts &g, const float (&rm)[4],
                                               const float (&pf)[4], const float (&ph)[4], float (&vo)[4]) {
#pragma clang fp contract(off)
  float fast_backaverage = st.fast_backaverage, hang_backaverage = st.hang_backaverage;
  float volts = st.volts, save_volts = st.save_volts;
  int state = st.state, decay_type = st.decay_type, hang_counter = st.hang_counter;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float ring_max = rm[k];
    fast_backaverage = pf[k] + g.onemfast_backmult * fast_backaverage;  // :525
    hang_backaverage = ph[k] + g.onemhang_backmult * hang_backaverage;  // :526
    if (hang_counter > 0) --hang_counter;                                // :543
    if (ring_max >= volts) {  // every state attacks the same way; 2, 3, 4 remember where from
      if (state >= 2) save_volts = volts;
      state = 0;
      volts += (ring_max - volts) * g.attack_mult;
    } else if (state == 0) {  // :549-566
      if (volts > g.pop_ratio * fast_backaverage) {
        state = 1;
        volts += (ring_max - volts) * g.fast_decay_mult;
      } else if (hang_backaverage > g.hang_level) {  // hang_enable = 1, :458
        state = 2;
        hang_counter = g.hang_count;
        decay_type = 1;
      } else {
        state = 3;
        volts += (ring_max - volts) * g.decay_mult;
        decay_type = 0;
      }
    } else if (state == 1) {  // :569-590
      if (volts > save_volts) {
        volts += (ring_max - volts) * g.fast_decay_mult;
      } else if (hang_counter > 0) {
        state = 2;
      } else if (decay_type == 0) {
        state = 3;
        volts += (ring_max - volts) * g.decay_mult;
      } else {
        state = 4;
        volts += (ring_max - volts) * g.hang_decay_mult;
      }
    } else if (state == 2) {  // :593-604
      if (hang_counter == 0) {
        state = 4;
        volts += (ring_max - volts) * g.hang_decay_mult;
      }
    } else if (state == 3) {  // :607-615; the .05 literal is a double
      volts = (float)((double)volts + (double)((ring_max - volts) * g.decay_mult) * .05);
    } else {  // :618-626
      volts += (ring_max - volts) * g.hang_decay_mult;
    }
    if (volts < g.min_volts) volts = g.min_volts;  // :629
    vo[k] = volts;
  }
  st = AgcState{fast_backaverage, hang_backaverage, volts, save_volts, state, decay_type, hang_counter};
}

// State 3's decay step inside the fast block: 1 = the reference's double-precision expression (shipped); 0 = two FMAs
// that bracket it + the slow block where they differ (round 4 experiment: bit-identical over tools/agc_decay_check.py's
// 24 streams, and no faster -- 32.2-32.4 against 31.9-32.1 us per frame on bench.py's ssb_agc, 36.3 against 33.5 on pure
// noise, where one block in ten then takes the slow path: the chain is paced by more than this step)
#ifndef T41RX_AGC_DECAY64
#define T41RX_AGC_DECAY64 1
#endif
// Lane masks as plain 64-bit scalars.  Written out by hand because the compiler, given bools,
// rebuilds them as 0/1 integers in VGPRs every time two of them meet in a select: with these
// three wrappers a comparison is one VALU instruction with an SGPR-pair result, the logic between
// masks is scalar ALU, and a select is one v_cndmask.
typedef unsigned long long lanemask;
__device__ __forceinline__ lanemask lanes_ge(float a, float b) {
  lanemask m;
  asm("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
  return m;
}
__device__ __forceinline__ lanemask lanes_gt(float a, float b) {
  lanemask m;
  asm("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
  return m;
}
__device__ __forceinline__ float pick(lanemask m, float if_set, float if_clear) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
  return r;
}

// What the straight-line path needs to know about a lane's state, kept in registers between
// blocks (recomputed only after a slow block):
//   stay  : volts += (ring_max - volts) * stay while the lane remains in its decay state
//           (state 2 = hang: 0, volts rests)
//   thr   : state 1 leaves fast decay once volts <= save_volts; -inf for the others
//   in0   : lanes in state 0 -- any step that does not attack is a decision, i.e. not for this path
//   pend  : lanes in states 2, 3, 4 that have not attacked yet: their first attack records save_volts
//   is3   : state 3 adds its decay step in double (DSP_Fn.cpp:614);  is2: hang
struct AgcLane {
  float stay, thr;
  lanemask in0, pend, is3, is2;
};
__device__ __forceinline__ AgcLane agc_lane_of(const AgcState &st, const AgcConsts &g) {
  const int s = st.state;
  AgcLane d;
  d.stay = g.hang_decay_mult;
  d.stay = (s == 1) ? g.fast_decay_mult : d.stay;
  d.stay = (s == 2) ? 0.0f : d.stay;
  d.stay = (s == 3) ? g.decay_mult : d.stay;
  d.thr = (s == 1) ? st.save_volts : -__builtin_inff();
  d.in0 = __builtin_amdgcn_ballot_w64(s == 0);
  d.pend = __builtin_amdgcn_ballot_w64(s >= 2);
  d.is3 = __builtin_amdgcn_ballot_w64(s == 3);
  d.is2 = __builtin_amdgcn_ballot_w64(s == 2);
  return d;
}

// The same four steps under the assumption described above; a clear bit in the result marks a
// lane whose assumption failed somewhere in the block (its results are then meaningless).
// HAS3: some lane is in state 3.
template <bool HAS3>
__device__ __forceinline__ lanemask agc_fast_block(AgcState &st, AgcLane &d, const AgcConsts &g, const float (&rm)[4],
                                                   const float (&pf)[4], const float (&ph)[4], float (&vo)[4]) {
#pragma clang fp contract(off)
  // the hang counter cannot run out inside the block if more than four steps are left
  lanemask ok = ~(d.is2 & __builtin_amdgcn_ballot_w64(st.hang_counter <= 4));
  lanemask in0 = d.in0, pend = d.pend;
  float volts = st.volts, save_volts = st.save_volts;
  f2 back = f2{st.fast_backaverage, st.hang_backaverage};
  const f2 onem = f2{g.onemfast_backmult, g.onemhang_backmult};
  float attack_mult = g.attack_mult, min_volts = g.min_volts;
  asm volatile("" : "+v"(attack_mult), "+v"(min_volts));  // v_cndmask / v_max operands: keep them in VGPRs
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f2 aged = onem * back;  // :525-526
    back = f2{pf[k], ph[k]} + aged;
    const float ring_max = rm[k];
    const lanemask ge = lanes_ge(ring_max, volts);
    const lanemask gt = lanes_gt(volts, d.thr);
    ok &= ge | (gt & ~in0);
    save_volts = pick(ge & pend, volts, save_volts);  // the first attack out of 2, 3, 4
    pend &= ~ge;
    in0 |= ge;
    const float step = (ring_max - volts) * pick(ge, attack_mult, d.stay);
    float next = volts + step;
    if (HAS3) {
#if T41RX_AGC_DECAY64
      const float next3 = (float)((double)volts + (double)step * .05);
      next = pick(d.is3 & ~ge, next3, next);
#else
      // State 3's decay step, (float)((double)volts + (double)step * .05) (DSP_Fn.cpp:614): four dependent double-precision
      // instructions (two conversions in, a multiply, an add, a conversion out) in a chain that is paced by its
      // instruction count.  Sandwich instead: for floats c_lo < .05 < c_hi the products step * c are exact in double, the
      // double sum and both roundings are monotonic, so the reference's value lies between fl(volts + step * c_lo) and
      // fl(volts + step * c_hi) -- two FMAs.  Where the two agree that IS the reference's value; where they differ (a
      // rounding boundary between them: about |step| / |volts| / 10 of the steps, i.e. < 1e-4) the block is redone by
      // agc_slow_block(), which has the expression as written.  c_lo / c_hi are the SECOND neighbours of .05 either side:
      // an FMA rounds once where the reference rounds to double first, which can move the result by one ulp when the sum
      // sits within 2^-53 of a boundary -- a shift 10^7 times smaller than the one the extra neighbour adds.
      const float r_lo = __builtin_fmaf(step, __uint_as_float(0x3d4ccccbu), volts);
      const float next3 = __builtin_fmaf(step, __uint_as_float(0x3d4cccceu), volts);
      lanemask differ;
      asm("v_cmp_neq_f32_e64 %0, %1, %2" : "=s"(differ) : "v"(r_lo), "v"(next3));
      ok &= ~(differ & d.is3 & ~ge);
      next = pick(d.is3 & ~ge, next3, next);
#endif
    }
    asm("v_max_f32 %0, %1, %2" : "=v"(volts) : "v"(next), "v"(min_volts));  // :629 (no NaNs here)
    vo[k] = volts;
  }
  st.fast_backaverage = back.x;
  st.hang_backaverage = back.y;
  st.volts = volts;
  st.save_volts = save_volts;
  st.state = __float_as_int(pick(in0, __int_as_float(0), __int_as_float(st.state)));
  const int hc = st.hang_counter - 4;
  st.hang_counter = hc > 0 ? hc : 0;
  d.pend = pend;
  d.in0 = in0;
  return ok;
}

// Round 4: the same four steps with the shortest dependent path per step this arithmetic allows.  agc_fast_block()
// above decides first (compare -> mask -> select the multiplier) and then computes, and takes state 3's decay step
// through four double-precision instructions: ten dependent instructions and a VALU -> SGPR -> VALU hop per step, on
// an in-order wave.  Here both candidates are computed from the difference at once -- attack: volts + diff * attack_mult,
// stay: volts + diff * stay, the very products and sums the reference would have formed on either branch -- and selected
// when the comparison, issued beside the subtraction, has long returned; state 3's decay value comes from two FMAs that
// bracket the reference's double expression (c_lo < .05 < c_hi, second float neighbours: see T41RX_AGC_DECAY64 above for
// why their agreement proves the value) instead of through double precision; and everything that only feeds masks
// (the assumption check, save_volts, the state bookkeeping) waits for the end of the block, off the chain.  Dependent
// path per step: subtract, multiply, add / FMA, two selects, max.  `sand` returns the lanes whose bracket did not close
// in some step (their block is redone by agc_fast_block<true>, which has the double expression); the other result
// is agc_fast_block()'s `ok`.
template <bool HAS3>
__device__ __forceinline__ lanemask agc_fast_block_s(AgcState &st, AgcLane &d, const AgcConsts &g, const float (&rm)[4],
                                                     const float (&pf)[4], const float (&ph)[4], float (&vo)[4], lanemask &sand) {
#pragma clang fp contract(off)
  float volts = st.volts;
  f2 back = f2{st.fast_backaverage, st.hang_backaverage};
  const f2 onem = f2{g.onemfast_backmult, g.onemhang_backmult};
  float attack_mult = g.attack_mult, min_volts = g.min_volts;
  asm volatile("" : "+v"(attack_mult), "+v"(min_volts));  // v_cndmask / v_max operands: keep them in VGPRs
  const float c_lo = __uint_as_float(0x3d4ccccbu), c_hi = __uint_as_float(0x3d4cccceu);
  float vin[4];
  lanemask ge[4], gt[4], df[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f2 aged = onem * back;  // :525-526
    back = f2{pf[k], ph[k]} + aged;
    vin[k] = volts;
    const float diff = rm[k] - volts;
    ge[k] = lanes_ge(rm[k], volts);
    gt[k] = lanes_gt(volts, d.thr);
    const float sa = diff * attack_mult, ss = diff * d.stay;
    const float na = volts + sa;
    float cand = volts + ss;
    df[k] = 0;
    if (HAS3) {
      const float r_lo = __builtin_fmaf(ss, c_lo, volts), r_hi = __builtin_fmaf(ss, c_hi, volts);
      lanemask differ;
      asm("v_cmp_neq_f32_e64 %0, %1, %2" : "=s"(differ) : "v"(r_lo), "v"(r_hi));
      df[k] = differ;
      cand = pick(d.is3, r_hi, cand);
    }
    const float next = pick(ge[k], na, cand);
    asm("v_max_f32 %0, %1, %2" : "=v"(volts) : "v"(next), "v"(min_volts));  // :629 (no NaNs here)
    vo[k] = volts;
  }
  // the block's bookkeeping, in step order
  lanemask ok = ~(d.is2 & __builtin_amdgcn_ballot_w64(st.hang_counter <= 4));  // the hang counter cannot run out inside the block
  lanemask in0 = d.in0, pend = d.pend, bad = 0;
  float save_volts = st.save_volts;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ok &= ge[k] | (gt[k] & ~in0);
    if (HAS3) bad |= df[k] & d.is3 & ~ge[k];
    save_volts = pick(ge[k] & pend, vin[k], save_volts);  // the first attack out of 2, 3, 4
    pend &= ~ge[k];
    in0 |= ge[k];
  }
  sand = bad;
  st.fast_backaverage = back.x;
  st.hang_backaverage = back.y;
  st.volts = volts;
  st.save_volts = save_volts;
  st.state = __float_as_int(pick(in0, __int_as_float(0), __int_as_float(st.state)));
  const int hc = st.hang_counter - 4;
  st.hang_counter = hc > 0 ? hc : 0;
  d.pend = pend;
  d.in0 = in0;
  return ok;
}
#ifndef T41RX_AGC_SPEC
#define T41RX_AGC_SPEC 1  // 0: round 3's fast block only (A/B, tools/agc_decay_check.py)
#endif
// one block of four steps by the fastest form that is exact for it
__device__ __forceinline__ lanemask agc_block(AgcState &t, AgcLane &dt, const AgcState &st, const AgcLane &d, const AgcConsts &g,
                                              const float (&rm)[4], const float (&pf)[4], const float (&ph)[4], float (&vo)[4]) {
  const bool has3 = (d.is3 & ~d.in0) != 0;
#if T41RX_AGC_SPEC
  lanemask sand = 0;
  lanemask ok = has3 ? agc_fast_block_s<true>(t, dt, g, rm, pf, ph, vo, sand) : agc_fast_block_s<false>(t, dt, g, rm, pf, ph, vo, sand);
  if (sand != 0) {  // a rounding boundary between the bracketing values somewhere: the double expression decides
    t = st;
    dt = d;
    ok = agc_fast_block<true>(t, dt, g, rm, pf, ph, vo);
  }
  return ok;
#else
  return has3 ? agc_fast_block<true>(t, dt, g, rm, pf, ph, vo) : agc_fast_block<false>(t, dt, g, rm, pf, ph, vo);
#endif
}

// Round 4, second step: the block of agc_chain_pipe().  The duty wave issues one instruction per ~5-6 cycles whatever it
// is, so what paces the chain is the COUNT of instructions per block (150 with agc_fast_block_s(): 122 in the block, the
// rest moves) and every LDS round trip the compiler's waits expose (~150-185 cycles each behind fifteen other waves'
// traffic).  This form has ~100 instructions per block and no exposed round trip:
//   * the attack's and the stay's products are one v_pk_mul_f32: (diff, diff) x (attack_mult, stay);
//   * state 3's bracket and the other states' plain sum are ONE operation: fma(ss, c, volts) with c = c_hi for the lanes in
//     state 3 and c = 1 for the others -- fma(ss, 1, volts) IS fl(volts + ss) -- so a per-lane constant pair (c_lo, c_hi) or
//     (1, 1) makes both bracketing values one v_pk_fma_f32 for every lane, and the select by state, the second sum and
//     the block's dispatch on "some lane is in state 3" are gone (for the other lanes the two halves are equal by
//     construction, so the bracket's verdict needs no mask either: df0 | df1 | df2 | df3);
//   * the masks' bookkeeping is done per block: in the usual block no lane attacks and none is in state 0, nothing about
//     the states changes, and because volts does not rise in such a block the last step's comparison with the fast decay's
//     threshold covers all four; otherwise the step-by-step form below it runs (same masks, same order);
//   * the steps need the four ring maxima only: the NEXT block's (the next chunk's first, from the other half of the
//     double-buffered stage) are requested at the top of a block and taken over at its end, four moves; the back-averages'
//     operands are requested at the top of the block they belong to and used behind its steps.  (Reading the ring maxima at
//     the block's end into the registers the steps have just finished with -- no move at all -- exposes an LDS round trip
//     per block: measured, no gain);
//   * the chunk's volts stay in registers (agc_chain_pipe: `keep`) instead of a write to the stage and a read back.
// Every float operation is one of agc_fast_block_s()'s on the same operands, or an FMA by 1 in place of a sum: bit-identical
// (T41RX_AGC_PHASED=0 builds the form above; tools/agc_decay_check.py, the pipelined == barrier tests and
// tools/pipe_soak.py -- 89 529 random runs, profiles/r04_pipe_soak2.json -- compare them).
// Timing experiments on one box (wrong results, T41RX_AGC_X), before the bookkeeping went per block, 33.6 us per frame:
// without the back-averages 31.3, without the bookkeeping 30.2, without the bracket 32.0, without all three 29.4; after:
// 30.2 -> 29.9 / 29.5 / 29.7 / 28.8 -- what is left of the period is the sixteen waves' own work.
#ifndef T41RX_AGC_PHASED
#define T41RX_AGC_PHASED 1
#endif
#ifndef T41RX_AGC_X
#define T41RX_AGC_X 0  // timing experiments (wrong results): 1 no back-averages, 2 no bookkeeping, 4 no bracket
#endif
struct AgcLaneP {
  f2 mult;  // (attack_mult, stay)
  f2 cb;    // state 3: the second float neighbours of .05 either side (see above); the others: (1, 1)
};
__device__ __forceinline__ AgcLaneP agc_lanep_of(const AgcState &st, const AgcLane &d, float attack_mult) {
  const bool s3 = st.state == 3;
  AgcLaneP lp;
  lp.mult = f2{attack_mult, d.stay};
  lp.cb = f2{s3 ? __uint_as_float(0x3d4ccccbu) : 1.0f, s3 ? __uint_as_float(0x3d4cccceu) : 1.0f};
  return lp;
}
__device__ __forceinline__ lanemask agc_block_phased(AgcState &st, AgcLane &d, const AgcConsts &g, const AgcLaneP &lp, float min_volts,
                                                     const float4 r4, const float4 a4, f2 backmult /* (fast, hang) */,
                                                     float (&vo)[4], lanemask &sand) {
#pragma clang fp contract(off)
  const float rm[4] = {r4.x, r4.y, r4.z, r4.w};
  const float am[4] = {a4.x, a4.y, a4.z, a4.w};  // |popped|: the back-averages take backmult * |popped| (:525-526)
  const f2 onem = f2{g.onemfast_backmult, g.onemhang_backmult};
  float volts = st.volts;
  f2 back = f2{st.fast_backaverage, st.hang_backaverage};
  lanemask ge[4], df = 0;
  float vin[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    vin[k] = volts;
    const float diff = rm[k] - volts;
    ge[k] = lanes_ge(rm[k], volts);
    const f2 s2 = splat(diff) * lp.mult;  // (diff * attack_mult, diff * stay)
    const float na = volts + s2.x;        // the attack's value
    float cand;
    if (!(T41RX_AGC_X & 4)) {
      const f2 r = pk_fma(splat(s2.y), lp.cb, splat(volts));  // state 3: the bracket; the others: volts + ss twice
      lanemask differ;
      asm("v_cmp_neq_f32_e64 %0, %1, %2" : "=s"(differ) : "v"(r.x), "v"(r.y));
      df |= differ;
      cand = r.y;
    } else {
      cand = volts + s2.y;
    }
    const float next = pick(ge[k], na, cand);
    asm("v_max_f32 %0, %1, %2" : "=v"(volts) : "v"(next), "v"(min_volts));  // :629 (no NaNs here)
    vo[k] = volts;
    if (k >= 2 && !(T41RX_AGC_X & 1)) {  // two steps behind: their operands were requested at the top of the block
      back = splat(am[2 * (k - 2)]) * backmult + onem * back;  // :525-526
      back = splat(am[2 * (k - 2) + 1]) * backmult + onem * back;
    }
  }
  lanemask ok = ~(d.is2 & __builtin_amdgcn_ballot_w64(st.hang_counter <= 4));  // the hang counter cannot run out inside the block
  if (!(T41RX_AGC_X & 2)) {
    const lanemask any = ge[0] | ge[1] | ge[2] | ge[3];
    if (__builtin_expect((any | d.in0) == 0, 1)) {
      // no attack, no lane in state 0: every lane stays in its decay state unless its fast decay has reached save_volts --
      // volts does not rise without an attack (diff < 0, multipliers >= 0, monotonic roundings), so "volts > thr" before
      // the last step implies it before the three others -- EXCEPT through the clamp of :629: a block that starts below
      // min_volts (a live set_params / set_coeffs raised it, or a restored checkpoint) is lifted to it at step 0, so
      // vin[0] may sit at or below the threshold while vin[1..3] = min_volts sit above it; once clamped the three are
      // equal, so the first and the last comparison cover every step
      ok &= lanes_gt(vin[0], d.thr) & lanes_gt(vin[3], d.thr);
    } else {
      lanemask in0 = d.in0, pend = d.pend;
      float save_volts = st.save_volts;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ok &= ge[k] | (lanes_gt(vin[k], d.thr) & ~in0);
        save_volts = pick(ge[k] & pend, vin[k], save_volts);  // the first attack out of 2, 3, 4
        pend &= ~ge[k];
        in0 |= ge[k];
      }
      st.save_volts = save_volts;
      st.state = __float_as_int(pick(in0, __int_as_float(0), __int_as_float(st.state)));
      d.pend = pend;
      d.in0 = in0;
    }
  }
  sand = df;
  st.fast_backaverage = back.x;
  st.hang_backaverage = back.y;
  st.volts = volts;
  const int hc = st.hang_counter - 4;
  st.hang_counter = hc > 0 ? hc : 0;
  return ok;
}

// ---- AMDecodeSAM's loop (Demod.cpp:69-117) for one channel per lane: zs = the channel's 256 complex samples
// (audio replaces the real parts), T = arm_sin_f32's table in LDS, ms = the channel's kStMisc words.
// As written there: the fade leveler's time constants are exp(-1 / 24000 * tau) = exp(0) = 1 (integer
// division), so it adds dc_insert - dc = 0 to the audio; ApproxAtan2 returns +-2 pi where +-pi/2 is
// meant.  arm_sin_f32 / arm_cos_f32: 512-entry table, linear interpolation (CMSIS-DSP >= 1.4.5).
// (index, fract) first and the table reads of sine and cosine together: the two evaluations are
// independent, and the loop is one dependent instruction after another
struct SamIdx { unsigned index; float fract; };
__device__ __forceinline__ SamIdx sam_table_index(float in) {
#pragma clang fp contract(off)
  int n = (int)in;
  if (in < 0.0f) n--;
  in = in - (float)n;
  float findex = 512.0f * in;
  unsigned index = (unsigned)findex & 0xffffu;  // (uint16_t)
  if (index >= 512u) {
    index = 0;
    findex -= 512.0f;
  }
  return SamIdx{index, findex - (float)index};
}
// The same for 0 <= in < 2, where the PLL keeps its arguments (phase in [0, 2 pi] times 0.159154943092f <= 1.0000001,
// + 0.25 for the cosine): there n = (int)in is 0 or 1 and in - n exact, i.e. v_fract_f32; `in < 0` never holds; and
// 512 (in - n) <= 512 - 2^-15 < 512, so the index never reaches 512 and the wrap is dead code.  Same index, same
// fraction, 5 instructions instead of 14 (checked against the form above on every float of the range:
// tests/test_sam.py::test_pll_table_index_short_form).
__device__ __forceinline__ SamIdx sam_table_index_pos(float in) {
#pragma clang fp contract(off)
  const float findex = 512.0f * __builtin_amdgcn_fractf(in);
  const unsigned index = (unsigned)findex;
  return SamIdx{index, findex - (float)index};
}
__device__ __forceinline__ float sam_atan(float z) {  // ApproxAtan, Utility.cpp:298-302
#pragma clang fp contract(off)
  const float n1 = 0.97239411f, n2 = -0.19194795f;
  return (n1 + n2 * z * z) * z;
}
__device__ __forceinline__ float sam_atan2(float y, float x) {  // ApproxAtan2, Demod.cpp:148-197, branch-free
#pragma clang fp contract(off)
  const float kPi = 3.1415926535897932384626433832795f, kTpi = 6.283185307179586476925286766559f;
  const bool xg = fabsf(x) > fabsf(y);           // the branch that divides y / x; else x / y
  const float t = sam_atan((xg ? y : x) / (xg ? x : y));  // one division, the operands the taken branch has
  const float rx = x > 0.0f ? t : (y >= 0.0f ? t + kPi : t - kPi);
  const float ry = y > 0.0f ? -t + kTpi : -t - kTpi;
  const float r0 = y > 0.0f ? kTpi : (y < 0.0f ? -kTpi : 0.0f);  // x == 0
  return x != 0.0f ? (xg ? rx : ry) : r0;
}
// AMDecodeSAM's loop for one channel per lane: the PLL's state and one step.
// The phase of step i + 1 is phase_i + fil_out_(i-1): it does not wait for step i's detector.  So the sine / cosine
// of the NEXT step (index arithmetic, four table reads, two interpolations) are evaluated while this step's
// products, arctangent (an IEEE division) and loop filter run: two independent dependency chains per iteration
// instead of one twice as long.  Same operations on the same values as the loop as written.
#ifndef T41RX_SAM_DEFER
#define T41RX_SAM_DEFER 1  // 0: the compiler's own placement of the interpolation (A/B)
#endif
struct SamPll {
  const float *T;  // arm_sin_f32's table (LDS)
  float omega_min, omega_max, g1, g2;
  float phzerror, fil_out, omega2, Sin, Cos;
  __device__ __forceinline__ void sincos(float ph, float &S, float &C) const {
#pragma clang fp contract(off)
    // (0 <= ph <= 2 pi: kept by the wrap below, by the power-on state and by t41rx_set_state's check)
    const SamIdx is = sam_table_index_pos(ph * 0.159154943092f), ic = sam_table_index_pos(ph * 0.159154943092f + 0.25f);
    const float sa = T[is.index], sb = T[is.index + 1], ca = T[ic.index], cb = T[ic.index + 1];
    S = (1.0f - is.fract) * sa + is.fract * sb;
    C = (1.0f - ic.fract) * ca + ic.fract * cb;
  }
  __device__ __forceinline__ void load(const float *Tab, const float *ms, CoefPtr cf0) {
    const CoefPtr c = fresh_coef(cf0);
    T = Tab;
    omega_min = c->sc[kScSamWmin], omega_max = c->sc[kScSamWmax], g1 = c->sc[kScSamG1], g2 = c->sc[kScSamG2];
    phzerror = ms[kMiscSamPhz], fil_out = ms[kMiscSamFil], omega2 = ms[kMiscSamOmega];
    sincos(phzerror, Sin, Cos);
  }
  __device__ __forceinline__ float step(cf z) {  // Demod.cpp:69-117 for one sample; returns the audio
#pragma clang fp contract(off)
    const float kTpi = 6.283185307179586476925286766559f;
    float phznext = phzerror + fil_out;  // (fil_out: still the previous step's = this step's del_out)
    // (the source's two `while` loops: |del_out| <= g1 (2 pi + pi / 4) + omega_max < 1.2, so one pass each)
    if (phznext >= kTpi) phznext -= kTpi;
    if (phznext < 0.0f) phznext += kTpi;
    // The next step's table entries are REQUESTED here and interpolated behind this step's detector (round 4): the
    // interpolation right behind the request, as the compiler places it when left alone, waits out an LDS round trip in
    // every step -- the longest single item of a step on the duty wave, whose fifteen neighbours keep the LDS pipe busy.
    const SamIdx is = sam_table_index_pos(phznext * 0.159154943092f), ic = sam_table_index_pos(phznext * 0.159154943092f + 0.25f);
    const float sa = T[is.index], sb = T[is.index + 1], ca = T[ic.index], cb = T[ic.index + 1];
#if T41RX_SAM_DEFER
    __builtin_amdgcn_sched_barrier(0);
#endif
    const float ai = Cos * z.x, bi = Sin * z.x, aq = Cos * z.y, bq = Sin * z.y;
    const float corr0 = +ai + bq, corr1 = -bi + aq;
    const float audio = (ai - bi) + (aq + bq);
    const float det = sam_atan2(corr1, corr0);
    omega2 = omega2 + g2 * det;
    omega2 = __builtin_amdgcn_fmed3f(omega2, omega_min, omega_max);  // Demod.cpp's if / else-if clamp (omega_min < omega_max, no NaNs): one instruction, no branches
    fil_out = g1 * det + omega2;
    phzerror = phznext;
#if T41RX_SAM_DEFER
    __builtin_amdgcn_sched_barrier(0);
#endif
    Sin = (1.0f - is.fract) * sa + is.fract * sb;  // sincos(phznext), second half
    Cos = (1.0f - ic.fract) * ca + ic.fract * cb;
    return audio;
  }
  __device__ __forceinline__ void store(float *ms) const {
    ms[kMiscSamPhz] = phzerror;
    ms[kMiscSamFil] = fil_out;
    ms[kMiscSamOmega] = omega2;
  }
};
// zs = the channel's 256 complex samples (audio replaces the real parts), T = arm_sin_f32's table in LDS, ms = the channel's kStMisc words
__device__ __forceinline__ void sam_chain(float *zs, const float *T, float *ms, CoefPtr cf0, bool store) {
  SamPll pll;
  pll.load(T, ms, cf0);
  cf zn = *reinterpret_cast<const cf *>(zs);
  for (int i = 0; i < 256; ++i) {
    const cf z = zn;
    if (i < 255) zn = *reinterpret_cast<const cf *>(zs + 2 * i + 2);  // ahead of the dependent chain
    zs[2 * i] = pll.step(z);
  }
  if (store) pll.store(ms);
}

template <typename AL>
__device__ __forceinline__ void agc_chain(float *sl, CoefPtr cf0, int lane STAMP_PARAMS) {
  constexpr int kAgR = AL::R, kAgS = AL::S;
  const CoefPtr c = fresh_coef(cf0);
  AgcConsts g;
  g.attack_mult = c->agc[kAgcAttackMult];
  g.decay_mult = c->agc[kAgcDecayMult];
  g.fast_decay_mult = c->agc[kAgcFastDecayMult];
  g.hang_decay_mult = c->agc[kAgcHangDecayMult];
  g.onemfast_backmult = c->agc[kAgcOnemFastBackmult];
  g.onemhang_backmult = c->agc[kAgcOnemHangBackmult];
  g.min_volts = c->agc[kAgcMinVolts];
  g.hang_level = c->agc[kAgcHangLevel];
  g.pop_ratio = c->agc[kAgcPopRatio];
  g.hang_count = (int)c->agc[kAgcHangCount];
  const float4 sf = lds4(sl + kAgS);
  const int4 si = *reinterpret_cast<const int4 *>(sl + kAgS + 4);
  AgcState st{sf.x, sf.y, sf.z, sf.w, si.x, si.y, si.z};
  AgcLane d = agc_lane_of(st, g);
  float4 nr4 = lds4(sl + kAgR), npa = lds4(sl + AL::pofs(0)), npb = lds4(sl + AL::pofs(4));
  for (int b = 0; b < 64; ++b) {
    const float4 r4 = nr4, pa = npa, pb = npb;
    if (b < 63) {  // the next four steps' operands, ahead of the dependent chain
      nr4 = lds4(sl + kAgR + 4 * b + 4);
      npa = lds4(sl + AL::pofs(8 * b + 8));
      npb = lds4(sl + AL::pofs(8 * b + 12));
    }
    const float rm[4] = {r4.x, r4.y, r4.z, r4.w};
    const float pf[4] = {pa.x, pa.z, pb.x, pb.z}, ph[4] = {pa.y, pa.w, pb.y, pb.w};
    float vo[4];
    AgcState t = st;
    AgcLane dt = d;
    const lanemask ok = agc_block(t, dt, st, d, g, rm, pf, ph, vo);
    if (~ok != 0) {  // some lane changes state other than by an attack
      STAMP(24);  // chain: fast blocks
      // only those lanes redo the block (the others' results stand): the lanes that share a
      // channel take the same branches, so the switch runs without divergence among them
      if (((~ok >> lane) & 1ull) != 0) {
        t = st;
        agc_slow_block(t, g, rm, pf, ph, vo);
      }
      dt = agc_lane_of(t, g);
      STAMP(25);  // chain: slow blocks
#ifdef T41RX_STAMP
      if (lane == 26) stamp_acc += 1;  // number of slow blocks
#endif
    }
    st = t;
    d = dt;
    *reinterpret_cast<float4 *>(sl + kAgR + 4 * b) = make_float4(vo[0], vo[1], vo[2], vo[3]);
  }
  *reinterpret_cast<float4 *>(sl + kAgS) = make_float4(st.fast_backaverage, st.hang_backaverage, st.volts, st.save_volts);
  *reinterpret_cast<int4 *>(sl + kAgS + 4) = make_int4(st.state, st.decay_type, st.hang_counter, 0);
}

// v[4 + j] = inverse FFT output sample i = lane + 64 j (the valid half); agst = this lane's
// float4 of the channel's AGC record (lanes 0..49 delay line, 50..51 state words).
// og[k] = AGC output sample 4 lane + k.
// slices: the first wave slice of the workgroup, NW / SLICE: waves per workgroup / floats per slice.
template <typename AL, int NW, int SLICE>
__device__ __forceinline__ void agc_apply(const cf (&v)[8], float4 agst, float *lds, float *slices, float *st_ag,
                                          CoefPtr cf0, int lane, int wv, int nvalid, cf (&og)[4] STAMP_PARAMS) {
  constexpr int kAgZ = AL::Z, kAgA = AL::A, kAgG = AL::G, kAgR = AL::R, kAgS = AL::S;
  wave_sync();
  if (lane < 50) *reinterpret_cast<float4 *>(lds + kAgZ + 4 * lane) = agst;
  else if (lane < 52) *reinterpret_cast<float4 *>(lds + kAgS + 4 * (lane - 50)) = agst;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    *reinterpret_cast<cf *>(lds + kAgZ + 2 * (100 + lane + 64 * j)) = v[4 + j];
    lds[kAgA + 100 + lane + 64 * j] = agc_mag(v[4 + j]);
  }
  if (lane < 50) *reinterpret_cast<float2 *>(lds + kAgA + 2 * lane) = make_float2(agc_mag(cf{agst.x, agst.y}), agc_mag(cf{agst.z, agst.w}));
  wave_sync();
  {  // maxima of the aligned groups of four
    float4 t = lds4(lds + kAgA + 4 * lane);
    lds[kAgG + lane] = fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w));
    if (lane < 25) {
      t = lds4(lds + kAgA + 256 + 4 * lane);
      lds[kAgG + 64 + lane] = fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w));
    }
  }
  wave_sync();
  {
    // sample i = 4 lane + j pops entry [i + 3] and its window is entries [i + 4 .. i + 100]:
    // the tail of group lane + 1, the 23 whole groups lane + 2 .. lane + 24, the head of the
    // lane's own new group
    const CoefPtr c = fresh_coef(cf0);
    const float fast_backmult = c->agc[kAgcFastBackmult], hang_backmult = c->agc[kAgcHangBackmult];
    const float4 nv = lds4(lds + kAgA + 100 + 4 * lane);
    const float4 g1 = lds4(lds + kAgA + 4 + 4 * lane);
    const float ao0 = lds[kAgA + 3 + 4 * lane];
    float C = lds[kAgG + lane + 2];
#pragma unroll
    for (int q = 3; q <= 24; ++q) C = fmaxf(C, lds[kAgG + lane + q]);
    const float s3 = g1.w, s2 = fmaxf(g1.z, s3), s1 = fmaxf(g1.y, s2), s0 = fmaxf(g1.x, s1);
    const float p0 = nv.x, p1 = fmaxf(p0, nv.y), p2 = fmaxf(p1, nv.z), p3 = fmaxf(p2, nv.w);
    *reinterpret_cast<float4 *>(lds + kAgR + 4 * lane) =
        make_float4(fmaxf(fmaxf(s0, C), p0), fmaxf(fmaxf(s1, C), p1), fmaxf(fmaxf(s2, C), p2), fmaxf(fmaxf(s3, C), p3));
    const float ao[4] = {ao0, g1.x, g1.y, g1.z};
    {
#pragma clang fp contract(off)
      *reinterpret_cast<float4 *>(lds + AL::pofs(8 * lane)) =
          make_float4(fast_backmult * ao[0], hang_backmult * ao[0], fast_backmult * ao[1], hang_backmult * ao[1]);
      *reinterpret_cast<float4 *>(lds + AL::pofs(8 * lane + 4)) =
          make_float4(fast_backmult * ao[2], hang_backmult * ao[2], fast_backmult * ao[3], hang_backmult * ao[3]);
    }
  }
  STAMP(19);  // AGC: magnitudes, look-ahead maximum
  if (kAgcCoop) {
    __syncthreads();
    STAMP(20);  // AGC: barrier 1
    // wave 0 runs the chains of all the workgroup's channels, one lane per channel (4-wave
    // workgroups: the hardware places the first waves of the workgroups sharing a CU on different
    // SIMDs, HW_ID dump of the -DT41RX_STAMP build, so the chains of a CU do not compete for issue slots)
    const int cw = 0;
    // All 64 lanes stay enabled (lane l redoes channel l mod nvalid): measured on MI355X
    // (tools/ubench/exec_mask.hip), VALU instructions of a wave with 16 or fewer active lanes
    // take 3-4x longer than with 32 or more.
    // The chain is the frame's critical path -- three waves wait for it -- and one dependent
    // instruction at a time: it gets the top issue priority (the other phases of these kernels
    // stay at 2 and below), or every one of its ~10 k instructions queues behind the parallel
    // phases of the other workgroups' waves on its SIMD (51 k cycles per frame, stamps).
    if (wv == cw) {
      PRIO(3);
      agc_chain<AL>(slices + (nvalid == NW ? (lane & (NW - 1)) : lane % nvalid) * SLICE, cf0, lane STAMP_ARGS);
      PRIO(1);
    }
    STAMP(21);  // AGC: the serial chain (chain wave only)
    __syncthreads();
    STAMP(22);  // AGC: barrier 2 (= waiting for the chain, for the other waves)
  } else {
    wave_sync();
    agc_chain<AL>(lds, cf0, lane STAMP_ARGS);
    wave_sync();
  }
  {
    const CoefPtr c = fresh_coef(cf0);
    const float inv_max_input = c->agc[kAgcInvMaxInput], out_target = c->agc[kAgcOutTarget], slope_constant = c->agc[kAgcSlopeConstant];
    const float4 vv = lds4(lds + kAgR + 4 * lane);
    const float vk[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const cf z = *reinterpret_cast<const cf *>(lds + kAgZ + 2 * (3 + 4 * lane + k));
      const float mult = agc_mult(vk[k], inv_max_input, out_target, slope_constant);
      og[k] = cf{z.x * mult, z.y * mult};
    }
  }
  // the record for the next frame: the newest 100 inputs and the state words
  if (lane < 50) *reinterpret_cast<float4 *>(st_ag + 4 * lane) = lds4(lds + kAgZ + 512 + 4 * lane);
  else if (lane < 52) *reinterpret_cast<float4 *>(st_ag + kAgcHistFloats + 4 * (lane - 50)) = lds4(lds + kAgS + 4 * (lane - 50));
  wave_sync();
  STAMP(23);  // AGC: gain from volts, scaling, record store
}

// ------------------------------------------------------------------------------------------
// AGC on, pipelined (rx512_kernel<..., PIPE = true>): the serial chain of frame g runs WHILE the
// other waves of the workgroup work on the front end of frame g + 1 / g + 2 and the back end of
// frame g - 1, instead of between two workgroup barriers.
//   * Geometry of the AGC-off kernel: one resident 16-wave workgroup per CU, one channel per wave,
//     filter memories on chip for the whole launch.
//   * The chain of frame g is run by ONE wave -- the first that gets to its duty point for g (a
//     compare-and-swap on the next unclaimed frame): the one furthest ahead, which is sure to be
//     waiting when the previous chain ends and can best afford to fall a chain behind -- for all
//     the workgroup's channels, one lane per channel, 16 channels' worth of instruction issue for
//     the price of one.  Its inputs (look-ahead maxima, |popped sample|) and
//     outputs (volts) and the popped samples the gain is applied to travel through a per-channel
//     ring of three slots in global memory (RxArgs::agc_pipe; the slices have no LDS left, and
//     the AGC's delay line rides in four registers): a wave's program per iteration f is
//         front end + AGC preparation of frame f     -> slot f mod 3, ready[f mod 3] += 1
//         (if it takes the duty for f - 1) chain of f - 1 -> waits for ready == channels and done == f - 1
//         gain + demodulator + back end of f - 2     -> waits for done > f - 2 (its loads are
//                                                       requested ahead of the preparation when the
//                                                       chain is done by then, which is the rule)
//     The back end trails by TWO frames: the wave that ran a chain is one chain (~60 k cycles)
//     behind the others from then on, and the next chain needs ITS channel's inputs too -- with a
//     single frame of slack that lag would sit on the chain's critical path every frame.
//   * Flags: five words of LDS behind the FFT twiddles; waits are bounded spins (a logic error
//     then shows as wrong samples in the parity tests, not as a hung GPU).  Release / acquire at
//     workgroup scope: the waves of a workgroup share the CU's vector memory path and L1, which
//     keeps their global accesses in order, so the fences cost an LDS wait and no vmcnt(0).
// Every value is computed by the same instructions as in agc_apply / agc_chain: bit-identical
// (tools/agc_pipe_probe.py, tests/test_gpu_parity.py::test_agc_pipelined_equals_barrier_form).
// Measured (MI355X, 4096 channels x 32 frames, tools/agc_pipe_round.sh, -DT41RX_PIPE_STAT counters in shader-clock
// cycles): 33-34 us per frame against 39-40 for the barrier form.  A chain takes 57-64 k cycles (175-200 per step +
// staging + ~8 k exposed at its start), a wave's front end 38 k, preparation 9 k, back end 10 k per frame; with
// the waits and a sixteenth of a chain ~68 k cycles = the measured period at the ~2.1 GHz sustained under this load.
// The slots cost ~10 KiB of fabric traffic per channel-frame on top of the 24 KiB of samples (they miss L2: the
// 4096-channel working set is 5 MiB per XCD), but removing half of it in a timing experiment bought 5 %: what binds
// is the chain's own latency (the protocol's model with these phase lengths: 60 k cycles per frame,
// tests/test_pipe_protocol_model.py) together with instruction issue (2528 VALU instructions per wave-frame against
// the AGC-off kernel's 1767 at the same 62 % VALU utilisation: every phase is stretched, the chain included).
// ------------------------------------------------------------------------------------------
constexpr int kPipeSlots = 3, kPipeSlotFloats = 1024;  // ring_max -> volts [256] | |popped| [256] | popped re [256] | popped im [256] (AM only)
constexpr int kPipeFlags = 1008;                        // float index in the table area: ready[3], done, next frame to claim
#ifndef T41RX_PIPE_CLAIM
#define T41RX_PIPE_CLAIM 1  // 0: the duty rotates (frame g -> wave g mod channels); measured 1.2 % slower
#endif
#ifndef T41RX_PIPE_SPINCAP
#define T41RX_PIPE_SPINCAP (1 << 20)  // (tools: a build with 1 exercises the time-out report)
#endif
constexpr int kPipeSpinCap = T41RX_PIPE_SPINCAP;

// -DT41RX_PIPE_STAT (diagnostic build, tools/build_variant.sh pstat -DT41RX_PIPE_STAT; T41RX_PIPE_STAT=1 prints them when
// the context is destroyed): 16 cycle counters per wave behind the slots -- [0] chain [1] chains [2] slow blocks [3] waiting
// for a chain's results [4] duty wave waiting before its chain [5] blocks [6] chain: staging [7] chain: the steps
// [8] front end [9] AGC preparation [10] back end [11] iterations
#ifdef T41RX_PIPE_STAT
#define PIPE_STAT_T0() const unsigned long long pipe_t0 = __builtin_readcyclecounter()
#define PIPE_STAT_ADD(k) do { if (lane == 0) pipe_stat[k] += __builtin_readcyclecounter() - pipe_t0; } while (0)  // (a wave's own eight counters)
#define PIPE_STAT_INC(k, n) do { if (lane == 0) pipe_stat[k] += (unsigned long long)(n); } while (0)
#else
#define PIPE_STAT_T0() do {} while (0)
#define PIPE_STAT_ADD(k) do {} while (0)
#define PIPE_STAT_INC(k, n) do {} while (0)
#endif
__device__ __forceinline__ unsigned pipe_flag_read(const unsigned *p) {
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// (err: a counter behind the slots.  A wait that runs out -- it cannot, unless the protocol is broken -- is counted
// there and the wave goes on: the host reports it at the next synchronising call, rx_host.cpp, instead of a hung GPU)
__device__ __forceinline__ void pipe_wait_ge(const unsigned *p, unsigned target, unsigned *err) {
  int it = 0;
  for (; it < kPipeSpinCap; ++it) {
    if (pipe_flag_read(p) >= target) break;
    __builtin_amdgcn_s_sleep(2);
  }
  if (it == kPipeSpinCap && (threadIdx.x & 63) == 0) atomicAdd(err, 1u);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// first half of agc_apply: magnitudes, look-ahead maxima; what the chain and the gain need -> slot
template <typename AL, bool NEED_IM>
// arec: the delay line's magnitudes (computed a frame ago as that frame's newest: carried, not recomputed)
__device__ __forceinline__ float4 agc_prep_pipe(const cf (&v)[8], float4 agst, float2 &arec, float *lds, float *slot, CoefPtr cf0, int lane) {
  constexpr int kAgZ = AL::Z, kAgA = AL::A, kAgG = AL::G;
  wave_sync();
  if (lane < 50) *reinterpret_cast<float4 *>(lds + kAgZ + 4 * lane) = agst;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    *reinterpret_cast<cf *>(lds + kAgZ + 2 * (100 + lane + 64 * j)) = v[4 + j];
    lds[kAgA + 100 + lane + 64 * j] = agc_mag(v[4 + j]);
  }
  if (lane < 50) *reinterpret_cast<float2 *>(lds + kAgA + 2 * lane) = arec;
  wave_sync();
  {
    float4 t = lds4(lds + kAgA + 4 * lane);
    lds[kAgG + lane] = fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w));
    if (lane < 25) {
      t = lds4(lds + kAgA + 256 + 4 * lane);
      lds[kAgG + 64 + lane] = fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w));
    }
  }
  wave_sync();
  {
    const float4 nv = lds4(lds + kAgA + 100 + 4 * lane);
    const float4 g1 = lds4(lds + kAgA + 4 + 4 * lane);
    const float ao0 = lds[kAgA + 3 + 4 * lane];
    float C = lds[kAgG + lane + 2];
#pragma unroll
    for (int q = 3; q <= 24; ++q) C = fmaxf(C, lds[kAgG + lane + q]);
    const float s3 = g1.w, s2 = fmaxf(g1.z, s3), s1 = fmaxf(g1.y, s2), s0 = fmaxf(g1.x, s1);
    const float p0 = nv.x, p1 = fmaxf(p0, nv.y), p2 = fmaxf(p1, nv.z), p3 = fmaxf(p2, nv.w);
    *reinterpret_cast<float4 *>(slot + 4 * lane) =
        make_float4(fmaxf(fmaxf(s0, C), p0), fmaxf(fmaxf(s1, C), p1), fmaxf(fmaxf(s2, C), p2), fmaxf(fmaxf(s3, C), p3));
    *reinterpret_cast<float4 *>(slot + 256 + 4 * lane) = make_float4(ao0, g1.x, g1.y, g1.z);
    cf z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = *reinterpret_cast<const cf *>(lds + kAgZ + 2 * (3 + 4 * lane + k));
    *reinterpret_cast<float4 *>(slot + 512 + 4 * lane) = make_float4(z[0].x, z[1].x, z[2].x, z[3].x);
    if (NEED_IM) *reinterpret_cast<float4 *>(slot + 768 + 4 * lane) = make_float4(z[0].y, z[1].y, z[2].y, z[3].y);
  }
  // the delay line for the next frame: the newest 100 inputs, kept in registers (lanes 0..49; the state words are the chain's)
  const float4 rec = lds4(lds + kAgZ + 512 + 4 * (lane < 50 ? lane : 0));
  arec = *reinterpret_cast<const float2 *>(lds + kAgA + 256 + 2 * (lane < 50 ? lane : 0));
  wave_sync();
  return rec;
}

// agc_chain for one channel per lane.  The operands wait in the channels' slots, i.e. in L2 / MALL / HBM behind the
// streaming traffic of fifteen other waves: several microseconds away.  The duty wave therefore moves them through
// its own LDS scratch (free between its AGC preparation and its back end) in chunks of 16 steps for all the
// channels at once -- lane (channel, quarter) loads one float4 of ring_max and one of |popped| per chunk --, requested
// T41RX_PIPE_AHEAD chunks ahead; the first ones before the wait for the previous frame's chain.
//   grp   : slot (frame g) of the workgroup's first channel; channel c's is kPipeSlots * kPipeSlotFloats * c further
//   stw0  : the eight state words of the workgroup's first channel, channel c's stride floats further
//   stage : kPipeStageFloats of LDS
constexpr int kPipeChunk = 16, kPipeChStride = 52;  // (T41RX_AGC_PHASED=0) per channel in the stage: ring_max -> volts [16] | (b x) pairs [32] | 4 pad
constexpr int kPipeChStrideP = 36;                  // per channel and half: ring_max [16] | |popped| [16] | 4 pad (conflict-free float4 rows)
constexpr int kPipeHalfFloats = 16 * kPipeChStrideP;
#ifndef T41RX_AGC_PHASED
#define T41RX_AGC_PHASED 1
#endif
constexpr int kPipeStageFloats = T41RX_AGC_PHASED ? 2 * kPipeHalfFloats : 16 * kPipeChStride;
#ifndef T41RX_PIPE_AHEAD
#define T41RX_PIPE_AHEAD 2  // chunks requested ahead (measured: 1, 2, 3, 4 within 1 %; 6 spills and is 23 % slower)
#endif
#if T41RX_AGC_PHASED
// Round 4: the chunk's operands are staged ONE CHUNK AHEAD into the other half of a double buffer (ring maxima and the raw
// |popped| -- 36 floats per channel and half instead of 52 with the two products, which the chain forms itself: one
// v_pk_mul_f32 per step, off the dependent path), so no block waits for an LDS write -> read round trip any more (the first
// block of every chunk did, ~185 cycles behind fifteen other waves' LDS traffic); the chunk's volts never touch the stage
// (see `keep`).  Chunk k + 1 is written at the top of chunk k and read a chunk later: the LDS unit executes one wave's
// instructions in order, the compiler is held by wave_sync().
__device__ __forceinline__ void agc_chain_pipe(float *grp, float *stw0, size_t stride, float *stage, const unsigned *done, unsigned g,
                                               int nvalid, CoefPtr cf0, int lane, unsigned long long *pipe_stat, unsigned *err) {
  const CoefPtr c = fresh_coef(cf0);
  AgcConsts gc;
  gc.attack_mult = c->agc[kAgcAttackMult];
  gc.decay_mult = c->agc[kAgcDecayMult];
  gc.fast_decay_mult = c->agc[kAgcFastDecayMult];
  gc.hang_decay_mult = c->agc[kAgcHangDecayMult];
  gc.onemfast_backmult = c->agc[kAgcOnemFastBackmult];
  gc.onemhang_backmult = c->agc[kAgcOnemHangBackmult];
  gc.min_volts = c->agc[kAgcMinVolts];
  gc.hang_level = c->agc[kAgcHangLevel];
  gc.pop_ratio = c->agc[kAgcPopRatio];
  gc.hang_count = (int)c->agc[kAgcHangCount];
  float fast_backmult = c->agc[kAgcFastBackmult], hang_backmult = c->agc[kAgcHangBackmult];
  float attack_mult_v = gc.attack_mult, min_volts_v = gc.min_volts;
  asm volatile("" : "+v"(fast_backmult), "+v"(hang_backmult), "+v"(attack_mult_v), "+v"(min_volts_v));  // VOP3P / v_max operands: VGPRs
  const f2 backmult = f2{fast_backmult, hang_backmult};
  const int ch = (nvalid == 16) ? (lane & 15) : (lane & 15) % nvalid;  // this lane's channel, as a loader and as a chain
  const int q = lane >> 4;                                              // the float4 of a chunk it moves
  float *gsrc = grp + (size_t)ch * (kPipeSlots * kPipeSlotFloats) + 4 * q;
  float *sw0 = stage + ch * kPipeChStrideP;
  constexpr int NCH = 256 / kPipeChunk;
  // chunks 0 and 1 requested before the wait for the previous frame's chain; from then on chunk k + 3 at the top of chunk k
  float4 pr0 = *reinterpret_cast<const float4 *>(gsrc), pa0 = *reinterpret_cast<const float4 *>(gsrc + 256);
  float4 pr1 = *reinterpret_cast<const float4 *>(gsrc + kPipeChunk), pa1 = *reinterpret_cast<const float4 *>(gsrc + 256 + kPipeChunk);
  {
    PIPE_STAT_T0();
    pipe_wait_ge(done, g, err);  // the previous frame's chain has left the state words
    PIPE_STAT_ADD(4);
  }
  PIPE_STAT_T0();
  float *stw = stw0 + (size_t)ch * stride;
  const float4 sf = *reinterpret_cast<const float4 *>(stw);
  const int4 si = *reinterpret_cast<const int4 *>(stw + 4);
  wave_sync();
  *reinterpret_cast<float4 *>(sw0 + 4 * q) = pr0;  // chunk 0 -> half 0
  *reinterpret_cast<float4 *>(sw0 + 16 + 4 * q) = pa0;
  pr0 = pr1;
  pa0 = pa1;
  pr1 = *reinterpret_cast<const float4 *>(gsrc + kPipeChunk * 2);
  pa1 = *reinterpret_cast<const float4 *>(gsrc + 256 + kPipeChunk * 2);
  wave_sync();
#ifdef T41RX_PIPE_STAT
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  if (lane == 0) pipe_stat[12] += __builtin_readcyclecounter() - pipe_t0;  // the state words' (and every older request's) round trip
  unsigned long long acc_stage = 0, acc_comp = 0, acc_slow = 0;
#endif
  AgcState st{sf.x, sf.y, sf.z, sf.w, si.x, si.y, si.z};
  AgcLane d = agc_lane_of(st, gc);
  AgcLaneP lp = agc_lanep_of(st, d, attack_mult_v);
  float4 r4 = lds4(sw0);  // chunk 0's first ring maxima; from then on every block requests its successor's
  // (rolled loops: one copy of the four-step block instead of sixteen; the register ring of requested chunks rotates by
  // moves.  Measured in round 3: the inner loop unrolled, four copies, runs 12 % faster per step and the kernel 6 % slower
  // (two copies: 1.5 % slower) -- the other waves' front and back ends share the instruction cache)
#pragma nounroll
  for (int k = 0; k < NCH; ++k) {
    float *sw = sw0 + (k & 1) * kPipeHalfFloats;
    {
#ifdef T41RX_PIPE_STAT
      const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
      // chunk k + 1 -> the other half (read from the top of the next iteration on); chunk k + 3 requested into the freed
      // registers (clamped: the last iterations re-read chunk 15, whose volts are stored after they have read it -- the
      // request is unconditional so that the registers are one value, not a merge)
      const float4 r4n = pr0, a4n = pa0;
      pr0 = pr1;
      pa0 = pa1;
      {
        const int kn = k + 3 < NCH ? k + 3 : NCH - 1;
        pr1 = *reinterpret_cast<const float4 *>(gsrc + kPipeChunk * kn);
        pa1 = *reinterpret_cast<const float4 *>(gsrc + 256 + kPipeChunk * kn);
      }
      wave_sync();
      if (k + 1 < NCH) {
        float *swn = sw0 + ((k + 1) & 1) * kPipeHalfFloats;
        *reinterpret_cast<float4 *>(swn + 4 * q) = r4n;
        *reinterpret_cast<float4 *>(swn + 16 + 4 * q) = a4n;
      }
      wave_sync();
#ifdef T41RX_PIPE_STAT
      acc_stage += __builtin_readcyclecounter() - ts0;  // staging the next chunk's operands
#endif
    }
#ifdef T41RX_PIPE_STAT
    const unsigned long long ts1 = __builtin_readcyclecounter();
#endif
    float4 keep = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma nounroll
    for (int b = 0; b < 4; ++b) {
      // the back-averages' operands of this block: requested now, used two steps further down; and the next block's ring
      // maxima -- the chunk's last block requests the NEXT chunk's first ones from the other half (the last chunk reads
      // whatever is there: unconditional, so nr4 is one value, not a merge) --, taken over at the end of the block: four
      // moves, and no LDS round trip between two blocks or two chunks
      const float4 a4 = lds4(sw + 16 + 4 * b);
      const float4 nr4 = lds4(b < 3 ? sw + 4 * (b + 1) : sw0 + ((k + 1) & 1) * kPipeHalfFloats);
      float vo[4];
      AgcState t = st;
      AgcLane dt = d;
      lanemask sand = 0;
      lanemask ok = agc_block_phased(t, dt, gc, lp, min_volts_v, r4, a4, backmult, vo, sand);
      if (__builtin_expect((sand | ~ok) != 0, 0)) {  // rare: this block again, by the forms that have every case
#pragma clang fp contract(off)
        const float rm[4] = {r4.x, r4.y, r4.z, r4.w};
        const float pf[4] = {fast_backmult * a4.x, fast_backmult * a4.y, fast_backmult * a4.z, fast_backmult * a4.w};
        const float ph[4] = {hang_backmult * a4.x, hang_backmult * a4.y, hang_backmult * a4.z, hang_backmult * a4.w};
        if (sand != 0) {  // a rounding boundary between the bracketing values somewhere: the double expression decides
          t = st;
          dt = d;
          ok = agc_fast_block<true>(t, dt, gc, rm, pf, ph, vo);
        }
        if (~ok != 0) {
#ifdef T41RX_PIPE_STAT
          acc_slow += 1;
#endif
          if (((~ok >> lane) & 1ull) != 0) {
            t = st;
            agc_slow_block(t, gc, rm, pf, ph, vo);
          }
          dt = agc_lane_of(t, gc);
          lp = agc_lanep_of(t, dt, attack_mult_v);
        }
      }
      st = t;
      d = dt;
      // The four lanes of a channel have computed the same four values; the lane whose quarter of the chunk this block is
      // keeps them for the chunk's store.  (Round 3 wrote them to the stage and read the chunk back: the compiler's wait
      // for the next block's operands then also waited for that write -- an LDS round trip, ~150 cycles, in every block.)
      {
        const lanemask mine = 0xffffull << (16 * b);
        keep.x = pick(mine, vo[0], keep.x);
        keep.y = pick(mine, vo[1], keep.y);
        keep.z = pick(mine, vo[2], keep.z);
        keep.w = pick(mine, vo[3], keep.w);
      }
      asm volatile("" : "+v"(vo[3]));  // (the take-over below stays behind the steps)
      r4 = nr4;
    }
#ifdef T41RX_PIPE_STAT
    acc_comp += __builtin_readcyclecounter() - ts1;  // the chunk's 16 steps
#endif
    *reinterpret_cast<float4 *>(gsrc + kPipeChunk * k) = keep;  // volts in ring_max's place
  }
  if (lane < nvalid) {
    *reinterpret_cast<float4 *>(stw) = make_float4(st.fast_backaverage, st.hang_backaverage, st.volts, st.save_volts);
    *reinterpret_cast<int4 *>(stw + 4) = make_int4(st.state, st.decay_type, st.hang_counter, 0);
  }
  PIPE_STAT_ADD(0);
  PIPE_STAT_INC(1, 1);
  PIPE_STAT_INC(5, 64);
#ifdef T41RX_PIPE_STAT
  PIPE_STAT_INC(2, acc_slow);
  PIPE_STAT_INC(6, acc_stage);
  PIPE_STAT_INC(7, acc_comp);
#endif
}
#else
__device__ __forceinline__ void agc_chain_pipe(float *grp, float *stw0, size_t stride, float *stage, const unsigned *done, unsigned g,
                                               int nvalid, CoefPtr cf0, int lane, unsigned long long *pipe_stat, unsigned *err) {
  const CoefPtr c = fresh_coef(cf0);
  AgcConsts gc;
  gc.attack_mult = c->agc[kAgcAttackMult];
  gc.decay_mult = c->agc[kAgcDecayMult];
  gc.fast_decay_mult = c->agc[kAgcFastDecayMult];
  gc.hang_decay_mult = c->agc[kAgcHangDecayMult];
  gc.onemfast_backmult = c->agc[kAgcOnemFastBackmult];
  gc.onemhang_backmult = c->agc[kAgcOnemHangBackmult];
  gc.min_volts = c->agc[kAgcMinVolts];
  gc.hang_level = c->agc[kAgcHangLevel];
  gc.pop_ratio = c->agc[kAgcPopRatio];
  gc.hang_count = (int)c->agc[kAgcHangCount];
  float fast_backmult = c->agc[kAgcFastBackmult], hang_backmult = c->agc[kAgcHangBackmult];
  asm volatile("" : "+v"(fast_backmult), "+v"(hang_backmult));
  const int ch = (nvalid == 16) ? (lane & 15) : (lane & 15) % nvalid;  // this lane's channel, as a loader and as a chain
  const int q = lane >> 4;                                              // the float4 of a chunk it moves
  float *gsrc = grp + (size_t)ch * (kPipeSlots * kPipeSlotFloats) + 4 * q;
  float *sw = stage + ch * kPipeChStride;
  constexpr int AH = T41RX_PIPE_AHEAD;
  float4 pr[AH], pa[AH];
#pragma unroll
  for (int u = 0; u < AH; ++u) {
    pr[u] = *reinterpret_cast<const float4 *>(gsrc + kPipeChunk * u);
    pa[u] = *reinterpret_cast<const float4 *>(gsrc + 256 + kPipeChunk * u);
  }
  {
    PIPE_STAT_T0();
    pipe_wait_ge(done, g, err);  // the previous frame's chain has left the state words
    PIPE_STAT_ADD(4);
  }
  PIPE_STAT_T0();
  float *stw = stw0 + (size_t)ch * stride;
  const float4 sf = *reinterpret_cast<const float4 *>(stw);
  const int4 si = *reinterpret_cast<const int4 *>(stw + 4);
#ifdef T41RX_PIPE_STAT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) pipe_stat[12] += __builtin_readcyclecounter() - pipe_t0;  // the state words' (and every older request's) round trip
#endif
  AgcState st{sf.x, sf.y, sf.z, sf.w, si.x, si.y, si.z};
  AgcLane d = agc_lane_of(st, gc);
#ifdef T41RX_PIPE_STAT
  unsigned long long acc_stage = 0, acc_comp = 0, acc_slow = 0;
#endif
  // (rolled loops: one copy of the four-step block -- ~1300 instructions -- instead of sixteen; the register ring
  // of requested chunks rotates by moves.  Measured: the inner loop unrolled, four copies, runs 12 % faster per step
  // and the kernel 6 % slower (two copies: 1.5 % slower) -- the other waves' front and back ends share the instruction cache)
#pragma nounroll
  for (int k = 0; k < 256 / kPipeChunk; ++k) {
    {
      const float4 r4c = pr[0], a4c = pa[0];
#pragma unroll
      for (int u = 0; u + 1 < AH; ++u) {
        pr[u] = pr[u + 1];
        pa[u] = pa[u + 1];
      }
      if (k + AH < 256 / kPipeChunk) {
        pr[AH - 1] = *reinterpret_cast<const float4 *>(gsrc + kPipeChunk * (k + AH));
        pa[AH - 1] = *reinterpret_cast<const float4 *>(gsrc + 256 + kPipeChunk * (k + AH));
      }
      wave_sync();
#ifdef T41RX_PIPE_STAT
      const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
      *reinterpret_cast<float4 *>(sw + 4 * q) = r4c;
      {
#pragma clang fp contract(off)
        *reinterpret_cast<float4 *>(sw + 16 + 8 * q) = make_float4(fast_backmult * a4c.x, hang_backmult * a4c.x, fast_backmult * a4c.y, hang_backmult * a4c.y);
        *reinterpret_cast<float4 *>(sw + 16 + 8 * q + 4) = make_float4(fast_backmult * a4c.z, hang_backmult * a4c.z, fast_backmult * a4c.w, hang_backmult * a4c.w);
      }
      wave_sync();
#ifdef T41RX_PIPE_STAT
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc_stage += __builtin_readcyclecounter() - ts0;  // staging the chunk's operands
#endif
    }
#ifdef T41RX_PIPE_STAT
    const unsigned long long ts1 = __builtin_readcyclecounter();
#endif
    // The next block's operands are requested at the top of a block and taken over at its END, behind the four steps:
    // by then they have long landed.  (Round 3 took them over at the top of the next block, i.e. waited out an LDS round
    // trip per block behind fifteen other waves' traffic; the request is unconditional -- the last block re-reads its
    // own -- so the registers are not a merge of old and new values.)
    float4 r4 = lds4(sw), pa4 = lds4(sw + 16), pb4 = lds4(sw + 20);
#pragma nounroll
    for (int b = 0; b < 4; ++b) {
      const int bn = b < 3 ? b + 1 : 3;
      float4 nr4 = lds4(sw + 4 * bn), npa = lds4(sw + 16 + 8 * bn), npb = lds4(sw + 16 + 8 * bn + 4);
      const float rm[4] = {r4.x, r4.y, r4.z, r4.w};
      const float pf[4] = {pa4.x, pa4.z, pb4.x, pb4.z}, ph[4] = {pa4.y, pa4.w, pb4.y, pb4.w};
      float vo[4];
      AgcState t = st;
      AgcLane dt = d;
      const lanemask ok = agc_block(t, dt, st, d, gc, rm, pf, ph, vo);
      if (~ok != 0) {
#ifdef T41RX_PIPE_STAT
        acc_slow += 1;
#endif
        if (((~ok >> lane) & 1ull) != 0) {
          t = st;
          agc_slow_block(t, gc, rm, pf, ph, vo);
        }
        dt = agc_lane_of(t, gc);
      }
      st = t;
      d = dt;
      *reinterpret_cast<float4 *>(sw + 4 * b) = make_float4(vo[0], vo[1], vo[2], vo[3]);  // (every lane of a channel writes the same)
      asm volatile("" : "+v"(vo[3]));  // (the take-over below stays behind the steps)
      r4 = nr4;
      pa4 = npa;
      pb4 = npb;
    }
    wave_sync();
#ifdef T41RX_PIPE_STAT
    acc_comp += __builtin_readcyclecounter() - ts1;  // the chunk's 16 steps
#endif
    *reinterpret_cast<float4 *>(gsrc + kPipeChunk * k) = lds4(sw + 4 * q);  // volts in ring_max's place
  }
  if (lane < nvalid) {
    *reinterpret_cast<float4 *>(stw) = make_float4(st.fast_backaverage, st.hang_backaverage, st.volts, st.save_volts);
    *reinterpret_cast<int4 *>(stw + 4) = make_int4(st.state, st.decay_type, st.hang_counter, 0);
  }
  PIPE_STAT_ADD(0);
  PIPE_STAT_INC(1, 1);
  PIPE_STAT_INC(5, 64);
#ifdef T41RX_PIPE_STAT
  PIPE_STAT_INC(2, acc_slow);
  PIPE_STAT_INC(6, acc_stage);
  PIPE_STAT_INC(7, acc_comp);
#endif
}

#endif

// last part of agc_apply: og[k] = popped sample 4 lane + k times the gain from volts
struct AgcGainIn { float4 vv, zr, zi; };  // requested ahead of the AGC preparation of the front end's frame, which hides the round trip
template <bool NEED_IM>
__device__ __forceinline__ AgcGainIn agc_gain_request(const float *slot, int lane) {
  AgcGainIn r;
  r.vv = *reinterpret_cast<const float4 *>(slot + 4 * lane);
  r.zr = *reinterpret_cast<const float4 *>(slot + 512 + 4 * lane);
  r.zi = make_float4(0, 0, 0, 0);
  if (NEED_IM) r.zi = *reinterpret_cast<const float4 *>(slot + 768 + 4 * lane);
  return r;
}
__device__ __forceinline__ void agc_gain_pipe(const AgcGainIn &in, CoefPtr cf0, cf (&og)[4]) {
  const CoefPtr c = fresh_coef(cf0);
  const float inv_max_input = c->agc[kAgcInvMaxInput], out_target = c->agc[kAgcOutTarget], slope_constant = c->agc[kAgcSlopeConstant];
  const float4 vv = in.vv, zr = in.zr, zi = in.zi;
  const float vk[4] = {vv.x, vv.y, vv.z, vv.w}, re[4] = {zr.x, zr.y, zr.z, zr.w}, im[4] = {zi.x, zi.y, zi.z, zi.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float mult = agc_mult(vk[k], inv_max_input, out_target, slope_constant);
    og[k] = cf{re[k] * mult, im[k] * mult};
  }
}

// ---- the synchronous detector (AGC off) on the same pipeline: the PLL of frame g on the duty wave, one lane per
// channel.  Slot: the frame's 256 complex samples (fixed gain applied) in time order [512] | the audio [256].
// The duty wave's scratch holds arm_sin_f32's table (read from the L2-resident constant table per chain: the
// resident geometry has no LDS for it) and the 16 channels' chunk of 16 samples.
// Measured (4096 channels): 72.6 -> 67.0 us per frame at 32 frames per launch, 74.5 -> 71.1 at 8: the loop itself is
// what binds (~510 cycles per step on one wave; here it shares its SIMD and the LDS pipe with three busy waves and
// runs ~65 us per frame), the pipeline only takes the other waves' 20 us off the path.
constexpr int kPipeSamZ = 520, kPipeSamStride = 36;  // stage: table [516 + pad] | per channel 16 complex + 4 pad
constexpr int kPipeSamStageFloats = kPipeSamZ + 16 * kPipeSamStride;
__device__ __forceinline__ void sam_prep_pipe(const cf (&v)[8], float fixed_gain, float *slot, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<cf *>(slot + 2 * (lane + 64 * j)) = v[4 + j] * splat(fixed_gain);
}
//   grp : slot (frame g) of the workgroup's first channel; ms0 : its kStMisc words, channel c's stride floats further
__device__ __forceinline__ void sam_chain_pipe(float *grp, float *ms0, size_t stride, float *stage, const float *tab, const unsigned *done,
                                               unsigned g, int nvalid, CoefPtr cf0, int lane, unsigned *err) {
  const int ch = (nvalid == 16) ? (lane & 15) : (lane & 15) % nvalid;
  const int q = lane >> 4;
  float *gsrc = grp + (size_t)ch * (kPipeSlots * kPipeSlotFloats);
  float *sw = stage + kPipeSamZ + ch * kPipeSamStride;
  constexpr int AH = T41RX_PIPE_AHEAD;
  float4 p0[AH], p1[AH];  // a chunk = 16 complex = 8 float4 per channel: this lane moves float4 q and q + 4
#pragma unroll
  for (int u = 0; u < AH; ++u) {
    p0[u] = *reinterpret_cast<const float4 *>(gsrc + 32 * u + 4 * q);
    p1[u] = *reinterpret_cast<const float4 *>(gsrc + 32 * u + 16 + 4 * q);
  }
  wave_sync();
  for (int i = lane; i < 516; i += 64) stage[i] = tab[i];
  pipe_wait_ge(done, g, err);  // the previous frame's loop has left the PLL words
  wave_sync();
  float *ms = ms0 + (size_t)ch * stride;
  SamPll pll;
  pll.load(stage, ms, cf0);
#pragma nounroll
  for (int k = 0; k < 16; ++k) {
    {
      const float4 a0 = p0[0], a1 = p1[0];
#pragma unroll
      for (int u = 0; u + 1 < AH; ++u) {
        p0[u] = p0[u + 1];
        p1[u] = p1[u + 1];
      }
      if (k + AH < 16) {
        p0[AH - 1] = *reinterpret_cast<const float4 *>(gsrc + 32 * (k + AH) + 4 * q);
        p1[AH - 1] = *reinterpret_cast<const float4 *>(gsrc + 32 * (k + AH) + 16 + 4 * q);
      }
      wave_sync();
      *reinterpret_cast<float4 *>(sw + 4 * q) = a0;
      *reinterpret_cast<float4 *>(sw + 16 + 4 * q) = a1;
      wave_sync();
    }
    cf zn = *reinterpret_cast<const cf *>(sw);
#pragma nounroll
    for (int i = 0; i < 16; ++i) {
      const cf z = zn;
      if (i < 15) zn = *reinterpret_cast<const cf *>(sw + 2 * i + 2);
      sw[2 * i] = pll.step(z);  // (every lane of a channel writes the same)
    }
    wave_sync();
    *reinterpret_cast<float4 *>(gsrc + 512 + 16 * k + 4 * q) = make_float4(sw[8 * q], sw[8 * q + 2], sw[8 * q + 4], sw[8 * q + 6]);
  }
  if (lane < nvalid) pll.store(ms);
}

constexpr int kModeSsb = 0, kModeAm = 1, kModeNfm = 2, kModeSam = 3;  // kernel template MODE
// the 4-wave geometry (Geo's second parameter): AGC on, and the synchronous detector, whose PLL is a serial
// chain run by one wave per workgroup like the AGC's
constexpr bool geo4(int mode, bool agc) { return agc || mode == kModeSam; }

// ------------------------------------------------------------------------------------------
// The fused kernel, FFT_LENGTH = 512
// ------------------------------------------------------------------------------------------
// PART 0: the whole chain for FFT_LENGTH 512.  PART 1 / PART 2 are the two ends of the
// FFT_LENGTH 4096 pipeline (front: loads .. /8 decimation + level adjust -> `mid`; back:
// `aud24` -> interpolators -> store): a 16384-sample frame is 8 consecutive 2048-sample segments
// for them, the 4096-point fast convolution in between is fastconv4096_kernel.
// PLAIN: band gain 1, |IQ amplitude correction| 1 and IQ phase correction 0 (the firmware defaults:
// bands[].RFgain 1, gwv.cpp:71-72).  Those stages then vanish from the instruction stream: the one
// thing left, the reference's I <- -I (Process.cpp:166), is folded into the sign of the RF-gain
// multiply of I, which is exact because the DC high-pass in between is linear and negation is exact
// (the I chain's carry is negated with it).
// AGC: AGCMode != 0 (see agc_apply); the demodulator then works on lane-contiguous samples.
// WQ15: the firmware's own sample format either side of the path -- q15 blocks from the
// AudioRecordQueues in (arm_q15_to_float, Process.cpp:102-111) and arm_float_to_q15 out
// (Process.cpp:936): a.I / a.Q / a.out then point at int16 samples, same [channel][frame*2048]
// layout.  The conversions are exact (x / 32768 folds into the RF-gain factor, a power of two)
// resp. CMSIS' truncating, saturating float -> q15.
// LDS geometry of rx512_kernel (floats).
// PART 0 -- the fused FFT_LENGTH 512 chain -- runs as ONE 16-wave workgroup per CU that owns all
// 160 KiB of LDS: [tw1, tw2 1024 | 16 wave slices of 2496].  A wave keeps its channel's streaming
// state ON CHIP across the frames of a launch: the /4 and /2 delay lines stay where the history
// rolls leave them (X[0..68), the first 12 slots of either Y1 plane), the overlap-save block and the x2 interpolator history
// have their own slots (OV, H1), the x4 history and the NCO / DC scalars live in registers.  HBM
// state is read before the first frame and written after the last.  Everything the back end needs
// as scratch (FFT exchange, AGC, x2 window, output transposition in two halves) therefore avoids
// those regions: it lives in X[80 ..) (and, for the AGC, the free parts of Y1).  Only the FFT twiddles are staged in
// LDS; the filter mask comes from the (L2-resident) constant table per frame: that is what makes
// the slices fit.
// PART 1 / 2 -- the two ends of the long-FFT pipeline -- keep 4-wave workgroups, 4 per CU:
// [mask, tw1, tw2 2032 | 4 slices of 2052], scratch from the start of the slice.
#ifndef T41RX_RESIDENT
#define T41RX_RESIDENT 1  // 0 (experiments): the fused kernel with the 4-wave geometry and per-frame HBM state
#endif
// AGC on (PART 0): back to 4-wave workgroups and per-frame state in HBM.  The serial gain law
// runs on one wave of the workgroup between two workgroup barriers; with 16 waves behind one
// barrier the whole CU stops for every chain (47.6 us per 4096 x 2048 frame), four independent
// workgroups per CU keep the other twelve waves busy (40.6 us).
template <int PART, bool AGC = false>
struct Geo {
  static constexpr bool kResident = (PART == 0) && !AGC && T41RX_RESIDENT;
  static constexpr int kWaves = kResident ? 16 : 4;
  static constexpr int kTab = kResident ? 1024 : kLdsTabFloats;
  static constexpr int kTw1 = kResident ? 0 : kLdsTabTw1;        // float2 units within the tables
  static constexpr int kTw2 = kResident ? 448 : kLdsTabTw2;
  static constexpr int kSlice = kResident ? 2496 : kLdsFloatsPerWave;
  static constexpr int kTotal = kTab + kWaves * kSlice;
  static constexpr int kX = 0;
  static constexpr int kXF = kResident ? 1348 : kXFloats;  // 2 * (xpad(539) + 1)
  static constexpr int kY1 = kXF;
  static constexpr int kOV = kY1 + kY1Floats;   // resident: overlap-save "previous" block, 256 complex in [j][lane] order
  static constexpr int kH1 = kOV + 512;         // resident: x2 interpolator history, 24 floats
  static constexpr int kScr = kResident ? 80 : 0;        // FFT exchange (1152), overlap assembly, output transposition
  static constexpr int kI1 = kScr;                       // x2 interpolator window: 24 history + 256 new
};
static_assert(!Geo<0>::kResident || Geo<0>::kTotal * sizeof(float) == 160 * 1024, "PART 0: one workgroup owns the CU's LDS");
static_assert(Geo<1>::kTotal * sizeof(float) == 40960, "PART 1/2: four workgroups per CU");
static_assert(!Geo<0>::kResident || (Geo<0>::kH1 + 24 <= Geo<0>::kSlice && Geo<0>::kScr + 8 * kFftRow * 2 <= Geo<0>::kXF &&
                                     Geo<0>::kI1 + 284 <= Geo<0>::kXF && Geo<0>::kScr >= 68), "resident LDS layout");

// SEGPAR (PART 1 without NFM, PART 2 without AGC / AM): one wave per (channel, run of a.seg_run
// consecutive 2048-sample segments) instead of one wave per channel looping over all its segments
// -- the host picks the run length so that the long-FFT pipeline, whose 1024-channel batch
// otherwise leaves 12 of a CU's 16 wave slots empty, fills the chip once or twice over.
// A wave that does not start at the first segment of the call rebuilds the filter memories it
// needs from the PRECEDING input instead of receiving them from its predecessor: front end = one
// extra sub-block (the previous segment's last 512 samples) through DC high-pass, mixer and /4
// decimator, which yields the /4 history, the last 48 /4 outputs (= the /2 history) and both DC
// high-pass chain states exactly (a1^512 ~ 1e-35, the oscillator phase is closed form and past its
// start-up transient); back end = the previous segment's last 28 audio samples.  The channel's
// state is written by the wave that READ it (the one that starts the call), from the call's last
// samples in the same way: a wave of a later run may execute before that one has started.
#ifdef T41RX_CLK
// Diagnostic build only (-DT41RX_CLK, tools/clock_probe.py): every wave leaves the shader-clock and the constant
// 100 MHz counter's ticks between its start and its end here (its clock under this load = their ratio x 100 MHz).
__device__ __attribute__((visibility("default"))) unsigned long long g_t41_clk[4 * 8192];  // per wave: shader cycles, 100 MHz ticks, start tick, HW_ID | XCC_ID << 32
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
#else
#define T41RX_CLK_BEGIN() do {} while (0)
#define T41RX_CLK_END(wave_id) do {} while (0)
#endif
template <int MODE, bool DEBUG, int PART, bool PLAIN, bool AGC = false, bool WQ15 = false, bool SEGPAR = false, bool PIPE = false>
__global__ __launch_bounds__((Geo<PART, geo4(MODE, AGC) && !PIPE>::kWaves * 64), 4) void rx512_kernel(const RxArgs a) {
  T41RX_CLK_BEGIN();
  static_assert(!SEGPAR || (PART == 1 && MODE != kModeNfm) || (PART == 2 && MODE == kModeSsb && !AGC), "SEGPAR variants");
  static_assert(!PIPE || ((AGC || MODE == kModeSam) && PART == 0 && !DEBUG && !SEGPAR && T41RX_RESIDENT),
                "PIPE: the pipelined variants -- AGC on (see agc_prep_pipe), the synchronous detector with the AGC off (sam_chain_pipe), or both (PSA)");
  constexpr bool PSAM = PIPE && MODE == kModeSam && !AGC;
  // round 4: the synchronous detector behind the AGC -- TWO serial chains per frame, each on a duty wave of its own,
  // four frames deep: front end + AGC preparation (f), AGC chain (f - 1), gain + hand-over to the PLL (f - 2), PLL
  // chain (f - 3), interpolators and stores (f - 4).  Two instances of the same three-slot protocol in a row.
  constexpr bool PSA = PIPE && MODE == kModeSam && AGC;
  // input sub-blocks of the NEXT frame requested across the back end (the pipelined kernels hold them across the
  // preparation, a chain and the back end of an older frame: registers that spill there)
  constexpr int kPF = PSAM ? T41RX_PIPE_PF_SAM : PIPE ? T41RX_PIPE_PF : T41RX_PF;
  typedef Geo<PART, geo4(MODE, AGC) && !PIPE> G;
  constexpr bool KEEP = G::kResident;  // streaming state stays on chip across the frames of a launch
  constexpr int NW = G::kWaves;
  constexpr int kX = G::kX, kY1 = G::kY1, kScr = G::kScr, kI1 = G::kI1;
  __shared__ __attribute__((aligned(16))) float smem[G::kTotal];
  constexpr int L = 2048, D = 256, N = 512;
  // per-channel state record size follows fft_length = 512 * (segments per frame)
  const int seg = (PART == 0) ? 1 : a.seg;
  const size_t state_stride = state_floats(512 * seg);
  // `lane` is re-defined through an opaque asm at every phase boundary (FRESH_LANE): addresses
  // derived from it are then recomputed per phase (one or two VALU instructions each) instead of
  // being hoisted out of the frame loop, kept live through every other phase and spilled.
  int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int job = blockIdx.x * NW + wv;                       // SEGPAR: (channel, run) pairs, run fastest
  const int runs = SEGPAR ? (a.nframes + a.seg_run - 1) / a.seg_run : 1;
  const int ch = SEGPAR ? job / runs : job;
  const int seg0 = SEGPAR ? (job - ch * runs) * a.seg_run : 0;                                      // first segment / frame this wave runs
  const int seg1 = SEGPAR ? (seg0 + a.seg_run < a.nframes ? seg0 + a.seg_run : a.nframes) : a.nframes;  // one past the last

  // mask + twiddles are staged in LDS once per workgroup (the only workgroup barrier).  The
  // staging runs AFTER the first frame's global loads have been issued, so its latency and the
  // barrier overlap with the HBM latency of the input instead of preceding it.
  auto stage_tables = [&]() {
    const float4 *src = reinterpret_cast<const float4 *>(a.tab);
    float4 *dst = reinterpret_cast<float4 *>(smem);
    if (PART == 1) {
      // the 4096 front end has no FFT: the mask's place holds the oscillator's (cos, sin) table,
      // whose per-sub-block lookup is otherwise a global load nothing hides at 4 waves per CU
      if (threadIdx.x < 128) dst[threadIdx.x] = reinterpret_cast<const float4 *>(a.tab + kTabSinCos)[threadIdx.x];
    } else if (KEEP) {
      // tw1 [7][64] and tw2 compacted to [7][8] -- 4032 B; the mask is read from the (L2-resident)
      // constant table per frame instead: its loads are issued ahead of the forward FFT, which
      // hides them, whereas the twiddles are needed the moment the FFT starts
      if (threadIdx.x < 224) dst[threadIdx.x] = src[kTabTw1 / 2 + threadIdx.x];
      else if (threadIdx.x >= 256 && threadIdx.x < 256 + 56)
        reinterpret_cast<float2 *>(smem)[G::kTw2 + threadIdx.x - 256] =
            a.tab[kTabTw2 + 64 * ((threadIdx.x - 256) >> 3) + ((threadIdx.x - 256) & 7)];
      else if (PIPE && threadIdx.x >= 320 && threadIdx.x < 332)
        reinterpret_cast<unsigned *>(smem)[kPipeFlags + threadIdx.x - 320] = 0u;  // ready[3], done, claim (see agc_prep_pipe); PSA: the PLL stage's five behind them
    } else if (MODE == kModeSam) {
      // the mask is read from the L2-resident table (as the resident kernels do); its place holds
      // arm_sin_f32's 513-entry table for the PLL's per-lane look-ups
      for (int i = threadIdx.x; i < 516; i += 256) smem[i] = reinterpret_cast<const float *>(a.tab + kTabSam)[i];
      for (int i = 512 / 2 + threadIdx.x; i < (512 + 448) / 2; i += 256) dst[i] = src[i];  // tw1
      if (threadIdx.x < 56)
        reinterpret_cast<float2 *>(smem)[kLdsTabTw2 + threadIdx.x] = a.tab[kTabTw2 + 64 * (threadIdx.x >> 3) + (threadIdx.x & 7)];
    } else {
      for (int i = threadIdx.x; i < (512 + 448) / 2; i += 256) dst[i] = src[i];  // mask, tw1
      if (threadIdx.x < 56)  // tw2[q][l1] = table entry [q][lane = l1]
        reinterpret_cast<float2 *>(smem)[kLdsTabTw2 + threadIdx.x] = a.tab[kTabTw2 + 64 * (threadIdx.x >> 3) + (threadIdx.x & 7)];
    }
    __syncthreads();
  };
  // per-lane constants of the DC high-pass scan
  const float2 hp8 = a.tab[kTabHp8 + lane];
  const float2 hp4 = a.tab[kTabHp4 + lane];
  if (ch >= a.nchan) {  // ragged last workgroup: help with the staging, meet the barrier, leave (SEGPAR: ch = job / segments)
    if (PART != 2) stage_tables();
    return;
  }

  const cf *ltab = reinterpret_cast<const cf *>(smem);
  float *lds = smem + G::kTab + wv * G::kSlice;
  float *st = a.state + (size_t)(T41RX_ABLATE == 9 ? (ch & 15) : ch) * state_stride;
  // coefficients are read-only for the kernel: constant address space -> scalar (SMEM) loads,
  // re-derived through an opaque asm per phase so the compiler keeps the tap loads next to
  // their use instead of hoisting all 180 of them (and spilling SGPRs).
  const CoefPtr cf0 = (CoefPtr)a.coef;
  const NcoPtr nco = (NcoPtr)(a.nco + ch);
  const float2 *__restrict__ tab = a.tab;

  // per-channel NCO constants and state (wave-uniform).  Only the LOADS are issued here; the
  // values are made uniform (which waits for them) after the first frame's input loads are in
  // flight, so the kernel's cold start is one memory round trip, not a chain of them.
  // Long FFT: the oscillator state is kept twice and the copies alternate from call to call
  // (a.nco_rd = the one to read; the other one is written).  With one wave per segment every wave
  // of a channel needs the phase the call STARTED with, and a wave may start -- on another XCD --
  // after the wave that ends the call has already stored the new one.
  const NcoState *ncs_rd = reinterpret_cast<const NcoState *>(st + kStNco) + (PART == 0 ? 0 : a.nco_rd);
  NcoState *ncs = reinterpret_cast<NcoState *>(st + kStNco) + (PART == 0 ? 0 : (a.nco_rd ^ 1));
  const uint64_t raw_dphi = nco->phase_inc;
  const double raw_rs = nco->r_star_sq;
  const uint64_t raw_phase = ncs_rd->phase;
  const double raw_r = ncs_rd->r;
  const float raw_dc = st[kStMisc + kMiscDc];
  uint64_t dphi = 0, phase0 = 0;
  double osc_r = 1.0;
  float dc_carry = 0.0f;
  bool transient = false;

  f2 dc2 = splat(0.0f);  // DC high-pass carries, see below
  // input registers, two sub-blocks in flight (even / odd).  They live across iterations because
  // the 4096 front end (PART 1: 8 segments per frame, only 4 waves per CU to hide anything)
  // requests the NEXT segment's first two sub-blocks while it finishes the current one.
  float4 pI0[2], pI1[2], pQ0[2], pQ1[2];
  float4 tailN = make_float4(0, 0, 0, 0);  // KEEP: the next frame's last 256 I samples, 4 per lane
  float2 tailNq = make_float2(0, 0);       // (WQ15: as q15 words)
  // PART 2 (4096 back end, same situation): interpolator histories and the next segment's audio
  cf nfm_carry = splat(0.0f);  // PART 1, NFM: the previous segment's last complex sample
  float4 hist1c = make_float4(0, 0, 0, 0);
  float hist2c = 0.0f, audn[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float4 agrec = make_float4(0, 0, 0, 0);  // PIPE: the AGC's delay line (its last 100 inputs), lanes 0..49, across the frames of a launch
  float2 agmag = make_float2(0, 0);        // ... and its magnitudes
#ifdef T41RX_STAMP
  unsigned long long stamp_acc = 0, stamp_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
  {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (lane == 27) stamp_acc = hwid | ((unsigned long long)(xcc & 0xf) << 32);  // placement of this wave
    unsigned long long rt;                // constant-rate counter (100 MHz): start / end of the wave
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
    if (lane == 28) stamp_acc = rt;
  }
#endif
  // SEGPAR back end: the x2 interpolator's outputs for the four audio samples that END at `end`
  // (what the wave owning them computes for its inputs 252..255 from the window end[-28 .. -1];
  // every lane computes all eight: uniform addresses).  xp[1..7] = the x4 interpolator's history.
  auto x2_tail = [&](const float *end, float (&xp)[8]) {
    float wp[28];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const float4 t = *reinterpret_cast<const float4 *>(end - 28 + 4 * i);
      wp[4 * i] = t.x;
      wp[4 * i + 1] = t.y;
      wp[4 * i + 2] = t.z;
      wp[4 * i + 3] = t.w;
    }
    f2 up[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) up[u] = splat(0.0f);
#pragma unroll
    for (int b = 0; b < 24; b += 8) {
      float ci[16];
      load_taps<16>(ci, (CFloatPtr)cf0->int1 + 2 * b);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int t = 0; t < 8; ++t) up[u] = pk_fma(splat(wp[u + b + t + 1]), f2{ci[1 + 2 * t], ci[2 * t]}, up[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xp[2 * u] = up[u].x;
      xp[2 * u + 1] = up[u].y;
    }
  };
  // PIPE: the back end trails the front end by two frames (see agc_prep_pipe): two more iterations
  constexpr int kSkew = PSA ? 4 : PIPE ? 2 : 0;
  for (int f = seg0; f < seg1 + kSkew; ++f) {
#ifdef T41RX_PIPE_STAT
    unsigned long long ps_t = __builtin_readcyclecounter();  // [8] front end [9] AGC preparation [10] back end [11] iterations
#endif
    if (AGC) PRIO(2); else PRIO(3);  // (AGC on: 3 is the serial chain's, see agc_apply)
    FRESH_LANE();
    const bool first_iter = (f == seg0);
    // sample offset of (channel, frame) in I / Q / audio: RxArgs::chan_stride / frame_stride (channel-major
    // [channel][frame][2048]: nframes * 2048 and 2048; time-major [frame][channel][2048]: 2048 and nchan * 2048)
    const size_t chbase = (size_t)(T41RX_ABLATE == 9 ? (ch & 15) : ch) * (size_t)a.chan_stride;
    const size_t fbase = chbase + (size_t)f * (size_t)a.frame_stride;
    const int fb = PIPE ? (f >= kSkew ? f - kSkew : 0) : f;  // the frame the back end works on
    const size_t fbase_o = chbase + (size_t)fb * (size_t)a.frame_stride;
    const size_t fstep = WQ15 ? (size_t)a.frame_stride / 2 : (size_t)a.frame_stride;  // this channel's next frame, in float slots
    // (WQ15: two samples per float slot, so sample offsets halve)
    const float *__restrict__ gI = a.I + (WQ15 ? fbase / 2 : fbase);
    const float *__restrict__ gQ = a.Q + (WQ15 ? fbase / 2 : fbase);
    float *__restrict__ gO = a.out + (WQ15 ? fbase_o / 2 : fbase_o);

    constexpr bool CONTIG = (MODE == kModeAm) || (AGC && MODE != kModeSam) || PSAM || PSA;  // aud[j] = sample 4 lane + j instead of lane + 64 j
    float aud[4];                            // 4 demodulated samples @24 kS/s
    float4 agst = make_float4(0, 0, 0, 0);   // AGC record (delay line + state words), one float4 per lane
    float4 hist1 = make_float4(0, 0, 0, 0);  // x2 interpolator history (lanes 0..5)
    float hist2 = 0.0f;                      // x4 interpolator history (lane i = entry i)
    cf v[8];  // FFT registers; v[0..3] = previous block, v[4..7] = new block / valid half of the result
    // back half of the long-FFT pipeline with AM or the AGC on: the fast convolution hands over the
    // complex valid half (no gain applied) and the AGC / demodulator below run here per segment
    constexpr bool LONGC = (PART == 2) && (MODE == kModeAm || AGC);
    if (PART == 2 && LONGC) {
      const cf *yc = reinterpret_cast<const cf *>(a.aud24) + ((size_t)ch * a.nframes + f) * D;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 + j] = yc[lane + 64 * j];
      if (lane < 6) hist1 = *reinterpret_cast<const float4 *>(st + kStInt1 + 4 * lane);
      if (lane < 8) hist2 = st[kStInt2 + lane];
      if (AGC && lane < 52) agst = *reinterpret_cast<const float4 *>(st + st_agc(512 * seg) + 4 * lane);
    } else if (PART == 2) {
      // back half of the 4096 pipeline: this segment's 256 audio samples come from the
      // fast-convolution kernel
      const float *au = a.aud24 + ((size_t)ch * a.nframes + f) * D;
      if (SEGPAR && first_iter && f > 0) {
        // this wave starts inside the call: both interpolator histories come from the previous
        // segment's audio -- the x2 history is its last 23 samples as they are, the x4 history the
        // last 7 outputs of the x2 interpolator (x2_tail)
#pragma unroll
        for (int j = 0; j < 4; ++j) aud[j] = au[lane + 64 * j];
        if (lane < 6) hist1 = *reinterpret_cast<const float4 *>(au - 24 + 4 * lane);
        float xp[8];
        x2_tail(au, xp);
        hist2 = xp[1];  // lane i = entry i (i = 1..7)
#pragma unroll
        for (int i = 2; i < 8; ++i) hist2 = (lane == i) ? xp[i] : hist2;
      } else if (first_iter) {
#pragma unroll
        for (int j = 0; j < 4; ++j) aud[j] = au[lane + 64 * j];
        if (lane < 6) hist1 = *reinterpret_cast<const float4 *>(st + kStInt1 + 4 * lane);
        if (lane < 8) hist2 = st[kStInt2 + lane];
      } else {  // requested / kept during the previous segment
#pragma unroll
        for (int j = 0; j < 4; ++j) aud[j] = audn[j];
        hist1 = hist1c;
        hist2 = hist2c;
      }
      if (f + 1 < seg1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) audn[j] = au[D + lane + 64 * j];
      }
    } else if (!PIPE || f < seg1) {
      // ---- first loads of the frame, issued in the order they are needed (vmcnt retires in
      // order): sub-block 0, the I tail for Q's DC-block start state, the delay lines, then
      // sub-block 1.  Input prefetch runs TWO sub-blocks ahead (two register sets, even / odd).
      // PART 1, second and later segments of a call: the inputs are already on their way and the
      // delay lines are still in LDS where the history rolls left them.  KEEP (PART 0), second and
      // later frames of a launch: the delay lines likewise.
      // KEEP: the same for the inputs -- the next frame's first two sub-blocks and its I tail are
      // requested while the current frame's last two sub-blocks are processed and arrive under its
      // back end, so a wave does not sit out a memory round trip at every frame start.
      const bool hist_carried = (PART == 1 || KEEP) && !first_iter;
      const bool carried = (PART == 1 || (KEEP && kPF >= 1)) && !first_iter;   // sub-block 0 (and the tail)
      const bool carried1 = (PART == 1 || (KEEP && kPF >= 2)) && !first_iter;  // sub-block 1
      const bool preroll = SEGPAR && first_iter && f > 0;  // this wave rebuilds its filter memories from the preceding input
      if (preroll) {  // requested first, into register set 1 (sub-block 1 is requested once the pre-roll is done)
        if (!WQ15) {
          pI0[1] = ldg_stream(gI - 512 + 8 * lane);
          pI1[1] = ldg_stream(gI - 512 + 8 * lane + 4);
          pQ0[1] = ldg_stream(gQ - 512 + 8 * lane);
          pQ1[1] = ldg_stream(gQ - 512 + 8 * lane + 4);
        } else {
          pI0[1] = ldg_stream(gI - 256 + 4 * lane);
          pQ0[1] = ldg_stream(gQ - 256 + 4 * lane);
        }
      }
      float4 tailI;
      if (KEEP && carried) {
        tailI = WQ15 ? make_float4(q15_lo(tailNq.x), q15_hi(tailNq.x), q15_lo(tailNq.y), q15_hi(tailNq.y)) : tailN;
      } else if (!WQ15) {
        if (!carried) {
          pI0[0] = ldg_stream(gI + 8 * lane);
          pI1[0] = ldg_stream(gI + 8 * lane + 4);
          pQ0[0] = ldg_stream(gQ + 8 * lane);
          pQ1[0] = ldg_stream(gQ + 8 * lane + 4);
        }
        tailI = *reinterpret_cast<const float4 *>(gI + (L - 256) + 4 * lane);
      } else {  // 8 samples = 16 bytes per lane and array
        if (!carried) {
          pI0[0] = ldg_stream(gI + 4 * lane);
          pQ0[0] = ldg_stream(gQ + 4 * lane);
        }
        const float2 t = *reinterpret_cast<const float2 *>(gI + (L - 256) / 2 + 2 * lane);
        tailI = make_float4(q15_lo(t.x), q15_hi(t.x), q15_lo(t.y), q15_hi(t.y));
      }
      float4 h1 = make_float4(0, 0, 0, 0), h2 = make_float4(0, 0, 0, 0);
      float4 ovl0 = make_float4(0, 0, 0, 0), ovl1 = ovl0, ovl2 = ovl0;  // KEEP, first frame: overlap block, x2 history
      if (!hist_carried && !preroll) {
        if (lane < 14) h1 = *reinterpret_cast<const float4 *>(st + kStDec1 + 4 * lane);
        if (lane < 24) h2 = *reinterpret_cast<const float4 *>(st + kStDec2 + 4 * lane);
        if (KEEP) {  // the rest of the channel's record: overlap block, interpolator histories
          ovl0 = *reinterpret_cast<const float4 *>(st + kStOverlap + 4 * lane);
          ovl1 = *reinterpret_cast<const float4 *>(st + kStOverlap + 256 + 4 * lane);
          if (lane < 6) ovl2 = *reinterpret_cast<const float4 *>(st + kStInt1 + 4 * lane);
          if (lane < 8) hist2c = st[kStInt2 + lane];
        }
      }
      // gains (Process.cpp:117-134, 165-166).  g_band and -IQAmp are folded into one factor on I
      // (exact whenever either is +-1, which is the firmware default; one rounding otherwise)
      float g_rf, g_rf_i, g_hp, g_hp_i, iq_phase_neg = 0.0f, iq_phase_pos = 0.0f;
      f2 g_iq = splat(1.0f);
      {
        g_rf = a.g_rf;
        if (WQ15) g_rf *= 1.0f / 32768.0f;  // arm_q15_to_float
        // PLAIN: sign of the I path (-1 when the IQ amplitude correction applies, Process.cpp:165-173)
        g_rf_i = (PLAIN && a.iq_corr_on) ? -g_rf : g_rf;
        g_hp = g_rf * (float)kHpB0;  // what the samples are multiplied by: the DC high-pass takes b0 x (dc_highpass)
        g_hp_i = (PLAIN && a.iq_corr_on) ? -g_hp : g_hp;
        if (!PLAIN) {
          const float gb = a.g_band;
          const bool iq_on = a.iq_corr_on != 0;
          g_iq = f2{iq_on ? gb * a.neg_iq_amp : gb, gb};
          const float ph = iq_on ? a.iq_phase : 0.0f;
          iq_phase_neg = ph < 0.0f ? ph : 0.0f;
          iq_phase_pos = ph > 0.0f ? ph : 0.0f;
        }
      }
      STAMP(16);  // prologue a: issue + scalar (SMEM) gains
      if (first_iter) stage_tables();
      STAMP(17);  // prologue b: table staging + workgroup barrier (first vmcnt wait)
      if (preroll) {
        // (register set 1 still holds the pre-roll)
      } else if (!WQ15) {
        if (!carried1) {
          pI0[1] = ldg_stream(gI + 512 + 8 * lane);
          pI1[1] = ldg_stream(gI + 512 + 8 * lane + 4);
          pQ0[1] = ldg_stream(gQ + 512 + 8 * lane);
          pQ1[1] = ldg_stream(gQ + 512 + 8 * lane + 4);
        }
      } else if (!carried1) {
        pI0[1] = ldg_stream(gI + 256 + 4 * lane);
        pQ0[1] = ldg_stream(gQ + 256 + 4 * lane);
      }

      // ---- delay lines -> LDS (first frame of a launch / every segment-0; afterwards they are
      // where the history rolls left them)
      wave_sync();
      if (!hist_carried && !preroll) {
        if (lane < 14) *reinterpret_cast<float4 *>(lds + kX + 2 * xpad(2 * lane)) = h1;
        if (lane < 24) *reinterpret_cast<float4 *>(lds + kY1 + y1slot(lane)) = h2;
        if (KEEP) {
          *reinterpret_cast<float4 *>(lds + G::kOV + 4 * lane) = ovl0;
          *reinterpret_cast<float4 *>(lds + G::kOV + 256 + 4 * lane) = ovl1;
          if (lane < 6) *reinterpret_cast<float4 *>(lds + G::kH1 + 4 * lane) = ovl2;
        }
      }
      STAMP(18);  // prologue c: delay lines -> LDS
      if (first_iter) {
        dphi = uniform_u64(raw_dphi);
        phase0 = uniform_u64(raw_phase) + (uint64_t)f * (uint64_t)L * dphi;  // (f = 0 unless SEGPAR: closed form)
        osc_r = uniform_f64(raw_r);
        dc_carry = uniform_f32(raw_dc);
        // the amplitude loop's start-up lasts ~300 samples after a reset: over before segment 1
        transient = !preroll && fabs(osc_r * osc_r - uniform_f64(raw_rs)) > 1e-13;
      }
      // SEGPAR: filter memories rebuilt from 512 input samples (already in registers) that END where
      // the oscillator phase is phase_end: the /4 history goes to X, the last 48 /4 outputs to the
      // /2 history slots of Y1; returns the states of the two DC high-pass chains after them.
      auto rebuild_from = [&](float4 rI0, float4 rI1, float4 rQ0, float4 rQ1, uint64_t phase_end) -> f2 {
        cf z[8];
        if (!WQ15) {
          z[0] = cf{rI0.x * g_hp_i, rQ0.x * g_hp};
          z[1] = cf{rI0.y * g_hp_i, rQ0.y * g_hp};
          z[2] = cf{rI0.z * g_hp_i, rQ0.z * g_hp};
          z[3] = cf{rI0.w * g_hp_i, rQ0.w * g_hp};
          z[4] = cf{rI1.x * g_hp_i, rQ1.x * g_hp};
          z[5] = cf{rI1.y * g_hp_i, rQ1.y * g_hp};
          z[6] = cf{rI1.z * g_hp_i, rQ1.z * g_hp};
          z[7] = cf{rI1.w * g_hp_i, rQ1.w * g_hp};
        } else {
          const float wi[4] = {rI0.x, rI0.y, rI0.z, rI0.w}, wq[4] = {rQ0.x, rQ0.y, rQ0.z, rQ0.w};
  #pragma unroll
          for (int k = 0; k < 4; ++k) {
            z[2 * k] = cf{q15_lo(wi[k]) * g_hp_i, q15_lo(wq[k]) * g_hp};
            z[2 * k + 1] = cf{q15_hi(wi[k]) * g_hp_i, q15_hi(wq[k]) * g_hp};
          }
        }
        f2 dcs = splat(0.0f);
        dc_highpass<8>(z, dcs, lane, hp8.x, hp8.y);  // from rest: 512 samples on, its memory of the start is a1^512
        if (!PLAIN) {
  #pragma unroll
          for (int k = 0; k < 8; ++k) {
            z[k] *= g_iq;
            z[k].y = fmaf(iq_phase_neg, z[k].x, z[k].y);
            z[k].x = fmaf(iq_phase_pos, z[k].y, z[k].x);
          }
        }
        {
          const uint64_t P = phase_end - (uint64_t)(511 - 8 * lane) * dphi;  // sample -512 + 8 lane, + 1
          const float2 t = reinterpret_cast<const float2 *>(smem)[(int)(P >> 56)];  // (PART 1: the table is in LDS)
          const uint32_t u = (uint32_t)(P >> 24);
          const float ang = (float)u * (float)(6.283185307179586476925 / 256.0 / 4294967296.0);
          const float a2 = ang * ang;
          const float sn = ang * fmaf(a2, -1.0f / 6.0f, 1.0f);
          const float cs = fmaf(a2, fmaf(a2, 1.0f / 24.0f, -0.5f), 1.0f);
          const cf base = cmul(cf{t.x, t.y}, cf{cs, sn});
  #pragma unroll
          for (int k = 0; k < 8; ++k) {
            const cf osc = cmul_s(base, cf{nco->wk[k][0], nco->wk[k][1]});
            z[k] = cmulc(z[k], osc);
          }
        }
        wave_sync();
        float *xw = lds + kX + 20 * lane;
  #pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<float4 *>(xw + 2 * (xpad(28 + 2 * i))) = make_float4(z[2 * i].x, z[2 * i].y, z[2 * i + 1].x, z[2 * i + 1].y);
        wave_sync();
        cf o1[2];
        auto pidx = [](int o) { return xpad(o); };
        fir_pair<kDec1Taps, 1, 5, 18, 6>(xw, pidx, (CFloatPtr)cf0->dec1, o1[0], o1[1]);
        // its first outputs saw no history and are dropped; the last 48 (lanes 40..63) are the /2 history
        float4 hh = make_float4(0, 0, 0, 0);
        if (lane < 14) hh = lds4(lds + kX + 2 * xpad(512 + 2 * lane));
        wave_sync();
        if (lane < 14) *reinterpret_cast<float4 *>(lds + kX + 2 * xpad(2 * lane)) = hh;
        if (lane >= 40) *reinterpret_cast<float4 *>(lds + kY1 + y1slot(lane - 40)) = make_float4(o1[0].x, o1[0].y, o1[1].x, o1[1].y);
        wave_sync();
        return dcs;
      };
      f2 dc_pre = splat(0.0f);  // SEGPAR: states of the two DC high-pass chains at the end of the preceding segment
      if (preroll) {
        const float4 rI0 = pI0[1], rI1 = WQ15 ? pI0[1] : pI1[1], rQ0 = pQ0[1], rQ1 = WQ15 ? pQ0[1] : pQ1[1];
        // now that set 1 is consumed: this segment's sub-block 1
        if (!WQ15) {
          pI0[1] = ldg_stream(gI + 512 + 8 * lane);
          pI1[1] = ldg_stream(gI + 512 + 8 * lane + 4);
          pQ0[1] = ldg_stream(gQ + 512 + 8 * lane);
          pQ1[1] = ldg_stream(gQ + 512 + 8 * lane + 4);
        } else {
          pI0[1] = ldg_stream(gI + 256 + 4 * lane);
          pQ0[1] = ldg_stream(gQ + 256 + 4 * lane);
        }
        dc_pre = rebuild_from(rI0, rI1, rQ0, rQ1, phase0);
        // a frame's first segment: the shared biquad comes from the previous frame's Q (Process.cpp:127-128)
        if ((f & (seg - 1)) == 0) dc_carry = uniform_f32(dc_pre.y);
      }

      // ---- Q's DC-block start state = state after ALL of this frame's I (one shared biquad
      // instance runs over I then Q, Process.cpp:127-128).  a1^256 ~ 3e-18, so the last 256 I
      // samples decide it.
      // (carry of the I chain, carry of the Q chain).  One biquad instance filters the whole
      // frame's I and then its Q (Process.cpp:127-128): for the 4096 pipeline a frame is 8
      // segments, so the I chain runs on across segments and the Q chain starts from the state
      // after the frame's LAST I samples.
      if (PART == 0 || (f & (seg - 1)) == 0) {
        float4 tailF = tailI;
        if (PART != 0) {  // the frame's last 256 I samples are seg segments further on
          if (!WQ15) {
            tailF = *reinterpret_cast<const float4 *>(gI + (seg * L - 256) + 4 * lane);
          } else {
            const float2 t = *reinterpret_cast<const float2 *>(gI + (seg * L - 256) / 2 + 2 * lane);
            tailF = make_float4(q15_lo(t.x), q15_hi(t.x), q15_lo(t.y), q15_hi(t.y));
          }
        }
        const float x[4] = {tailF.x * g_rf, tailF.y * g_rf, tailF.z * g_rf, tailF.w * g_rf};
        dc2 = f2{(g_rf_i != g_rf) ? -dc_carry : dc_carry, dc_highpass_end_state<4>(x, hp4.x, hp4.y)};
      } else if (preroll) {
        dc2 = dc_pre;  // inside a frame both chains simply run on
      }

      STAMP(15);  // prologue d: NCO/DC state uniformisation + Q's DC-block start state
      cf y2[2][2];  // /8 outputs of this frame: m = 128*round + 2*lane + e
      cf o1x[2] = {splat(0.0f), splat(0.0f)};  // (T41RX_LOO 13: the last /4 outputs, a register source for the /2 window)
      // (cos, sin) table entries of this lane's first sample of the four sub-blocks.  All four
      // are requested HERE and nowhere later: vector-memory results return in issue order, so a
      // table read issued between two input requests could only be used once every older input
      // request has landed -- it would cut the two-sub-block prefetch distance to nothing.
      float2 osc_tab[4];
  #pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        const uint64_t P = phase0 + (uint64_t)(512 * sb + 8 * lane + 1) * dphi;
        osc_tab[sb] = (PART == 1) ? reinterpret_cast<const float2 *>(smem)[(int)(P >> 56)] : tab[kTabSinCos + (int)(P >> 56)];
      }

  #pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
  #pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int s = 2 * rd + h;
          // -- RF gain (Process.cpp:117-119); the multiply also interleaves I and Q into pairs
          cf z[8];
          if (!WQ15) {
            z[0] = cf{pI0[h].x * g_hp_i, pQ0[h].x * g_hp};
            z[1] = cf{pI0[h].y * g_hp_i, pQ0[h].y * g_hp};
            z[2] = cf{pI0[h].z * g_hp_i, pQ0[h].z * g_hp};
            z[3] = cf{pI0[h].w * g_hp_i, pQ0[h].w * g_hp};
            z[4] = cf{pI1[h].x * g_hp_i, pQ1[h].x * g_hp};
            z[5] = cf{pI1[h].y * g_hp_i, pQ1[h].y * g_hp};
            z[6] = cf{pI1[h].z * g_hp_i, pQ1[h].z * g_hp};
            z[7] = cf{pI1[h].w * g_hp_i, pQ1[h].w * g_hp};
          } else {  // arm_q15_to_float (x / 32768, exact) is part of g_rf here
            const float wi[4] = {pI0[h].x, pI0[h].y, pI0[h].z, pI0[h].w}, wq[4] = {pQ0[h].x, pQ0[h].y, pQ0[h].z, pQ0[h].w};
  #pragma unroll
            for (int k = 0; k < 4; ++k) {
              z[2 * k] = cf{q15_lo(wi[k]) * g_hp_i, q15_lo(wq[k]) * g_hp};
              z[2 * k + 1] = cf{q15_hi(wi[k]) * g_hp_i, q15_hi(wq[k]) * g_hp};
            }
          }
          if (s < 2) {  // refill this register set with the sub-block after next
            if (!WQ15) {
              const int o = 512 * (s + 2) + 8 * lane;
              pI0[h] = ldg_stream(gI + o);
              pI1[h] = ldg_stream(gI + o + 4);
              pQ0[h] = ldg_stream(gQ + o);
              pQ1[h] = ldg_stream(gQ + o + 4);
            } else {
              const int o = 256 * (s + 2) + 4 * lane;
              pI0[h] = ldg_stream(gI + o);
              pQ0[h] = ldg_stream(gQ + o);
            }
          } else if (PART == 1 || (KEEP && kPF >= 1)) {  // the next segment's / frame's sub-blocks 0 and 1
            // Requested UNCONDITIONALLY (behind the launch's last frame: from the constant table, 8 KiB of L2-resident
            // values nobody uses -- no fabric traffic): a request under `if (f + 1 < seg1)` makes the register set a merge
            // of old and new values -- 16 copies per sub-block -- and hipcc's wait for the sub-block in front of it a
            // vmcnt(0), since it cannot count on a younger request having been issued (ISA of round 3's kernel).
            {
              static_assert(kTabEntries512 * 2 >= 2048 + 8, "the stand-in source of the last frame's prefetch covers a frame's offsets");
              const bool more = f + 1 < seg1;
              const float *nI = more ? gI + fstep : reinterpret_cast<const float *>(tab);
              const float *nQ = more ? gQ + fstep : reinterpret_cast<const float *>(tab);
              if (KEEP && s == 3) {  // and the I tail that decides the next frame's Q start state
                if (!WQ15) {
                  tailN = *reinterpret_cast<const float4 *>(nI + (L - 256) + 4 * lane);
                } else {  // (raw q15 words; converted when used, not here: that would wait for them)
                  tailNq = *reinterpret_cast<const float2 *>(nI + (L - 256) / 2 + 2 * lane);
                }
              }
              if (KEEP && kPF < 2 && s == 3) {
                // sub-block 1 is requested at the top of the next frame
              } else if (!WQ15) {
                const int o = 512 * (s - 2) + 8 * lane;
                pI0[h] = ldg_stream(nI + o);
                pI1[h] = ldg_stream(nI + o + 4);
                pQ0[h] = ldg_stream(nQ + o);
                pQ1[h] = ldg_stream(nQ + o + 4);
              } else {
                const int o = 256 * (s - 2) + 4 * lane;
                pI0[h] = ldg_stream(nI + o);
                pQ0[h] = ldg_stream(nQ + o);
              }
            }
          } else if (s == 3 && !KEEP) {  // last sub-block: prefetch the overlap-save "previous" block instead
            const cf *ov = reinterpret_cast<const cf *>(st + kStOverlap);
  #pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ov[64 * j + lane];
          }
          STAMP(s == 0 ? 14 : 0);  // wait for the sub-block's global loads + gain/interleave (14: first sub-block)
        // -- DC high-pass (127-128), band gain (133-134) / IQ amplitude (166)
          if (!T41RX_CUT(6)) dc_highpass<8>(z, dc2, lane, hp8.x, hp8.y);
          if (!PLAIN) {
            // band gain / IQ amplitude (Process.cpp:133-134, 166) and IQ phase correction
            // (Utility.cpp:178-187: phi < 0 mixes I into Q, phi > 0 mixes Q into I), branch-free:
            // at most one of the two phase factors is non-zero
  #pragma unroll
            for (int k = 0; k < 8; ++k) {
              z[k] *= g_iq;
              z[k].y = fmaf(iq_phase_neg, z[k].x, z[k].y);
              z[k].x = fmaf(iq_phase_pos, z[k].y, z[k].x);
            }
          }
          STAMP(1);  // DC high-pass, gains, IQ correction
        // -- oscillator for my 8 samples.  Osc_n = V_n * W has phase phase0 + (n+1) dphi.
          const int n0 = 512 * s + 8 * lane;
          if (DEBUG && a.dbg_pre) {  // what CalcZoom1Magn() sees (Process.cpp:185), input of the display FFT
            float *dp = a.dbg_pre + ((size_t)ch * a.nframes + f) * (2 * L);
  #pragma unroll
            for (int k = 0; k < 8; ++k) {
              dp[n0 + k] = z[k].x;
              dp[L + n0 + k] = z[k].y;
            }
          }
          if (transient) {
            // start-up of the amplitude loop g = 1.95 - |V|^2 (Freq_Shift.cpp:130-134): replay the
            // scalar recurrence (wave-uniform); each lane scales its own 8 samples by
            // |Osc_n| / A* = |V_n| / r* (the mix below is linear, so scaling first is equivalent)
            const NcoPtr nt = fresh_nco(nco);
            const double r_star_sq = uniform_f64(nt->r_star_sq);
            const double w_abs = uniform_f64(nt->w_abs);
            const double inv_r = 1.0 / sqrt(r_star_sq);
            double r = osc_r;
            float amp[8];
  #pragma unroll
            for (int k = 0; k < 8; ++k) amp[k] = 1.0f;
            for (int g = 0; g < 64; ++g) {
  #pragma unroll
              for (int k = 0; k < 8; ++k) {
                if (g == lane) amp[k] = (float)(r * inv_r);
                r = r * (1.95 - r * r) * w_abs;
              }
              if (fabs(r * r - r_star_sq) <= 1e-13) break;
            }
            osc_r = r;
            transient = fabs(osc_r * osc_r - r_star_sq) > 1e-13;
  #pragma unroll
            for (int k = 0; k < 8; ++k) z[k] *= splat(amp[k]);
          }
          // base phasor of my 8 samples from the 64-bit phase: 8-bit table entry (requested at the
          // top of the frame, see there) x 32-bit Taylor remainder
          cf base;
          {
            const uint64_t P = phase0 + (uint64_t)(n0 + 1) * dphi;
            const uint32_t u = (uint32_t)(P >> 24);
            const float ang = (float)u * (float)(6.283185307179586476925 / 256.0 / 4294967296.0);
            const float a2 = ang * ang;
            const float sn = ang * fmaf(a2, -1.0f / 6.0f, 1.0f);
            const float cs = fmaf(a2, fmaf(a2, 1.0f / 24.0f, -0.5f), 1.0f);
            base = cmul(cf{osc_tab[s].x, osc_tab[s].y}, cf{cs, sn});
          }
          // -- Fs/4 shift (x j^n, Freq_Shift.cpp:42-65) and NCO mix (Freq_Shift.cpp:138-139):
          //    (I' + jQ') = (I + jQ) j^k conj(Osc_k) = (I + jQ) conj(base wk''),  wk'' = wk (-j)^k
          //    (the host pre-rotates the per-channel constants, so the Fs/4 shift costs nothing)
          // (PART 1: plain pointer, so the 16 scalar loads are hoisted out of the sub-block loop)
          const NcoPtr ncw = (PART == 1) ? nco : fresh_nco(nco);
  #pragma unroll
          for (int k = 0; k < 8; ++k) {
            const cf w = cf{ncw->wk[k][0], ncw->wk[k][1]};
            if (T41RX_CUT(5)) continue;
            const cf osc = cmul_s(base, w);
            z[k] = cmulc(z[k], osc);
          }
          if (DEBUG && a.dbg_nco) {
            float *dn = a.dbg_nco + ((size_t)ch * a.nframes + f) * (2 * L);
  #pragma unroll
            for (int k = 0; k < 8; ++k) {
              dn[n0 + k] = z[k].x;
              dn[L + n0 + k] = z[k].y;
            }
          }
          STAMP(2);  // oscillator + mix
        // -- stage into LDS, then decimate by 4 (28 taps): outputs m = 2*lane, 2*lane+1
          wave_sync();
          float *xw = lds + kX + 20 * lane;  // lane stride: 8 complex + 1 pad slot = 20 floats
  #pragma unroll
          for (int i = 0; i < 4; ++i)  // logical 28 + 8 lane + 2 i  ->  xpad() - 10 lane is a constant
            *reinterpret_cast<float4 *>(xw + 2 * (xpad(28 + 2 * i))) =
                make_float4(z[2 * i].x, z[2 * i].y, z[2 * i + 1].x, z[2 * i + 1].y);
          wave_sync();
          cf o1[2];
          // arm_fir_decimate_f32: y[m] = sum_i c[i] * state[4m + i]; state[i] = buf[i + 1]
          {
            auto pidx = [](int o) { return xpad(o); };  // window-relative, identical for every lane
            if (T41RX_LOO == 14) {
              fir_pair<kDec1Taps, 1, 5, 18, 6>(xw, pidx, (CFloatPtr)cf0->dec1, o1[0], o1[1], z);
            } else if (!T41RX_CUT(4)) {
              // (round 4, measured and dropped: the window requested one group ahead of its use behind scheduling
              //  barriers, taps in 16-tap scalar loads -- 18 spilled registers, 26.7 against 22.4 us per frame)
              fir_pair<kDec1Taps, 1, 5, 18, 6>(xw, pidx, (CFloatPtr)cf0->dec1, o1[0], o1[1]);
            } else {
              o1[0] = *reinterpret_cast<cf *>(xw);
              o1[1] = *reinterpret_cast<cf *>(xw + 8);
            }
          }
          STAMP(3);  // LDS staging + /4 decimator
        // -- roll the /4 history (logical 512..539 -> 0..27) and append the /4 outputs
          {
            float4 hh = make_float4(0, 0, 0, 0);
            if (lane < 14) hh = lds4(lds + kX + 2 * xpad(512 + 2 * lane));
            wave_sync();
            if (lane < 14) *reinterpret_cast<float4 *>(lds + kX + 2 * xpad(2 * lane)) = hh;
            *reinterpret_cast<float4 *>(lds + kY1 + y1slot(24 + 64 * h + lane)) =
                make_float4(o1[0].x, o1[0].y, o1[1].x, o1[1].y);
          }
          if (T41RX_LOO == 13) { o1x[0] = o1[0]; o1x[1] = o1[1]; }
        }  // h
        STAMP(4);  // history roll
        if (rd == 1) { if (AGC) PRIO(1); else PRIO(2); }
      // ---- decimate by 2 (46 taps) over the 256 new /4 samples: m = 2*lane, 2*lane+1
        wave_sync();
        // y[m] = sum_i c[i] * state[2m + i]; state[i] = buf[i + 3]
        {
          // window-relative complex offset o (even) -> offset in the planes, relative to lds + kY1 + 4 lane
          auto planes = [](int o) { return y1slot(o >> 1) / 2; };
          if (T41RX_LOO == 13) {
            const cf src[8] = {y2[0][0], y2[0][1], o1x[0], o1x[1], y2[0][1], o1x[1], o1x[0], y2[0][0]};
            fir_pair<kDec2Taps, 3, 5, 26, 6>(lds + kY1 + 4 * lane, planes, (CFloatPtr)cf0->dec2, y2[rd][0], y2[rd][1], src);
          } else if (!T41RX_CUT(3)) {
            fir_pair<kDec2Taps, 3, 5, 26, 6>(lds + kY1 + 4 * lane, planes, (CFloatPtr)cf0->dec2, y2[rd][0], y2[rd][1]);
          } else {
            y2[rd][0] = *reinterpret_cast<cf *>(lds + kY1 + 4 * lane);
            y2[rd][1] = *reinterpret_cast<cf *>(lds + kY1 + 4 * lane + 2);
          }
        }
        STAMP(5);  // /2 decimator
      {  // roll the /2 history: logical 256..303 -> 0..47
          float4 hh = make_float4(0, 0, 0, 0);
          if (lane < 24) hh = lds4(lds + kY1 + y1slot(128 + lane));
          wave_sync();
          if (lane < 24) *reinterpret_cast<float4 *>(lds + kY1 + y1slot(lane)) = hh;
        }
      }  // rd
      phase0 += (uint64_t)L * dphi;
      if (PART == 0 || (f & (seg - 1)) == seg - 1) dc_carry = uniform_f32(dc2.y);  // the shared biquad ends the frame on Q

      STAMP(4);
      FRESH_LANE();
      // ---- !KEEP: delay lines back to HBM (the LDS copies are about to be reused as scratch).
      // Issue the small back-end loads now so the FFT hides their latency: interpolator histories,
      // the AGC record; KEEP: the filter mask of this lane (8 x 8 B from the L2-resident table).
      wave_sync();
      if (SEGPAR && seg0 == 0 && f == seg1 - 1 && seg1 < a.nframes) {
        // The channel's state is written by the wave that READ it -- the one of the call's first
        // segment -- and by no other: a wave of a later segment may run (on another XCD) before
        // this one has started.  What the state must hold is what the call's LAST samples leave
        // behind, so this wave rebuilds it from them exactly as the others rebuild theirs.
        const size_t last = (size_t)(a.nframes - 1 - f) * L;  // the call's last segment, relative to this one
        float4 rI0, rI1, rQ0, rQ1;
        if (!WQ15) {
          rI0 = ldg_stream(gI + last + 1536 + 8 * lane);
          rI1 = ldg_stream(gI + last + 1536 + 8 * lane + 4);
          rQ0 = ldg_stream(gQ + last + 1536 + 8 * lane);
          rQ1 = ldg_stream(gQ + last + 1536 + 8 * lane + 4);
        } else {
          rI0 = rI1 = ldg_stream(gI + last / 2 + 768 + 4 * lane);
          rQ0 = rQ1 = ldg_stream(gQ + last / 2 + 768 + 4 * lane);
        }
        phase0 += (uint64_t)(a.nframes - 1 - f) * (uint64_t)L * dphi;  // the oscillator phase after the call
        const f2 dc_end = rebuild_from(rI0, rI1, rQ0, rQ1, phase0);
        dc_carry = uniform_f32(dc_end.y);
      }
      if (!KEEP && (!SEGPAR || (seg0 == 0 && f == seg1 - 1))) {
        if (lane < 14) *reinterpret_cast<float4 *>(st + kStDec1 + 4 * lane) = lds4(lds + kX + 2 * xpad(2 * lane));
        if (lane < 24) *reinterpret_cast<float4 *>(st + kStDec2 + 4 * lane) = lds4(lds + kY1 + y1slot(lane));
        if (PART != 1) {
          if (lane < 6) hist1 = *reinterpret_cast<const float4 *>(st + kStInt1 + 4 * lane);
          if (lane < 8) hist2 = st[kStInt2 + lane];
        }
      } else {
        hist2 = hist2c;
      }
      if (PIPE) {
        if (first_iter && lane < 50) agrec = *reinterpret_cast<const float4 *>(st + st_agc(512) + 4 * lane);
      } else if (AGC && lane < 52) {
        agst = *reinterpret_cast<const float4 *>(st + st_agc(512) + 4 * lane);
      }
      wave_sync();

      // ---- level adjust (Process.cpp:481-492): folded into the /2 decimator's taps by the host (DevCoef::dec2)
      if (DEBUG && a.dbg_dec) {
        float *dd = a.dbg_dec + ((size_t)ch * a.nframes + f) * N;
  #pragma unroll
        for (int rd = 0; rd < 2; ++rd)
  #pragma unroll
          for (int e = 0; e < 2; ++e) {
            dd[128 * rd + 2 * lane + e] = y2[rd][e].x;
            dd[D + 128 * rd + 2 * lane + e] = y2[rd][e].y;
          }
      }

      // ---- NFM (Process.cpp:716-727): quadri-correlator discriminator on the 256 new complex
      // samples, hard limiter, then the demodulated REAL audio goes through the same overlap-save
      // filter with zero imaginary part (Process.cpp:765-816)
      if (MODE == kModeNfm) {
        // fmdemod_quadri_K (Demod.h:7) is a double: K * (float expr) / (float expr) in double
        constexpr double K = 0.340447550238101026565118445432744920253753662109375;
        const cf *ms = reinterpret_cast<const cf *>(st + kStMisc + kMiscNfmI);
        const cf last = ms[0];  // nfmdemod()'s "last sample", see the quirk note below
        // long FFT (PART 1): nfmdemod() sees the whole frame of 256 seg samples, this is one segment of it
        const bool frame_first = (PART == 0) || (f & (seg - 1)) == 0;
        float au[2][2];
  #pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          // previous complex sample of m = 128 rd + 2 lane: lane-1's odd sample; lane 0 wraps to the
          // previous round's last sample
          cf prev0 = cf{lane_up1(y2[rd][1].x), lane_up1(y2[rd][1].y)};
          if (rd == 1 && lane == 0)
            prev0 = cf{__int_as_float(__builtin_amdgcn_readlane(__float_as_int(y2[0][1].x), 63)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y2[0][1].y), 63))};
          if (PART == 1 && rd == 0 && lane == 0) prev0 = nfm_carry;  // the previous segment's last sample
          const cf cur0 = y2[rd][0], cur1 = y2[rd][1];
          // Demod.cpp:229-231: (qnow * ilast - inow * qlast) / (inow^2 + qnow^2)
          float num0 = cur0.y * prev0.x - cur0.x * prev0.y;
          const float den0 = cur0.x * cur0.x + cur0.y * cur0.y;
          const float num1 = cur1.y * cur0.x - cur1.x * cur0.y;
          const float den1 = cur1.x * cur1.x + cur1.y * cur1.y;
          if (rd == 0 && lane == 0 && frame_first)  // Demod.cpp:224: first sample of the frame uses the difference form
            num0 = cur0.x * (cur0.y - last.y) - cur0.y * (cur0.x - last.x);
          float a0 = (float)(K * (double)num0 / (double)den0);
          float a1 = (float)(K * (double)num1 / (double)den1);
          // Process.cpp:719-727: limiter, skips sample 0 of the frame
          if (!(rd == 0 && lane == 0 && frame_first)) {
            a0 = (1.0f < a0) ? 1.0f : a0;
            a0 = (-1.0f > a0) ? -1.0f : a0;
          }
          a1 = (1.0f < a1) ? 1.0f : a1;
          a1 = (-1.0f > a1) ? -1.0f : a1;
          au[rd][0] = a0;
          au[rd][1] = a1;
        }
        if (PART == 0 && a.nfm_atan) {
          // ---- nfm_demod = 1, the alternative the reference keeps commented out: fmdemod_atan_cf
          // (Demod.cpp:368-392) with ApproxAtan2 (Demod.cpp:148-197, its 2 pi for pi / 2 as written),
          // the limiter, then deemphasis_nfm_ff applied block-wise (Demod.cpp:328-344, Process.cpp:
          // 734-735): only the first 256 - 81 samples of a block are filtered, the rest of the
          // destination buffer still holds the decimated Q samples.
#pragma clang fp contract(off)
          constexpr float kPi = 3.1415926535897932384626433832795f, kTpi = 6.283185307179586476925286766559f;
          auto atan_poly = [](float z) { return (0.97239411f + -0.19194795f * z * z) * z; };  // ApproxAtan, Utility.cpp:298-302
          auto atan2_as_written = [&](float y, float x) {
            const bool wide = fabsf(x) > fabsf(y);
            const float z = wide ? y / x : x / y;
            const float t = atan_poly(z);
            const float r_wide = (x > 0.0f) ? t : (y >= 0.0f ? t + kPi : t - kPi);
            const float r_tall = (y > 0.0f) ? -t + kTpi : -t - kTpi;
            const float r_axis = (y > 0.0f) ? kTpi : (y < 0.0f ? -kTpi : 0.0f);
            return (x != 0.0f) ? (wide ? r_wide : r_tall) : r_axis;
          };
          float ph[2][2];
  #pragma unroll
          for (int rd = 0; rd < 2; ++rd)
  #pragma unroll
            for (int e = 0; e < 2; ++e) ph[rd][e] = atan2_as_written(y2[rd][e].y, y2[rd][e].x);
          const float last_phase = st[kStMisc + kMiscNfmPhase];
  #pragma unroll
          for (int rd = 0; rd < 2; ++rd) {
            float prev = lane_up1(ph[rd][1]);
            if (lane == 0) prev = (rd == 0) ? last_phase : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ph[0][1]), 63));
            float d[2] = {ph[rd][0] - prev, ph[rd][1] - ph[rd][0]};
  #pragma unroll
            for (int e = 0; e < 2; ++e) {
              if (d[e] < -kPi) d[e] += 2 * kPi;
              if (d[e] > kPi) d[e] -= 2 * kPi;
              float o = d[e] / kPi;
              if (!(rd == 0 && e == 0 && lane == 0)) {  // Process.cpp:719-727: the limiter skips sample 0
                o = (1.0f < o) ? 1.0f : o;
                o = (-1.0f > o) ? -1.0f : o;
              }
              au[rd][e] = o;
            }
          }
          if (lane == 63) st[kStMisc + kMiscNfmPhase] = ph[1][1];
          // de-emphasis: out[i] = sum_ti taps[ti] * in[i + ti], i < 175; samples through LDS in natural order
          float *ds = lds + kScr;
          wave_sync();
  #pragma unroll
          for (int rd = 0; rd < 2; ++rd) *reinterpret_cast<float2 *>(ds + 128 * rd + 2 * lane) = make_float2(au[rd][0], au[rd][1]);
          wave_sync();
  #pragma unroll
          for (int rd = 0; rd < 2; ++rd) {
            const int m0 = 128 * rd + 2 * lane;  // my samples m0, m0 + 1 share the window in[m0 .. m0 + 81]
            float acc0 = 0.0f, acc1 = 0.0f;      // taps in ascending order, separate multiply and add, as the reference's loop
            float tp[96];
  #pragma unroll
            for (int c = 0; c < 96; c += 16) {
              float chunk[16];
              load_taps<16>(chunk, (CFloatPtr)cf0->deemph + c);
  #pragma unroll
              for (int k = 0; k < 16; ++k) tp[c + k] = chunk[k];
  #pragma unroll
              for (int jj = c / 2; jj < c / 2 + 8; ++jj) {
                if (2 * jj > kDeemphTaps) continue;  // pairs (in[m0 + 2 jj], in[m0 + 2 jj + 1]), jj = 0..40
                const float2 w = *reinterpret_cast<const float2 *>(ds + m0 + 2 * jj);
                if (2 * jj < kDeemphTaps) acc0 += tp[2 * jj] * w.x;
                if (jj > 0) acc1 += tp[2 * jj - 1] * w.x;
                if (2 * jj + 1 < kDeemphTaps) acc0 += tp[2 * jj + 1] * w.y;
                if (2 * jj < kDeemphTaps) acc1 += tp[2 * jj] * w.y;
              }
            }
            au[rd][0] = (m0 < D - kDeemphTaps) ? acc0 : y2[rd][0].y;
            au[rd][1] = (m0 + 1 < D - kDeemphTaps) ? acc1 : y2[rd][1].y;
          }
          wave_sync();
        }
        // Demod.cpp:232-233 keeps floats [input_size-2], [input_size-1] of the interleaved buffer
        // as "last sample": that is complex sample 127 (m = 127: round 0, lane 63, odd), not 255
        // (frame of 256 seg samples: complex sample 128 seg - 1 = the last one of segment seg/2 - 1)
        if (PART == 0) {
          if (lane == 63) *reinterpret_cast<cf *>(st + kStMisc + kMiscNfmI) = y2[0][1];
        } else {
          if (lane == 63 && (f & (seg - 1)) == seg / 2 - 1) *reinterpret_cast<cf *>(st + kStMisc + kMiscNfmI) = y2[1][1];
          nfm_carry = cf{__int_as_float(__builtin_amdgcn_readlane(__float_as_int(y2[1][1].x), 63)),
                         __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y2[1][1].y), 63))};
        }
  #pragma unroll
        for (int rd = 0; rd < 2; ++rd)
  #pragma unroll
          for (int e = 0; e < 2; ++e) y2[rd][e] = cf{au[rd][e], 0.0f};
      }
      if (PART == 1) {  // front half of the 4096 pipeline: hand the 256 new /8 samples to the fast-conv kernel
        float *mid = a.mid + ((size_t)ch * a.nframes + f) * (2 * D);
  #pragma unroll
        for (int rd = 0; rd < 2; ++rd)
          *reinterpret_cast<float4 *>(mid + 2 * (128 * rd + 2 * lane)) =
              make_float4(y2[rd][0].x, y2[rd][0].y, y2[rd][1].x, y2[rd][1].y);
        continue;
      }


      // ---- overlap-save assemble (Process.cpp:498-522): v[0..3] = previous block, v[4..7] = new
      {
        cf *tb = reinterpret_cast<cf *>(lds + kScr);
  #pragma unroll
        for (int rd = 0; rd < 2; ++rd)
          *reinterpret_cast<float4 *>(lds + kScr + 2 * (128 * rd + 2 * lane)) =
              make_float4(y2[rd][0].x, y2[rd][0].y, y2[rd][1].x, y2[rd][1].y);
        wave_sync();
  #pragma unroll
        for (int j = 0; j < 4; ++j) v[4 + j] = tb[lane + 64 * j];
        if (KEEP) {  // the previous block waits in LDS; the new one takes its place
          cf *ov = reinterpret_cast<cf *>(lds + G::kOV);
  #pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = ov[64 * j + lane];
  #pragma unroll
          for (int j = 0; j < 4; ++j) ov[64 * j + lane] = v[4 + j];
        } else {
          cf *ov = reinterpret_cast<cf *>(st + kStOverlap);
  #pragma unroll
          for (int j = 0; j < 4; ++j) ov[64 * j + lane] = v[4 + j];
        }
      }

      STAMP(6);  // state save, level, overlap-save assemble
      FRESH_LANE();
      // ---- FFT, x mask, inverse FFT (Process.cpp:535-595).  The mask table is pre-scaled by 1/N.
      {
        cf tw1[7], tw2[7];
        cf mk[8];  // KEEP: FIR_filter_mask[lane + 64 r] / N, requested from the L2-resident table inside the forward FFT
        constexpr bool GMASK = KEEP || MODE == kModeSam;  // (on the 4-wave geometry, measured: +1.2 .. 2.5 % against mask and twiddles in LDS / registers)
        if (!GMASK) {
  #pragma unroll
          for (int q = 0; q < 7; ++q) {
            tw1[q] = ltab[G::kTw1 + 64 * q + lane];
            tw2[q] = ltab[G::kTw2 + 8 * q + (lane & 7)];
          }
        }
        if (!T41RX_CUT(2)) {
          if (GMASK) {
            fft512_ldstw<false>(v, ltab + G::kTw1 + lane, ltab + G::kTw2 + (lane & 7), lds + kScr, lane, [&]() {
  #pragma unroll
              for (int r = 0; r < 8; ++r) {
                const float2 t = tab[kTabMask + 64 * r + lane];
                mk[r] = cf{t.x, t.y};
              }
            });
          } else {
            fft512<false>(v, tw1, tw2, lds + kScr, lane);
          }
  #pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = cmul(v[r], GMASK ? mk[r] : ltab[kLdsTabMask + 64 * r + lane]);
          if (DEBUG && a.spect) {
            // ---- audio spectrum side output (Process.cpp:550-570 with updateDisplayFlag == 1):
            // audioSpectBuffer[1023 - k] = iFFT_buffer[k]^2 over the 1024 floats of the masked
            // spectrum.  The mask table carries 1/N (the reference applies it in the inverse FFT):
            // a power of two, so squaring after undoing it is exact.  v[r] = bin lane + 64 r.
            float *sp = a.spect + ((size_t)ch * a.nframes + f) * 1024;
            float best = -1.0f;
            int besti = 0;
  #pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int k = lane + 64 * r;
              const float re = v[r].x * 512.0f, im = v[r].y * 512.0f;
              const float e0 = im * im, e1 = re * re;  // buffer indices 1022 - 2k, 1023 - 2k
              *reinterpret_cast<float2 *>(sp + 1022 - 2 * k) = make_float2(e0, e1);
              // arm_max_f32: the first occurrence of the maximum = the smallest buffer index
              if (e1 >= best) { best = e1; besti = 1023 - 2 * k; }
              if (e0 >= best) { best = e0; besti = 1022 - 2 * k; }
            }
  #pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
              const float ob = __shfl_xor(best, m, 64);
              const int oi = __shfl_xor(besti, m, 64);
              if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
            }
            if (lane == 0) {
              float *mx = a.spect_max + ((size_t)ch * a.nframes + f) * 3;
              const float ave = (float)(.5 * (double)best + .5 * (double)st[kStMisc + kMiscMaxSqAve]);  // :570
              mx[0] = best;
              mx[1] = (float)besti;
              mx[2] = ave;
              st[kStMisc + kMiscMaxSqAve] = ave;
            }
          }
          if (GMASK)
            fft512_ldstw<true>(v, ltab + G::kTw1 + lane, ltab + G::kTw2 + (lane & 7), lds + kScr, lane, []() {});
          else
            fft512<true>(v, tw1, tw2, lds + kScr, lane);
        }
      }
    }
    FRESH_LANE();
    if (PART != 2 || LONGC) {
      // ---- AGC (Process.cpp:605 / :810).  Off: fixed gain on the valid half (DSP_Fn.cpp:494-502).
      // SSB/NFM: audio = Re
      const float fixed_gain = fresh_coef(cf0)->sc[kScFixedGain];
      cf og[4];
      if (PSA) {
        const int left = a.nchan - NW * (int)blockIdx.x;
        const int nvalid = left < NW ? left : NW;
        unsigned *fa = reinterpret_cast<unsigned *>(smem) + kPipeFlags;  // AGC stage: ready[3], done, claim
        unsigned *fs = fa + 5;                                           // PLL stage: the same five words
        unsigned *pipe_err = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned long long *>(a.agc_pipe + (size_t)a.nchan * kPipeSlots * kPipeSlotFloats) +
                                                          ((size_t)a.nchan + 15) * 16);
        // the PLL stage's slots lie behind the AGC stage's and the diagnostic words (rx_host.cpp allocates both)
        float *sam_slots = a.agc_pipe + (size_t)a.nchan * kPipeSlots * kPipeSlotFloats + ((size_t)a.nchan + 16) * 32;
        const size_t ch0 = (size_t)NW * blockIdx.x;
        auto claim = [&](unsigned *word, int g) -> bool {  // the first wave to get here takes frame g's chain
          unsigned won = 0u;
          if (lane == 0) {
            unsigned expect = (unsigned)g;
            won = __hip_atomic_compare_exchange_strong(word, &expect, (unsigned)(g + 1), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
          }
          return __builtin_amdgcn_readfirstlane(won) != 0u;
        };
        static_assert(!PSA || (NW == 16 && kScr + kPipeStageFloats <= G::kXF && kScr + kPipeSamStageFloats <= G::kXF), "chain staging inside the X scratch");
        // ---- stage A: this frame's AGC operands and popped samples -> the channel's slot
        if (f < seg1) {
          float *pslot = a.agc_pipe + ((size_t)ch * kPipeSlots + f % kPipeSlots) * kPipeSlotFloats;
          if (first_iter) agmag = make_float2(agc_mag(cf{agrec.x, agrec.y}), agc_mag(cf{agrec.z, agrec.w}));  // (later frames: carried)
          agrec = agc_prep_pipe<AgcLds<true>, true>(v, agrec, agmag, lds, pslot, cf0, lane);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) __hip_atomic_fetch_add(fa + f % kPipeSlots, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // ---- duty 1: the AGC chain of frame f - 1
        {
          const int g = f - 1;
          if (g >= seg0 && g < seg1 && claim(fa + 4, g)) {
            pipe_wait_ge(fa + g % kPipeSlots, (unsigned)nvalid, pipe_err);
            if (lane == 0) __hip_atomic_store(fa + g % kPipeSlots, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            PRIO(3);
            unsigned long long *pipe_stat = reinterpret_cast<unsigned long long *>(a.agc_pipe + (size_t)a.nchan * kPipeSlots * kPipeSlotFloats) + (size_t)job * 16;
            agc_chain_pipe(a.agc_pipe + (ch0 * kPipeSlots + g % kPipeSlots) * kPipeSlotFloats, a.state + ch0 * state_stride + st_agc(512) + kAgcHistFloats,
                           state_stride, lds + kScr, fa + 3, (unsigned)g, nvalid, cf0, lane, pipe_stat, pipe_err);
            PRIO(1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(fa + 3, (unsigned)(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        // ---- stage B: gain of frame f - 2 (its chain is done), the scaled samples -> the PLL stage's slot, time order
        {
          const int fm = f - 2;
          if (fm >= seg0 && fm < seg1) {
            pipe_wait_ge(fa + 3, (unsigned)(fm + 1), pipe_err);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const AgcGainIn gin = agc_gain_request<true>(a.agc_pipe + ((size_t)ch * kPipeSlots + fm % kPipeSlots) * kPipeSlotFloats, lane);
            agc_gain_pipe(gin, cf0, og);
            float *sslot = sam_slots + ((size_t)ch * kPipeSlots + fm % kPipeSlots) * kPipeSlotFloats;
            *reinterpret_cast<float4 *>(sslot + 8 * lane) = make_float4(og[0].x, og[0].y, og[1].x, og[1].y);  // samples 4 lane, 4 lane + 1
            *reinterpret_cast<float4 *>(sslot + 8 * lane + 4) = make_float4(og[2].x, og[2].y, og[3].x, og[3].y);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_fetch_add(fs + fm % kPipeSlots, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        // ---- duty 2: the PLL of frame f - 3
        {
          const int g = f - 3;
          if (g >= seg0 && g < seg1 && claim(fs + 4, g)) {
            pipe_wait_ge(fs + g % kPipeSlots, (unsigned)nvalid, pipe_err);
            if (lane == 0) __hip_atomic_store(fs + g % kPipeSlots, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            PRIO(3);
            sam_chain_pipe(sam_slots + (ch0 * kPipeSlots + g % kPipeSlots) * kPipeSlotFloats, a.state + ch0 * state_stride + kStMisc, state_stride,
                           lds + kScr, reinterpret_cast<const float *>(a.tab + kTabSam), fs + 3, (unsigned)g, nvalid, cf0, lane, pipe_err);
            PRIO(1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(fs + 3, (unsigned)(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        if (f < seg0 + kSkew) continue;  // nothing to finish yet
        // ---- stage C: the audio of frame f - 4
        pipe_wait_ge(fs + 3, (unsigned)(fb + 1), pipe_err);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        {
          const float4 au = *reinterpret_cast<const float4 *>(sam_slots + ((size_t)ch * kPipeSlots + fb % kPipeSlots) * kPipeSlotFloats + 512 + 4 * lane);
          aud[0] = au.x, aud[1] = au.y, aud[2] = au.z, aud[3] = au.w;
        }
        if (KEEP) hist2 = hist2c;
      } else if (PIPE) {
        const int left = a.nchan - NW * (int)blockIdx.x;
        const int nvalid = left < NW ? left : NW;
        unsigned *flags = reinterpret_cast<unsigned *>(smem) + kPipeFlags;
        constexpr bool NEED_IM = (MODE == kModeAm);
        unsigned long long *pipe_stat = reinterpret_cast<unsigned long long *>(a.agc_pipe + (size_t)a.nchan * kPipeSlots * kPipeSlotFloats) + (size_t)job * 16;
        (void)pipe_stat;
        unsigned *pipe_err = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned long long *>(a.agc_pipe + (size_t)a.nchan * kPipeSlots * kPipeSlotFloats) +
                                                          ((size_t)a.nchan + 15) * 16);
#ifdef T41RX_PIPE_STAT
        if (f < seg1 && lane == 0) {
          const unsigned long long now = __builtin_readcyclecounter();
          pipe_stat[8] += now - ps_t;
          pipe_stat[11] += 1;
          ps_t = now;
        }
#endif
        // the back end's frame: if its chain is done by now (the rule), volts and the popped samples are requested here,
        // ahead of the front end's AGC preparation, which hides the round trip (measured: 2 % of the kernel, although the
        // twelve registers it holds meanwhile spill); if not, behind it -- waiting HERE would put
        // the duty wave's preparation on the path from one chain to the next
        AgcGainIn gin{};
        const float *bslot = a.agc_pipe + ((size_t)ch * kPipeSlots + fb % kPipeSlots) * kPipeSlotFloats;
        const bool early = f >= seg0 + kSkew && pipe_flag_read(flags + 3) >= (unsigned)(fb + 1);
        if (early) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if (PSAM) gin.vv = *reinterpret_cast<const float4 *>(bslot + 512 + 4 * lane);  // the frame's audio
          else gin = agc_gain_request<NEED_IM>(bslot, lane);
        }
        if (f < seg1) {  // this frame's chain operands and popped samples -> the channel's slot
          float *pslot = a.agc_pipe + ((size_t)ch * kPipeSlots + f % kPipeSlots) * kPipeSlotFloats;
          if (PSAM) sam_prep_pipe(v, fixed_gain, pslot, lane);
          else {
            if (first_iter) agmag = make_float2(agc_mag(cf{agrec.x, agrec.y}), agc_mag(cf{agrec.z, agrec.w}));  // (later frames: carried)
            agrec = agc_prep_pipe<AgcLds<true>, NEED_IM>(v, agrec, agmag, lds, pslot, cf0, lane);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) __hip_atomic_fetch_add(flags + f % kPipeSlots, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef T41RX_PIPE_STAT
          if (lane == 0) pipe_stat[9] += __builtin_readcyclecounter() - ps_t;
#endif
        }
        const int g = f - 1;  // the frame whose chain is due
#if T41RX_PIPE_CLAIM
        // the duty goes to the first wave that gets here (the one furthest ahead: it is sure to be waiting when the previous
        // chain ends, and it can best afford to fall a chain behind) instead of rotating blindly
        bool duty = false;
        if (g >= seg0 && g < seg1) {
          // flags[4] = the next frame whose chain nobody has taken yet: frame g is taken by the one wave whose
          // compare-and-swap g -> g + 1 succeeds (a slower wave finds g + 1 or more there, whenever it arrives)
          unsigned won = 0u;
          if (lane == 0) {
            unsigned expect = (unsigned)g;
            won = __hip_atomic_compare_exchange_strong(flags + 4, &expect, (unsigned)(g + 1), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
          }
          duty = __builtin_amdgcn_readfirstlane(won) != 0u;
        }
        if (duty) {
#else
        if (g >= seg0 && g < seg1 && g % nvalid == wv) {
#endif
          {
            PIPE_STAT_T0();
            pipe_wait_ge(flags + g % kPipeSlots, (unsigned)nvalid, pipe_err);
            PIPE_STAT_ADD(4);
          }
          if (lane == 0) __hip_atomic_store(flags + g % kPipeSlots, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const size_t ch0 = (size_t)NW * blockIdx.x;
          PRIO(3);  // the critical path of the whole workgroup, one dependent instruction at a time
          static_assert(!PIPE || (NW == 16 && kScr + kPipeStageFloats <= G::kXF && kScr + kPipeSamStageFloats <= G::kXF),
                        "chain staging: 16 channels, inside the X scratch");
          if (PSAM)
            sam_chain_pipe(a.agc_pipe + (ch0 * kPipeSlots + g % kPipeSlots) * kPipeSlotFloats, a.state + ch0 * state_stride + kStMisc, state_stride,
                           lds + kScr, reinterpret_cast<const float *>(a.tab + kTabSam), flags + 3, (unsigned)g, nvalid, cf0, lane, pipe_err);
          else
            agc_chain_pipe(a.agc_pipe + (ch0 * kPipeSlots + g % kPipeSlots) * kPipeSlotFloats, a.state + ch0 * state_stride + st_agc(512) + kAgcHistFloats,
                           state_stride, lds + kScr, flags + 3, (unsigned)g, nvalid, cf0, lane, pipe_stat, pipe_err);
          PRIO(1);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) __hip_atomic_store(flags + 3, (unsigned)(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (f < seg0 + kSkew) continue;  // nothing to finish yet
        if (!early) {
          PIPE_STAT_T0();
          pipe_wait_ge(flags + 3, (unsigned)(fb + 1), pipe_err);
          PIPE_STAT_ADD(3);
          if (PSAM) gin.vv = *reinterpret_cast<const float4 *>(bslot + 512 + 4 * lane);
          else gin = agc_gain_request<NEED_IM>(bslot, lane);
        }
#ifdef T41RX_PIPE_STAT
        ps_t = __builtin_readcyclecounter();
#endif
        if (PSAM) {
          aud[0] = gin.vv.x, aud[1] = gin.vv.y, aud[2] = gin.vv.z, aud[3] = gin.vv.w;
        } else {
          agc_gain_pipe(gin, cf0, og);
        }
        if (KEEP) hist2 = hist2c;
      } else if (AGC) {
        const int left = a.nchan - NW * (int)blockIdx.x;
        agc_apply<AgcLds<KEEP>, NW, G::kSlice>(v, agst, lds, smem + G::kTab, st + st_agc(512 * seg), cf0, lane, wv,
                                               left < NW ? left : NW, og STAMP_ARGS);
      }
      if (PSAM || PSA) {
        // (the audio came out of the slot above)
      } else if (MODE == kModeSam) {
        // ---- synchronous AM, AMDecodeSAM() Demod.cpp:40-139: a PLL, one sample at a time.  Every wave
        // puts its channel's 256 complex samples in its slice in time order; wave 0 then runs the
        // loops of the workgroup's channels, one lane per channel (all lanes enabled, as in
        // agc_apply), and leaves the audio in place of the real parts.
        const int left = a.nchan - NW * (int)blockIdx.x;
        const int nvalid = left < NW ? left : NW;
        wave_sync();
  #pragma unroll
        for (int j = 0; j < 4; ++j) {
          const cf g = AGC ? og[j] : v[4 + j] * splat(fixed_gain);
          *reinterpret_cast<cf *>(lds + 2 * (AGC ? 4 * lane + j : lane + 64 * j)) = g;
        }
        __syncthreads();
        if (wv == 0) {
          PRIO(3);  // the frame's critical path, one dependent instruction at a time
          const int c = (nvalid == NW) ? (lane & (NW - 1)) : lane % nvalid;
          sam_chain(smem + G::kTab + c * G::kSlice, smem, a.state + (size_t)(NW * (int)blockIdx.x + c) * state_stride + kStMisc,
                    cf0, lane < nvalid);
          PRIO(1);
        }
        __syncthreads();
  #pragma unroll
        for (int j = 0; j < 4; ++j) aud[j] = lds[2 * (lane + 64 * j)];
        wave_sync();
      } else if (MODE != kModeAm) {
  #pragma unroll
        for (int j = 0; j < 4; ++j) aud[j] = AGC ? og[j].x : fixed_gain * v[4 + j].x;
      } else {
        // ---- AM (Process.cpp:697-707): AlphaBetaMag envelope (Utility.cpp:269-285), DC removal
        // w = m + 0.99 w_old, y = w - w_old, then biquad_lowpass1 (DF1).  Both recurrences run as
        // wave scans over lane-contiguous chunks of 4 samples.  No FMA contraction in this block:
        // the reference's arithmetic is separate multiplies and adds, and every instantiation of
        // the kernel (f32 / q15 entry, debug taps) must round alike.
#pragma clang fp contract(off)
  #pragma unroll
        for (int j = 0; j < 4; ++j) {
          const cf g = AGC ? og[j] : v[4 + j] * splat(fixed_gain);
          const float ai = fabsf(g.x), aq = fabsf(g.y);
          const float hi = fmaxf(ai, aq), lo = fminf(ai, aq);
          aud[j] = 0.960433870103f * hi + 0.397824734759f * lo;
        }
        float4 m4 = make_float4(aud[0], aud[1], aud[2], aud[3]);
        if (!AGC) {  // lane + 64 j -> 4 lane + j
          wave_sync();
  #pragma unroll
          for (int j = 0; j < 4; ++j) lds[kI1 + 24 + lane + 64 * j] = aud[j];
          wave_sync();
          m4 = lds4(lds + kI1 + 24 + 4 * lane);
        }
        const float m[4] = {m4.x, m4.y, m4.z, m4.w};
        float *ms = st + kStMisc;
        // -- DC block.  The reference accumulates w ~ 100x the signal in f32; here the scan runs in
        // f64 (no accumulation noise of its own), state kept as the reference's float wold
        const double ca = (double)0.99f;
        double wl[4];
        {
          double wprev = (lane == 0) ? (double)ms[kMiscWold] : 0.0;
  #pragma unroll
          for (int k = 0; k < 4; ++k) {
            wprev = (double)m[k] + ca * wprev;
            wl[k] = wprev;
          }
        }
        double B = wl[3];
        {
          const double a4 = ca * ca * ca * ca, a8 = a4 * a4, a16 = a8 * a8, a32 = a16 * a16;
          // ca^(4 ((lane&15)+1)), ca^(4 ((lane&31)+1)): per-lane constants from the table
          const double2 pw = *reinterpret_cast<const double2 *>(tab + kTabAm + 6 * lane);
          const double p15 = pw.x, p31 = pw.y;
          B = fma(a4, dpp_d<kDppRowShr1, 0xf, true>(B), B);
          B = fma(a8, dpp_d<kDppRowShr2, 0xf, true>(B), B);
          B = fma(a16, dpp_d<kDppRowShr4, 0xf, true>(B), B);
          B = fma(a32, dpp_d<kDppRowShr8, 0xf, true>(B), B);
          B = fma(p15, dpp_d<kDppRowBcast15, 0xa, false>(B), B);
          B = fma(p31, dpp_d<kDppRowBcast31, 0xc, false>(B), B);
          const double e = dpp_d<kDppWaveShr1, 0xf, true>(B);  // w just before my first sample (lane 0: already included)
          double wk_prev = (lane == 0) ? (double)ms[kMiscWold] : e;
          double apow = ca;
  #pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double wt = (lane == 0) ? wl[k] : wl[k] + apow * e;
            aud[k] = (float)(wt - wk_prev);
            wk_prev = wt;
            apow *= ca;
          }
          if (lane == 63) ms[kMiscWold] = (float)wk_prev;
        }
        // -- biquad_lowpass1, DF1: y = b0 x + b1 x1 + b2 x2 + a1 y1 + a2 y2 (a's pre-negated)
        {
          const CoefPtr c = fresh_coef(cf0);
          const float b0 = c->lp1[0], b1 = c->lp1[1], b2 = c->lp1[2], a1 = c->lp1[3], a2 = c->lp1[4];
          const float4 sv = *reinterpret_cast<const float4 *>(ms + kMiscLp1);  // x1, x2, y1, y2
          float xm1 = lane_up1(aud[3]), xm2 = lane_up1(aud[2]);
          if (lane == 0) {
            xm1 = sv.x;
            xm2 = sv.y;
          }
          float y[4];
          float s1 = (lane == 0) ? sv.z : 0.0f, s2 = (lane == 0) ? sv.w : 0.0f;  // y[n-1], y[n-2]
          {
            float x1 = xm1, x2 = xm2;
  #pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float u = b0 * aud[k] + b1 * x1 + b2 * x2;
              const float yy = u + a1 * s1 + a2 * s2;
              x2 = x1;
              x1 = aud[k];
              s2 = s1;
              s1 = yy;
              y[k] = yy;
            }
          }
          // state transition over one lane (4 samples): s_out = P s_in + (s1, s2), P = M^4,
          // M = [[a1, a2], [1, 0]]; scan with 2x2 matrix powers
          struct M2 { float a, b, c, d; };
          auto mm = [](M2 x, M2 y) { return M2{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d}; };
          const M2 M{a1, a2, 1.0f, 0.0f};
          const M2 Mq = mm(M, M);
          const M2 P1 = mm(Mq, Mq), P2 = mm(P1, P1), P4 = mm(P2, P2), P8 = mm(P4, P4);
          // (M^4)^((lane&15)+1), (M^4)^((lane&31)+1): per-lane constants from the table
          const float4 q15t = *reinterpret_cast<const float4 *>(tab + kTabAm + 6 * lane + 2);
          const float4 q31t = *reinterpret_cast<const float4 *>(tab + kTabAm + 6 * lane + 4);
          const M2 Q15{q15t.x, q15t.y, q15t.z, q15t.w}, Q31{q31t.x, q31t.y, q31t.z, q31t.w};
          auto step = [&](M2 P, float o1, float o2) {
            s1 = s1 + P.a * o1 + P.b * o2;
            s2 = s2 + P.c * o1 + P.d * o2;
          };
          {
            float o1 = dpp_f<kDppRowShr1, 0xf, true>(0.0f, s1), o2 = dpp_f<kDppRowShr1, 0xf, true>(0.0f, s2);
            step(P1, o1, o2);
            o1 = dpp_f<kDppRowShr2, 0xf, true>(0.0f, s1), o2 = dpp_f<kDppRowShr2, 0xf, true>(0.0f, s2);
            step(P2, o1, o2);
            o1 = dpp_f<kDppRowShr4, 0xf, true>(0.0f, s1), o2 = dpp_f<kDppRowShr4, 0xf, true>(0.0f, s2);
            step(P4, o1, o2);
            o1 = dpp_f<kDppRowShr8, 0xf, true>(0.0f, s1), o2 = dpp_f<kDppRowShr8, 0xf, true>(0.0f, s2);
            step(P8, o1, o2);
            o1 = dpp_f<kDppRowBcast15, 0xa, false>(0.0f, s1), o2 = dpp_f<kDppRowBcast15, 0xa, false>(0.0f, s2);
            step(Q15, o1, o2);
            o1 = dpp_f<kDppRowBcast31, 0xc, false>(0.0f, s1), o2 = dpp_f<kDppRowBcast31, 0xc, false>(0.0f, s2);
            step(Q31, o1, o2);
          }
          // carry entering my chunk = inclusive state of lane-1; fix up y_k += (M^(k+1) e)[0]
          const float e1 = lane_up1(s1), e2 = lane_up1(s2);
          M2 Mk = M;
  #pragma unroll
          for (int k = 0; k < 4; ++k) {
            y[k] += Mk.a * e1 + Mk.b * e2;
            Mk = mm(M, Mk);
          }
          if (lane == 63) *reinterpret_cast<float4 *>(ms + kMiscLp1) = make_float4(aud[3], aud[2], y[3], y[2]);
  #pragma unroll
          for (int k = 0; k < 4; ++k) aud[k] = y[k];
        }
      }
      if (DEBUG && a.dbg_demod) {
        float *dm = a.dbg_demod + ((size_t)ch * a.nframes + f) * D;
  #pragma unroll
        for (int j = 0; j < 4; ++j) dm[CONTIG ? 4 * lane + j : lane + 64 * j] = aud[j];
      }

    }
    if (T41RX_CUT(1)) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        // (1-KiB store instructions like the product's, so that the staged cuts time the arithmetic they
        // remove and not a worse store pattern; 8: the 16 x 64 B form)
        *reinterpret_cast<float4 *>(gO + (T41RX_ABLATE == 8 ? 32 * (16 * (u & 3) + (lane >> 2)) + 16 * (u >> 2) + 4 * (lane & 3)
                                                            : 4 * lane + 256 * u)) = make_float4(aud[0], aud[1], aud[2], aud[3]);
      continue;
    }
    if ((DEBUG || WQ15) && PART == 0 && a.aud_out) {
      // noise reduction / notch on (Process.cpp:841-866): those stages sit between the demodulator and the
      // interpolators and run in kernels of their own (nr_kernels.hip) on the whole call's audio; this kernel
      // hands over the 256 samples of the frame in time order and leaves the interpolator memories alone
      float *ao = a.aud_out + ((size_t)ch * a.nframes + fb) * D;
      if (CONTIG) {
        *reinterpret_cast<float4 *>(ao + 4 * lane) = make_float4(aud[0], aud[1], aud[2], aud[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) ao[lane + 64 * j] = aud[j];
      }
      continue;
    }
    FRESH_LANE();
    // ---- interpolate by 2 (48 taps, phase length 24): inputs n = 4 lane .. 4 lane + 3
    // LDS buf: [0] pad, [1..23] history, [24 + i] new sample i
    wave_sync();
    {
      float *ib = lds + kI1;
      if (KEEP && lane < 6) hist1 = lds4(lds + G::kH1 + 4 * lane);
      if (lane < 6) *reinterpret_cast<float4 *>(ib + 4 * lane) = hist1;
      if (CONTIG) {
        *reinterpret_cast<float4 *>(ib + 24 + 4 * lane) = make_float4(aud[0], aud[1], aud[2], aud[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) ib[24 + lane + 64 * j] = aud[j];
      }
    }
    wave_sync();
    STAMP(10);  // demod + x2 staging
    PRIO(0);
    f2 u1[4];  // outputs (2n, 2n+1) of input n = 4 lane + u
    {
      float w[28];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const float4 t = lds4(lds + kI1 + 4 * lane + 4 * i);
        w[4 * i] = t.x;
        w[4 * i + 1] = t.y;
        w[4 * i + 2] = t.z;
        w[4 * i + 3] = t.w;
      }
      if (KEEP) {  // next frame's history: to its LDS slot
        if (lane < 6) *reinterpret_cast<float4 *>(lds + G::kH1 + 4 * lane) = lds4(lds + kI1 + 256 + 4 * lane);
      } else if (lane < 6) {
        hist1c = lds4(lds + kI1 + 256 + 4 * lane);
        if (!SEGPAR || (seg0 == 0 && seg1 == a.nframes)) *reinterpret_cast<float4 *>(st + kStInt1 + 4 * lane) = hist1c;
      }
      // arm_fir_interpolate_f32: out[2n + j - 1] = sum_t state[n + t] * c[(2 - j) + 2 t]:
      // (out[2n], out[2n+1]) += state[n+t] * (c[2t+1], c[2t])  -- one packed FMA per tap
#pragma unroll
      for (int u = 0; u < 4; ++u) u1[u] = splat(0.0f);
#pragma unroll
      for (int b = 0; b < 24; b += 8) {
        float ci[16];
        load_taps<16>(ci, (CFloatPtr)cf0->int1 + 2 * b);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int t = 0; t < 8; ++t) u1[u] = pk_fma(splat(w[u + b + t + 1]), f2{ci[1 + 2 * t], ci[2 * t]}, u1[u]);
        }
      }
    }
    STAMP(11);  // x2 interpolator
    FRESH_LANE();
    // ---- interpolate by 4 (32 taps, phase length 8): inputs n = 8 lane .. 8 lane + 7; the
    // 7-sample history is the neighbouring lane's tail (lane 0: last frame's, from HBM)
    {
      float w[15];
      float c4[32];
      load_taps<32>(c4, (CFloatPtr)cf0->int2);  // (pre-multiplied by the volume factor)
      const float x1[8] = {u1[0].x, u1[0].y, u1[1].x, u1[1].y, u1[2].x, u1[2].y, u1[3].x, u1[3].y};
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const float up = lane_up1(x1[i + 1]);
        const float hs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hist2), i + 1));
        w[i] = (lane == 0) ? hs : up;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) w[7 + i] = x1[i];
      if (!KEEP && lane == 63 && (!SEGPAR || (seg0 == 0 && seg1 == a.nframes))) {
        *reinterpret_cast<float4 *>(st + kStInt2) = make_float4(0.0f, x1[1], x1[2], x1[3]);
        *reinterpret_cast<float4 *>(st + kStInt2 + 4) = make_float4(x1[4], x1[5], x1[6], x1[7]);
      }
      if (PART == 2 && SEGPAR && seg0 == 0 && f == seg1 - 1 && seg1 < a.nframes) {
        // the state is written by the wave that read it (see the front end): what the call's last
        // audio samples leave behind
        const float *end = a.aud24 + ((size_t)ch * a.nframes + a.nframes) * D;
        if (lane < 6) *reinterpret_cast<float4 *>(st + kStInt1 + 4 * lane) = *reinterpret_cast<const float4 *>(end - 24 + 4 * lane);
        float xp[8];
        x2_tail(end, xp);
        if (lane == 0) {
          *reinterpret_cast<float4 *>(st + kStInt2) = make_float4(0.0f, xp[1], xp[2], xp[3]);
          *reinterpret_cast<float4 *>(st + kStInt2 + 4) = make_float4(xp[4], xp[5], xp[6], xp[7]);
        }
      }
      if (PART == 2 || KEEP) {  // the same seven values, kept for the next segment / frame: lane i = entry i
#pragma unroll
        for (int i = 1; i < 8; ++i) {
          const float t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x1[i]), 63));
          hist2c = (lane == i) ? t : hist2c;
        }
      }
      // out[4n + j - 1] = sum_t state[n + t] * c[(4 - j) + 4 t],  state[n + t] = w[u + t]:
      // (out[4n], out[4n+1]) += w * (c[4t+3], c[4t+2]);  (out[4n+2], out[4n+3]) += w * (c[4t+1], c[4t])
      // A lane owns 32 consecutive output samples (128 B).  Storing them directly would be 8
      // instructions of 64 scattered 16-byte pieces each (2 M partial-line writes per launch), so
      // each float4 goes to an XOR-swizzled LDS slot first (slot 8 lane + (u ^ (lane & 7)):
      // conflict-free both for these row writes and for the column reads below) ...
      wave_sync();
      unsigned qw[2] = {0u, 0u};  // WQ15: the four packed samples of the even u
      // KEEP, f32 samples: the transposition needs 2048 floats, more than the slice has free, so it
      // takes the slice from its start and the resident state it covers waits in registers
      // meanwhile: the /4 history (lanes 0..13) and the /2 history (lanes 16..39) share one
      // float4, the part of the overlap block below float 2048 (lanes 0..22) takes another.
      // (Two half-size transpositions instead -- 64-byte store segments -- cost 8..17 % of the
      // whole kernel: measured, tools/build_variant.sh -DT41RX_X_HALFTR=1.)
      constexpr bool PARK = KEEP && !WQ15 && !T41RX_X_HALFTR;
      constexpr int kOvPark = (2048 - G::kOV + 3) / 4;  // float4s of the overlap block below float 2048
      static_assert(!PARK || (kOvPark > 0 && kOvPark <= 64 && G::kH1 >= 2048), "parking layout");
      float4 park_h = make_float4(0, 0, 0, 0), park_o = make_float4(0, 0, 0, 0);
      if (PARK) {
        if (lane < 14) park_h = lds4(lds + kX + 2 * xpad(2 * lane));
        else if (lane >= 16 && lane < 40) park_h = lds4(lds + kY1 + y1slot(lane - 16));
        if (lane < kOvPark) park_o = lds4(lds + G::kOV + 4 * lane);
        wave_sync();
      }
      float *tr = PARK ? lds : lds + kScr;
      constexpr bool HALFTR = (KEEP && !PARK) || T41RX_X_HALFTR;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        f2 o01 = splat(0.0f), o23 = splat(0.0f);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const f2 x = splat(w[u + t]);
          o01 = pk_fma(x, f2{c4[4 * t + 3], c4[4 * t + 2]}, o01);
          o23 = pk_fma(x, f2{c4[4 * t + 1], c4[4 * t]}, o23);
        }
        // ---- volume (Process.cpp:929): DF * VolumeToAmplification() is folded into the x4 taps by the host
        // (DevCoef::int2), one rounding per tap instead of one per output
        if (!WQ15 && !HALFTR) {
          *reinterpret_cast<float4 *>(tr + 4 * (8 * lane + (u ^ (lane & 7)))) = make_float4(o01.x, o01.y, o23.x, o23.y);
        } else if (!WQ15) {
          // (experiment T41RX_X_HALFTR) a 1024-float transposition buffer: the 32 outputs of a lane
          // go out in two halves of 16 = 64 contiguous bytes per lane: slot 4 lane + ((u & 3) ^
          // swizzle), swizzle = (lane >> 1) & 3, and a store instruction then writes 16 rows of 64 B
          *reinterpret_cast<float4 *>(tr + 4 * (4 * lane + ((u & 3) ^ ((lane >> 1) & 3)))) = make_float4(o01.x, o01.y, o23.x, o23.y);
          if ((u & 3) == 3) {
            wave_sync();
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // float4 F = 64 i + lane of this half: row F >> 2, column F & 3
              const int row = 16 * i + (lane >> 2);
              const float4 t = lds4(tr + 4 * (4 * row + ((lane & 3) ^ ((row >> 1) & 3))));
              stg_stream(gO + 32 * row + 16 * (u >> 2) + 4 * (lane & 3), t);
            }
            wave_sync();
          }
        } else if ((u & 1) == 0) {  // arm_float_to_q15 (Process.cpp:936)
          qw[0] = q15_pack2(o01.x, o01.y);
          qw[1] = q15_pack2(o23.x, o23.y);
        } else {
          // 8 samples = one 16-byte piece; a lane has 4 of them: slot 4 lane + (piece ^ swizzle)
          const int piece = u >> 1;
          *reinterpret_cast<uint4 *>(tr + 4 * (4 * lane + (piece ^ ((lane >> 1) & 3)))) =
              make_uint4(qw[0], qw[1], q15_pack2(o01.x, o01.y), q15_pack2(o23.x, o23.y));
        }
      }
      wave_sync();
      STAMP(12);  // x4 interpolator + LDS transpose writes
      // ... and every global store instruction then writes 1 KiB of consecutive addresses:
      // float4 index F = 64 i + lane lives in row F >> 3 = 8 i + (lane >> 3), column lane & 7
      if (!WQ15 && !HALFTR) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = 8 * i + (lane >> 3);
          const float4 t = lds4(tr + 4 * (8 * row + ((lane & 7) ^ (row & 7))));
          stg_stream(gO + 256 * i + 4 * lane, t);
        }
      } else if (WQ15) {  // 4 pieces per row: piece F = 64 i + lane is row F >> 2, column lane & 3
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 16 * i + (lane >> 2);
          const float4 t = lds4(tr + 4 * (4 * row + ((lane & 3) ^ ((row >> 1) & 3))));
          stg_stream(gO + 256 * i + 4 * lane, t);
        }
      }
      if (PARK) {  // the resident state returns to its place
        wave_sync();
        if (lane < 14) *reinterpret_cast<float4 *>(lds + kX + 2 * xpad(2 * lane)) = park_h;
        else if (lane >= 16 && lane < 40) *reinterpret_cast<float4 *>(lds + kY1 + y1slot(lane - 16)) = park_h;
        if (lane < kOvPark) *reinterpret_cast<float4 *>(lds + G::kOV + 4 * lane) = park_o;
      }
    }
    STAMP(13);  // transposed reads + global stores
#ifdef T41RX_PIPE_STAT
    if (PIPE && lane == 0)
      (reinterpret_cast<unsigned long long *>(a.agc_pipe + (size_t)a.nchan * kPipeSlots * kPipeSlotFloats) + (size_t)job * 16)[10] += __builtin_readcyclecounter() - ps_t;
#endif
  }  // frames
#ifdef T41RX_STAMP
  // stamps go behind the demod tap's data: dbg_demod must be [nchan*nframes*256 floats | nchan*64 uint64]
  {
    unsigned long long rt;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
    if (lane == 29) stamp_acc = rt;
  }
  if (a.dbg_demod && PART != 2)  // (4096 pipeline: the front kernel's stamps)
    reinterpret_cast<unsigned long long *>(a.dbg_demod + (size_t)a.nchan * a.nframes * D)[(size_t)ch * 64 + lane] = stamp_acc;
#endif

  if (PIPE && AGC && lane < 50) *reinterpret_cast<float4 *>(st + st_agc(512) + 4 * lane) = agrec;
  if (KEEP) {  // the channel's record goes back to HBM once per launch
    wave_sync();
    if (lane < 14) *reinterpret_cast<float4 *>(st + kStDec1 + 4 * lane) = lds4(lds + kX + 2 * xpad(2 * lane));
    if (lane < 24) *reinterpret_cast<float4 *>(st + kStDec2 + 4 * lane) = lds4(lds + kY1 + y1slot(lane));
    *reinterpret_cast<float4 *>(st + kStOverlap + 4 * lane) = lds4(lds + G::kOV + 4 * lane);
    *reinterpret_cast<float4 *>(st + kStOverlap + 256 + 4 * lane) = lds4(lds + G::kOV + 256 + 4 * lane);
    if (lane < 6) *reinterpret_cast<float4 *>(st + kStInt1 + 4 * lane) = lds4(lds + G::kH1 + 4 * lane);
    if (lane < 8) st[kStInt2 + lane] = (lane == 0) ? 0.0f : hist2c;
  }
  if (PART != 2 && lane == 0) {
    if (!SEGPAR) {
      ncs->phase = phase0;
      ncs->r = osc_r;
      st[kStMisc + kMiscDc] = dc_carry;
    } else if (seg0 == 0) {  // (phase0 / dc_carry: advanced to the end of the call above)
      ncs->phase = phase0;
      ncs->r = osc_r;
      st[kStMisc + kMiscDc] = dc_carry;
    }
  }
  T41RX_CLK_END(job);
}
#ifdef T41RX_CLK
extern "C" __attribute__((visibility("default"))) int t41rx_debug_read_clk(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_t41_clk), sizeof(unsigned long long) * (size_t)n);
}
#endif

// ------------------------------------------------------------------------------------------
// FFT_LENGTH 4096 (BASELINE config 4, a synthetic generalisation: the firmware is compiled for
// 512; 1024 and 2048 likewise with R = 2, 4 in place of 8): overlap-save fast convolution, one
// channel per 4-wave workgroup.  4096 = 8 x 512:
//   pass 1 (DIF radix-8 over p, x[k' + 512 p]): DFT8, twiddle W4096^(k' q)  -> Z[q][k'] in place
//   pass 2 per q: fft512 over k' -> X[q + 8 m]; x mask; inverse fft512 over m -> W[q][k'] in place
//   pass 3 (inverse of pass 1): conj twiddle, inverse DFT8 over q -> y[k' + 512 p], natural order
// The 4096-point working array lives in LDS (32 KiB, every access is lane-contiguous), the
// 512-point sub-FFTs are the same register/LDS-exchange code as the 512 path.  Each pass is 8
// independent pieces (column blocks / rows): wave w takes pieces w and w + 4, workgroup barriers
// separate the passes.  (One wave per channel left a CU with 4 latency-bound waves: 28.5 us for
// 1024 channels; four per channel: 3 workgroups = 12 waves per CU.)
// ------------------------------------------------------------------------------------------
// The working array is R rows of 512 complex; a row is padded to the size of the fft512 exchange
// scratch (8 x kFftRow complex = 576), because while a wave holds a row in registers for its
// 512-point FFTs the row's own LDS is free to be that scratch: no separate scratch buffers,
// 36 KiB per workgroup at R = 8, four workgroups (16 waves) per CU.
constexpr int kFcRow = 8 * kFftRow;  // complex units per padded row
static_assert(kFcRow >= 512, "row padding");
constexpr int fc_lds_floats(int R) { return 2 * kFcRow * R + 2 * (448 + 56); }  // working array + the 512-point FFT's twiddles

// Waves per channel and LDS budget.  The kernel is bound by the latency of its barrier-separated
// phases, not by any throughput (stamps: a workgroup alone on a CU is hardly faster than one of
// three, and 8 waves per channel bought 3 %), so what counts is how many channels a CU holds at
// once -- and that every channel of the 1024-channel batch is resident in ONE round (3 per CU =
// 768 slots left a second round at a third of the occupancy: 16 frame-times where 10.7 would do).
// Fused kernel, R = 8: four workgroups per CU = 40 KiB each = [working array 36 KiB | twiddles
// 4 KiB] and nothing else: the frame's audio, then the output transposition buffers and the x4
// interpolator's boundary samples alias the working array (one more barrier), and what crosses the
// frames (24 audio samples, 7 x2 outputs) waits in two registers of wave 0.
#ifndef T41RX_FC_PRIO
#define T41RX_FC_PRIO 1
#endif
#ifndef T41RX_FC_X2
#define T41RX_FC_X2 1  // pass 2: the wave's two rows in lockstep
#endif
#ifndef T41RX_FCABL
#define T41RX_FCABL 0  // timing experiments: 1 no output stores, 2 no 512-point FFTs, 4 no input loads, 8 no x4 arithmetic, 16 no x2 arithmetic
#endif
constexpr int kFcWaves = 4;
constexpr int fc_arr_floats(int R) { return 2 * kFcRow * R < kFcWaves * 2048 + 8 * (R + 1) ? kFcWaves * 2048 + 8 * (R + 1) : 2 * kFcRow * R; }  // >= four transposition buffers + YT
constexpr int fcb_lds_floats(int R) { return fc_arr_floats(R) + 2 * (448 + 56); }
static_assert(fcb_lds_floats(8) * 4 * 4 <= 160 * 1024, "four fused workgroups per CU");
template <int R, bool INV>
__device__ __forceinline__ void dft_r(cf (&v)[R]) {
  if constexpr (R == 8) {
    dft8<INV>(v);
  } else if constexpr (R == 4) {
    const cf a0 = v[0] + v[2], a1 = v[0] - v[2], a2 = v[1] + v[3];
    // -j (v1 - v3) forward, +j (v1 - v3) inverse
    const cf d = v[1] - v[3];
    v[0] = a0 + a2;
    v[2] = a0 - a2;
    v[1] = INV ? add_pj(a1, d) : add_mj(a1, d);
    v[3] = INV ? add_mj(a1, d) : add_pj(a1, d);
  } else {
    const cf a0 = v[0] + v[1], a1 = v[0] - v[1];
    v[0] = a0;
    v[1] = a1;
  }
}

// N = 512 R, R = 2, 4, 8 (FFT_LENGTH 1024, 2048, 4096).  CPLX: hand the complex valid half on as it
// is (AM, AGC on: the back kernel applies the AGC / gain and demodulates), else the SSB audio
// fixed_gain * Re.
#ifndef T41RX_FC_WAVES
#define T41RX_FC_WAVES 4  // waves per SIMD the register allocation is held to (four workgroups per CU)
#endif
// BACK: SSB / NFM audio with the fixed gain goes straight on through the x2 / x4 interpolators and
// out (Process.cpp:917-931) instead of to the `aud24` scratch and a third kernel: pass 3 leaves
// the frame's N/2 audio samples in LDS (where the working array was), then every wave runs whole
// 256-sample segments of the back end (s = wave, wave + 4) with the arithmetic of rx512_kernel's,
// the x4 interpolator's 7-sample history crossing the segment boundaries through LDS.
template <int R, bool CPLX, bool BACK = false>
__global__ __launch_bounds__((64 * kFcWaves), T41RX_FC_WAVES) void fastconv_kernel(const RxArgs a) {
  static_assert(!(BACK && CPLX), "the fused back end takes real audio");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int N = 512 * R, D = N / 2;
  constexpr int NWV = kFcWaves, NT = 64 * NWV;  // waves / threads per channel
  constexpr int H = 8 / NWV;                       // column blocks (passes 1, 3) / rows (pass 2) / segments (back end) per wave
  int lane = threadIdx.x & 63;  // (re-defined per phase, see FRESH_LANE)
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ch = blockIdx.x;
  if (ch >= a.nchan) return;
  cf *A = reinterpret_cast<cf *>(smem);
  // behind the working array: the 512-point FFT's twiddles (tw1 [7][64], tw2 compacted to [7][8]),
  // read at the point of use -- the registers they would occupy hold what must not wait for L2
  constexpr int kArr = BACK ? fc_arr_floats(R) : 2 * kFcRow * R;
  cf *ltw = reinterpret_cast<cf *>(smem + kArr);
  float *AU = smem;                    // BACK, from pass 3 to the x2 interpolator: [0] pad, [1..23] history, [24 + i] audio sample i of the frame
  float *YT = smem + kFcWaves * 2048;  // BACK, behind the transposition buffers: [s][0] pad, [s][1..7] = the last 7 x2 outputs before segment s
  static_assert(!BACK || 24 + D + 8 <= kFcWaves * 2048, "the audio fits where the working array was");
  // BACK, wave 0: lane i < 24 = audio history entry i ([0] pad), lane i < 8 = x2 history entry i
  float hi_reg = 0.0f, yt_reg = 0.0f;
  float *st = a.state + (size_t)ch * state_floats(N);
  const cf *twN = reinterpret_cast<const cf *>(a.tab4k);                    // [R-1][512]
  const cf *maskN = reinterpret_cast<const cf *>(a.tab4k) + (R - 1) * 512;  // [R][512]
  const cf *tab = reinterpret_cast<const cf *>(a.tab);
  const float fixed_gain = ((CoefPtr)a.coef)->sc[kScFixedGain];

  for (int i = threadIdx.x; i < 448; i += NT) ltw[i] = tab[kTabTw1 + i];
  if (threadIdx.x < 56) ltw[448 + threadIdx.x] = tab[kTabTw2 + 64 * (threadIdx.x >> 3) + (threadIdx.x & 7)];
  if (BACK) {  // interpolator memories of the channel
    if (threadIdx.x < 24) hi_reg = st[kStInt1 + threadIdx.x];
    if (threadIdx.x < 8) yt_reg = st[kStInt2 + threadIdx.x];
  }
  // the outer radix-R pass' twiddles of this wave's two column blocks (r = wv, wv + 4): re-read
  // from L2 for passes 1 and 3 of every frame, requested ahead of the barrier in front of the pass
  // (pass 2 wants the 28 registers they would hold)
  cf twp[H][R > 1 ? R - 1 : 1];
  auto load_twp = [&]() {
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int q = 1; q < R; ++q) twp[h][q - 1] = twN[512 * (q - 1) + lane + 64 * (wv + NWV * h)];
  };

#ifdef T41RX_STAMP
  unsigned long long stamp_acc = 0, stamp_last, stamp_t0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t0)::"memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
  cf x1[H][R];
  auto load_x1 = [&](int f) {
    const cf *mid = reinterpret_cast<const cf *>(a.mid + ((size_t)ch * a.nframes4k + f) * (2 * D));
    const cf *prev = (f == 0) ? reinterpret_cast<const cf *>(st + kStOverlap) : mid - D;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int k = lane + 64 * (wv + NWV * h);
#pragma unroll
      for (int p = 0; p < R; ++p) {
        const int e = k + 512 * p;  // index into [previous | new]
        if (T41RX_FCABL & 4) x1[h][p] = cf{1.0f + lane, (float)f};
        else x1[h][p] = (p < R / 2) ? prev[e] : mid[e - D];
      }
    }
  };
  for (int f = 0; f < a.nframes4k; ++f) {
    FRESH_LANE();
#if T41RX_FC_PRIO
    // issue priority falls with progress: the arbiter favours the oldest waves, so without it the
    // first workgroup of a CU finishes long before the last (108 .. 195 us, stamps), which then
    // runs alone -- and alone a workgroup is latency-bound
    switch ((4 * f) / a.nframes4k) {
      case 0: PRIO(3); break;
      case 1: PRIO(2); break;
      case 2: PRIO(1); break;
      default: PRIO(0); break;
    }
#endif
    // ---- overlap-save assemble (Process.cpp:498-522): [previous N/2 | new N/2].  Inside a call the
    // previous block is the preceding frame's `mid` (just read, L2-warm); the state record supplies
    // it for the call's first frame and receives the last frame's block.  Pass 1 takes its inputs
    // x[k + 512 p] straight from there (8 bytes per lane, 512 per instruction): the array is first
    // written with pass 1's results.
    // (requesting the next frame's inputs before the back end instead -- the registers are there --
    // measured 1..3 % slower: the loads queue up behind the back end's 64 KiB of stores)
    load_x1(f);
    STAMP(2);
    load_twp();
    STAMP(0);         // (tail of the previous frame: stores issued, history rolled)
    __syncthreads();  // the previous frame's back end is done with the array (first frame: the twiddles are staged)
    STAMP(1);
    FRESH_LANE();
    // ---- pass 1
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int k = lane + 64 * (wv + NWV * h);
      cf v[R];
#pragma unroll
      for (int p = 0; p < R; ++p) v[p] = x1[h][p];
      if (f == a.nframes4k - 1) {  // next call's "previous"
#pragma unroll
        for (int p = R / 2; p < R; ++p) reinterpret_cast<cf *>(st + kStOverlap)[k + 512 * p - D] = v[p];
      }
      dft_r<R, false>(v);
#pragma unroll
      for (int q = 1; q < R; ++q) v[q] = cmul(v[q], twp[h][q - 1]);
#pragma unroll
      for (int q = 0; q < R; ++q) A[k + kFcRow * q] = v[q];
    }
    // the filter mask of this wave's rows of pass 2 (q = wv, wv + 4): requested here, so its L2
    // round trip runs under the barrier and the forward FFTs
    cf mk[H][8];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int q = wv + NWV * h;
      if (q < R) {
#pragma unroll
        for (int r = 0; r < 8; ++r) mk[h][r] = maskN[512 * q + lane + 64 * r];
      }
    }
    STAMP(5);
    __syncthreads();
    STAMP(6);
    FRESH_LANE();
    // ---- pass 2: 512-point FFT, mask (pre-scaled by 1/N), inverse 512-point FFT, per q
    if constexpr (R == 8 && H == 2 && T41RX_FC_X2) {  // both rows of the wave in lockstep
      const int q0 = wv, q1 = wv + NWV;
      cf v[8], u[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = A[kFcRow * q0 + lane + 64 * r];
#pragma unroll
      for (int r = 0; r < 8; ++r) u[r] = A[kFcRow * q1 + lane + 64 * r];
      float *xv = smem + 2 * kFcRow * q0, *xu = smem + 2 * kFcRow * q1;  // the rows themselves (now in registers) are the exchange scratch
      fft512_ldstw_x2<false>(v, u, ltw + lane, ltw + 448 + (lane & 7), xv, xu, lane);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        v[r] = cmul(v[r], mk[0][r]);
        u[r] = cmul(u[r], mk[1][r]);
      }
      fft512_ldstw_x2<true>(v, u, ltw + lane, ltw + 448 + (lane & 7), xv, xu, lane);
      wave_sync();
#pragma unroll
      for (int r = 0; r < 8; ++r) A[kFcRow * q0 + lane + 64 * r] = v[r];
#pragma unroll
      for (int r = 0; r < 8; ++r) A[kFcRow * q1 + lane + 64 * r] = u[r];
    } else {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int q = wv + NWV * h;
        if (q < R) {
          cf v[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = A[kFcRow * q + lane + 64 * r];
          float *xbuf = smem + 2 * kFcRow * q;  // the row itself (now in registers) is the exchange scratch
          wave_sync();
          if (!(T41RX_FCABL & 2)) fft512_ldstw<false>(v, ltw + lane, ltw + 448 + (lane & 7), xbuf, lane, []() {});
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = cmul(v[r], mk[h][r]);
          if (!(T41RX_FCABL & 2)) fft512_ldstw<true>(v, ltw + lane, ltw + 448 + (lane & 7), xbuf, lane, []() {});
          wave_sync();
#pragma unroll
          for (int r = 0; r < 8; ++r) A[kFcRow * q + lane + 64 * r] = v[r];
        }
      }
    }
    load_twp();
    STAMP(7);
    __syncthreads();
    STAMP(8);
    FRESH_LANE();
    // ---- pass 3; AGC off: fixed gain (DSP_Fn.cpp:494-502); SSB: audio = Re of the valid half
    float *au = a.aud24 + ((size_t)ch * a.nframes4k + f) * D;
    float y3[H][R / 2];  // BACK: the audio samples k + 512 j of this wave's column blocks
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int k = lane + 64 * (wv + NWV * h);
      cf v[R];
#pragma unroll
      for (int q = 0; q < R; ++q) v[q] = A[k + kFcRow * q];
#pragma unroll
      for (int q = 1; q < R; ++q) v[q] = cmulc(v[q], twp[h][q - 1]);
      dft_r<R, true>(v);
#pragma unroll
      for (int p = R / 2; p < R; ++p) {
        if (CPLX)
          reinterpret_cast<cf *>(a.aud24)[((size_t)ch * a.nframes4k + f) * D + k + 512 * (p - R / 2)] = v[p];
        else if (BACK)
          y3[h][p - R / 2] = fixed_gain * v[p].x;
        else
          au[k + 512 * (p - R / 2)] = fixed_gain * v[p].x;
      }
    }
    if (BACK) {
      const CoefPtr cf0 = (CoefPtr)a.coef;
      float ci[48];  // the x2 interpolator's taps: requested here, so the scalar loads' latency runs under the barrier
      load_taps<48>(ci, (CFloatPtr)fresh_coef(cf0)->int1);
      STAMP(9);
      __syncthreads();  // every wave has read its columns: the working array is free
      FRESH_LANE();
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int j = 0; j < R / 2; ++j) AU[24 + lane + 64 * (wv + NWV * h) + 512 * j] = y3[h][j];
      if (threadIdx.x < 24) AU[threadIdx.x] = hi_reg;
      if (threadIdx.x < 8) YT[threadIdx.x] = yt_reg;
      __syncthreads();  // the frame's audio is complete
      STAMP(10);
      FRESH_LANE();
      if (threadIdx.x < 24) hi_reg = AU[D + threadIdx.x];  // the next frame's history, before the transposition takes the place
      // ---- interpolate by 2 (48 taps, phase length 24), segment s: inputs n = 4 lane .. 4 lane + 3
      f2 u1[H][4];  // outputs (2n, 2n+1) of input n = 4 lane + u
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int sg = wv + NWV * h;
        if (sg < R) {
          const float *ib = AU + 256 * sg;
          float w[28];
#pragma unroll
          for (int i = 0; i < 7; ++i) {
            const float4 t = lds4(ib + 4 * lane + 4 * i);
            w[4 * i] = t.x;
            w[4 * i + 1] = t.y;
            w[4 * i + 2] = t.z;
            w[4 * i + 3] = t.w;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) u1[h][u] = splat(0.0f);
#pragma unroll
          for (int b = 0; b < 24; b += 8) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
              for (int t = 0; t < ((T41RX_FCABL & 16) ? 1 : 8); ++t) u1[h][u] = pk_fma(splat(w[u + b + t + 1]), f2{ci[2 * b + 1 + 2 * t], ci[2 * b + 2 * t]}, u1[h][u]);
            }
          }
          if (lane == 63) {  // what the next segment's x4 interpolator remembers
            *reinterpret_cast<float4 *>(YT + 8 * (sg + 1)) = make_float4(0.0f, u1[h][0].y, u1[h][1].x, u1[h][1].y);
            *reinterpret_cast<float4 *>(YT + 8 * (sg + 1) + 4) = make_float4(u1[h][2].x, u1[h][2].y, u1[h][3].x, u1[h][3].y);
          }
        }
      }
      float c4[32];  // the x4 interpolator's taps, likewise
      load_taps<32>(c4, (CFloatPtr)fresh_coef(cf0)->int2);
      STAMP(11);
      __syncthreads();
      STAMP(12);
      FRESH_LANE();
      if (f == a.nframes4k - 1) {  // the channel's interpolator memories after the call
        if (threadIdx.x < 24) st[kStInt1 + threadIdx.x] = hi_reg;
        else if (threadIdx.x >= 64 && threadIdx.x < 72) st[kStInt2 + threadIdx.x - 64] = YT[8 * R + threadIdx.x - 64];
      }
      // ---- interpolate by 4 (32 taps, phase length 8): inputs n = 8 lane .. 8 lane + 7; volume;
      // LDS transposition (2048 floats of the idle working array per wave); 1-KiB stores
      float *gOf = a.out + ((size_t)ch * a.nframes4k + f) * (size_t)(8 * D);
      float *tr = smem + 2048 * wv;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int sg = wv + NWV * h;
        if (sg < R) {
          float w[15];
          const float x1[8] = {u1[h][0].x, u1[h][0].y, u1[h][1].x, u1[h][1].y, u1[h][2].x, u1[h][2].y, u1[h][3].x, u1[h][3].y};
#pragma unroll
          for (int i = 0; i < 7; ++i) {
            const float up = lane_up1(x1[i + 1]);
            const float hs = YT[8 * sg + i + 1];
            w[i] = (lane == 0) ? hs : up;
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) w[7 + i] = x1[i];
          wave_sync();  // the buffer's previous readers are done
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            f2 o01 = splat(0.0f), o23 = splat(0.0f);
#pragma unroll
            for (int t = 0; t < ((T41RX_FCABL & 8) ? 1 : 8); ++t) {
              const f2 x = splat(w[u + t]);
              o01 = pk_fma(x, f2{c4[4 * t + 3], c4[4 * t + 2]}, o01);
              o23 = pk_fma(x, f2{c4[4 * t + 1], c4[4 * t]}, o23);
            }
            *reinterpret_cast<float4 *>(tr + 4 * (8 * lane + (u ^ (lane & 7)))) = make_float4(o01.x, o01.y, o23.x, o23.y);
          }
          wave_sync();
          float *gO = gOf + 2048 * sg;
#pragma unroll
          for (int i = 0; i < 8; ++i) {  // float4 F = 64 i + lane: row F >> 3 = the source lane, column lane & 7
            const int row = 8 * i + (lane >> 3);
            const float4 t = lds4(tr + 4 * (8 * row + ((lane & 7) ^ (row & 7))));
            if (!(T41RX_FCABL & 1) || t.x == 123.456f) stg_stream(gO + 256 * i + 4 * lane, t);
          }
        }
      }
      if (threadIdx.x < 8) yt_reg = YT[8 * R + threadIdx.x];  // the frame's last x2 outputs: the next frame's history
    }
  }
#ifdef T41RX_STAMP
  STAMP(13);
  {
    unsigned long long rt;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
    if (lane == 28) stamp_acc = stamp_t0;
    if (lane == 29) stamp_acc = rt;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (lane == 27) stamp_acc = hwid | ((unsigned long long)(xcc & 0xf) << 32);
  }
  // behind the front kernel's stamps: [nchan][waves][64] uint64
  if (a.dbg_demod)
    reinterpret_cast<unsigned long long *>(a.dbg_demod + (size_t)a.nchan * a.nframes * 256)[(size_t)a.nchan * 64 + ((size_t)ch * NWV + wv) * 64 + lane] = stamp_acc;
#endif
}

// ------------------------------------------------------------------------------------------
// FFT_LENGTH 4096, SSB, f32 samples, AGC off: the WHOLE chain in one kernel.  The two-kernel pipeline
// above moves every decimated sample through HBM once (`mid`: 16 KiB written and 32 KiB read back per
// channel-frame, a quarter more traffic than the path needs) and its front kernel takes longer than
// the fast convolution it feeds.  Here the workgroup that owns a channel's fast convolution also runs
// its front end: wave w takes segments 2w and 2w + 1 of the frame (2 x 2048 input samples) through DC
// high-pass, mixer and the two decimators with the segment-parallel kernel's arithmetic -- a wave
// that does not start the frame rebuilds the filter memories it needs from the 512 input samples in
// front of its first segment (L2-warm: the neighbouring wave is reading them), wave 0 takes them from
// the channel's record, where wave 3 left them at the end of the previous frame -- with its LDS slice
// in the idle working array; the 2 x 256 decimated samples of a wave cross to the radix-8 pass'
// column layout through that array too, and the previous block (the overlap-save "old" half) waits
// in 16 registers per lane in exactly the layout pass 1 wants.  From pass 1 on: fastconv_kernel<8,
// false, true>.  HBM sees the frame once in and once out.
// ------------------------------------------------------------------------------------------
#ifndef T41RX_FF_X2
#define T41RX_FF_X2 1  // pass 2: the wave's two rows in lockstep (costs registers)
#endif
// Register budget: four workgroups per CU hold the kernel to 128 registers.  Measured on MI355X (1024 channels x 32
// frames, us per frame; two-kernel pipeline 57.6): two input sub-blocks in flight + the previous block in registers
// 56.4 (42 registers spilled), two in flight + the previous block in the record 56.4 (13 spilled), ONE in flight + the
// record 54.2 (none spilled); three workgroups per CU (168 registers, nothing spilled, 768 channel slots) 60.6.
#ifndef T41RX_FF_PF
#define T41RX_FF_PF 1  // input sub-blocks in flight ahead of the one being worked on (1: one register set, 2: two)
#endif
#ifndef T41RX_FF_PREV_GLOBAL
#define T41RX_FF_PREV_GLOBAL 1  // the "previous" block waits in the channel's record (L2) instead of 16 registers per lane
#endif
typedef float f2n __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf ldg_stream2(const cf *p) {  // 8-byte load that does not look at this CU's L1
  const f2n t = __builtin_nontemporal_load(reinterpret_cast<const f2n *>(p));
  return cf{t.x, t.y};
}
template <bool PLAIN>
__global__ __launch_bounds__((64 * kFcWaves), T41RX_FC_WAVES) void fastconv_fused_kernel(const RxArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = 8, N = 512 * R, D = N / 2, L = 2048;
  constexpr int NWV = kFcWaves, H = 8 / NWV;
  static_assert(NWV == 4 && H == 2, "written for four waves per channel");
  static_assert(NWV * kLdsFloatsPerWave <= 2 * kFcRow * R, "the front end's slices fit in the working array");
  int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ch = blockIdx.x;
  if (ch >= a.nchan) return;
  T41RX_CLK_BEGIN();
  cf *A = reinterpret_cast<cf *>(smem);
  constexpr int kArr = fc_arr_floats(R);
  cf *ltw = reinterpret_cast<cf *>(smem + kArr);
  float *AU = smem;
  float *YT = smem + kFcWaves * 2048;
  float *lds = smem + wv * kLdsFloatsPerWave;  // front end: this wave's slice (X | Y1), inside the working array
  constexpr int kX = 0, kY1 = kXFloats;
  float hi_reg = 0.0f, yt_reg = 0.0f;
  float *st = a.state + (size_t)ch * state_floats(N);
  const cf *twN = reinterpret_cast<const cf *>(a.tab4k);
  const cf *maskN = reinterpret_cast<const cf *>(a.tab4k) + (R - 1) * 512;
  const float2 *__restrict__ tab = a.tab;
  const CoefPtr cf0 = (CoefPtr)a.coef;
  const NcoPtr nco = (NcoPtr)(a.nco + ch);
  const float fixed_gain = cf0->sc[kScFixedGain];
  const float2 hp8 = tab[kTabHp8 + lane];
  const float2 hp4 = tab[kTabHp4 + lane];

  for (int i = threadIdx.x; i < 448; i += 64 * NWV) ltw[i] = reinterpret_cast<const cf *>(tab)[kTabTw1 + i];
  if (threadIdx.x < 56) ltw[448 + threadIdx.x] = reinterpret_cast<const cf *>(tab)[kTabTw2 + 64 * (threadIdx.x >> 3) + (threadIdx.x & 7)];
  if (threadIdx.x < 24) hi_reg = st[kStInt1 + threadIdx.x];
  if (threadIdx.x < 8) yt_reg = st[kStInt2 + threadIdx.x];

  // oscillator: phase at the call's start (the copy the host names), rotation per sample, amplitude loop
  const NcoState *ncs_rd = reinterpret_cast<const NcoState *>(st + kStNco) + a.nco_rd;
  NcoState *ncs_wr = reinterpret_cast<NcoState *>(st + kStNco) + (a.nco_rd ^ 1);
  const uint64_t dphi = uniform_u64(nco->phase_inc);
  const uint64_t phase_call = uniform_u64(ncs_rd->phase);
  double osc_r = uniform_f64(ncs_rd->r);
  bool transient = (wv == 0) && fabs(osc_r * osc_r - uniform_f64(nco->r_star_sq)) > 1e-13;
  // gains (Process.cpp:117-134, 165-166); the DC high-pass takes b0 x
  float g_rf = a.g_rf, iq_phase_neg = 0.0f, iq_phase_pos = 0.0f;
  f2 g_iq = splat(1.0f);
  if (!PLAIN) {
    const float gb = a.g_band;
    const bool iq_on = a.iq_corr_on != 0;
    g_iq = f2{iq_on ? gb * a.neg_iq_amp : gb, gb};
    const float ph = iq_on ? a.iq_phase : 0.0f;
    iq_phase_neg = ph < 0.0f ? ph : 0.0f;
    iq_phase_pos = ph > 0.0f ? ph : 0.0f;
  }
  const float g_rf_i = (PLAIN && a.iq_corr_on) ? -g_rf : g_rf;
  const float g_hp = g_rf * (float)kHpB0;
  const float g_hp_i = (PLAIN && a.iq_corr_on) ? -g_hp : g_hp;

  // the overlap-save "previous" block in pass 1's layout: element k + 512 p, p < 4, k = lane + 64 (wv + 4 h)
  // (T41RX_FF_PREV_GLOBAL: it waits in the channel's record instead -- every lane re-reads next frame exactly the
  // elements it wrote itself, 16 KiB per channel that stay in L2 -- which keeps 16 registers free through the front end)
  cf prevx[H][R / 2];
  if (!T41RX_FF_PREV_GLOBAL) {
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int p = 0; p < R / 2; ++p)
        prevx[h][p] = reinterpret_cast<const cf *>(st + kStOverlap)[lane + 64 * (wv + NWV * h) + 512 * p];
  }
  cf twp[H][R - 1];
  auto load_twp = [&]() {
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int q = 1; q < R; ++q) twp[h][q - 1] = twN[512 * (q - 1) + lane + 64 * (wv + NWV * h)];
  };
  const float *gIc = a.I + (size_t)ch * a.nframes * L;  // (a.nframes counts 2048-sample segments)
  const float *gQc = a.Q + (size_t)ch * a.nframes * L;

  for (int f = 0; f < a.nframes4k; ++f) {
    FRESH_LANE();
#if T41RX_FC_PRIO
    switch ((4 * f) / a.nframes4k) {
      case 0: PRIO(3); break;
      case 1: PRIO(2); break;
      case 2: PRIO(1); break;
      default: PRIO(0); break;
    }
#endif
    __syncthreads();  // the previous frame's back end is done with the working array (first frame: the twiddles are staged)
    // =========================== front end: segments 2 wv, 2 wv + 1 of this frame ===========================
    cf ynew[2][2][2];  // [segment][round][even / odd]: /8 outputs m = 128 round + 2 lane + e of the segment
    {
      const int s0 = R * f + 2 * wv;  // first segment of this wave's run, counted from the call's start
      const float *gI = gIc + (size_t)s0 * L, *gQ = gQc + (size_t)s0 * L;
      uint64_t phase0 = phase_call + (uint64_t)s0 * (uint64_t)L * dphi;
      f2 dc2;
      // ---- filter memories at the start of the run
      auto stage_sub = [&](const float4 &rI0, const float4 &rI1, const float4 &rQ0, const float4 &rQ1, cf (&z)[8]) {
        z[0] = cf{rI0.x * g_hp_i, rQ0.x * g_hp};
        z[1] = cf{rI0.y * g_hp_i, rQ0.y * g_hp};
        z[2] = cf{rI0.z * g_hp_i, rQ0.z * g_hp};
        z[3] = cf{rI0.w * g_hp_i, rQ0.w * g_hp};
        z[4] = cf{rI1.x * g_hp_i, rQ1.x * g_hp};
        z[5] = cf{rI1.y * g_hp_i, rQ1.y * g_hp};
        z[6] = cf{rI1.z * g_hp_i, rQ1.z * g_hp};
        z[7] = cf{rI1.w * g_hp_i, rQ1.w * g_hp};
      };
      auto iq_corr = [&](cf (&z)[8]) {
        if (!PLAIN) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            z[k] *= g_iq;
            z[k].y = fmaf(iq_phase_neg, z[k].x, z[k].y);
            z[k].x = fmaf(iq_phase_pos, z[k].y, z[k].x);
          }
        }
      };
      // mixer for the 8 samples of this lane whose first one has oscillator phase P (Freq_Shift.cpp:94-141 + :42-65)
      auto mix = [&](cf (&z)[8], uint64_t P, float2 t) {
        const uint32_t u = (uint32_t)(P >> 24);
        const float ang = (float)u * (float)(6.283185307179586476925 / 256.0 / 4294967296.0);
        const float a2 = ang * ang;
        const float sn = ang * fmaf(a2, -1.0f / 6.0f, 1.0f);
        const float cs = fmaf(a2, fmaf(a2, 1.0f / 24.0f, -0.5f), 1.0f);
        const cf base = cmul(cf{t.x, t.y}, cf{cs, sn});
        const NcoPtr ncw = fresh_nco(nco);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const cf osc = cmul_s(base, cf{ncw->wk[k][0], ncw->wk[k][1]});
          z[k] = cmulc(z[k], osc);
        }
      };
      // input registers, two sub-blocks in flight
      float4 pI0[2], pI1[2], pQ0[2], pQ1[2];
      auto request = [&](int set, const float *pi, const float *pq) {
        if (T41RX_FF_PF < 2) set = 0;
        pI0[set] = ldg_stream(pi + 8 * lane);
        pI1[set] = ldg_stream(pi + 8 * lane + 4);
        pQ0[set] = ldg_stream(pq + 8 * lane);
        pQ1[set] = ldg_stream(pq + 8 * lane + 4);
      };
      float dc_carry = 0.0f;
      if (wv == 0) {
        // the frame's first segment: memories from the channel's record (the call's first frame: as the last call
        // left them) or from the hand-over slot where wave 3 left them at the end of the previous frame (two slots
        // in the `mid` scratch, alternating, so that wave 3 of THIS frame never writes what this wave still reads;
        // its stores were drained before the barrier above, and these loads do not look at this CU's L1)
        const float *src = (f == 0) ? st : a.mid + ((size_t)ch * 2 + ((f - 1) & 1)) * 256;
        request(0, gI, gQ);
        float4 h1 = make_float4(0, 0, 0, 0), h2 = make_float4(0, 0, 0, 0);
        if (lane < 14) h1 = ldg_stream(src + kStDec1 + 4 * lane);
        if (lane < 24) h2 = ldg_stream(src + kStDec2 + 4 * lane);
        const float4 dcs = ldg_stream(src + kStMisc);  // (kMiscDc first)
        if (T41RX_FF_PF >= 2) request(1, gI + 512, gQ + 512);
        wave_sync();
        if (lane < 14) *reinterpret_cast<float4 *>(lds + kX + 2 * xpad(2 * lane)) = h1;
        if (lane < 24) *reinterpret_cast<float4 *>(lds + kY1 + y1slot(lane)) = h2;
        wave_sync();
        dc_carry = uniform_f32(dcs.x);
      } else {
        // pre-roll: the 512 samples in front of the run through DC high-pass, mixer and /4 decimator
        request(1, gI - 512, gQ - 512);
        if (T41RX_FF_PF >= 2) request(0, gI, gQ);
        cf z[8];
        stage_sub(pI0[T41RX_FF_PF >= 2 ? 1 : 0], pI1[T41RX_FF_PF >= 2 ? 1 : 0], pQ0[T41RX_FF_PF >= 2 ? 1 : 0], pQ1[T41RX_FF_PF >= 2 ? 1 : 0], z);
        if (T41RX_FF_PF >= 2) request(1, gI + 512, gQ + 512);
        else request(0, gI, gQ);
        f2 dcs = splat(0.0f);
        dc_highpass<8>(z, dcs, lane, hp8.x, hp8.y);  // from rest: 512 samples on, its memory of the start is a1^512
        iq_corr(z);
        {
          const uint64_t P = phase0 - (uint64_t)(511 - 8 * lane) * dphi;
          mix(z, P, tab[kTabSinCos + (int)(P >> 56)]);
        }
        wave_sync();
        float *xw = lds + kX + 20 * lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<float4 *>(xw + 2 * (xpad(28 + 2 * i))) = make_float4(z[2 * i].x, z[2 * i].y, z[2 * i + 1].x, z[2 * i + 1].y);
        wave_sync();
        cf o1[2];
        auto pidx = [](int o) { return xpad(o); };
        fir_pair<kDec1Taps, 1, 5, 18, 6>(xw, pidx, (CFloatPtr)cf0->dec1, o1[0], o1[1]);
        float4 hh = make_float4(0, 0, 0, 0);
        if (lane < 14) hh = lds4(lds + kX + 2 * xpad(512 + 2 * lane));
        wave_sync();
        if (lane < 14) *reinterpret_cast<float4 *>(lds + kX + 2 * xpad(2 * lane)) = hh;
        if (lane >= 40) *reinterpret_cast<float4 *>(lds + kY1 + y1slot(lane - 40)) = make_float4(o1[0].x, o1[0].y, o1[1].x, o1[1].y);
        wave_sync();
        dc2 = dcs;  // inside a frame both chains of the shared biquad simply run on
      }
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) {
        if (wv == 0 && sg == 0) {
          // Q's DC-block start state = the state after ALL of the frame's I (one shared instance runs over I then Q,
          // Process.cpp:127-128): a1^256 ~ 3e-18, so the frame's last 256 I samples decide it
          const float4 tailF = *reinterpret_cast<const float4 *>(gI + (R * L - 256) + 4 * lane);
          const float x[4] = {tailF.x * g_rf, tailF.y * g_rf, tailF.z * g_rf, tailF.w * g_rf};
          dc2 = f2{(g_rf_i != g_rf) ? -dc_carry : dc_carry, dc_highpass_end_state<4>(x, hp4.x, hp4.y)};
        }
        float2 osc_tab[4];
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
          const uint64_t P = phase0 + (uint64_t)(512 * sb + 8 * lane + 1) * dphi;
          osc_tab[sb] = tab[kTabSinCos + (int)(P >> 56)];
        }
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
          for (int hh2 = 0; hh2 < 2; ++hh2) {
            const int sb = 2 * rd + hh2;
            cf z[8];
            stage_sub(pI0[T41RX_FF_PF >= 2 ? hh2 : 0], pI1[T41RX_FF_PF >= 2 ? hh2 : 0], pQ0[T41RX_FF_PF >= 2 ? hh2 : 0], pQ1[T41RX_FF_PF >= 2 ? hh2 : 0], z);
            // the sub-block after next / the next one (this segment's, or the run's second segment's)
            if (T41RX_FF_PF >= 2) {
              if (!(sg == 1 && sb >= 2)) request(hh2, gI + L * sg + 512 * (sb + 2), gQ + L * sg + 512 * (sb + 2));
            } else if (!(sg == 1 && sb == 3)) {
              request(0, gI + L * sg + 512 * (sb + 1), gQ + L * sg + 512 * (sb + 1));
            }
            dc_highpass<8>(z, dc2, lane, hp8.x, hp8.y);
            iq_corr(z);
            const int n0 = 512 * sb + 8 * lane;
            if (transient) {  // start-up of the oscillator's amplitude loop (Freq_Shift.cpp:130-134), first samples after a reset
              const NcoPtr nt = fresh_nco(nco);
              const double r_star_sq = uniform_f64(nt->r_star_sq);
              const double w_abs = uniform_f64(nt->w_abs);
              const double inv_r = 1.0 / sqrt(r_star_sq);
              double r = osc_r;
              float amp[8];
#pragma unroll
              for (int k = 0; k < 8; ++k) amp[k] = 1.0f;
              for (int g = 0; g < 64; ++g) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                  if (g == lane) amp[k] = (float)(r * inv_r);
                  r = r * (1.95 - r * r) * w_abs;
                }
                if (fabs(r * r - r_star_sq) <= 1e-13) break;
              }
              osc_r = r;
              transient = fabs(osc_r * osc_r - r_star_sq) > 1e-13;
#pragma unroll
              for (int k = 0; k < 8; ++k) z[k] *= splat(amp[k]);
            }
            mix(z, phase0 + (uint64_t)(n0 + 1) * dphi, osc_tab[sb]);
            wave_sync();
            float *xw = lds + kX + 20 * lane;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              *reinterpret_cast<float4 *>(xw + 2 * (xpad(28 + 2 * i))) = make_float4(z[2 * i].x, z[2 * i].y, z[2 * i + 1].x, z[2 * i + 1].y);
            wave_sync();
            cf o1[2];
            {
              auto pidx = [](int o) { return xpad(o); };
              fir_pair<kDec1Taps, 1, 5, 18, 6>(xw, pidx, (CFloatPtr)cf0->dec1, o1[0], o1[1]);
            }
            {
              float4 hh = make_float4(0, 0, 0, 0);
              if (lane < 14) hh = lds4(lds + kX + 2 * xpad(512 + 2 * lane));
              wave_sync();
              if (lane < 14) *reinterpret_cast<float4 *>(lds + kX + 2 * xpad(2 * lane)) = hh;
              *reinterpret_cast<float4 *>(lds + kY1 + y1slot(24 + 64 * hh2 + lane)) = make_float4(o1[0].x, o1[0].y, o1[1].x, o1[1].y);
            }
          }
          wave_sync();
          {
            auto planes = [](int o) { return y1slot(o >> 1) / 2; };
            fir_pair<kDec2Taps, 3, 5, 26, 6>(lds + kY1 + 4 * lane, planes, (CFloatPtr)cf0->dec2, ynew[sg][rd][0], ynew[sg][rd][1]);
          }
          {
            float4 hh = make_float4(0, 0, 0, 0);
            if (lane < 24) hh = lds4(lds + kY1 + y1slot(128 + lane));
            wave_sync();
            if (lane < 24) *reinterpret_cast<float4 *>(lds + kY1 + y1slot(lane)) = hh;
          }
        }
        phase0 += (uint64_t)L * dphi;
      }
      if (wv == NWV - 1) {
        // the frame ends here: the memories the next frame's first segment starts from (and, behind the call's last
        // frame, the channel's state) go to the record; the shared DC biquad ends the frame on Q
        wave_sync();
        float *dst = (f == a.nframes4k - 1) ? st : a.mid + ((size_t)ch * 2 + (f & 1)) * 256;
        if (lane < 14) *reinterpret_cast<float4 *>(dst + kStDec1 + 4 * lane) = lds4(lds + kX + 2 * xpad(2 * lane));
        if (lane < 24) *reinterpret_cast<float4 *>(dst + kStDec2 + 4 * lane) = lds4(lds + kY1 + y1slot(lane));
        if (lane == 0) dst[kStMisc + kMiscDc] = uniform_f32(dc2.y);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // in L2 before the next frame's barrier lets wave 0 read them
      }
    }
    __syncthreads();  // every wave is done with its slice: the array takes the new block in time order
    FRESH_LANE();
    {
      cf *Nb = reinterpret_cast<cf *>(smem);  // new block: sample n = 256 s + m at Nb[n]
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd)
          *reinterpret_cast<float4 *>(smem + 2 * (256 * (2 * wv + sg) + 128 * rd + 2 * lane)) =
              make_float4(ynew[sg][rd][0].x, ynew[sg][rd][0].y, ynew[sg][rd][1].x, ynew[sg][rd][1].y);
      __syncthreads();
      cf x1[H][R];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int k = lane + 64 * (wv + NWV * h);
#pragma unroll
        for (int p = 0; p < R / 2; ++p) {
          x1[h][p] = T41RX_FF_PREV_GLOBAL ? ldg_stream2(reinterpret_cast<const cf *>(st + kStOverlap) + k + 512 * p) : prevx[h][p];
          x1[h][R / 2 + p] = Nb[k + 512 * p];
        }
      }
      load_twp();
      __syncthreads();  // the new block is in registers everywhere: pass 1 may write the array
      FRESH_LANE();
      // ---- pass 1
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int k = lane + 64 * (wv + NWV * h);
        cf v[R];
#pragma unroll
        for (int p = 0; p < R; ++p) v[p] = x1[h][p];
#pragma unroll
        for (int p = 0; p < R / 2; ++p) prevx[h][p] = x1[h][R / 2 + p];  // next frame's "previous"
        if (T41RX_FF_PREV_GLOBAL || f == a.nframes4k - 1) {
#pragma unroll
          for (int p = R / 2; p < R; ++p) reinterpret_cast<cf *>(st + kStOverlap)[k + 512 * p - D] = v[p];
        }
        dft_r<R, false>(v);
#pragma unroll
        for (int q = 1; q < R; ++q) v[q] = cmul(v[q], twp[h][q - 1]);
#pragma unroll
        for (int q = 0; q < R; ++q) A[k + kFcRow * q] = v[q];
      }
    }
    cf mk[H][8];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int q = wv + NWV * h;
#pragma unroll
      for (int r = 0; r < 8; ++r) mk[h][r] = maskN[512 * q + lane + 64 * r];
    }
    __syncthreads();
    FRESH_LANE();
    // ---- pass 2
    if (T41RX_FF_X2) {
      const int q0 = wv, q1 = wv + NWV;
      cf v[8], u[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = A[kFcRow * q0 + lane + 64 * r];
#pragma unroll
      for (int r = 0; r < 8; ++r) u[r] = A[kFcRow * q1 + lane + 64 * r];
      float *xv = smem + 2 * kFcRow * q0, *xu = smem + 2 * kFcRow * q1;
      fft512_ldstw_x2<false>(v, u, ltw + lane, ltw + 448 + (lane & 7), xv, xu, lane);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        v[r] = cmul(v[r], mk[0][r]);
        u[r] = cmul(u[r], mk[1][r]);
      }
      fft512_ldstw_x2<true>(v, u, ltw + lane, ltw + 448 + (lane & 7), xv, xu, lane);
      wave_sync();
#pragma unroll
      for (int r = 0; r < 8; ++r) A[kFcRow * q0 + lane + 64 * r] = v[r];
#pragma unroll
      for (int r = 0; r < 8; ++r) A[kFcRow * q1 + lane + 64 * r] = u[r];
    } else {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int q = wv + NWV * h;
        cf v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = A[kFcRow * q + lane + 64 * r];
        float *xbuf = smem + 2 * kFcRow * q;
        wave_sync();
        fft512_ldstw<false>(v, ltw + lane, ltw + 448 + (lane & 7), xbuf, lane, []() {});
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = cmul(v[r], mk[h][r]);
        fft512_ldstw<true>(v, ltw + lane, ltw + 448 + (lane & 7), xbuf, lane, []() {});
        wave_sync();
#pragma unroll
        for (int r = 0; r < 8; ++r) A[kFcRow * q + lane + 64 * r] = v[r];
      }
    }
    load_twp();
    __syncthreads();
    FRESH_LANE();
    // ---- pass 3; fixed gain (DSP_Fn.cpp:494-502); SSB: audio = Re of the valid half
    float y3[H][R / 2];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int k = lane + 64 * (wv + NWV * h);
      cf v[R];
#pragma unroll
      for (int q = 0; q < R; ++q) v[q] = A[k + kFcRow * q];
#pragma unroll
      for (int q = 1; q < R; ++q) v[q] = cmulc(v[q], twp[h][q - 1]);
      dft_r<R, true>(v);
#pragma unroll
      for (int p = R / 2; p < R; ++p) y3[h][p - R / 2] = fixed_gain * v[p].x;
    }
    {
      float ci[48];
      load_taps<48>(ci, (CFloatPtr)fresh_coef(cf0)->int1);
      __syncthreads();
      FRESH_LANE();
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int j = 0; j < R / 2; ++j) AU[24 + lane + 64 * (wv + NWV * h) + 512 * j] = y3[h][j];
      if (threadIdx.x < 24) AU[threadIdx.x] = hi_reg;
      if (threadIdx.x < 8) YT[threadIdx.x] = yt_reg;
      __syncthreads();
      FRESH_LANE();
      if (threadIdx.x < 24) hi_reg = AU[D + threadIdx.x];
      f2 u1[H][4];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int sg = wv + NWV * h;
        const float *ib = AU + 256 * sg;
        float w[28];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const float4 t = lds4(ib + 4 * lane + 4 * i);
          w[4 * i] = t.x;
          w[4 * i + 1] = t.y;
          w[4 * i + 2] = t.z;
          w[4 * i + 3] = t.w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) u1[h][u] = splat(0.0f);
#pragma unroll
        for (int b = 0; b < 24; b += 8) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int t = 0; t < 8; ++t) u1[h][u] = pk_fma(splat(w[u + b + t + 1]), f2{ci[2 * b + 1 + 2 * t], ci[2 * b + 2 * t]}, u1[h][u]);
          }
        }
        if (lane == 63) {
          *reinterpret_cast<float4 *>(YT + 8 * (sg + 1)) = make_float4(0.0f, u1[h][0].y, u1[h][1].x, u1[h][1].y);
          *reinterpret_cast<float4 *>(YT + 8 * (sg + 1) + 4) = make_float4(u1[h][2].x, u1[h][2].y, u1[h][3].x, u1[h][3].y);
        }
      }
      float c4[32];
      load_taps<32>(c4, (CFloatPtr)fresh_coef(cf0)->int2);
      __syncthreads();
      FRESH_LANE();
      if (f == a.nframes4k - 1) {
        if (threadIdx.x < 24) st[kStInt1 + threadIdx.x] = hi_reg;
        else if (threadIdx.x >= 64 && threadIdx.x < 72) st[kStInt2 + threadIdx.x - 64] = YT[8 * R + threadIdx.x - 64];
      }
      float *gOf = a.out + ((size_t)ch * a.nframes4k + f) * (size_t)(8 * D);
      float *tr = smem + 2048 * wv;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int sg = wv + NWV * h;
        float w[15];
        const float x1v[8] = {u1[h][0].x, u1[h][0].y, u1[h][1].x, u1[h][1].y, u1[h][2].x, u1[h][2].y, u1[h][3].x, u1[h][3].y};
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const float up = lane_up1(x1v[i + 1]);
          const float hs = YT[8 * sg + i + 1];
          w[i] = (lane == 0) ? hs : up;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) w[7 + i] = x1v[i];
        wave_sync();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          f2 o01 = splat(0.0f), o23 = splat(0.0f);
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const f2 x = splat(w[u + t]);
            o01 = pk_fma(x, f2{c4[4 * t + 3], c4[4 * t + 2]}, o01);
            o23 = pk_fma(x, f2{c4[4 * t + 1], c4[4 * t]}, o23);
          }
          *reinterpret_cast<float4 *>(tr + 4 * (8 * lane + (u ^ (lane & 7)))) = make_float4(o01.x, o01.y, o23.x, o23.y);
        }
        wave_sync();
        float *gO = gOf + 2048 * sg;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = 8 * i + (lane >> 3);
          const float4 t = lds4(tr + 4 * (8 * row + ((lane & 7) ^ (row & 7))));
          stg_stream(gO + 256 * i + 4 * lane, t);
        }
      }
      if (threadIdx.x < 8) yt_reg = YT[8 * R + threadIdx.x];
    }
  }
  // the oscillator after the call (the other copy: see rx512_kernel)
  if (threadIdx.x == 0) {
    ncs_wr->phase = phase_call + (uint64_t)a.nframes * (uint64_t)L * dphi;
    ncs_wr->r = osc_r;
  }
  T41RX_CLK_END(ch * NWV + wv);
}

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

// ------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------
template <int MODE>
static hipError_t launch512(const RxArgs &a, hipStream_t s, bool debug) {
  // One 16-wave workgroup per CU (all 160 KiB of LDS, declared statically by the kernel): a
  // 4096-channel batch is one full, balanced wave of work on 256 CUs, and every wave keeps its
  // channel for all the frames of the launch.
  // (AGC on: 4-wave workgroups, see Geo)
#define T41RX_GO(DBG, PLN, AGCv, Q15v)                                                                   \
  hipLaunchKernelGGL((rx512_kernel<MODE, DBG, 0, PLN, AGCv, Q15v>),                                      \
                     dim3((a.nchan + Geo<0, AGCv>::kWaves - 1) / Geo<0, AGCv>::kWaves), dim3(Geo<0, AGCv>::kWaves * 64), 0, s, a)
  // AGC on, calls of four frames or more without taps: the pipelined variant (agc_prep_pipe); shorter calls have
  // nothing to overlap and take the barrier form, which computes the same values (T41RX_AGC_PIPE=0: experiments, tests)
  static const bool pipe_env = [] {
    const char *e = std::getenv("T41RX_AGC_PIPE");
    return !e || std::atoi(e) != 0;
  }();
  if constexpr (MODE != kModeSam) {
    if (a.agc && a.agc_pipe && !debug && a.nframes >= 4 && pipe_env) {
#define T41RX_GOP(PLN, Q15v)                                                                             \
  hipLaunchKernelGGL((rx512_kernel<MODE, false, 0, PLN, true, Q15v, false, true>), dim3((a.nchan + Geo<0>::kWaves - 1) / Geo<0>::kWaves), \
                     dim3(Geo<0>::kWaves * 64), 0, s, a)
      if (a.q15) {
        if (a.plain) T41RX_GOP(true, true); else T41RX_GOP(false, true);
      } else {
        if (a.plain) T41RX_GOP(true, false); else T41RX_GOP(false, false);
      }
#undef T41RX_GOP
      return hipGetLastError();
    }
  }
  if (a.q15 && debug) {  // q15 samples either side with the side outputs / stage taps (round 4; general front end)
    if (a.agc)
      T41RX_GO(true, false, true, true);
    else
      T41RX_GO(true, false, false, true);
  } else if (a.q15) {  // the firmware's q15 sample format either side
    if (a.agc) {
      if (a.plain)
        T41RX_GO(false, true, true, true);
      else
        T41RX_GO(false, false, true, true);
    } else if (a.plain)
      T41RX_GO(false, true, false, true);
    else
      T41RX_GO(false, false, false, true);
  } else if (a.agc) {
    if (debug)
      T41RX_GO(true, false, true, false);
    else if (a.plain)
      T41RX_GO(false, true, true, false);
    else
      T41RX_GO(false, false, true, false);
  } else if (debug)
    T41RX_GO(true, false, false, false);
  else if (a.plain)
    T41RX_GO(false, true, false, false);
  else
    T41RX_GO(false, false, false, false);
#undef T41RX_GO
  return hipGetLastError();
}

// FFT_LENGTH 512 R (R = 2, 4, 8): front half (R segments per frame) -> N-point fast convolution -> back half
static hipError_t launch_long(const RxArgs &a, int mode, hipStream_t s) {
  // FFT_LENGTH 4096, SSB audio with the fixed gain, f32 samples: the whole chain in one kernel (T41RX_FUSE_FRONT=0
  // or an explicit T41RX_SEG_RUN select the two-kernel pipeline: experiments, and the tests that compare the two)
  static const bool fuse_front_env = [] {
    const char *e = std::getenv("T41RX_FUSE_FRONT");
    return !e || std::atoi(e) != 0;
  }();
  if (a.seg == 8 && mode != T41RX_DEMOD_NFM && mode != T41RX_DEMOD_AM && !a.agc && !a.q15 && fuse_front_env && !std::getenv("T41RX_SEG_RUN")) {
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
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const bool cplx = a.agc || mode == T41RX_DEMOD_AM;
  // real audio with the fixed gain, f32 samples out: the interpolators run behind pass 3 of the
  // fast convolution (no `aud24` round trip, no third kernel)
  static const bool fuse_env = [] { const char *e = std::getenv("T41RX_FUSE_BACK"); return !e || std::atoi(e) != 0; }();  // experiments
  const bool fused = !cplx && !a.q15 && fuse_env;
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
  e = hipGetLastError();
  if (e != hipSuccess || fused) return e;
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

hipError_t launch_rx(const RxArgs &a, int fft_length, int mode, hipStream_t s) {
  const bool debug = a.dbg_nco || a.dbg_dec || a.dbg_demod || a.spect || a.dbg_pre || a.aud_out;  // side outputs (and the NR hand-over) ride on the tap kernels
  if (fft_length == 1024 || fft_length == 2048 || fft_length == 4096) {
    if (a.seg * 512 != fft_length) return hipErrorInvalidValue;
    return launch_long(a, mode, s);
  }
  if (fft_length != 512) return hipErrorInvalidValue;
  switch (mode) {
    case T41RX_DEMOD_USB:
    case T41RX_DEMOD_LSB:
      return launch512<kModeSsb>(a, s, debug);
    case T41RX_DEMOD_AM:
      return launch512<kModeAm>(a, s, debug);
    case T41RX_DEMOD_NFM:
      return launch512<kModeNfm>(a, s, debug);
    case T41RX_DEMOD_SAM: {  // the general front end
      static const bool pipe_env = [] {
        const char *e = std::getenv("T41RX_AGC_PIPE");
        return !e || std::atoi(e) != 0;
      }();
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
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace t41

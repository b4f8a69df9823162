// t41_sdr_amd/csrc/rx_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the T41 RX hot path.
//
// One 64-lane wavefront runs one whole ProcessIQData() call (Process.cpp:70-944) for one
// channel: 2048 complex f32 samples in -> 2048 real f32 samples out, every stage fused, so HBM
// sees the frame once (16 KiB in + 8 KiB out) plus the ~3 KiB per-channel streaming state.
// A 256-thread workgroup is four independent waves (no workgroup barriers anywhere); each
// wave owns a private LDS slice and synchronises with itself only (LDS is in-order per wave).
// 16 waves per CU x 256 CUs = 4096 channels in flight = BASELINE config 2's batch.
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
//   AGC off (fixed gain)          DSP_Fn.cpp:494-502
//   SSB demod                     Process.cpp:616-624,688-694
//   interpolate x2 (48 taps)      Process.cpp:917            int1 section
//   interpolate x4 (32 taps)      Process.cpp:920            int2 section (lane shuffles)
//   volume                        Process.cpp:929
//
// No MFMA: FIR taps and FFT butterflies are not dense contractions (BASELINE north_star).
#include <hip/hip_runtime.h>

#include "rx_internal.hpp"
#include "rx_kernels.hpp"

namespace t41 {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() {
  // Orders this wave's LDS traffic (other lanes' writes -> my reads).  The LDS unit executes
  // one wave's instructions in issue order, so no s_waitcnt or workgroup barrier is needed:
  // only the COMPILER must not move memory operations across this point.
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);  // also pin ALU work: keeps register live ranges per phase
  asm volatile("" ::: "memory");
}

struct alignas(8) cf {
  float x, y;
};
__device__ __forceinline__ cf cmul(cf a, cf b) {
  return cf{fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x)};
}
__device__ __forceinline__ cf cmulc(cf a, cf b) {  // a * conj(b)
  return cf{fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -(a.x * b.y))};
}
__device__ __forceinline__ cf cadd(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }

// 8-point DFT in registers, natural order in and out.  INV selects e^{+j...}.
template <bool INV>
__device__ __forceinline__ void dft8(cf (&v)[8]) {
  constexpr float kR = 0.70710678118654752440f;
  const cf a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]);
  const cf a2 = cadd(v[2], v[6]), a3 = csub(v[2], v[6]);
  const cf a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
  const cf a6 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
  const cf b0 = cadd(a0, a2), b1 = csub(a0, a2);
  const cf b2 = cadd(a4, a6), b3 = csub(a4, a6);
  // multiply by -j (forward) / +j (inverse)
  auto rot = [](cf z) { return INV ? cf{-z.y, z.x} : cf{z.y, -z.x}; };
  const cf ja3 = rot(a3), ja7 = rot(a7), jb3 = rot(b3);
  const cf c0 = cadd(a1, ja3), c1 = csub(a1, ja3);
  const cf d0 = cadd(a5, ja7), d1 = csub(a5, ja7);
  // W8^1 * d0 and W8^3 * d1
  cf e0, e1;
  if (!INV) {
    e0 = cf{(d0.x + d0.y) * kR, (d0.y - d0.x) * kR};
    e1 = cf{(d1.y - d1.x) * kR, -(d1.x + d1.y) * kR};
  } else {
    e0 = cf{(d0.x - d0.y) * kR, (d0.x + d0.y) * kR};
    e1 = cf{-(d1.x + d1.y) * kR, (d1.x - d1.y) * kR};
  }
  v[0] = cadd(b0, b2);
  v[4] = csub(b0, b2);
  v[2] = cadd(b1, jb3);
  v[6] = csub(b1, jb3);
  v[1] = cadd(c0, e0);
  v[5] = csub(c0, e0);
  v[3] = cadd(c1, e1);
  v[7] = csub(c1, e1);
}

// LDS exchange buffer row stride for the FFT transposes (in complex elements): 64 + 8 keeps
// both the row-major writes and the 8-strided reads bank-conflict-free for ds_*_b64.
constexpr int kFftRow = 72;

// 512-point complex FFT held as 8 points per lane: lane l register r <-> element l + 64 r,
// on input AND output (natural order both ways, no bit-reversal pass).
//   stage 1: DFT8 over r (stride 64), twiddle W512^(l q)
//   stage 2: DFT8 over bits 3..5 of l, twiddle W64^((l&7) q2)
//   stage 3: DFT8 over bits 0..2 of l
// tw1/tw2: this lane's 7+7 forward twiddles (INV conjugates them on the fly).
template <bool INV>
__device__ __forceinline__ void fft512(cf (&v)[8], const cf (&tw1)[7], const cf (&tw2)[7],
                                       float *__restrict__ xbuf, int lane) {
  cf *xb = reinterpret_cast<cf *>(xbuf);
  dft8<INV>(v);
#pragma unroll
  for (int q = 1; q < 8; ++q) v[q] = INV ? cmulc(v[q], tw1[q - 1]) : cmul(v[q], tw1[q - 1]);
  // exchange 1: (reg q, lane l1 + 8 k2) -> (reg k2, lane l1 + 8 q)
  wave_sync();
#pragma unroll
  for (int q = 0; q < 8; ++q) xb[q * kFftRow + lane] = v[q];
  wave_sync();
  {
    const int l1 = lane & 7, q = lane >> 3;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) v[k2] = xb[q * kFftRow + l1 + 8 * k2];
  }
  dft8<INV>(v);
#pragma unroll
  for (int q = 1; q < 8; ++q) v[q] = INV ? cmulc(v[q], tw2[q - 1]) : cmul(v[q], tw2[q - 1]);
  // exchange 2: (reg q2, lane l1 + 8 q) -> (reg l1, lane q + 8 q2)
  wave_sync();
  {
    const int l1 = lane & 7, q = lane >> 3;
#pragma unroll
    for (int q2 = 0; q2 < 8; ++q2) xb[q2 * kFftRow + q + 8 * l1] = v[q2];
  }
  wave_sync();
  {
    const int q = lane & 7, q2 = lane >> 3;
#pragma unroll
    for (int l1 = 0; l1 < 8; ++l1) v[l1] = xb[q2 * kFftRow + q + 8 * l1];
  }
  dft8<INV>(v);
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

// GFX9 DPP controls: data moves between lanes inside the VALU, no LDS round trip
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppWaveShr1 = 0x138, kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_f(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf, BOUND));
}
// value of lane-1 (lane 0 gets 0)
__device__ __forceinline__ float lane_up1(float v) { return dpp_f<kDppWaveShr1, 0xf, true>(0.0f, v); }

// Inclusive wave scan of the affine carry map d_out = A d_in + B with the same A = a1^n on
// every lane: 4 row_shr steps inside each 16-lane row, then row_bcast:15 / row_bcast:31 to
// stitch the rows.  m15 = A^((lane&15)+1), m31 = A^((lane&31)+1) (per-lane constants).
template <int n>
__device__ __forceinline__ float hp_scan(float B, float m15, float m31) {
  constexpr HpTab<n> T{};
  B = fmaf(T.scanA[0], dpp_f<kDppRowShr1, 0xf, true>(0.0f, B), B);
  B = fmaf(T.scanA[1], dpp_f<kDppRowShr2, 0xf, true>(0.0f, B), B);
  B = fmaf(T.scanA[2], dpp_f<kDppRowShr4, 0xf, true>(0.0f, B), B);
  B = fmaf(T.scanA[3], dpp_f<kDppRowShr8, 0xf, true>(0.0f, B), B);
  B = fmaf(m15, dpp_f<kDppRowBcast15, 0xa, false>(0.0f, B), B);
  B = fmaf(m31, dpp_f<kDppRowBcast31, 0xc, false>(0.0f, B), B);
  return B;
}

// Runs the recurrence over `n` consecutive samples per lane (lane-major: lane l owns samples
// l*n .. l*n+n-1), all 64 lanes in parallel: local pass with zero carry, wave scan of the
// carries, rank-1 fix-up.  `carry` (wave-uniform) is the filter state entering lane 0 and is
// replaced by the state leaving lane 63.
template <int n>
__device__ __forceinline__ void dc_highpass(float (&x)[n], float &carry, int lane, float m15, float m31) {
  constexpr HpTab<n> T{};
  const float b0 = (float)kHpB0, b1 = (float)kHpB1, a1 = (float)kHpA1;
  float d = (lane == 0) ? carry : 0.0f;
#pragma unroll
  for (int k = 0; k < n; ++k) {
    const float y = fmaf(b0, x[k], d);
    d = fmaf(a1, y, b1 * x[k]);
    x[k] = y;
  }
  const float B = hp_scan<n>(d, m15, m31);
  const float e = lane_up1(B);
#pragma unroll
  for (int k = 0; k < n; ++k) x[k] = fmaf(T.pw[k], e, x[k]);
  carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(B), 63));
}

// filter state after `n` samples per lane when only the end state matters (zero start state)
template <int n>
__device__ __forceinline__ float dc_highpass_end_state(const float (&x)[n], float m15, float m31) {
  const float c = (float)(kHpB1 + kHpA1 * kHpB0), a1 = (float)kHpA1;
  float d = 0.0f;
#pragma unroll
  for (int k = 0; k < n; ++k) d = fmaf(a1, d, c * x[k]);
  const float B = hp_scan<n>(d, m15, m31);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(B), 63));
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
// LDS layout of one wave (floats)
// ------------------------------------------------------------------------------------------
// X  : post-NCO samples of one 512-sample sub-block + 28-entry history, I and Q.
//      logical index j: [0] pad, [1..27] history, [28+k] new sample k.  4 floats of padding
//      every 128 keep the 32-byte-strided ds_read_b128 of the /4 decimator conflict-free.
// Y1 : /4 decimator outputs of two sub-blocks (256) + 48-entry history, I and Q.
//      logical j: [0..2] pad, [3..47] history, [48+m] new.
constexpr int kXStride = 560;
constexpr int kXI = 0, kXQ = kXStride;
constexpr int kY1Len = 304;
constexpr int kY1I = 2 * kXStride, kY1Q = kY1I + kY1Len;
constexpr int kLdsFloatsPerWave = kY1Q + kY1Len;  // 1728 floats = 6912 B
static_assert(kLdsFloatsPerWave >= 8 * kFftRow * 2, "FFT exchange buffer must fit");
// workgroup-shared copy of the constant tables the FFT needs (float2 units, same order as the
// global table): mask[8][64], tw1[7][64], tw2[7][64]
constexpr int kLdsTabFloats = 2 * kTabSinCos;  // everything before the sin/cos table
constexpr int kLdsWaveBase = kLdsTabFloats;
__device__ __forceinline__ int xpad(int j) { return j + ((j >> 7) << 2); }

typedef const __attribute__((address_space(4))) DevCoef *CoefPtr;
__device__ __forceinline__ CoefPtr fresh_coef(CoefPtr p) {
  asm volatile("" : "+s"(p));
  return p;
}

__device__ __forceinline__ float4 lds4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

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


// Two adjacent outputs of a decimating FIR from one lane-contiguous LDS window:
//   acc0 = sum_i c[i] * w[OFF0 + i],  acc1 = sum_i c[i] * w[OFF1 + i],  i = 0..NT-1 in order
// (arm_fir_decimate_f32's tap order), where w[] is NLOAD float4 reads starting at `win`
// (IDX maps the logical float index of each float4 to its padded LDS position).
// The window is streamed: every value is consumed right after its ds_read_b128, taps arrive in
// 8-wide scalar-load chunks just before first use, and a wave-scope fence every GROUP loads
// keeps the compiler from hoisting the whole window into registers.
template <int NT, int OFF0, int OFF1, int NLOAD, int GROUP, typename IDX>
__device__ __forceinline__ void fir_pair(const float *win, IDX idx, CFloatPtr taps, float &acc0,
                                         float &acc1) {
  constexpr int NTP = (NT + 7) & ~7;
  float tc[NTP];
  acc0 = 0.0f;
  acc1 = 0.0f;
#pragma unroll
  for (int l = 0; l < NLOAD; ++l) {
    if (l > 0 && (l % GROUP) == 0) {
      // tie the accumulators to the instruction order: everything issued so far is consumed
      // before the next group of loads is issued (bounded live ranges)
      asm volatile("" : "+v"(acc0), "+v"(acc1)::"memory");
    }
    const float4 t = lds4(win + idx(4 * l));
    const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = 4 * l + j;
      const int i0 = x - OFF0, i1 = x - OFF1;
      if (i0 >= 0 && i0 < NT) {
        if ((i0 & 7) == 0) {
          float chunk[8];
          load_taps<8>(chunk, taps + i0);
#pragma unroll
          for (int q = 0; q < 8; ++q) tc[i0 + q] = chunk[q];
        }
        acc0 = fmaf(tc[i0], tv[j], acc0);
      }
      if (i1 >= 0 && i1 < NT) acc1 = fmaf(tc[i1], tv[j], acc1);
    }
  }
}

typedef const __attribute__((address_space(4))) ChanNco *NcoPtr;
__device__ __forceinline__ NcoPtr fresh_nco(NcoPtr p) {
  asm volatile("" : "+s"(p));
  return p;
}


// ------------------------------------------------------------------------------------------
// The fused kernel, FFT_LENGTH = 512
// ------------------------------------------------------------------------------------------
template <int MODE, bool DEBUG>
__global__ __launch_bounds__(256, 4) void rx512_kernel(const RxArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int L = 2048, D = 256, N = 512;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ch = blockIdx.x * 4 + wv;

  // ---- stage mask + twiddles in LDS once per workgroup (the only workgroup barrier)
  {
    const float4 *src = reinterpret_cast<const float4 *>(a.tab);
    float4 *dst = reinterpret_cast<float4 *>(smem);
#pragma unroll
    for (int i = threadIdx.x; i < kLdsTabFloats / 4; i += 256) dst[i] = src[i];
  }
  // per-lane constants of the DC high-pass scan
  const float2 hp8 = a.tab[kTabHp8 + lane];
  const float2 hp4 = a.tab[kTabHp4 + lane];
  __syncthreads();
  if (ch >= a.nchan) return;  // whole wave leaves; no further workgroup barriers

  const cf *ltab = reinterpret_cast<const cf *>(smem);
  float *lds = smem + kLdsWaveBase + wv * kLdsFloatsPerWave;
  float *st = a.state + (size_t)ch * state_floats(N);
  // coefficients are read-only for the kernel: constant address space -> scalar (SMEM) loads,
  // re-derived through an opaque asm per phase so the compiler keeps the tap loads next to
  // their use instead of hoisting all 180 of them (and spilling SGPRs).
  const CoefPtr cf0 = (CoefPtr)a.coef;
  const NcoPtr nco = (NcoPtr)(a.nco + ch);
  const float2 *__restrict__ tab = a.tab;

  // per-channel NCO constants and state (wave-uniform)
  const uint64_t dphi = uniform_u64(nco->phase_inc);
  NcoState *ncs = reinterpret_cast<NcoState *>(st + kStNco);
  uint64_t phase0 = uniform_u64(ncs->phase);
  double osc_r = uniform_f64(ncs->r);
  float dc_carry = uniform_f32(st[kStMisc + kMiscDc]);
  bool transient;
  {
    const double rs = uniform_f64(nco->r_star_sq);
    transient = fabs(osc_r * osc_r - rs) > 1e-13;
  }

  for (int f = 0; f < a.nframes; ++f) {
    const size_t fbase = ((size_t)ch * a.nframes + f) * L;
    const float *__restrict__ gI = a.I + fbase;
    const float *__restrict__ gQ = a.Q + fbase;
    float *__restrict__ gO = a.out + fbase;

    // ---- first loads of the frame
    float4 nI0 = *reinterpret_cast<const float4 *>(gI + 8 * lane);
    float4 nI1 = *reinterpret_cast<const float4 *>(gI + 8 * lane + 4);
    float4 nQ0 = *reinterpret_cast<const float4 *>(gQ + 8 * lane);
    float4 nQ1 = *reinterpret_cast<const float4 *>(gQ + 8 * lane + 4);
    const float4 tailI = *reinterpret_cast<const float4 *>(gI + (L - 256) + 4 * lane);

    // ---- delay lines HBM -> LDS (every frame is self-contained: load state, run, store state)
    wave_sync();
    if (lane < 7) {
      *reinterpret_cast<float4 *>(lds + kXI + 4 * lane) = *reinterpret_cast<const float4 *>(st + kStDec1I + 4 * lane);
      *reinterpret_cast<float4 *>(lds + kXQ + 4 * lane) = *reinterpret_cast<const float4 *>(st + kStDec1Q + 4 * lane);
    }
    if (lane < 12) {
      *reinterpret_cast<float4 *>(lds + kY1I + 4 * lane) = *reinterpret_cast<const float4 *>(st + kStDec2I + 4 * lane);
      *reinterpret_cast<float4 *>(lds + kY1Q + 4 * lane) = *reinterpret_cast<const float4 *>(st + kStDec2Q + 4 * lane);
    }

    // gains (Process.cpp:117-134, 165-166).  g_band and -IQAmp are folded into one factor on I
    // (exact whenever either is +-1, which is the firmware default; one rounding otherwise)
    float g_rf, g_i, g_q, iq_phase;
    {
      const CoefPtr c = fresh_coef(cf0);
      g_rf = c->sc[kScRfGain];
      const float gb = c->sc[kScBandGain];
      const bool iq_on = c->sc[kScIqCorrOn] != 0.0f;
      g_i = iq_on ? gb * c->sc[kScNegIqAmp] : gb;
      g_q = gb;
      iq_phase = iq_on ? c->sc[kScIqPhase] : 0.0f;
    }

    // ---- Q's DC-block start state = state after ALL of this frame's I (one shared biquad
    // instance runs over I then Q, Process.cpp:127-128).  a1^256 ~ 3e-18, so the last 256 I
    // samples decide it.
    float dc_carry_i = dc_carry;
    float dc_carry_q;
    {
      const float x[4] = {tailI.x * g_rf, tailI.y * g_rf, tailI.z * g_rf, tailI.w * g_rf};
      dc_carry_q = dc_highpass_end_state<4>(x, hp4.x, hp4.y);
    }

    float y2I[2][2], y2Q[2][2];  // /8 outputs of this frame: m = 128*round + 2*lane + e
    cf v[8];                     // FFT registers; v[0..3] = previous block, prefetched below

#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        const int s = 2 * rd + h;
        float xi[8] = {nI0.x, nI0.y, nI0.z, nI0.w, nI1.x, nI1.y, nI1.z, nI1.w};
        float xq[8] = {nQ0.x, nQ0.y, nQ0.z, nQ0.w, nQ1.x, nQ1.y, nQ1.z, nQ1.w};
        if (s < 3) {  // prefetch the next sub-block
          const int o = 512 * (s + 1) + 8 * lane;
          nI0 = *reinterpret_cast<const float4 *>(gI + o);
          nI1 = *reinterpret_cast<const float4 *>(gI + o + 4);
          nQ0 = *reinterpret_cast<const float4 *>(gQ + o);
          nQ1 = *reinterpret_cast<const float4 *>(gQ + o + 4);
        } else {  // last sub-block: prefetch the overlap-save "previous" block instead
          const float2 *ov = reinterpret_cast<const float2 *>(st + kStOverlap);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float2 t = ov[64 * j + lane];
            v[j] = cf{t.x, t.y};
          }
        }
        // -- RF gain, DC high-pass, band gain / IQ amplitude
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          xi[k] *= g_rf;
          xq[k] *= g_rf;
        }
        dc_highpass<8>(xi, dc_carry_i, lane, hp8.x, hp8.y);
        dc_highpass<8>(xq, dc_carry_q, lane, hp8.x, hp8.y);
        if (g_i != 1.0f) {
#pragma unroll
          for (int k = 0; k < 8; ++k) xi[k] *= g_i;
        }
        if (g_q != 1.0f) {
#pragma unroll
          for (int k = 0; k < 8; ++k) xq[k] *= g_q;
        }
        // -- IQ phase correction (Utility.cpp:178-187)
        if (iq_phase < 0.0f) {
#pragma unroll
          for (int k = 0; k < 8; ++k) xq[k] = fmaf(iq_phase, xi[k], xq[k]);
        } else if (iq_phase > 0.0f) {
#pragma unroll
          for (int k = 0; k < 8; ++k) xi[k] = fmaf(iq_phase, xq[k], xi[k]);
        }
        // -- oscillator for my 8 samples.  Osc_n = V_n * W has phase phase0 + (n+1) dphi.
        const int n0 = 512 * s + 8 * lane;
        float amp[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) amp[k] = 1.0f;
        if (transient) {
          // start-up of the amplitude loop g = 1.95 - |V|^2 (Freq_Shift.cpp:130-134): replay the
          // scalar recurrence (wave-uniform); each lane keeps its own 8 values.  |Osc_n| / A* =
          // |V_n| / r*.
          const NcoPtr nt = fresh_nco(nco);
          const double r_star_sq = uniform_f64(nt->r_star_sq);
          const double w_abs = uniform_f64(nt->w_abs);
          const double inv_r = 1.0 / sqrt(r_star_sq);
          double r = osc_r;
          for (int g = 0; g < 64; ++g) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              if (g == lane) amp[k] = (float)(r * inv_r);
              r = r * (1.95 - r * r) * w_abs;
            }
            if (fabs(r * r - r_star_sq) <= 1e-13) break;
          }
          osc_r = r;
        }
        cf base;
        {
          const uint64_t P = phase0 + (uint64_t)(n0 + 1) * dphi;
          const float2 t = tab[kTabSinCos + (int)(P >> 56)];
          const uint32_t u = (uint32_t)(P >> 24);
          const float ang = (float)u * (float)(6.283185307179586476925 / 256.0 / 4294967296.0);
          const float a2 = ang * ang;
          const float sn = ang * fmaf(a2, -1.0f / 6.0f, 1.0f);
          const float cs = fmaf(a2, fmaf(a2, 1.0f / 24.0f, -0.5f), 1.0f);
          base = cmul(cf{t.x, t.y}, cf{cs, sn});
        }
        // -- Fs/4 shift (x j^n, Freq_Shift.cpp:42-65) and NCO mix (Freq_Shift.cpp:138-139):
        //    (I' + jQ') = (Iex + jQex) * conj(Osc)
        float mi[8], mq[8];
        const NcoPtr ncw = fresh_nco(nco);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const cf w = cf{ncw->wk[k][0], ncw->wk[k][1]};
          cf osc = cmul(base, w);
          if (transient) {
            osc.x *= amp[k];
            osc.y *= amp[k];
          }
          float ex_i, ex_q;
          switch (k & 3) {
            case 0: ex_i = xi[k]; ex_q = xq[k]; break;
            case 1: ex_i = -xq[k]; ex_q = xi[k]; break;
            case 2: ex_i = -xi[k]; ex_q = -xq[k]; break;
            default: ex_i = xq[k]; ex_q = -xi[k]; break;
          }
          mi[k] = fmaf(ex_i, osc.x, ex_q * osc.y);
          mq[k] = fmaf(ex_q, osc.x, -(ex_i * osc.y));
        }
        if (transient) {
          const double rs = uniform_f64(fresh_nco(nco)->r_star_sq);
          transient = fabs(osc_r * osc_r - rs) > 1e-13;
        }
        if (DEBUG && a.dbg_nco) {
          float *dn = a.dbg_nco + ((size_t)ch * a.nframes + f) * (2 * L);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            dn[n0 + k] = mi[k];
            dn[L + n0 + k] = mq[k];
          }
        }
        // -- stage into LDS, then decimate by 4 (28 taps): outputs m = 2*lane, 2*lane+1
        wave_sync();
        {
          const int j0 = xpad(28 + 8 * lane), j1 = xpad(32 + 8 * lane);
          *reinterpret_cast<float4 *>(lds + kXI + j0) = make_float4(mi[0], mi[1], mi[2], mi[3]);
          *reinterpret_cast<float4 *>(lds + kXI + j1) = make_float4(mi[4], mi[5], mi[6], mi[7]);
          *reinterpret_cast<float4 *>(lds + kXQ + j0) = make_float4(mq[0], mq[1], mq[2], mq[3]);
          *reinterpret_cast<float4 *>(lds + kXQ + j1) = make_float4(mq[4], mq[5], mq[6], mq[7]);
        }
        wave_sync();
        float o1[2][2];  // [I/Q][e]
        // arm_fir_decimate_f32: y[m] = sum_i c[i] * state[4m + i]; state[i] = buf[i + 1]
        {
          const int wbase = xpad(8 * lane);
          // a lane's 36-float window [8 lane, 8 lane + 36) can cross one 128-float pad
          // boundary: use the exact padded index per float4
          auto pidx = [&](int o) { return xpad(8 * lane + o) - wbase; };
          fir_pair<kDec1Taps, 1, 5, 9, 5>(lds + kXI + wbase, pidx, (CFloatPtr)cf0->dec1, o1[0][0], o1[0][1]);
          wave_sync();
          fir_pair<kDec1Taps, 1, 5, 9, 5>(lds + kXQ + wbase, pidx, (CFloatPtr)cf0->dec1, o1[1][0], o1[1][1]);
        }
        // -- roll the /4 history (logical 512..539 -> 0..27) and append the /4 outputs
        {
          float4 hI = make_float4(0, 0, 0, 0), hQ = make_float4(0, 0, 0, 0);
          if (lane < 7) {
            hI = lds4(lds + kXI + xpad(512 + 4 * lane));
            hQ = lds4(lds + kXQ + xpad(512 + 4 * lane));
          }
          wave_sync();
          if (lane < 7) {
            *reinterpret_cast<float4 *>(lds + kXI + 4 * lane) = hI;
            *reinterpret_cast<float4 *>(lds + kXQ + 4 * lane) = hQ;
          }
          const int j = 48 + 128 * h + 2 * lane;
          *reinterpret_cast<float2 *>(lds + kY1I + j) = make_float2(o1[0][0], o1[0][1]);
          *reinterpret_cast<float2 *>(lds + kY1Q + j) = make_float2(o1[1][0], o1[1][1]);
        }
      }  // h
      // ---- decimate by 2 (46 taps) over the 256 new /4 samples: m = 2*lane, 2*lane+1
      wave_sync();
      // y[m] = sum_i c[i] * state[2m + i]; state[i] = buf[i + 3]
      {
        auto lin = [](int o) { return o; };
        fir_pair<kDec2Taps, 3, 5, 13, 5>(lds + kY1I + 4 * lane, lin, (CFloatPtr)cf0->dec2, y2I[rd][0], y2I[rd][1]);
        wave_sync();
        fir_pair<kDec2Taps, 3, 5, 13, 5>(lds + kY1Q + 4 * lane, lin, (CFloatPtr)cf0->dec2, y2Q[rd][0], y2Q[rd][1]);
      }
      {  // roll the /2 history: logical 256..303 -> 0..47
        float4 hI = make_float4(0, 0, 0, 0), hQ = make_float4(0, 0, 0, 0);
        if (lane < 12) {
          hI = lds4(lds + kY1I + 256 + 4 * lane);
          hQ = lds4(lds + kY1Q + 256 + 4 * lane);
        }
        wave_sync();
        if (lane < 12) {
          *reinterpret_cast<float4 *>(lds + kY1I + 4 * lane) = hI;
          *reinterpret_cast<float4 *>(lds + kY1Q + 4 * lane) = hQ;
        }
      }
    }  // rd
    phase0 += (uint64_t)L * dphi;
    dc_carry = dc_carry_q;  // the shared biquad ends the frame on Q

    // ---- delay lines back to HBM (the LDS copies are about to be reused as scratch); issue
    // the small back-end history loads now so the FFT hides their latency
    wave_sync();
    if (lane < 7) {
      *reinterpret_cast<float4 *>(st + kStDec1I + 4 * lane) = lds4(lds + kXI + 4 * lane);
      *reinterpret_cast<float4 *>(st + kStDec1Q + 4 * lane) = lds4(lds + kXQ + 4 * lane);
    }
    if (lane < 12) {
      *reinterpret_cast<float4 *>(st + kStDec2I + 4 * lane) = lds4(lds + kY1I + 4 * lane);
      *reinterpret_cast<float4 *>(st + kStDec2Q + 4 * lane) = lds4(lds + kY1Q + 4 * lane);
    }
    float4 hist1 = make_float4(0, 0, 0, 0);
    if (lane < 6) hist1 = *reinterpret_cast<const float4 *>(st + kStInt1 + 4 * lane);
    float hist2 = 0.0f;
    if (lane < 8) hist2 = st[kStInt2 + lane];
    wave_sync();

    // ---- level adjust (Process.cpp:481-492)
    const float level = fresh_coef(cf0)->sc[kScLevel];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        y2I[rd][e] *= level;
        y2Q[rd][e] *= level;
      }
    if (DEBUG && a.dbg_dec) {
      float *dd = a.dbg_dec + ((size_t)ch * a.nframes + f) * N;
#pragma unroll
      for (int rd = 0; rd < 2; ++rd)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          dd[128 * rd + 2 * lane + e] = y2I[rd][e];
          dd[D + 128 * rd + 2 * lane + e] = y2Q[rd][e];
        }
    }

    // ---- overlap-save assemble (Process.cpp:498-522): v[0..3] = previous block, v[4..7] = new
    {
      cf *tb = reinterpret_cast<cf *>(lds);
#pragma unroll
      for (int rd = 0; rd < 2; ++rd)
        *reinterpret_cast<float4 *>(lds + 2 * (128 * rd + 2 * lane)) =
            make_float4(y2I[rd][0], y2Q[rd][0], y2I[rd][1], y2Q[rd][1]);
      wave_sync();
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 + j] = tb[lane + 64 * j];
      float2 *ov = reinterpret_cast<float2 *>(st + kStOverlap);
#pragma unroll
      for (int j = 0; j < 4; ++j) ov[64 * j + lane] = make_float2(v[4 + j].x, v[4 + j].y);
    }

    // ---- FFT, x mask, inverse FFT (Process.cpp:535-595).  The mask table is pre-scaled by 1/N.
    {
      cf tw1[7], tw2[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        tw1[q] = ltab[kTabTw1 + 64 * q + lane];
        tw2[q] = ltab[kTabTw2 + 64 * q + lane];
      }
      fft512<false>(v, tw1, tw2, lds, lane);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = cmul(v[r], ltab[kTabMask + 64 * r + lane]);
      fft512<true>(v, tw1, tw2, lds, lane);
    }

    // ---- AGC off: fixed gain on the valid half (DSP_Fn.cpp:494-502); SSB: audio = Re
    float aud[4];
    const float fixed_gain = fresh_coef(cf0)->sc[kScFixedGain];
#pragma unroll
    for (int j = 0; j < 4; ++j) aud[j] = fixed_gain * v[4 + j].x;
    if (DEBUG && a.dbg_demod) {
      float *dm = a.dbg_demod + ((size_t)ch * a.nframes + f) * D;
#pragma unroll
      for (int j = 0; j < 4; ++j) dm[lane + 64 * j] = aud[j];
    }

    // ---- interpolate by 2 (48 taps, phase length 24): inputs n = 4 lane .. 4 lane + 3
    // LDS buf: [0] pad, [1..23] history, [24 + i] new sample i
    wave_sync();
    {
      if (lane < 6) *reinterpret_cast<float4 *>(lds + 4 * lane) = hist1;
#pragma unroll
      for (int j = 0; j < 4; ++j) lds[24 + lane + 64 * j] = aud[j];
    }
    wave_sync();
    float u1[8];
    {
      float w[28];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const float4 t = lds4(lds + 4 * lane + 4 * i);
        w[4 * i] = t.x;
        w[4 * i + 1] = t.y;
        w[4 * i + 2] = t.z;
        w[4 * i + 3] = t.w;
      }
      if (lane < 6) *reinterpret_cast<float4 *>(st + kStInt1 + 4 * lane) = lds4(lds + 256 + 4 * lane);
      // arm_fir_interpolate_f32: out[2n + j - 1] = sum_t state[n + t] * c[(2 - j) + 2 t]
#pragma unroll
      for (int i = 0; i < 8; ++i) u1[i] = 0.0f;
#pragma unroll
      for (int b = 0; b < 24; b += 8) {
        float ci[16];
        load_taps<16>(ci, (CFloatPtr)cf0->int1 + 2 * b);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float x = w[u + b + t + 1];
            u1[2 * u] = fmaf(x, ci[1 + 2 * t], u1[2 * u]);
            u1[2 * u + 1] = fmaf(x, ci[2 * t], u1[2 * u + 1]);
          }
        }
      }
    }
    // ---- interpolate by 4 (32 taps, phase length 8): inputs n = 8 lane .. 8 lane + 7; the
    // 7-sample history is the neighbouring lane's tail (lane 0: last frame's, from HBM)
    {
      float w[15];
      const float out_scale = fresh_coef(cf0)->sc[kScOutScale];
      float c4[32];
      load_taps<32>(c4, (CFloatPtr)cf0->int2);
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const float up = lane_up1(u1[i + 1]);
        const float h = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hist2), i + 1));
        w[i] = (lane == 0) ? h : up;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) w[7 + i] = u1[i];
      if (lane == 63) {
        *reinterpret_cast<float4 *>(st + kStInt2) = make_float4(0.0f, u1[1], u1[2], u1[3]);
        *reinterpret_cast<float4 *>(st + kStInt2 + 4) = make_float4(u1[4], u1[5], u1[6], u1[7]);
      }
      // out[4n + j - 1] = sum_t state[n + t] * c[(4 - j) + 4 t],  state[n + t] = w[u + t]
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float x = w[u + t];
#pragma unroll
          for (int j = 1; j <= 4; ++j) o[j - 1] = fmaf(x, c4[(4 - j) + 4 * t], o[j - 1]);
        }
        // ---- volume (Process.cpp:929) and store
        *reinterpret_cast<float4 *>(gO + 32 * lane + 4 * u) =
            make_float4(o[0] * out_scale, o[1] * out_scale, o[2] * out_scale, o[3] * out_scale);
      }
    }
  }  // frames

  if (lane == 0) {
    ncs->phase = phase0;
    ncs->r = osc_r;
    st[kStMisc + kMiscDc] = dc_carry;
  }
}

// ------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------
template <int MODE>
static hipError_t launch512(const RxArgs &a, hipStream_t s, bool debug) {
  const int grid = (a.nchan + 3) / 4;
  // 40 KiB of dynamic LDS per workgroup pins residency at exactly 4 workgroups (16 waves)
  // per CU, so a 4096-channel batch is one full, balanced wave of work on 256 CUs.
  const size_t lds = 40960;
  static_assert((kLdsTabFloats + 4 * kLdsFloatsPerWave) * sizeof(float) <= 40960, "LDS slice too large");
  if (debug)
    hipLaunchKernelGGL((rx512_kernel<MODE, true>), dim3(grid), dim3(256), lds, s, a);
  else
    hipLaunchKernelGGL((rx512_kernel<MODE, false>), dim3(grid), dim3(256), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_rx(const RxArgs &a, int fft_length, int mode, hipStream_t s) {
  const bool debug = a.dbg_nco || a.dbg_dec || a.dbg_demod;
  if (fft_length != 512) return hipErrorInvalidValue;
  switch (mode) {
    case T41RX_DEMOD_USB:
    case T41RX_DEMOD_LSB:
      return launch512<0>(a, s, debug);
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace t41

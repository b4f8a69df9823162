// t41_sdr_amd/csrc/wave_fft.hpp -- device helpers shared by the gfx950 kernels of this library:
// packed-FP32 complex arithmetic, the register-resident 512-point FFT of one wavefront, DPP moves.
// (Product code; included by rx_kernels.hip and nr_kernels.hip.)
#pragma once
#include <hip/hip_runtime.h>

namespace t41 {

__device__ __forceinline__ void wave_sync() {
  // Orders this wave's LDS traffic (other lanes' writes -> my reads).  The LDS unit executes
  // one wave's instructions in issue order, so no s_waitcnt or workgroup barrier is needed:
  // only the COMPILER must not move memory operations across this point.
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);  // also pin ALU work: keeps register live ranges per phase
  asm volatile("" ::: "memory");
}


typedef float f2 __attribute__((ext_vector_type(2)));
typedef f2 cf;  // .x = re / I, .y = im / Q, one even-aligned VGPR pair

__device__ __forceinline__ f2 splat(float s) { return f2{s, s}; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// complex a*b: (ax bx - ay by, ax by + ay bx) in two packed instructions
__device__ __forceinline__ cf cmul(cf a, cf b) {
  cf t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}
// complex a*conj(b): (ax bx + ay by, ay bx - ax by)
__device__ __forceinline__ cf cmulc(cf a, cf b) {
  cf t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}
// same as cmul with b wave-uniform in an SGPR pair
__device__ __forceinline__ cf cmul_s(cf a, cf b_uniform) {
  cf t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "s"(b_uniform));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(t));
  return r;
}
// a + (-j) b = (ax + by, ay - bx)   and   a + j b = (ax - by, ay + bx)
__device__ __forceinline__ cf add_mj(cf a, cf b) {
  cf r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ cf add_pj(cf a, cf b) {
  cf r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// 8-point DFT in registers, natural order in and out.  INV selects e^{+j...}.  26 packed ops.
template <bool INV>
__device__ __forceinline__ void dft8(cf (&v)[8]) {
  constexpr float kR = 0.70710678118654752440f;
  const cf a0 = v[0] + v[4], a1 = v[0] - v[4];
  const cf a2 = v[2] + v[6], a3 = v[2] - v[6];
  const cf a4 = v[1] + v[5], a5 = v[1] - v[5];
  const cf a6 = v[3] + v[7], a7 = v[3] - v[7];
  const cf b0 = a0 + a2, b1 = a0 - a2;
  const cf b2 = a4 + a6, b3 = a4 - a6;
  // forward: W4 = -j, W8 = (1-j)/sqrt2, W8^3 = (-1-j)/sqrt2; inverse: conjugates
  const cf c0 = INV ? add_pj(a1, a3) : add_mj(a1, a3);
  const cf c1 = INV ? add_mj(a1, a3) : add_pj(a1, a3);
  const cf d0 = INV ? add_pj(a5, a7) : add_mj(a5, a7);
  const cf d1 = INV ? add_mj(a5, a7) : add_pj(a5, a7);
  // W8 d0 = kR (d0 + (-j) d0) fwd / kR (d0 + j d0) inv;  W8^3 d1 = -kR (d1 + j d1) fwd / -kR (d1 + (-j) d1) inv
  const cf t0 = INV ? add_pj(d0, d0) : add_mj(d0, d0);
  const cf t1 = INV ? add_mj(d1, d1) : add_pj(d1, d1);
  v[0] = b0 + b2;
  v[4] = b0 - b2;
  v[2] = INV ? add_pj(b1, b3) : add_mj(b1, b3);
  v[6] = INV ? add_mj(b1, b3) : add_pj(b1, b3);
  v[1] = pk_fma(t0, splat(kR), c0);
  v[5] = pk_fma(t0, splat(-kR), c0);
  v[3] = pk_fma(t1, splat(-kR), c1);
  v[7] = pk_fma(t1, splat(kR), c1);
}

// LDS exchange buffer row stride for the FFT transposes (in complex elements): 64 + 8 keeps
// both the row-major writes and the 8-strided reads bank-conflict-free for ds_*_b64.
constexpr int kFftRow = 72;
// Second exchange: element (q, l1) of a row sits at q + 8 l1 + (l1 & 6).  The extra term spreads
// the ds_write_b64 of 16 consecutive lanes (q in {2g, 2g+1}, l1 = 0..7) over all 16 8-byte bank
// pairs (plain 8 l1 is 4-way conflicted: 8 l1 mod 16 has two values); rows stay disjoint (max 69).
__device__ __forceinline__ constexpr int fft_x2(int l1) { return 8 * l1 + (l1 & 6); }

// The first exchange of the 512-point transform -- register index <-> lane bits 3..5, three independent bit
// swaps -- runs inside the VALU instead of through LDS: v_permlane32_swap (bit 5), v_permlane16_swap (bit 4), and
// for bit 3 a pair of bank-masked DPP row shifts by 8.  40 VALU instructions per transform in place of 8
// ds_write_b64 + 8 ds_read_b64 (+ 3 for addresses) and two LDS ordering points.  Measured on MI355X in the fused
// FFT_LENGTH 512 kernel (interleaved rounds, tools/ablation_table.py, profiles/r03_ablation_ssb.md): 0.9 % faster
// than the LDS form (-DT41RX_FFT_X1_PERM=0) -- the kernel is co-limited by VALU and LDS, and the LDS relief wins.
// The second exchange moves lane bits 0..2 AND swaps the lane's two octal digits; the same trade costs ~85 VALU
// instructions there (no swap instruction below 16 lanes) and stays in LDS.
#ifndef T41RX_FFT_X1_PERM
#define T41RX_FFT_X1_PERM 1
#endif
__device__ __forceinline__ void lane_swap32(float &a, float &b) {  // lanes 32..63 of a <-> lanes 0..31 of b
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void lane_swap16(float &a, float &b) {  // odd 16-lane rows of a <-> even rows of b
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void lane_swap8(float &a, float &b) {  // lanes 8..15 of every row of a <-> lanes 0..7 of b
  const float t = b;
  b = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(b), __float_as_int(a), 0x108 /* row_shl:8 */, 0xf, 0x3, false));
  a = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(a), __float_as_int(t), 0x118 /* row_shr:8 */, 0xf, 0xc, false));
}
// (reg q, lane l1 + 8 k2) -> (reg k2, lane l1 + 8 q)
__device__ __forceinline__ void fft_exchange1_perm(cf (&v)[8]) {
  float re[8], im[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    re[q] = v[q].x;
    im[q] = v[q].y;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    lane_swap32(re[q], re[q + 4]);
    lane_swap32(im[q], im[q + 4]);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (q & 2) continue;
    lane_swap16(re[q], re[q + 2]);
    lane_swap16(im[q], im[q + 2]);
  }
#pragma unroll
  for (int q = 0; q < 8; q += 2) {
    lane_swap8(re[q], re[q + 1]);
    lane_swap8(im[q], im[q + 1]);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = cf{re[q], im[q]};
}


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
#if T41RX_FFT_X1_PERM
  fft_exchange1_perm(v);
#else
  wave_sync();
#pragma unroll
  for (int q = 0; q < 8; ++q) xb[q * kFftRow + lane] = v[q];
  wave_sync();
  {
    const int l1 = lane & 7, q = lane >> 3;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) v[k2] = xb[q * kFftRow + l1 + 8 * k2];
  }
#endif
  dft8<INV>(v);
#pragma unroll
  for (int q = 1; q < 8; ++q) v[q] = INV ? cmulc(v[q], tw2[q - 1]) : cmul(v[q], tw2[q - 1]);
  // exchange 2: (reg q2, lane l1 + 8 q) -> (reg l1, lane q + 8 q2)
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


// GFX9 DPP controls: data moves between lanes inside the VALU, no LDS round trip
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppWaveShr1 = 0x138, kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_f(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf, BOUND));
}
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ f2 dpp_f2(f2 src) {
  return f2{dpp_f<CTRL, ROW_MASK, BOUND>(0.0f, src.x), dpp_f<CTRL, ROW_MASK, BOUND>(0.0f, src.y)};
}
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ double dpp_d(double x) {
  const long long u = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)u, CTRL, ROW_MASK, 0xf, BOUND);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, ROW_MASK, 0xf, BOUND);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// value of lane-1 (lane 0 gets 0)
__device__ __forceinline__ float lane_up1(float v) { return dpp_f<kDppWaveShr1, 0xf, true>(0.0f, v); }


}  // namespace t41

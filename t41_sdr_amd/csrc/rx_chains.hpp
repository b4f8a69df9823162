// t41_sdr_amd/csrc/rx_chains.hpp -- the serial stages of the receive path as wave programs: the look-ahead AGC's gain law (DSP_Fn.cpp:504-631; barrier and pipelined forms) and the synchronous detector's PLL (Demod.cpp:40-139).
#pragma once
#include "rx_device.hpp"

namespace t41 {

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
// (T41RX_AGC_X: timing experiments with wrong results, rx_experiments.hpp)
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
#if T41RX_AGC_R04CHECK  // (experiment build: round 4's unsound short-cut, to show that the tests and the soak catch it)
      ok &= lanes_gt(vin[3], d.thr);
#else
      ok &= lanes_gt(vin[0], d.thr) & lanes_gt(vin[3], d.thr);
#endif
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

// max of the 23 group maxima p[0 .. 22] (the look-ahead window's whole groups).  As a loop `C = fmaxf(C, p[q])` hipcc
// requests them one pair at a time -- ds_read2_b32, wait, maximum, next: twelve LDS round trips in a row, each with a VALU
// addition for an address that does not fit ds_read2's offset field (round 5, ISA of the AGC kernels).  Here the base is
// opaque (so the offsets are small), the requests go out in two batches and the maxima are taken three at a time.
// (fmaxf: the result does not depend on the order.)
__device__ __forceinline__ float agc_group_max(const float *p) {
#if T41RX_AGC_GROUPMAX
  typedef const __attribute__((address_space(3))) float *LdsFloats;  // (kept an LDS pointer: a laundered generic one is read by flat loads)
  LdsFloats lp = (LdsFloats)p;
  asm volatile("" : "+v"(lp));
  float x[12], y[11];
#pragma unroll
  for (int q = 0; q < 12; ++q) x[q] = lp[q];
  asm volatile("" ::: "memory");
#pragma unroll
  for (int q = 0; q < 11; ++q) y[q] = lp[12 + q];
  asm volatile("" ::: "memory");
  float m = fmaxf(fmaxf(x[0], x[1]), x[2]);
#pragma unroll
  for (int q = 3; q + 1 < 12; q += 2) m = fmaxf(fmaxf(m, x[q]), x[q + 1]);
  m = fmaxf(m, x[11]);
#pragma unroll
  for (int q = 0; q + 1 < 11; q += 2) m = fmaxf(fmaxf(m, y[q]), y[q + 1]);
  return fmaxf(m, y[10]);
#else
  float C = p[0];
#pragma unroll
  for (int q = 1; q <= 22; ++q) C = fmaxf(C, p[q]);
  return C;
#endif
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
    const float C = agc_group_max(lds + kAgG + lane + 2);
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
// Round 5, measured and left off (T41RX_PIPE_KEEP_RE = 1 builds it; SSB / NFM, AM needs the imaginary parts too): the popped
// samples' real parts -- written by this lane at the preparation of frame f and read back by the SAME lane at the gain of
// frame f, two iterations later -- waiting in registers of their owner (a ring of three float4 in the kernel) instead of
// in the slot, which then holds the chain's operands only: 6 instead of 9 KiB touched per channel, 3 MiB per XCD instead of
// 4.6 -- VERDICT r04's remedy for the slots' fabric traffic.  Result (interleaved A/B, profiles/r05_ab_keepre.txt): the
// kernel spills 18 instead of 10 VGPRs (up to 40 in the q15 / general variants), runs 29.2 against 29.0 us per frame, and
// the fabric traffic falls from 1.306 to 1.288 x only: the slots do not stay in L2 next to 3 GB of streamed samples
// whatever their size (nontemporal accesses keep their lines in the XCD's L2 too), so a smaller cyclic set buys nothing.
#ifndef T41RX_PIPE_KEEP_RE
#define T41RX_PIPE_KEEP_RE 0
#endif
template <typename AL, bool NEED_IM, bool KEEP_RE = false>
// arec: the delay line's magnitudes (computed a frame ago as that frame's newest: carried, not recomputed)
__device__ __forceinline__ float4 agc_prep_pipe(const cf (&v)[8], float4 agst, float2 &arec, float *lds, float *slot, CoefPtr cf0, int lane,
                                               float4 *re_out = nullptr) {
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
    const float C = agc_group_max(lds + kAgG + lane + 2);
    const float s3 = g1.w, s2 = fmaxf(g1.z, s3), s1 = fmaxf(g1.y, s2), s0 = fmaxf(g1.x, s1);
    const float p0 = nv.x, p1 = fmaxf(p0, nv.y), p2 = fmaxf(p1, nv.z), p3 = fmaxf(p2, nv.w);
    *reinterpret_cast<float4 *>(slot + 4 * lane) =
        make_float4(fmaxf(fmaxf(s0, C), p0), fmaxf(fmaxf(s1, C), p1), fmaxf(fmaxf(s2, C), p2), fmaxf(fmaxf(s3, C), p3));
    *reinterpret_cast<float4 *>(slot + 256 + 4 * lane) = make_float4(ao0, g1.x, g1.y, g1.z);
    cf z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = *reinterpret_cast<const cf *>(lds + kAgZ + 2 * (3 + 4 * lane + k));
    if (KEEP_RE) *re_out = make_float4(z[0].x, z[1].x, z[2].x, z[3].x);
    else *reinterpret_cast<float4 *>(slot + 512 + 4 * lane) = make_float4(z[0].x, z[1].x, z[2].x, z[3].x);
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
template <bool NEED_IM, bool KEEP_RE = false>
__device__ __forceinline__ AgcGainIn agc_gain_request(const float *slot, int lane, float4 re_kept = make_float4(0, 0, 0, 0)) {
  AgcGainIn r;
  r.vv = *reinterpret_cast<const float4 *>(slot + 4 * lane);
  r.zr = KEEP_RE ? re_kept : *reinterpret_cast<const float4 *>(slot + 512 + 4 * lane);
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


}  // namespace t41

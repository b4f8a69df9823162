// t41_sdr_amd/csrc/rx512_kernel.hpp -- the fused FFT_LENGTH 512 kernel rx512_kernel<MODE, DEBUG, PART, PLAIN, AGC, WQ15, SEGPAR, PIPE> (ProcessIQData(), Process.cpp:70-944, one wave per channel) and its LDS geometry.
#pragma once
#include "rx_chains.hpp"

namespace t41 {

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
  constexpr bool kKeepPl = T41RX_PHASE_SPLIT && T41RX_KEEP_PL && KEEP && !AGC && (MODE != kModeSam || T41RX_SAM_KEEP_PL);
  uint64_t pl_keep = 0;  // (8 lane + 1) dphi, the lane's part of the oscillator phases of a frame
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
  // PIPE, SSB / NFM (T41RX_PIPE_KEEP_RE): the popped samples' real parts of the three frames between preparation and gain
  float4 zre0 = make_float4(0, 0, 0, 0), zre1 = zre0, zre2 = zre0;
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
      load_taps<16>(ci, (CFloatPtr)cf0, kCoInt1 + 2 * b);
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
    if (AGC) PRIO(2); else if (T41RX_PRIO_AGE == 2 && KEEP) { if (wv < 8) PRIO(2); else PRIO(3); } else PRIO(3);  // (AGC on: 3 is the serial chain's, see agc_apply)
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
          const LaneOff lof = fresh_off(8 * lane);
          pI0[1] = ldg_stream(gI - 512, lof);
          pI1[1] = ldg_stream(gI - 512, lof, 4);
          pQ0[1] = ldg_stream(gQ - 512, lof);
          pQ1[1] = ldg_stream(gQ - 512, lof, 4);
        } else {
          const LaneOff lof = fresh_off(4 * lane);
          pI0[1] = ldg_stream(gI - 256, lof);
          pQ0[1] = ldg_stream(gQ - 256, lof);
        }
      }
      float4 tailI;
      if (KEEP && carried) {
        tailI = WQ15 ? make_float4(q15_lo(tailNq.x), q15_hi(tailNq.x), q15_lo(tailNq.y), q15_hi(tailNq.y)) : tailN;
      } else if (!WQ15) {
        if (!carried) {
          const LaneOff lof = fresh_off(8 * lane);
          pI0[0] = ldg_stream(gI, lof);
          pI1[0] = ldg_stream(gI, lof, 4);
          pQ0[0] = ldg_stream(gQ, lof);
          pQ1[0] = ldg_stream(gQ, lof, 4);
        }
        tailI = ldg4(gI + (L - 256), fresh_off(4 * lane));
      } else {  // 8 samples = 16 bytes per lane and array
        if (!carried) {
          const LaneOff lof = fresh_off(4 * lane);
          pI0[0] = ldg_stream(gI, lof);
          pQ0[0] = ldg_stream(gQ, lof);
        }
        const float2 t = *reinterpret_cast<const float2 *>(gI + (L - 256) / 2 + 2 * lane);
        tailI = make_float4(q15_lo(t.x), q15_hi(t.x), q15_lo(t.y), q15_hi(t.y));
      }
      // (loaded and, behind the table staging, stored under the same conditions: no start values needed -- as zeros they
      // are written on every frame, 20 v_mov_b32 hoisted above the branch)
      float4 h1 = any_float4(), h2 = any_float4();
      float4 ovl0 = any_float4(), ovl1 = any_float4(), ovl2 = any_float4();  // KEEP, first frame: overlap block, x2 history
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
          const LaneOff lof = fresh_off(8 * lane);
          pI0[1] = ldg_stream(gI + 512, lof);
          pI1[1] = ldg_stream(gI + 512, lof, 4);
          pQ0[1] = ldg_stream(gQ + 512, lof);
          pQ1[1] = ldg_stream(gQ + 512, lof, 4);
        }
      } else if (!carried1) {
        const LaneOff lof = fresh_off(4 * lane);
        pI0[1] = ldg_stream(gI + 256, lof);
        pQ0[1] = ldg_stream(gQ + 256, lof);
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
        if (kKeepPl) {
          unsigned n1 = (unsigned)(8 * lane + 1);
          asm volatile("" : "+v"(n1));  // (or the product is if-converted out of this block and worked out on every frame again)
          pl_keep = (uint64_t)n1 * dphi;
        }
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
        dc_highpass<8, !AGC && (MODE != kModeSam || T41RX_SAM_SCAN_FUSED)>(z, dcs, lane, hp8.x, hp8.y);  // from rest: 512 samples on, its memory of the start is a1^512
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
        fir_pair<kDec1Taps, 1, 5, 18, 6>(xw, pidx, cf0, kCoDec1, o1[0], o1[1]);
        // its first outputs saw no history and are dropped; the last 48 (lanes 40..63) are the /2 history
        float4 hh = any_float4();
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
          const LaneOff lof = fresh_off(8 * lane);
          pI0[1] = ldg_stream(gI + 512, lof);
          pI1[1] = ldg_stream(gI + 512, lof, 4);
          pQ0[1] = ldg_stream(gQ + 512, lof);
          pQ1[1] = ldg_stream(gQ + 512, lof, 4);
        } else {
          const LaneOff lof = fresh_off(4 * lane);
          pI0[1] = ldg_stream(gI + 256, lof);
          pQ0[1] = ldg_stream(gQ + 256, lof);
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
            tailF = ldg4(gI + (seg * L - 256), fresh_off(4 * lane));
          } else {
            const float2 t = *reinterpret_cast<const float2 *>(gI + (seg * L - 256) / 2 + 2 * lane);
            tailF = make_float4(q15_lo(t.x), q15_hi(t.x), q15_lo(t.y), q15_hi(t.y));
          }
        }
        const float x[4] = {tailF.x * g_rf, tailF.y * g_rf, tailF.z * g_rf, tailF.w * g_rf};
        dc2 = f2{(g_rf_i != g_rf) ? -dc_carry : dc_carry, dc_highpass_end_state<4, !AGC && (MODE != kModeSam || T41RX_SAM_SCAN_FUSED)>(x, hp4.x, hp4.y)};
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
      // (the sample number is widened UNSIGNED -- as an int it is sign-extended, lane's range being unknown behind
      // FRESH_LANE, and the 64-bit product costs two more VALU instructions -- and only the lane's part is a vector
      // product: the sub-block's part is the same for every lane, scalar arithmetic.  Bits 24..55 of each phase are
      // kept for the sub-block that needs them instead of being worked out again there.)
      float2 osc_tab[4];
      uint32_t osc_u[4];
      {
        // (the lane's part does not change from frame to frame: where registers are to spare it is worked out once --
        // two 64-bit multiply-adds and two 32-bit multiplies, quarter-rate instructions, per frame otherwise)
        const uint64_t Pl = kKeepPl ? pl_keep : (uint64_t)(unsigned)(8 * lane + 1) * dphi;
  #pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
  #if T41RX_PHASE_SPLIT
          const uint64_t P = (phase0 + (uint64_t)(512 * sb) * dphi) + Pl;
  #else
          const uint64_t P = phase0 + (uint64_t)(512 * sb + 8 * lane + 1) * dphi;
  #endif
          osc_u[sb] = (uint32_t)(P >> 24);
          osc_tab[sb] = (PART == 1) ? reinterpret_cast<const float2 *>(smem)[(int)(P >> 56)] : ldg2(tab + kTabSinCos, (unsigned)(P >> 56));
        }
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
              const int o = 512 * (s + 2);
              const LaneOff lof = fresh_off(8 * lane);
              pI0[h] = ldg_stream(gI + o, lof);
              pI1[h] = ldg_stream(gI + o, lof, 4);
              pQ0[h] = ldg_stream(gQ + o, lof);
              pQ1[h] = ldg_stream(gQ + o, lof, 4);
            } else {
              const int o = 256 * (s + 2);
              const LaneOff lof = fresh_off(4 * lane);
              pI0[h] = ldg_stream(gI + o, lof);
              pQ0[h] = ldg_stream(gQ + o, lof);
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
                  tailN = ldg4(nI + (L - 256), fresh_off(4 * lane));
                } else {  // (raw q15 words; converted when used, not here: that would wait for them)
                  tailNq = *reinterpret_cast<const float2 *>(nI + (L - 256) / 2 + 2 * lane);
                }
              }
              if (KEEP && kPF < 2 && s == 3) {
                // sub-block 1 is requested at the top of the next frame
              } else if (!WQ15) {
                const int o = 512 * (s - 2);
                const LaneOff lof = fresh_off(8 * lane);
                pI0[h] = ldg_stream(nI + o, lof);
                pI1[h] = ldg_stream(nI + o, lof, 4);
                pQ0[h] = ldg_stream(nQ + o, lof);
                pQ1[h] = ldg_stream(nQ + o, lof, 4);
              } else {
                const int o = 256 * (s - 2);
                const LaneOff lof = fresh_off(4 * lane);
                pI0[h] = ldg_stream(nI + o, lof);
                pQ0[h] = ldg_stream(nQ + o, lof);
              }
            }
          } else if (s == 3 && !KEEP) {  // last sub-block: prefetch the overlap-save "previous" block instead
            const cf *ov = reinterpret_cast<const cf *>(st + kStOverlap);
  #pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ov[64 * j + lane];
          }
          STAMP(s == 0 ? 14 : 0);  // wait for the sub-block's global loads + gain/interleave (14: first sub-block)
        // -- DC high-pass (127-128), band gain (133-134) / IQ amplitude (166)
          if (!T41RX_CUT(6)) dc_highpass<8, !AGC && (MODE != kModeSam || T41RX_SAM_SCAN_FUSED)>(z, dc2, lane, hp8.x, hp8.y);
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
  #if T41RX_PHASE_SPLIT
            const uint32_t u = osc_u[s];
  #else
            const uint64_t P = phase0 + (uint64_t)(n0 + 1) * dphi;
            const uint32_t u = (uint32_t)(P >> 24);
  #endif
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
              fir_pair<kDec1Taps, 1, 5, 18, 6>(xw, pidx, cf0, kCoDec1, o1[0], o1[1], z);
            } else if (!T41RX_CUT(4)) {
              // (round 4, measured and dropped: the window requested one group ahead of its use behind scheduling
              //  barriers, taps in 16-tap scalar loads -- 18 spilled registers, 26.7 against 22.4 us per frame)
              // (round 5: window entries 28.. are the lane's own new samples, still in z: three of the 17 reads come from
              // there -- in the kernels without the AGC (with it: neutral for SSB, 1 % slower behind SAM; SAM alone gains 2.8 %
              // although it spills two registers for it, profiles/r05_ab_regtail_agc.txt))
              if (T41RX_DEC1_REGTAIL && (!AGC || T41RX_DEC1_REGTAIL_AGC)) fir_pair<kDec1Taps, 1, 5, 18, 6, 14>(xw, pidx, cf0, kCoDec1, o1[0], o1[1], nullptr, z);
              else fir_pair<kDec1Taps, 1, 5, 18, 6>(xw, pidx, cf0, kCoDec1, o1[0], o1[1]);
            } else {
              o1[0] = *reinterpret_cast<cf *>(xw);
              o1[1] = *reinterpret_cast<cf *>(xw + 8);
            }
          }
          STAMP(3);  // LDS staging + /4 decimator
        // -- roll the /4 history (logical 512..539 -> 0..27) and append the /4 outputs
          {
            float4 hh = any_float4();
            if (lane < 14) hh = lds4(lds + kX + 2 * xpad(512 + 2 * lane));
            wave_sync();
            if (lane < 14) *reinterpret_cast<float4 *>(lds + kX + 2 * xpad(2 * lane)) = hh;
            *reinterpret_cast<float4 *>(lds + kY1 + y1slot(24 + 64 * h + lane)) =
                make_float4(o1[0].x, o1[0].y, o1[1].x, o1[1].y);
          }
          if (T41RX_LOO == 13) { o1x[0] = o1[0]; o1x[1] = o1[1]; }
        }  // h
        STAMP(4);  // history roll
        if (rd == 1) {
          if (AGC) PRIO(1);
          else if (T41RX_PRIO_AGE == 1 && KEEP) { if (wv < 8) PRIO(2); else PRIO(3); }
          else if (T41RX_PRIO_AGE == 2 && KEEP) { if (wv < 8) PRIO(1); else PRIO(2); }
          else PRIO(2);
        }
      // ---- decimate by 2 (46 taps) over the 256 new /4 samples: m = 2*lane, 2*lane+1
        wave_sync();
        // y[m] = sum_i c[i] * state[2m + i]; state[i] = buf[i + 3]
        {
          // window-relative complex offset o (even) -> offset in the planes, relative to lds + kY1 + 4 lane
          auto planes = [](int o) { return y1slot(o >> 1) / 2; };
          if (T41RX_LOO == 13) {
            const cf src[8] = {y2[0][0], y2[0][1], o1x[0], o1x[1], y2[0][1], o1x[1], o1x[0], y2[0][0]};
            fir_pair<kDec2Taps, 3, 5, 26, 6>(lds + kY1 + 4 * lane, planes, cf0, kCoDec2, y2[rd][0], y2[rd][1], src);
          } else if (!T41RX_CUT(3)) {
            fir_pair<kDec2Taps, 3, 5, 26, 6>(lds + kY1 + 4 * lane, planes, cf0, kCoDec2, y2[rd][0], y2[rd][1]);
          } else {
            y2[rd][0] = *reinterpret_cast<cf *>(lds + kY1 + 4 * lane);
            y2[rd][1] = *reinterpret_cast<cf *>(lds + kY1 + 4 * lane + 2);
          }
        }
        STAMP(5);  // /2 decimator
      {  // roll the /2 history: logical 256..303 -> 0..47
          float4 hh = any_float4();
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
          const LaneOff lof = fresh_off(8 * lane);
          rI0 = ldg_stream(gI + last + 1536, lof);
          rI1 = ldg_stream(gI + last + 1536, lof, 4);
          rQ0 = ldg_stream(gQ + last + 1536, lof);
          rQ1 = ldg_stream(gQ + last + 1536, lof, 4);
        } else {
          const LaneOff lof = fresh_off(4 * lane);
          rI0 = rI1 = ldg_stream(gI + last / 2 + 768, lof);
          rQ0 = rQ1 = ldg_stream(gQ + last / 2 + 768, lof);
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
            // taps in ascending order, separate multiply and add, as the reference's loop.  Round 5: the two outputs of a
            // lane ride in ONE register pair -- step k multiplies in[m0 + k] by (taps[k], taps[k - 1]), a v_pk_mul_f32 and a
            // v_pk_add_f32 where there were two multiplies and two adds (same operations on the same values in the same
            // order; a tap that does not exist is a zero, whose product leaves its sum as it is): 656 -> 328 VALU
            // instructions per frame
            f2 acc = splat(0.0f);  // (acc0, acc1)
            float tp[96];
  #pragma unroll
            for (int c = 0; c < 96; c += 16) {
              float chunk[16];
              load_taps<16>(chunk, (CFloatPtr)cf0, kCoDeemph + c);
  #pragma unroll
              for (int k = 0; k < 16; ++k) tp[c + k] = chunk[k];
  #pragma unroll
              for (int jj = c / 2; jj < c / 2 + 8; ++jj) {
                if (2 * jj > kDeemphTaps) continue;  // pairs (in[m0 + 2 jj], in[m0 + 2 jj + 1]), jj = 0..40
                const float2 w = *reinterpret_cast<const float2 *>(ds + m0 + 2 * jj);
                static_assert(kDeemphTaps == 81, "2 jj <= 80 below: tap 2 jj always exists");
                const float t_m1 = (jj > 0) ? tp[2 * jj - 1] : 0.0f;
                const float t_p1 = (2 * jj + 1 < kDeemphTaps) ? tp[2 * jj + 1] : 0.0f;
                f2 pr = f2{tp[2 * jj], t_m1} * splat(w.x);
                acc = acc + pr;
                pr = f2{t_p1, tp[2 * jj]} * splat(w.y);
                acc = acc + pr;
              }
            }
            au[rd][0] = (m0 < D - kDeemphTaps) ? acc.x : y2[rd][0].y;
            au[rd][1] = (m0 + 1 < D - kDeemphTaps) ? acc.y : y2[rd][1].y;
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
        constexpr bool KEEP_RE = T41RX_PIPE_KEEP_RE && !NEED_IM && !PSAM;  // re(f - 2) = zre2 here, re(f) -> zre0 below
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
          else gin = agc_gain_request<NEED_IM, KEEP_RE>(bslot, lane, zre2);
        }
        if (f < seg1) {  // this frame's chain operands and popped samples -> the channel's slot
          float *pslot = a.agc_pipe + ((size_t)ch * kPipeSlots + f % kPipeSlots) * kPipeSlotFloats;
          if (PSAM) sam_prep_pipe(v, fixed_gain, pslot, lane);
          else {
            if (first_iter) agmag = make_float2(agc_mag(cf{agrec.x, agrec.y}), agc_mag(cf{agrec.z, agrec.w}));  // (later frames: carried)
            agrec = agc_prep_pipe<AgcLds<true>, NEED_IM, KEEP_RE>(v, agrec, agmag, lds, pslot, cf0, lane, &zre0);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) __hip_atomic_fetch_add(flags + f % kPipeSlots, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef T41RX_PIPE_STAT
          if (lane == 0) pipe_stat[9] += __builtin_readcyclecounter() - ps_t;
#endif
        }
        const float4 re_gain = zre2;  // the gain below works on frame f - 2 ...
        if (KEEP_RE) {                // ... and the ring moves on for the next iteration (every iteration, also the first two)
          zre2 = zre1;
          zre1 = zre0;
        }
        (void)re_gain;
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
          else gin = agc_gain_request<NEED_IM, KEEP_RE>(bslot, lane, re_gain);
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
    if (T41RX_PRIO_AGE == 1 && KEEP && !AGC) { if (wv < 8) PRIO(0); else PRIO(1); } else PRIO(0);
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
      // (w[0] is padding nobody reads; told so, hipcc narrows the seven 16-byte reads to the 27 floats behind it:
      // 13 ds_read2_b32 + 1 ds_read_b32 at 4-byte alignment, each with a VALU add for an address that no longer fits
      // the offset field.  Kept alive, the reads stay ds_read_b128 at immediate offsets.)
      asm volatile("" ::"v"(w[0]));
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
        load_taps<16>(ci, (CFloatPtr)cf0, kCoInt1 + 2 * b);
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
      load_taps<32>(c4, (CFloatPtr)cf0, kCoInt2);  // (pre-multiplied by the volume factor)
      const float x1[8] = {u1[0].x, u1[0].y, u1[1].x, u1[1].y, u1[2].x, u1[2].y, u1[3].x, u1[3].y};
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const float up = lane_up1(x1[i + 1]);
        const float hs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hist2), i + 1));
  #if T41RX_WRITELANE  // (one v_writelane_b32 instead of the move + select `lane == 0 ? hs : up` compiles to)
        w[i] = write_lane<0>(up, hs);
  #else
        w[i] = (lane == 0) ? hs : up;
  #endif
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
  #if T41RX_WRITELANE  // (instead of move + compare + select per entry)
  #define T41RX_HIST2_ENTRY(i) hist2c = write_lane<i>(hist2c, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x1[i]), 63)));
        T41RX_HIST2_ENTRY(1) T41RX_HIST2_ENTRY(2) T41RX_HIST2_ENTRY(3) T41RX_HIST2_ENTRY(4) T41RX_HIST2_ENTRY(5) T41RX_HIST2_ENTRY(6) T41RX_HIST2_ENTRY(7)
  #undef T41RX_HIST2_ENTRY
  #else
#pragma unroll
        for (int i = 1; i < 8; ++i) {
          const float t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x1[i]), 63));
          hist2c = (lane == i) ? t : hist2c;
        }
  #endif
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
      float4 park_h = any_float4(), park_o = any_float4();
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
              stg_stream(gO + 16 * (u >> 2), fresh_off(32 * row + 4 * (lane & 3)), t);
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
        const LaneOff lof = fresh_off(4 * lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = 8 * i + (lane >> 3);
          const float4 t = lds4(tr + 4 * (8 * row + ((lane & 7) ^ (row & 7))));
          stg_stream(gO + 256 * i, lof, t);
        }
      } else if (WQ15) {  // 4 pieces per row: piece F = 64 i + lane is row F >> 2, column lane & 3
        const LaneOff lof = fresh_off(4 * lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 16 * i + (lane >> 2);
          const float4 t = lds4(tr + 4 * (4 * row + ((lane & 3) ^ ((row >> 1) & 3))));
          stg_stream(gO + 256 * i, lof, t);
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

}  // namespace t41

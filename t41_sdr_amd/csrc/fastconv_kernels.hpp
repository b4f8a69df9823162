// t41_sdr_amd/csrc/fastconv_kernels.hpp -- FFT_LENGTH 1024 / 2048 / 4096 (BASELINE config 4): the N-point overlap-save fast convolution, fastconv_kernel<R, CPLX, BACK> and the one-kernel form fastconv_fused_kernel<PLAIN>.
#pragma once
#include "rx_device.hpp"

namespace t41 {

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
// (T41RX_FCABL: timing experiments with wrong results, rx_experiments.hpp)
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
    // (two scalar bases, one per h, and R - 1 lane offsets shared by both: a base per entry runs the kernel out of SGPRs)
    const LaneOff lo = fresh_off(2 * lane);
#pragma unroll
    for (int q = 1; q < R; ++q) {
      const LaneOff oq = LaneOff{lo.bytes + 4096u * (q - 1)};
#pragma unroll
      for (int h = 0; h < H; ++h) twp[h][q - 1] = ldg_cf(twN + 64 * (wv + NWV * h), oq);
    }
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
    const LaneOff lom = fresh_off(2 * lane);
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int q = wv + NWV * h;
      if (q < R) {
#pragma unroll
        for (int r = 0; r < 8; ++r) mk[h][r] = ldg_cf(maskN + 512 * q, lom, 64 * r);
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
      load_taps<48>(ci, (CFloatPtr)cf0, kCoInt1);
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
          asm volatile("" ::"v"(w[0]));  // (rx512_kernel.hpp: or the 16-byte reads are narrowed to ds_read2_b32 + address adds)
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
      load_taps<32>(c4, (CFloatPtr)cf0, kCoInt2);
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
          const LaneOff lof = fresh_off(4 * lane);
#pragma unroll
          for (int i = 0; i < 8; ++i) {  // float4 F = 64 i + lane: row F >> 3 = the source lane, column lane & 7
            const int row = 8 * i + (lane >> 3);
            const float4 t = lds4(tr + 4 * (8 * row + ((lane & 7) ^ (row & 7))));
            if (!(T41RX_FCABL & 1) || t.x == 123.456f) stg_stream(gO + 256 * i, lof, t);
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
// Round 5: with the tap pointers laundered at their base and the wave-index offsets re-derived per phase (FRESH_WV) the
// kernel spills 40 SGPRs instead of 85, and the balance tips: the previous block IN REGISTERS (12 VGPRs spilled, 36 B of
// scratch per lane that stay in L2) measures 51.4-51.6 against 53.1-53.3 us per frame interleaved on one box, and the fabric
// traffic falls from 1.153 to 1.077 x the algorithmic bytes (the 16 KiB round trip through the record is gone):
// profiles/r05_ab_prevreg.txt.  (Round 3, 85 spilled SGPRs, two sub-blocks in flight: 56.4 against 54.2 the other way.)
// T41RX_FF_REGTAIL=1: the /4 decimator's window tail from registers here too (rx512_kernel.hpp; measured: 49.78 against
// 49.68 us per frame and one more spilled register, HBM traffic 1.075 -> 1.082 x: off)
#ifndef T41RX_FF_REGTAIL
#define T41RX_FF_REGTAIL 0
#endif
#ifndef T41RX_FF_PREV_GLOBAL
#define T41RX_FF_PREV_GLOBAL 0  // 1: the "previous" block waits in the channel's record (L2) instead of 16 registers per lane
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
  int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (re-defined per phase like `lane`, see FRESH_WV)
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
    // (two scalar bases, one per h, and R - 1 lane offsets shared by both: a base per entry runs the kernel out of SGPRs)
    const LaneOff lo = fresh_off(2 * lane);
#pragma unroll
    for (int q = 1; q < R; ++q) {
      const LaneOff oq = LaneOff{lo.bytes + 4096u * (q - 1)};
#pragma unroll
      for (int h = 0; h < H; ++h) twp[h][q - 1] = ldg_cf(twN + 64 * (wv + NWV * h), oq);
    }
  };
  const float *gIc = a.I + (size_t)ch * a.nframes * L;  // (a.nframes counts 2048-sample segments)
  const float *gQc = a.Q + (size_t)ch * a.nframes * L;

  for (int f = 0; f < a.nframes4k; ++f) {
    FRESH_LANE(); FRESH_WV(wv);
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
        const LaneOff lof = fresh_off(8 * lane);
        pI0[set] = ldg_stream(pi, lof);
        pI1[set] = ldg_stream(pi, lof, 4);
        pQ0[set] = ldg_stream(pq, lof);
        pQ1[set] = ldg_stream(pq, lof, 4);
      };
      float dc_carry = 0.0f;
      if (wv == 0) {
        // the frame's first segment: memories from the channel's record (the call's first frame: as the last call
        // left them) or from the hand-over slot where wave 3 left them at the end of the previous frame (two slots
        // in the `mid` scratch, alternating, so that wave 3 of THIS frame never writes what this wave still reads;
        // its stores were drained before the barrier above, and these loads do not look at this CU's L1)
        const float *src = (f == 0) ? st : a.mid + ((size_t)ch * 2 + ((f - 1) & 1)) * 256;
        request(0, gI, gQ);
        float4 h1 = any_float4(), h2 = any_float4();
        const LaneOff lo4 = fresh_off(4 * lane);
        if (lane < 14) h1 = ldg_stream(src + kStDec1, lo4);
        if (lane < 24) h2 = ldg_stream(src + kStDec2, lo4);
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
          mix(z, P, ldg2(tab + kTabSinCos, (unsigned)(P >> 56)));
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
        float4 hh = any_float4();
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
          const float4 tailF = ldg4(gI + (R * L - 256), fresh_off(4 * lane));
          const float x[4] = {tailF.x * g_rf, tailF.y * g_rf, tailF.z * g_rf, tailF.w * g_rf};
          dc2 = f2{(g_rf_i != g_rf) ? -dc_carry : dc_carry, dc_highpass_end_state<4>(x, hp4.x, hp4.y)};
        }
        float2 osc_tab[4];
        uint64_t osc_p[4];  // (rx512_kernel.hpp: unsigned sample numbers, the sub-block's part of the phase scalar)
        {
          const uint64_t Pl = (uint64_t)(unsigned)(8 * lane + 1) * dphi;
#pragma unroll
          for (int sb = 0; sb < 4; ++sb) {
#if T41RX_PHASE_SPLIT
            osc_p[sb] = (phase0 + (uint64_t)(512 * sb) * dphi) + Pl;
#else
            osc_p[sb] = phase0 + (uint64_t)(512 * sb + 8 * lane + 1) * dphi;
#endif
            osc_tab[sb] = ldg2(tab + kTabSinCos, (unsigned)(osc_p[sb] >> 56));
          }
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
#if T41RX_PHASE_SPLIT
            mix(z, osc_p[sb], osc_tab[sb]);
#else
            mix(z, phase0 + (uint64_t)(n0 + 1) * dphi, osc_tab[sb]);
#endif
            wave_sync();
            float *xw = lds + kX + 20 * lane;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              *reinterpret_cast<float4 *>(xw + 2 * (xpad(28 + 2 * i))) = make_float4(z[2 * i].x, z[2 * i].y, z[2 * i + 1].x, z[2 * i + 1].y);
            wave_sync();
            cf o1[2];
            {
              auto pidx = [](int o) { return xpad(o); };
              if (T41RX_DEC1_REGTAIL && T41RX_FF_REGTAIL) fir_pair<kDec1Taps, 1, 5, 18, 6, 14>(xw, pidx, cf0, kCoDec1, o1[0], o1[1], nullptr, z);  // (rx512_kernel.hpp)
              else fir_pair<kDec1Taps, 1, 5, 18, 6>(xw, pidx, cf0, kCoDec1, o1[0], o1[1]);
            }
            {
              float4 hh = any_float4();
              if (lane < 14) hh = lds4(lds + kX + 2 * xpad(512 + 2 * lane));
              wave_sync();
              if (lane < 14) *reinterpret_cast<float4 *>(lds + kX + 2 * xpad(2 * lane)) = hh;
              *reinterpret_cast<float4 *>(lds + kY1 + y1slot(24 + 64 * hh2 + lane)) = make_float4(o1[0].x, o1[0].y, o1[1].x, o1[1].y);
            }
          }
          wave_sync();
          {
            auto planes = [](int o) { return y1slot(o >> 1) / 2; };
            fir_pair<kDec2Taps, 3, 5, 26, 6>(lds + kY1 + 4 * lane, planes, cf0, kCoDec2, ynew[sg][rd][0], ynew[sg][rd][1]);
          }
          {
            float4 hh = any_float4();
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
    FRESH_LANE(); FRESH_WV(wv);
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
      FRESH_LANE(); FRESH_WV(wv);
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
    const LaneOff lom = fresh_off(2 * lane);
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int q = wv + NWV * h;
#pragma unroll
      for (int r = 0; r < 8; ++r) mk[h][r] = ldg_cf(maskN + 512 * q, lom, 64 * r);
    }
    __syncthreads();
    FRESH_LANE(); FRESH_WV(wv);
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
    FRESH_LANE(); FRESH_WV(wv);
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
      load_taps<48>(ci, (CFloatPtr)cf0, kCoInt1);
      __syncthreads();
      FRESH_LANE(); FRESH_WV(wv);
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int j = 0; j < R / 2; ++j) AU[24 + lane + 64 * (wv + NWV * h) + 512 * j] = y3[h][j];
      if (threadIdx.x < 24) AU[threadIdx.x] = hi_reg;
      if (threadIdx.x < 8) YT[threadIdx.x] = yt_reg;
      __syncthreads();
      FRESH_LANE(); FRESH_WV(wv);
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
        asm volatile("" ::"v"(w[0]));  // (rx512_kernel.hpp: or the 16-byte reads are narrowed to ds_read2_b32 + address adds)
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
      load_taps<32>(c4, (CFloatPtr)cf0, kCoInt2);
      __syncthreads();
      FRESH_LANE(); FRESH_WV(wv);
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
        // (the segment's eight history words as two 16-byte reads; [0] is padding and is kept alive on purpose --
        // rx512_kernel.hpp, x2 window: without it the reads are narrowed to ds_read2_b32 at one address register each)
        const float4 ya = lds4(YT + 8 * sg), yb = lds4(YT + 8 * sg + 4);
        asm volatile("" ::"v"(ya.x));
        const float hsv[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const float up = lane_up1(x1v[i + 1]);
          w[i] = (lane == 0) ? hsv[i + 1] : up;
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
        const LaneOff lof = fresh_off(4 * lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = 8 * i + (lane >> 3);
          const float4 t = lds4(tr + 4 * (8 * row + ((lane & 7) ^ (row & 7))));
          stg_stream(gO + 256 * i, lof, t);
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


}  // namespace t41

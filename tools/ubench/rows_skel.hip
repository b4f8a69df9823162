// The memory-only skeleton of the fused receive kernel as a free-standing micro-benchmark: what does each
// property of its access pattern cost against a bare copy of the same rows and against a linear sweep?
// ("bisect the skeleton against stream_rows".)
//
// 4096 waves, one per channel; a wave streams through its own rows of I, Q (reads) and O (writes),
// [channel][T * 2048] floats as the C ABI lays them out.  Per frame: four 512-sample sub-blocks of I and Q
// (2 KiB each), DEPTH sub-blocks requested ahead, the frame's 2048 output floats stored in 1-KiB instructions.
// Template switches select: the lane layout of a sub-block load (PAIR: lane l takes samples 8 l .. 8 l + 7 with two
// 16-byte loads, the kernel's layout; CONTIG: 16 bytes at 16 l and at 1024 + 16 l), the prefetch depth, when the
// stores are issued (burst at the frame's end / two behind every sub-block), the cache policy bits of loads and
// stores (buffer instructions' sc0 / nt / sc1), an LDS round trip per sub-block, a block of dependent packed FMAs per
// sub-block (WORK: stands in for the arithmetic; with LAYOUT 2 every wave works on the same 16 channels' rows, i.e.
// cache-resident I/O, which times the arithmetic alone), the workgroup shape and the array layout.
//
// hipcc -O3 --offload-arch=gfx950 rows_skel.hip -o rows_skel && ./rows_skel [filter]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <type_traits>
#include <vector>

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum { PAIR = 0, CONTIG = 1 };
enum { BURST = 0, SPREAD = 1, NOSTORE = 2, NOLOAD = 3 };
enum { ROWS = 0, FMAJOR = 1, RESIDENT = 2 };
// cache policy (aux operand of the raw buffer intrinsics on gfx940+): bit 0 sc0, bit 1 nt, bit 4 sc1
enum { P_PLAIN = 0, P_SC0 = 1, P_NT = 2, P_SC1 = 16, P_SC0SC1 = 17, P_NTSC1 = 18, P_ALL = 19, P_SC0NT = 3 };

template <int LDPAT, int DEPTH, int STMODE, int LDPOL, int STPOL, int WORK, int LDSB, int WGW, int LAYOUT>
__global__ __launch_bounds__(WGW * 64) void skel(const float *I, const float *Q, float *O, int T, int nchan, float wa, float wb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ch0 = blockIdx.x * WGW + wv;
  if (ch0 >= nchan) return;
  const int ch = (LAYOUT == RESIDENT) ? (ch0 & 15) : ch0;
  const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc((void *)I, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((void *)Q, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void *)O, 0, 0x7fffffff, 0x00020000);
  // byte offset of frame f of this channel
  auto fofs = [&](int f) -> unsigned {
    if (LAYOUT == FMAJOR) return ((unsigned)f * (unsigned)nchan + (unsigned)ch) * 8192u;
    return ((unsigned)ch * (unsigned)T + (unsigned)f) * 8192u;
  };
  const unsigned v0 = (LDPAT == PAIR) ? 32u * lane : 16u * lane;
  constexpr unsigned kSecond = (LDPAT == PAIR) ? 16u : 1024u;
  u4 in[DEPTH][4];
  auto request = [&](int slot, int g) {  // sub-block g of the launch (4 per frame)
    const unsigned so = fofs(g >> 2) + 2048u * (g & 3);
    in[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rI, v0, so, LDPOL);
    in[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rI, v0 + kSecond, so, LDPOL);
    in[slot][2] = __builtin_amdgcn_raw_buffer_load_b128(rQ, v0, so, LDPOL);
    in[slot][3] = __builtin_amdgcn_raw_buffer_load_b128(rQ, v0 + kSecond, so, LDPOL);
  };
  constexpr int U = (DEPTH == 3) ? 12 : 4;  // sub-blocks per unrolled super-iteration (a multiple of 4 and of DEPTH)
  const int total = 4 * T;                  // T is a multiple of 3 when DEPTH == 3 (host)
  if (STMODE != NOLOAD) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) request(d, d);
  }
  u4 out[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = u4{(unsigned)lane, 1u, 2u, 3u};
  f2 w[4] = {f2{1.0f, 2.0f}, f2{3.0f, 4.0f}, f2{5.0f, 6.0f}, f2{7.0f, 8.0f}};
  float *slice = smem + wv * 2048;
  for (int g0 = 0; g0 < total; g0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int g = g0 + u, slot = u % DEPTH, s = u & 3;
      u4 a0, a1;
      if (STMODE != NOLOAD) {
        a0 = in[slot][0] + in[slot][2];
        a1 = in[slot][1] + in[slot][3];
        __builtin_amdgcn_sched_barrier(0);  // (the phases stay in program order: consume, request, work, store)
        request(slot, (g + DEPTH < total) ? g + DEPTH : total - 1);  // (unconditional: the tail re-reads the last sub-block)
      } else {
        a0 = u4{(unsigned)g, 0u, 1u, 2u};
        a1 = a0 + 1u;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (LDSB) {  // one round trip of the sub-block's 16 floats per lane through the wave's LDS slice
        *reinterpret_cast<u4 *>(slice + 8 * lane) = a0;
        *reinterpret_cast<u4 *>(slice + 8 * lane + 4) = a1;
        __builtin_amdgcn_wave_barrier();
        a0 = *reinterpret_cast<u4 *>(slice + 4 * lane);
        a1 = *reinterpret_cast<u4 *>(slice + 256 + 4 * lane);
        __builtin_amdgcn_wave_barrier();
      }
      if (WORK) {
        w[0].x += __uint_as_float(a0.x & 0x3fffffffu);
        for (int it = 0; it < WORK; ++it) {
#pragma unroll
          for (int c = 0; c < 4; ++c) w[c] = __builtin_elementwise_fma(w[c], f2{wa, wa}, f2{wb, wb});
        }
        a0.x ^= __float_as_uint(w[0].x + w[1].y + w[2].x + w[3].y) & 1u;
      }
      out[2 * s] = a0;
      out[2 * s + 1] = a1;
      __builtin_amdgcn_sched_barrier(0);
      const unsigned oo = fofs(g >> 2);
      if (STMODE == SPREAD) {
        __builtin_amdgcn_raw_buffer_store_b128(out[2 * s], rO, 16u * lane + 2048u * s, oo, STPOL);
        __builtin_amdgcn_raw_buffer_store_b128(out[2 * s + 1], rO, 16u * lane + 2048u * s + 1024u, oo, STPOL);
      } else if ((STMODE == BURST || STMODE == NOLOAD) && s == 3) {
#pragma unroll
        for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_buffer_store_b128(out[j], rO, 16u * lane + 1024u * j, oo, STPOL);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (STMODE == NOSTORE) {  // keep the loads alive
    u4 t = out[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) t += out[j];
    if (t.x == 0x12345678u && t.y == 0x9abcdef0u) __builtin_amdgcn_raw_buffer_store_b128(t, rO, 16u * lane, fofs(0), 0);
  }
}

// ---- E: the same skeleton with every vector-memory operation written in inline asm and every wait counted by hand, so
// that what is waited for is exactly what this file says (hipcc's own counting across the loop's back edge is
// conservative: it drains the previous frame's stores at the top of every frame -- the product kernel's steady state
// did the same, which is what these variants measure).  WAITTOP 0: exact waits (a sub-block's data only); 1: vmcnt(0)
// at the top of every frame after the first (= wait for the previous frame's 8 stores); 2: also vmcnt(0) before the
// frame's last sub-block (drains the next frame's first sub-block, requested one sub-block earlier).
#define T41_STR2(x) #x
#define T41_STR(x) T41_STR2(x)
template <int IMM>
__device__ __forceinline__ u4 asm_load(unsigned voff, __amdgpu_buffer_rsrc_t r) {
  u4 d;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%c3 nt" : "=v"(d) : "v"(voff), "s"(r), "n"(IMM) : "memory");
  return d;
}
template <int IMM>
__device__ __forceinline__ void asm_store(u4 v, unsigned voff, __amdgpu_buffer_rsrc_t r) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen offset:%c3 nt\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(r), "n"(IMM) : "memory");
}
template <int N>
__device__ __forceinline__ void asm_wait(u4 &a, u4 &b, u4 &c, u4 &d) {
  asm volatile("s_waitcnt vmcnt(%c4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int DEPTH, int WAITTOP, int WORK, int LDSB, int WGW, int LAYOUT = 0, int STAGGER = 0, int CHMAP = 0, int PRIO = 0, int XKIND = 0, int XN = 0>
__global__ __launch_bounds__(WGW * 64) void skel2(const float *I, const float *Q, float *O, int T, int nchan, float wa, float wb, int resident, unsigned long long *clk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // CHMAP 1: the channels of a workgroup lie nchan / WGW apart instead of next to each other
  const int ch0 = (CHMAP == 1) ? (int)(blockIdx.x + (unsigned)wv * (unsigned)(nchan / WGW)) : (int)(blockIdx.x * WGW + wv);
  if (ch0 >= nchan) return;
  const int ch = resident ? (ch0 & 15) : ch0;
  const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc((void *)I, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((void *)Q, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void *)O, 0, 0x7fffffff, 0x00020000);
  unsigned long long c0 = 0, r0 = 0;  // shader clock / constant 100 MHz clock at the wave's start
  if (clk) {
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  }
  // STAGGER: wave w of the workgroup starts (w / 4) * STAGGER * 64 cycles late (the four waves of a SIMD apart);
  // negative: (w % 4) * -STAGGER * 64 (the four SIMDs apart)
  if (STAGGER != 0) {
    const int steps = STAGGER > 0 ? (wv >> 2) * STAGGER : (wv & 3) * -STAGGER;
    for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(1);
  }
  if (PRIO == 1) {  // a static priority per wave of a SIMD
    if ((wv >> 2) == 0) __builtin_amdgcn_s_setprio(3);
    else if ((wv >> 2) == 1) __builtin_amdgcn_s_setprio(2);
    else if ((wv >> 2) == 2) __builtin_amdgcn_s_setprio(1);
  }
  // byte offset of sub-block g (4 per frame) of this channel.  LAYOUT 0: [channel][frame][2048] (the C ABI's rows);
  // 1: [frame][channel][2048] (time-major: the frames of a call stacked as the calls' own [channel][2048] buffers);
  // 2: [frame][sub-block][channel][512]; 3: [block of 16 channels][frame][16][2048] (a workgroup's channels together)
  auto sbofs = [&](int g) -> unsigned {
    const unsigned f = (unsigned)g >> 2, sb = (unsigned)g & 3u, C = (unsigned)nchan, c = (unsigned)ch;
    if (LAYOUT == 1) return (f * C + c) * 8192u + sb * 2048u;
    if (LAYOUT == 2) return ((f * 4u + sb) * C + c) * 2048u;
    if (LAYOUT == 3) return (((c >> 4) * (unsigned)T + f) * 16u + (c & 15u)) * 8192u + sb * 2048u;
    return (c * (unsigned)T + f) * 8192u + sb * 2048u;
  };
  const int total = 4 * T;
  u4 in[DEPTH][4];
  auto request = [&](int slot, int g) {
    const int gc = g < total ? g : total - 1;
    const unsigned vo = sbofs(gc) + 32u * lane;
    in[slot][0] = asm_load<0>(vo, rI);
    in[slot][1] = asm_load<16>(vo, rI);
    in[slot][2] = asm_load<0>(vo, rQ);
    in[slot][3] = asm_load<16>(vo, rQ);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) request(d, d);
  u4 out[8];
  f2 w[4] = {f2{1.0f, 2.0f}, f2{3.0f, 4.0f}, f2{5.0f, 6.0f}, f2{7.0f, 8.0f}};
  float *slice = smem + wv * 2048;
  constexpr int U = (DEPTH == 3) ? 12 : 4;
  // younger operations than sub-block g's loads when it is consumed: the (DEPTH - 1) sets requested after it and the 8
  // stores of every frame that ended between its request (behind step g - DEPTH) and now (ahead of step g)
  auto body = [&](auto first_c, int g0) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int g = g0 + u;
      constexpr int dummy = 0;
      (void)dummy;
      const int slot = u % DEPTH, s = u & 3;
      int stores = 0;
#pragma unroll
      for (int k = u - DEPTH; k <= u - 1; ++k) stores += ((k & 3) == 3 && (!FIRST || k >= 0)) ? 8 : 0;
      const int younger = 4 * (DEPTH - 1) + stores;
      if (WAITTOP >= 1 && s == 0 && !(FIRST && u == 0)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (WAITTOP >= 2 && s == 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // (the count is a compile-time constant per unrolled step)
      switch (younger) {
        case 0: asm_wait<0>(in[slot][0], in[slot][1], in[slot][2], in[slot][3]); break;
        case 4: asm_wait<4>(in[slot][0], in[slot][1], in[slot][2], in[slot][3]); break;
        case 8: asm_wait<8>(in[slot][0], in[slot][1], in[slot][2], in[slot][3]); break;
        case 12: asm_wait<12>(in[slot][0], in[slot][1], in[slot][2], in[slot][3]); break;
        case 16: asm_wait<16>(in[slot][0], in[slot][1], in[slot][2], in[slot][3]); break;
        case 20: asm_wait<20>(in[slot][0], in[slot][1], in[slot][2], in[slot][3]); break;
        default: asm_wait<0>(in[slot][0], in[slot][1], in[slot][2], in[slot][3]); break;
      }
      u4 a0 = in[slot][0] + in[slot][2];
      u4 a1 = in[slot][1] + in[slot][3];
      __builtin_amdgcn_sched_barrier(0);
      request(slot, g + DEPTH);
      __builtin_amdgcn_sched_barrier(0);
      if (LDSB) {
        *reinterpret_cast<u4 *>(slice + 8 * lane) = a0;
        *reinterpret_cast<u4 *>(slice + 8 * lane + 4) = a1;
        __builtin_amdgcn_wave_barrier();
        a0 = *reinterpret_cast<u4 *>(slice + 4 * lane);
        a1 = *reinterpret_cast<u4 *>(slice + 256 + 4 * lane);
        __builtin_amdgcn_wave_barrier();
      }
      if (LDSB > 1) {  // LDSB window-like reads (lane-contiguous, 20 floats apart as the kernel's), four per batch
        u4 acc = a1;
        for (int r = 0; r < LDSB; r += 4) {
          const float *wp = slice + 20 * lane + 16 * (r & 12);
          const u4 t0 = *reinterpret_cast<const u4 *>(wp), t1 = *reinterpret_cast<const u4 *>(wp + 4);
          const u4 t2 = *reinterpret_cast<const u4 *>(wp + 8), t3 = *reinterpret_cast<const u4 *>(wp + 12);
          acc ^= (t0 + t1) ^ (t2 + t3);
          asm volatile("" : "+v"(acc));
        }
        a1 = acc;
      }
      if (WORK) {
        w[0].x += __uint_as_float(a0.x & 0x3fffffffu);
        // XKIND / XN: XN extra instructions of one class per iteration beside the four packed FMAs (what does an
        // instruction of that class cost while the chip is holding its clock down?)
        float xs[4] = {w[0].x, w[1].x, w[2].x, w[3].x};
        u4 xl = a1;
        for (int it = 0; it < WORK; ++it) {
#pragma unroll
          for (int c = 0; c < 4; ++c) w[c] = __builtin_elementwise_fma(w[c], f2{wa, wa}, f2{wb, wb});
#pragma unroll
          for (int x = 0; x < XN; ++x) {
            if (XKIND == 1) w[x & 3] = __builtin_elementwise_fma(w[x & 3], f2{wb, wb}, f2{wa, wa});  // v_pk_fma_f32
            if (XKIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xs[x & 3]) : "v"(wa), "v"(wb));
            if (XKIND == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(xs[x & 3]) : "v"(wb));
            if (XKIND == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(xs[x & 3]) : "v"(lane));
            if (XKIND == 5) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(xs[x & 3]) : "v"(xs[(x + 1) & 3]));
            if (XKIND == 6) asm volatile("ds_read_b128 %0, %1" : "=v"(xl) : "v"(80u * lane + 8192u * wv) : "memory");
            if (XKIND == 7) asm volatile("ds_write_b128 %0, %1" ::"v"(80u * lane + 8192u * wv), "v"(xl) : "memory");
            if (XKIND == 8) asm volatile("s_nop 0");
            if (XKIND == 9) asm volatile("ds_read_b64 %0, %1" : "=v"(w[x & 3]) : "v"(80u * lane + 8192u * wv) : "memory");
          }
          if (XKIND == 6 || XKIND == 7 || XKIND == 9) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        a0.x ^= __float_as_uint(w[0].x + w[1].y + w[2].x + w[3].y + xs[0] + xs[1] + xs[2] + xs[3]) & 1u;
        a0.y ^= xl.x & 1u;
      }
      out[2 * s] = a0;
      out[2 * s + 1] = a1;
      __builtin_amdgcn_sched_barrier(0);
      if (s == 3) {
        if (LAYOUT == 2) {  // the frame's four 2-KiB output pieces lie a channel-row apart
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned vo = sbofs(g - 3 + q) + 16u * lane;
            asm_store<0>(out[2 * q], vo, rO);
            asm_store<1024>(out[2 * q + 1], vo, rO);
          }
        } else {
          const unsigned vo = sbofs(g - 3) + 16u * lane;
          asm_store<0>(out[0], vo, rO);
          asm_store<1024>(out[1], vo, rO);
          asm_store<2048>(out[2], vo, rO);
          asm_store<3072>(out[3], vo, rO);
          asm_store<0>(out[4], vo + 4096u, rO);
          asm_store<1024>(out[5], vo + 4096u, rO);
          asm_store<2048>(out[6], vo + 4096u, rO);
          asm_store<3072>(out[7], vo + 4096u, rO);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  body(std::true_type{}, 0);
  for (int g0 = U; g0 < total; g0 += U) body(std::false_type{}, g0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (clk) {
    unsigned long long c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    if (lane == 0) {
      clk[2 * ch0] = c1 - c0;
      clk[2 * ch0 + 1] = r1 - r0;
    }
  }
}

// the same bytes as one linear sweep (every wave 1 KiB of each array per step)
template <int LDPOL, int STPOL>
__global__ __launch_bounds__(1024) void linear(const float *I, const float *Q, float *O, int T, int nchan) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned w = blockIdx.x * 16 + wv, W = nchan;
  const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc((void *)I, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((void *)Q, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void *)O, 0, 0x7fffffff, 0x00020000);
  const unsigned steps = (unsigned)T * 8;  // 1 KiB steps per wave
  for (unsigned i = 0; i < steps; ++i) {
    const unsigned so = (i * W + w) * 1024u;
    const u4 a = __builtin_amdgcn_raw_buffer_load_b128(rI, 16u * lane, so, LDPOL);
    const u4 b = __builtin_amdgcn_raw_buffer_load_b128(rQ, 16u * lane, so, LDPOL);
    __builtin_amdgcn_raw_buffer_store_b128(a + b, rO, 16u * lane, so, STPOL);
  }
}

template <int MODE>  // 1 loads only, 2 stores only
__global__ __launch_bounds__(1024) void linear1(const float *I, const float *Q, float *O, int T, int nchan) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned w = blockIdx.x * 16 + wv, W = nchan;
  const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc((void *)I, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((void *)Q, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void *)O, 0, 0x7fffffff, 0x00020000);
  const unsigned steps = (unsigned)T * 8;
  u4 acc = u4{0u, 0u, 0u, 0u};
  for (unsigned i = 0; i < steps; ++i) {
    const unsigned so = (i * W + w) * 1024u;
    if (MODE == 1) {
      acc += __builtin_amdgcn_raw_buffer_load_b128(rI, 16u * lane, so, P_NT);
      acc += __builtin_amdgcn_raw_buffer_load_b128(rQ, 16u * lane, so, P_NT);
    } else {
      __builtin_amdgcn_raw_buffer_store_b128(u4{i, w, 1u, 2u}, rO, 16u * lane, so, P_NT);
    }
  }
  if (MODE == 1 && acc.x == 0x12345678u && acc.y == 0x9abcdef1u) __builtin_amdgcn_raw_buffer_store_b128(acc, rO, 16u * lane, 0, 0);
}

static float *gI, *gQ, *gO;
static const char *gfilter = nullptr;
static const int kT = 30, kChan = 4096;  // 30 frames: a multiple of 3 for DEPTH 3 (the product runs 32)

template <typename K>
static void timeit(const char *name, K launch, double bytes) {
  if (gfilter && !strstr(name, gfilter)) return;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  const int reps = 12;
  float best = 1e30f, sum = 0;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    best = ms < best ? ms : best;
    sum += ms;
  }
  const hipError_t err = hipGetLastError();
  printf("%-64s %8.1f us/launch (best %8.1f)  %6.2f us/frame  %5.2f TB/s%s\n", name, sum / 3 * 1e3, best * 1e3, sum / 3 * 1e3 / kT,
         bytes / (sum / 3) / 1e9, err == hipSuccess ? "" : "  ERROR");
  fflush(stdout);
}

template <int LDPAT, int DEPTH, int STMODE, int LDPOL, int STPOL, int WORK, int LDSB, int WGW, int LAYOUT>
static void run(const char *name, int lds_bytes) {
  auto kern = skel<LDPAT, DEPTH, STMODE, LDPOL, STPOL, WORK, LDSB, WGW, LAYOUT>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const double per = (STMODE == NOSTORE ? 8.0 : STMODE == NOLOAD ? 4.0 : 12.0);
  timeit(name, [&]() { hipLaunchKernelGGL(kern, dim3(kChan / WGW), dim3(WGW * 64), lds_bytes, 0, gI, gQ, gO, kT, kChan, 0.999f, 0.001f); },
         per * kChan * kT * 2048.0);
}

template <int DEPTH, int WAITTOP, int WORK, int LDSB, int WGW, int LAYOUT = 0, int STAGGER = 0, int CHMAP = 0, int PRIO = 0, int XKIND = 0, int XN = 0>
static void run2(const char *name, int lds_bytes, int resident) {
  auto kern = skel2<DEPTH, WAITTOP, WORK, LDSB, WGW, LAYOUT, STAGGER, CHMAP, PRIO, XKIND, XN>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  timeit(name, [&]() { hipLaunchKernelGGL(kern, dim3(kChan / WGW), dim3(WGW * 64), lds_bytes, 0, gI, gQ, gO, kT, kChan, 0.999f, 0.001f, resident, (unsigned long long *)nullptr); },
         12.0 * kChan * kT * 2048.0);
  if (gfilter && !strstr(name, gfilter)) return;
  // the shader clock the waves saw: one more launch with the two counters read at each wave's start and end
  static unsigned long long *dclk = nullptr;
  if (!dclk) (void)hipMalloc(&dclk, sizeof(unsigned long long) * 2 * kChan);
  for (int i = 0; i < 4; ++i)
    hipLaunchKernelGGL(kern, dim3(kChan / WGW), dim3(WGW * 64), lds_bytes, 0, gI, gQ, gO, kT, kChan, 0.999f, 0.001f, resident, dclk);
  std::vector<unsigned long long> h(2 * kChan);
  (void)hipMemcpy(h.data(), dclk, sizeof(unsigned long long) * 2 * kChan, hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  for (int c = 0; c < kChan; ++c)
    if (h[2 * c + 1]) ghz.push_back((double)h[2 * c] / (double)h[2 * c + 1] * 0.1);
  std::sort(ghz.begin(), ghz.end());
  if (!ghz.empty()) printf("      shader clock seen by the waves: median %.3f GHz (min %.3f max %.3f)\n", ghz[ghz.size() / 2], ghz.front(), ghz.back());
  fflush(stdout);
}

int main(int argc, char **argv) {
  if (argc > 1) gfilter = argv[1];
  const size_t n = (size_t)kChan * kT * 2048;
  if (hipMalloc(&gI, n * 4) != hipSuccess || hipMalloc(&gQ, n * 4) != hipSuccess || hipMalloc(&gO, n * 4) != hipSuccess) return 1;
  (void)hipMemset(gI, 1, n * 4);
  (void)hipMemset(gQ, 2, n * 4);
  const int L16 = 160 * 1024, L4 = 40 * 1024, L8 = 80 * 1024;
  const double b12 = 12.0 * kChan * kT * 2048.0;
  timeit("linear nt/nt", [&]() { hipLaunchKernelGGL((linear<P_NT, P_NT>), dim3(kChan / 16), dim3(1024), 0, 0, gI, gQ, gO, kT, kChan); }, b12);
  timeit("linear plain/plain", [&]() { hipLaunchKernelGGL((linear<P_PLAIN, P_PLAIN>), dim3(kChan / 16), dim3(1024), 0, 0, gI, gQ, gO, kT, kChan); }, b12);
  timeit("linear loads only", [&]() { hipLaunchKernelGGL((linear1<1>), dim3(kChan / 16), dim3(1024), 0, 0, gI, gQ, gO, kT, kChan); }, b12 * 2 / 3);
  timeit("linear stores only", [&]() { hipLaunchKernelGGL((linear1<2>), dim3(kChan / 16), dim3(1024), 0, 0, gI, gQ, gO, kT, kChan); }, b12 / 3);
  // ---- A: the bare copy of the kernel's rows and its variations
  run<PAIR, 2, BURST, P_NT, P_NT, 0, 0, 16, ROWS>("A0 rows PAIR D2 burst nt/nt 16w (the kernel's pattern)", L16);
  run<CONTIG, 2, BURST, P_NT, P_NT, 0, 0, 16, ROWS>("A1 rows CONTIG D2 burst nt/nt 16w", L16);
  run<PAIR, 1, BURST, P_NT, P_NT, 0, 0, 16, ROWS>("A2 rows PAIR D1", L16);
  run<PAIR, 3, BURST, P_NT, P_NT, 0, 0, 16, ROWS>("A3 rows PAIR D3", L16);
  run<PAIR, 4, BURST, P_NT, P_NT, 0, 0, 16, ROWS>("A4 rows PAIR D4", L16);
  run<PAIR, 2, SPREAD, P_NT, P_NT, 0, 0, 16, ROWS>("A5 rows PAIR D2 spread stores", L16);
  run<PAIR, 2, NOSTORE, P_NT, P_NT, 0, 0, 16, ROWS>("A6 rows PAIR D2 loads only (8 B/sample)", L16);
  run<PAIR, 2, NOLOAD, P_NT, P_NT, 0, 0, 16, ROWS>("A7 rows stores only (4 B/sample)", L16);
  run<PAIR, 2, BURST, P_NT, P_NT, 0, 0, 4, ROWS>("A8 rows PAIR D2 burst, 4-wave workgroups x 4 per CU", L4);
  run<PAIR, 4, BURST, P_NT, P_NT, 0, 0, 8, ROWS>("A9 rows PAIR D4 burst, 8 waves per CU", L16);
  run<PAIR, 2, BURST, P_NT, P_NT, 0, 0, 16, FMAJOR>("A10 frame-major PAIR D2 burst", L16);
  run<CONTIG, 4, BURST, P_NT, P_NT, 0, 0, 16, FMAJOR>("A11 frame-major CONTIG D4 burst", L16);
  // ---- B: cache policies (PAIR D2 burst 16w)
  run<PAIR, 2, BURST, P_PLAIN, P_NT, 0, 0, 16, ROWS>("B0 ld plain  st nt", L16);
  run<PAIR, 2, BURST, P_SC1, P_NT, 0, 0, 16, ROWS>("B1 ld sc1    st nt", L16);
  run<PAIR, 2, BURST, P_SC0SC1, P_NT, 0, 0, 16, ROWS>("B2 ld sc0sc1 st nt", L16);
  run<PAIR, 2, BURST, P_NTSC1, P_NT, 0, 0, 16, ROWS>("B3 ld nt sc1 st nt", L16);
  run<PAIR, 2, BURST, P_ALL, P_NT, 0, 0, 16, ROWS>("B4 ld sc0 nt sc1 st nt", L16);
  run<PAIR, 2, BURST, P_SC0NT, P_NT, 0, 0, 16, ROWS>("B5 ld sc0 nt st nt", L16);
  run<PAIR, 2, BURST, P_NT, P_PLAIN, 0, 0, 16, ROWS>("B6 ld nt st plain", L16);
  run<PAIR, 2, BURST, P_NT, P_SC1, 0, 0, 16, ROWS>("B7 ld nt st sc1", L16);
  run<PAIR, 2, BURST, P_NT, P_SC0SC1, 0, 0, 16, ROWS>("B8 ld nt st sc0sc1", L16);
  run<PAIR, 2, BURST, P_NT, P_NTSC1, 0, 0, 16, ROWS>("B9 ld nt st nt sc1", L16);
  run<PAIR, 2, BURST, P_NT, P_ALL, 0, 0, 16, ROWS>("B10 ld nt st sc0 nt sc1", L16);
  // ---- C: with an LDS round trip per sub-block
  run<PAIR, 2, BURST, P_NT, P_NT, 0, 1, 16, ROWS>("C0 rows PAIR D2 burst + LDS round trip", L16);
  // ---- D: with arithmetic (WORK x 4 dependent v_pk_fma_f32 per sub-block), alone (cache-resident rows) and with the rows
  run<PAIR, 2, BURST, P_NT, P_NT, 80, 0, 16, RESIDENT>("D0 work 80 resident (arithmetic alone)", L16);
  run<PAIR, 2, BURST, P_NT, P_NT, 80, 0, 16, ROWS>("D1 work 80 rows D2", L16);
  run<PAIR, 3, BURST, P_NT, P_NT, 80, 0, 16, ROWS>("D2 work 80 rows D3", L16);
  run<PAIR, 4, BURST, P_NT, P_NT, 80, 0, 16, ROWS>("D3 work 80 rows D4", L16);
  run<PAIR, 1, BURST, P_NT, P_NT, 80, 0, 16, ROWS>("D4 work 80 rows D1", L16);
  run<PAIR, 2, SPREAD, P_NT, P_NT, 80, 0, 16, ROWS>("D5 work 80 rows D2 spread stores", L16);
  run<CONTIG, 2, BURST, P_NT, P_NT, 80, 0, 16, ROWS>("D6 work 80 rows CONTIG D2", L16);
  run<PAIR, 2, BURST, P_NT, P_NT, 120, 0, 16, RESIDENT>("D7 work 120 resident", L16);
  run<PAIR, 2, BURST, P_NT, P_NT, 120, 0, 16, ROWS>("D8 work 120 rows D2", L16);
  run<PAIR, 2, BURST, P_NT, P_NT, 150, 0, 16, RESIDENT>("D9 work 150 resident", L16);
  run<PAIR, 2, BURST, P_NT, P_NT, 150, 0, 16, ROWS>("D10 work 150 rows D2", L16);
  run<PAIR, 4, BURST, P_NT, P_NT, 150, 0, 16, ROWS>("D11 work 150 rows D4", L16);
  (void)L8;
  // ---- E: hand-counted waits
  run2<2, 0, 0, 0, 16>("E0 asm D2 exact waits", L16, 0);
  run2<2, 1, 0, 0, 16>("E1 asm D2 vmcnt(0) at every frame's top (waits for the stores)", L16, 0);
  run2<2, 2, 0, 0, 16>("E2 asm D2 ... and before the last sub-block", L16, 0);
  run2<1, 0, 0, 0, 16>("E3 asm D1 exact", L16, 0);
  run2<3, 0, 0, 0, 16>("E4 asm D3 exact", L16, 0);
  run2<4, 0, 0, 0, 16>("E5 asm D4 exact", L16, 0);
  run2<2, 0, 0, 1, 16>("E6 asm D2 exact + LDS round trip", L16, 0);
  run2<2, 1, 0, 1, 16>("E7 asm D2 top wait + LDS round trip", L16, 0);
  run2<2, 0, 120, 0, 16>("F0 asm D2 exact, work 120 resident (arithmetic alone)", L16, 1);
  run2<2, 0, 120, 0, 16>("F1 asm D2 exact, work 120 rows", L16, 0);
  run2<2, 1, 120, 0, 16>("F2 asm D2 top wait, work 120 rows", L16, 0);
  run2<2, 2, 120, 0, 16>("F3 asm D2 top + last-sub-block waits, work 120 rows", L16, 0);
  run2<3, 0, 120, 0, 16>("F4 asm D3 exact, work 120 rows", L16, 0);
  run2<4, 0, 120, 0, 16>("F5 asm D4 exact, work 120 rows", L16, 0);
  run2<1, 0, 120, 0, 16>("F6 asm D1 exact, work 120 rows", L16, 0);
  run2<2, 0, 150, 0, 16>("G0 asm D2 exact, work 150 resident", L16, 1);
  run2<2, 0, 150, 0, 16>("G1 asm D2 exact, work 150 rows", L16, 0);
  run2<2, 1, 150, 0, 16>("G2 asm D2 top wait, work 150 rows", L16, 0);
  run2<2, 0, 90, 0, 16>("H0 asm D2 exact, work 90 resident", L16, 1);
  run2<2, 0, 90, 0, 16>("H1 asm D2 exact, work 90 rows", L16, 0);
  run2<2, 1, 90, 0, 16>("H2 asm D2 top wait, work 90 rows", L16, 0);
  // ---- L: layouts (hand-counted waits, D2)
  run2<2, 0, 0, 0, 16, 1>("L1 time-major [frame][channel][2048]", L16, 0);
  run2<2, 0, 0, 0, 16, 2>("L2 [frame][sub-block][channel][512]", L16, 0);
  run2<2, 0, 0, 0, 16, 3>("L3 [16-channel block][frame][16][2048]", L16, 0);
  run2<2, 0, 0, 0, 4, 1>("L4 time-major, 4-wave workgroups x 4", L4, 0);
  run2<2, 0, 0, 0, 8, 1>("L5 time-major, 8 waves per CU", L16, 0);
  run2<2, 0, 150, 0, 16, 1>("L6 time-major, work 150", L16, 0);
  run2<2, 0, 120, 0, 16, 1>("L7 time-major, work 120", L16, 0);
  run2<2, 0, 90, 0, 16, 1>("L8 time-major, work 90", L16, 0);
  run2<2, 0, 150, 0, 16, 2>("L9 sub-block-major, work 150", L16, 0);
  run2<2, 0, 120, 0, 16, 2>("L10 sub-block-major, work 120", L16, 0);
  run2<2, 0, 150, 0, 4, 1>("L11 time-major, 4-wave workgroups, work 150", L4, 0);
  run2<2, 0, 150, 0, 4, 0>("L12 rows, 4-wave workgroups, work 150", L4, 0);
  run2<1, 0, 150, 0, 16, 1>("L13 time-major D1, work 150", L16, 0);
  // ---- S: what about the workgroup's shape matters (rows, D2, work 150)
  run2<2, 0, 150, 0, 16, 0>("S0 16-wave workgroup (reference)", L16, 0);
  run2<2, 0, 150, 0, 4, 0>("S1 4-wave workgroups x 4, 40 KiB each", L4, 0);
  run2<2, 0, 150, 0, 4, 0>("S2 4-wave workgroups x 4, no LDS", 0, 0);
  run2<2, 0, 150, 0, 8, 0>("S3 8-wave workgroups x 2, 80 KiB each", L8, 0);
  run2<2, 0, 150, 0, 16, 0, 8>("S4 16-wave, SIMD-mates start 512 cycles apart", L16, 0);
  run2<2, 0, 150, 0, 16, 0, 64>("S5 16-wave, SIMD-mates start 4 k cycles apart", L16, 0);
  run2<2, 0, 150, 0, 16, 0, 256>("S6 16-wave, SIMD-mates start 16 k cycles apart", L16, 0);
  run2<2, 0, 150, 0, 16, 0, -64>("S7 16-wave, the four SIMDs start 4 k cycles apart", L16, 0);
  run2<2, 0, 150, 0, 16, 0, 0, 1>("S8 16-wave, a workgroup's channels 256 apart", L16, 0);
  run2<2, 0, 150, 0, 16, 0, 0, 0, 1>("S9 16-wave, static priorities 3/2/1/0 within a SIMD", L16, 0);
  run2<2, 0, 150, 0, 4, 0, 0, 1>("S10 4-wave workgroups, channels 1024 apart", L4, 0);
  run2<1, 0, 150, 0, 4, 0>("S11 4-wave workgroups D1", L4, 0);
  run2<2, 0, 0, 0, 16, 0, 0, 1>("S12 16-wave no work, channels 256 apart", L16, 0);
  run2<2, 0, 120, 0, 4, 0>("S13 4-wave workgroups work 120", L4, 0);
  run2<2, 0, 120, 0, 4, 1>("S14 4-wave workgroups work 120 time-major", L4, 0);
  // ---- K: the clock under each kind of load
  run2<2, 0, 150, 0, 16, 0>("K0 work 150 resident (arithmetic alone)", L16, 1);
  run2<2, 0, 0, 0, 16, 0>("K1 rows, no work (memory alone)", L16, 0);
  run2<2, 0, 150, 0, 16, 0>("K2 work 150 rows", L16, 0);
  run2<2, 0, 120, 0, 16, 0>("K3 work 120 rows", L16, 0);
  run2<2, 0, 90, 0, 16, 0>("K4 work 90 rows", L16, 0);
  run2<2, 0, 60, 0, 16, 0>("K5 work 60 rows", L16, 0);
  run2<2, 0, 150, 1, 16, 0>("K6 work 150 rows + LDS round trip", L16, 0);
  run2<2, 0, 150, 0, 16, 1>("K7 work 150 time-major", L16, 0);
  // ---- P: what an LDS read costs next to a packed FMA when the chip is holding its clock down (power)
  run2<2, 0, 100, 0, 16, 0>("P0 work 100 rows", L16, 0);
  run2<2, 0, 100, 64, 16, 0>("P1 work 100 rows + 64 ds_read_b128 per sub-block", L16, 0);
  run2<2, 0, 100, 128, 16, 0>("P2 work 100 rows + 128 ds_read_b128 per sub-block", L16, 0);
  run2<2, 0, 120, 0, 16, 0>("P3 work 120 rows", L16, 0);
  run2<2, 0, 120, 64, 16, 0>("P4 work 120 rows + 64 ds_read_b128", L16, 0);
  run2<2, 0, 140, 0, 16, 0>("P5 work 140 rows", L16, 0);
  run2<2, 0, 100, 64, 16, 0>("P6 work 100 resident + 64 ds_read_b128", L16, 1);
  run2<2, 0, 100, 0, 16, 0>("P7 work 100 resident", L16, 1);
  run2<2, 0, 160, 0, 16, 0>("P8 work 160 rows", L16, 0);
  run2<2, 0, 180, 0, 16, 0>("P9 work 180 rows", L16, 0);
  // ---- X: price list under the power cap: 100 x (4 v_pk_fma_f32 + XN x one instruction class) per sub-block, rows
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 0, 0>("X00 base: 100 x 4 v_pk_fma_f32", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 1, 2>("X01 + 2 v_pk_fma_f32", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 2, 2>("X02 + 2 v_fma_f32", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 3, 2>("X03 + 2 v_mov_b32", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 4, 2>("X04 + 2 v_add_u32", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 5, 2>("X05 + 2 v_mov_b32_dpp", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 6, 1>("X06 + 1 ds_read_b128", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 7, 1>("X07 + 1 ds_write_b128", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 8, 2>("X08 + 2 s_nop", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 9, 2>("X09 + 2 ds_read_b64", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 1, 4>("X11 + 4 v_pk_fma_f32", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 2, 4>("X12 + 4 v_fma_f32", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 3, 4>("X13 + 4 v_mov_b32", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 4, 4>("X14 + 4 v_add_u32", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 6, 2>("X16 + 2 ds_read_b128", L16, 0);
  run2<2, 0, 100, 0, 16, 0, 0, 0, 0, 0, 0>("X00 base again", L16, 0);
  return 0;
}

// Microbenchmark (round 4): what one step of the AGC chain costs a single wave -- alone on its SIMD and beside three
// waves that issue packed FMAs back to back (the duty wave's situation in rx512_kernel<..., PIPE>).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/chain_lat.hip -o gpurun_out/chain_lat && gpurun_out/chain_lat
// Cycles are s_memtime (shader clock) in the chain wave, per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long lanemask;
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lanemask lanes_ge(float a, float b) { lanemask m; asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ lanemask lanes_gt(float a, float b) { lanemask m; asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ float pick(lanemask m, float s, float c) { float r; asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(c), "v"(s), "s"(m)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// V = 0: six dependent plain VALU operations (sub mul add mov mov max), no masks
//     1: the phased step: sub, cmp_ge, cmp_gt, pk_mul, pk_add, pk_fma, cmp_neq, cndmask, cndmask, max
//     2: 1 + the step's mask bookkeeping (7 SALU, 1 cndmask)
//     3: the unpacked step (sub, 2 mul, 2 add, 2 fma, ...) + bookkeeping
//     4: 1 without the bracket (HAS3 = false): sub, cmp, cmp, pk_mul, pk_add, cndmask, max
//     5: dependent v_add only (latency of one VALU operation)
//     6: v_cmp -> v_cndmask only (the SGPR hop)
template <int V>
__device__ __forceinline__ void steps(float &volts, const float *rm, int n, f2 mult, float thr, float minv, lanemask is3, lanemask &acc, float &save) {
#pragma clang fp contract(off)
  const f2 cb = f2{__uint_as_float(0x3d4ccccbu), __uint_as_float(0x3d4cccceu)};
  lanemask in0 = acc, pend = ~acc, ok = ~0ull, df = 0;
#pragma unroll 4
  for (int k = 0; k < n; ++k) {
    const float x = rm[k & 3];
    if (V == 5) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(volts) : "v"(x)); continue; }
    if (V == 6) { const lanemask ge = lanes_ge(x, volts); volts = pick(ge, x, minv); continue; }
    if (V == 0) {
      float d = x - volts, s = d * mult.x, nv = volts + s;
      asm volatile("v_mov_b32 %0, %0" : "+v"(nv));
      asm volatile("v_mov_b32 %0, %0" : "+v"(nv));
      volts = vmax(nv, minv);
      continue;
    }
    const float vin = volts, diff = x - volts;
    const lanemask ge = lanes_ge(x, volts), gt = lanes_gt(volts, thr);
    float na, cand;
    if (V == 3) {
      const float sa = diff * mult.x, ss = diff * mult.y;
      na = volts + sa;
      cand = volts + ss;
      const float rl = __builtin_fmaf(ss, cb.x, volts), rh = __builtin_fmaf(ss, cb.y, volts);
      lanemask differ;
      asm volatile("v_cmp_neq_f32_e64 %0, %1, %2" : "=s"(differ) : "v"(rl), "v"(rh));
      df |= differ;
      cand = pick(is3, rh, cand);
    } else {
      const f2 s2 = f2{diff, diff} * mult, n2 = f2{volts, volts} + s2;
      na = n2.x;
      cand = n2.y;
      if (V != 4) {
        const f2 r = __builtin_elementwise_fma(f2{s2.y, s2.y}, cb, f2{volts, volts});
        lanemask differ;
        asm volatile("v_cmp_neq_f32_e64 %0, %1, %2" : "=s"(differ) : "v"(r.x), "v"(r.y));
        df |= differ;
        cand = pick(is3, r.y, cand);
      }
    }
    volts = vmax(pick(ge, na, cand), minv);
    if (V == 2 || V == 3) {
      ok &= ge | (gt & ~in0);
      save = pick(ge & pend, vin, save);
      pend &= ~ge;
      in0 |= ge;
    } else {
      ok &= gt;
    }
  }
  acc = ok ^ df ^ in0 ^ pend;
}

template <int V, bool BUSY>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters, float a, float b, float thr, lanemask m0) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {  // one chain wave per SIMD (waves are dealt to the SIMDs round robin)
    __builtin_amdgcn_s_setprio(3);
    float volts = 1.0f + 1e-3f * lane, save = 0.f;
    const float rm[4] = {a, b, a * 1.01f, b * 0.99f};
    lanemask acc = m0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) steps<V>(volts, rm, 64, f2{0.01f, 0.002f}, thr, 1e-6f, m0, acc, save);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
    out[blockIdx.x * 1024 + threadIdx.x] = volts + save + (float)(acc & 1);
    __syncthreads();  // (not reached by the busy waves before their loop ends: see below)
  } else if (BUSY) {
    f2 acc2[8];
    for (int i = 0; i < 8; ++i) acc2[i] = f2{(float)lane + i, (float)i};
    const f2 av = {a, a}, bv = {b * 1e-3f, b * 1e-3f};
    // a fixed amount of packed-FMA work, sized by the host to outlast the chain waves
    for (int it = 0; it < iters * 300; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc2[i] = __builtin_elementwise_fma(acc2[i], av, bv);
    }
    f2 t = acc2[0] + acc2[1] + acc2[2] + acc2[3] + acc2[4] + acc2[5] + acc2[6] + acc2[7];
    out[blockIdx.x * 1024 + threadIdx.x] = t.x + t.y;
    __syncthreads();
  } else {
    __syncthreads();
  }
}

template <int V, bool BUSY>
void run(float *d, unsigned long long *dc, const char *name) {
  const int iters = 400, blocks = 256;
  unsigned long long h[1024];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<V, BUSY>), dim3(blocks), dim3(1024), 0, 0, d, dc, iters, 1.5f, 0.5f, 0.25f, 0x00000000ffff0000ull);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 1024; ++i) s += (double)h[i];
  printf("%-74s %s  %7.1f cycles per step\n", name, BUSY ? "beside 3 busy waves" : "alone on its SIMD   ", s / 1024 / (iters * 64.0));
}
int main() {
  float *d;
  unsigned long long *dc;
  hipMalloc(&d, 256 * 1024 * sizeof(float));
  hipMalloc(&dc, 1024 * sizeof(unsigned long long));
#define BOTH(V, name) run<V, false>(d, dc, name); run<V, true>(d, dc, name)
  BOTH(5, "v_add -> v_add (one dependent VALU operation)");
  BOTH(6, "v_cmp -> v_cndmask (VALU -> SGPR -> VALU)");
  BOTH(0, "sub mul add mov mov max (six dependent VALU operations)");
  BOTH(4, "step without the bracket: sub cmp cmp pk_mul pk_add cndmask max");
  BOTH(1, "phased step: sub cmp cmp pk_mul pk_add pk_fma cmp_neq cndmask cndmask max");
  BOTH(2, "phased step + mask bookkeeping (7 SALU, 1 cndmask)");
  BOTH(3, "unpacked step (2 mul 2 add 2 fma) + mask bookkeeping");
  return 0;
}

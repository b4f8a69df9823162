// Microbenchmark: does a wave64 VALU instruction get cheaper when only a few lanes are active?
// One wave per SIMD, dependent and independent f32 chains, with lanes < ACTIVE enabled.
// Also: latency of v_cmp -> s_and -> v_cndmask chains (VALU -> SALU -> VALU hops) and of a
// scalar branch on a VALU compare, the building blocks of a single-wave serial state machine.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b, int active) {
  const int lane = threadIdx.x & 63;
  float acc[CH];
  for (int i = 0; i < CH; ++i) acc[i] = (float)threadIdx.x + i;
  if (lane < active) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 32 / CH; ++r)
#pragma unroll
        for (int i = 0; i < CH; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
    }
  }
  float s = 0;
  for (int i = 0; i < CH; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// v = (v >= t) ? v * a : v * b  -- compare, select, multiply: a 3-deep dependent loop
__global__ __launch_bounds__(256) void ksel(float *out, int iters, float a, float b, float t, int active) {
  const int lane = threadIdx.x & 63;
  float v = 1.0f + (float)lane * 1e-3f;
  if (lane < active) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float m = (v >= t) ? a : b;
        v = v * m;
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = v;
}
// the same with a wave-uniform branch on the compare
__global__ __launch_bounds__(256) void kbr(float *out, int iters, float a, float b, float t, int active) {
  const int lane = threadIdx.x & 63;
  float v = 1.0f + (float)lane * 1e-3f;
  if (lane < active) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (__builtin_amdgcn_ballot_w64(v >= t) != 0) v = v * a; else v = v * b + 1e-9f;
        asm volatile("" : "+v"(v));
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = v;
}
template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
  return ms;
}
int main() {
  float *d; hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
  const int iters = 20000;
  for (int active : {64, 32, 16, 4, 1}) {
    float m1 = timeit([&] { hipLaunchKernelGGL((k<1>), dim3(256), dim3(256), 0, 0, d, iters, 0.999f, 0.001f, active); });
    float m8 = timeit([&] { hipLaunchKernelGGL((k<8>), dim3(256), dim3(256), 0, 0, d, iters, 0.999f, 0.001f, active); });
    float ms = timeit([&] { hipLaunchKernelGGL(ksel, dim3(256), dim3(256), 0, 0, d, iters, 0.9999f, 1.0001f, 1.0f, active); });
    float mb = timeit([&] { hipLaunchKernelGGL(kbr, dim3(256), dim3(256), 0, 0, d, iters, 0.9999f, 1.0001f, 1.0f, active); });
    printf("active=%2d: dependent fma %.1f cyc, 8 independent chains %.1f cyc/instr, cmp+select+mul loop %.1f cyc/iter, cmp+branch+mul %.1f cyc/iter\n",
           active, m1 * 1e6 / (iters * 32.0) * 2.4, m8 * 1e6 / (iters * 32.0) * 2.4, ms * 1e6 / (iters * 16.0) * 2.4, mb * 1e6 / (iters * 16.0) * 2.4);
  }
  return 0;
}

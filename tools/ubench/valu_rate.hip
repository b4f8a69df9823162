// Microbenchmark: VALU issue rate per SIMD for v_fma_f32 vs v_pk_fma_f32 at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
  if (PK == 0) {
    float acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f2 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f2{(float)threadIdx.x + i, (float)i};
    f2 av = {a, a}, bv = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
    }
    f2 s = {0, 0};
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
  }
}

int main() {
  float *d;
  hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;  // 32 instr per iter
  for (int pk = 0; pk < 2; ++pk)
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
      int grid = 256 * wgs_per_cu;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
        else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double instr_per_wave = (double)iters * 32;
      double waves_per_simd = wgs_per_cu;  // 256 threads = 4 waves = 1 per SIMD
      double ns_per_instr_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
      double tflops = (double)grid * 256 * instr_per_wave * (pk ? 4 : 2) / (ms * 1e-3) / 1e12;
      printf("%s waves/SIMD=%d  time %.3f ms  %.3f ns per wave-instr per SIMD  (~%.2f cyc @2.4GHz)  %.1f TFLOP/s\n",
             pk ? "v_pk_fma_f32" : "v_fma_f32   ", wgs_per_cu, ms, ns_per_instr_simd, ns_per_instr_simd * 2.4, tflops);
    }
  return 0;
}

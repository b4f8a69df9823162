// Microbenchmark: dependent-chain latency of v_pk_fma_f32 / v_fma_f32 with 1, 2, 4 chains per
// wave, at 1 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int CH, int PK>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
  f2 acc[CH];
  for (int i = 0; i < CH; ++i) acc[i] = f2{(float)threadIdx.x + i, (float)i};
  f2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 32 / CH; ++r)
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        if (PK) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
        else acc[i].x = __builtin_fmaf(acc[i].x, a, b);
      }
  }
  f2 s = {0, 0};
  for (int i = 0; i < CH; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
template <int CH, int PK>
void run(float *d, int wps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CH, PK>), dim3(256 * wps), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s chains=%d waves/SIMD=%d: %.2f ns per instr per wave (%.1f cyc @2.4GHz)\n", PK ? "pk_fma" : "fma   ", CH, wps,
         ms * 1e6 / (iters * 32.0), ms * 1e6 / (iters * 32.0) * 2.4);
}
int main() {
  float *d; hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
  run<1, 1>(d, 1); run<2, 1>(d, 1); run<4, 1>(d, 1); run<8, 1>(d, 1);
  run<1, 0>(d, 1); run<2, 0>(d, 1); run<4, 0>(d, 1);
  run<1, 1>(d, 4); run<2, 1>(d, 4); run<4, 1>(d, 4);
  return 0;
}

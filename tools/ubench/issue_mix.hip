// Microbenchmark: do SALU / s_nop instructions share issue bandwidth with VALU on a SIMD?
// Each wave runs N iterations of {8 v_pk_fma (4 independent chains)} plus K extra scalar ops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND, int K>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b, int s0) {
  f2 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f2{(float)threadIdx.x + i, (float)i};
  f2 av = {a, a}, bv = {b, b};
  int s = s0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (KIND == 0) asm volatile("s_nop 0");
      if (KIND == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s));
      if (KIND == 2) asm volatile("v_mov_b32 %0, %0" : "+v"(acc[0].x));  // extra VALU for reference
    }
  }
  f2 t = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y + s;
}
template <int KIND, int K>
void run(float *d, const char *name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 40000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, K>), dim3(256 * 4), dim3(256), 0, 0, d, iters, 0.999f, 0.001f, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-8s extra=%d per 8 pk_fma: %.3f ms  -> %.2f ns per iteration per SIMD (4 waves)\n", name, K, ms, ms * 1e6 / iters);
}
int main() {
  float *d; hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
  run<0, 0>(d, "base"); run<0, 4>(d, "s_nop"); run<0, 8>(d, "s_nop");
  run<1, 4>(d, "s_add"); run<1, 8>(d, "s_add");
  run<2, 4>(d, "v_mov"); run<2, 8>(d, "v_mov");
  return 0;
}

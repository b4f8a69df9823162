// Microbenchmark: cost of VALU -> SGPR -> SALU -> VALU hops and of scalar instructions in a
// single-wave dependent loop (the shape of the AGC chain, rx_kernels.hip: agc_fast_block).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long lanemask;
__device__ __forceinline__ lanemask lanes_ge(float a, float b) { lanemask m; asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ float pick(lanemask m, float s, float c) { float r; asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(c), "v"(s), "s"(m)); return r; }
// VARIANT 0: cmp -> cndmask -> mul.  1: cmp -> s_and -> cndmask -> mul.  2: variant 1 + 6 more scalar
// instructions that depend on the compare (mask bookkeeping).  3: variant 0 + 6 independent scalar ops.
template <int VARIANT>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b, float t, lanemask m0) {
  const int lane = threadIdx.x & 63;
  float v = 1.0f + (float)lane * 1e-3f;
  lanemask acc = m0, acc2 = ~m0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      lanemask ge = lanes_ge(v, t);
      if (VARIANT == 1 || VARIANT == 2) ge &= m0;
      if (VARIANT == 2) {
        asm volatile("s_or_b64 %0, %0, %2\n\ts_andn2_b64 %1, %1, %2\n\ts_and_b64 %0, %0, %1\n\ts_or_b64 %1, %1, %2\n\ts_andn2_b64 %0, %0, %2\n\ts_or_b64 %0, %0, %1"
                     : "+s"(acc), "+s"(acc2) : "s"(ge));
      }
      if (VARIANT == 3) {
        asm volatile("s_or_b64 %0, %0, %1\n\ts_andn2_b64 %1, %1, %0\n\ts_and_b64 %0, %0, %1\n\ts_or_b64 %1, %1, %0\n\ts_andn2_b64 %0, %0, %1\n\ts_or_b64 %0, %0, %1"
                     : "+s"(acc), "+s"(acc2));
      }
      v = v * pick(ge, a, b);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = v + (float)(acc & 1) + (float)(acc2 & 1);
}
template <int V>
void run(float *d, const char *name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V>), dim3(256), dim3(256), 0, 0, d, iters, 0.9999f, 1.0001f, 1.0f, 0xffffffffffffffffull);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-58s %.1f cycles per iteration @2.4 GHz\n", name, ms * 1e6 / (iters * 16.0) * 2.4);
}
int main() {
  float *d; hipMalloc(&d, 256 * 256 * sizeof(float));
  run<0>(d, "v_cmp -> v_cndmask -> v_mul");
  run<1>(d, "v_cmp -> s_and -> v_cndmask -> v_mul");
  run<2>(d, "  ... + 6 scalar ops depending on the compare");
  run<3>(d, "v_cmp -> v_cndmask -> v_mul + 6 independent scalar ops");
  return 0;
}

// What does the hot path's ACCESS PATTERN cost against a linear sweep?  4096 waves, each streaming
// through its own rows of two input arrays and one output array ([channels][T * 2048] floats, as
// the C ABI lays them out): per frame of 2048 floats a wave reads I and Q in PIECE-float pieces
// with DEPTH pieces in flight and writes the frame's output in 1-KiB store instructions.
// hipcc -O3 --offload-arch=gfx950 stream_rows.hip -o stream_rows && ./stream_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4n __attribute__((ext_vector_type(4)));
#ifndef PLAIN_LOADS
#define PLAIN_LOADS 0
#endif
__device__ __forceinline__ float4 ldnt(const float4 *p) {
#if PLAIN_LOADS
  return *p;
#else
  const f4n t = __builtin_nontemporal_load(reinterpret_cast<const f4n *>(p));
  return make_float4(t.x, t.y, t.z, t.w);
#endif
}
__device__ __forceinline__ void stnt(float4 v, float4 *p) { __builtin_nontemporal_store(f4n{v.x, v.y, v.z, v.w}, reinterpret_cast<f4n *>(p)); }

// PAIR: a lane takes 32 contiguous bytes of a piece with two loads (the hot path's sub-block layout:
// lane l holds samples 8 l .. 8 l + 7), instead of 16 bytes at 1-KiB distance
template <int PIECE, int WG_WAVES, bool LINEAR, bool PAIR = false, bool FMAJOR = false>
__global__ __launch_bounds__(WG_WAVES * 64) void k(const float *__restrict__ I, const float *__restrict__ Q, float *__restrict__ O, int T, int nchan) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ch = blockIdx.x * WG_WAVES + wv;
  if (ch >= nchan) return;
  constexpr int V = PIECE / 256;  // float4 per lane and piece
  if (LINEAR) {
    // the same bytes as a linear sweep: wave w of W takes float4 index i * W * 64 + w * 64 + lane
    const size_t W = (size_t)nchan, total = (size_t)nchan * T * 512;  // float4s
    for (size_t i = (size_t)ch * 64 + lane; i < total; i += W * 64) {
      const float4 a = ldnt(reinterpret_cast<const float4 *>(I) + i);
      const float4 b = ldnt(reinterpret_cast<const float4 *>(Q) + i);
      float4 c = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
      stnt(c, reinterpret_cast<float4 *>(O) + i);
    }
    return;
  }
  // FMAJOR: [frame][channel][2048] instead of [channel][frame * 2048]
  const float4 *pi = reinterpret_cast<const float4 *>(I + (FMAJOR ? (size_t)ch * 2048 : (size_t)ch * T * 2048));
  const float4 *pq = reinterpret_cast<const float4 *>(Q + (FMAJOR ? (size_t)ch * 2048 : (size_t)ch * T * 2048));
  float4 *po = reinterpret_cast<float4 *>(O + (FMAJOR ? (size_t)ch * 2048 : (size_t)ch * T * 2048));
  constexpr int PPFr = 2048 / PIECE;
  auto pofs = [&](int piece) -> size_t {  // float4 offset of a piece
    if (!FMAJOR) return (size_t)piece * (PIECE / 4);
    return (size_t)(piece / PPFr) * ((size_t)nchan * 512) + (size_t)(piece % PPFr) * (PIECE / 4);
  };
  const int pieces = T * 2048 / PIECE;
  float4 a[2][V], b[2][V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    a[0][v] = ldnt(pi + (PAIR ? 128 * (v >> 1) + 2 * lane + (v & 1) : 64 * v + lane));
    b[0][v] = ldnt(pq + (PAIR ? 128 * (v >> 1) + 2 * lane + (v & 1) : 64 * v + lane));
  }
  float4 acc[8];
  for (int p = 0; p < pieces; p += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int nx = p + h + 1;
      if (nx < pieces) {
#pragma unroll
        for (int v = 0; v < V; ++v) a[h ^ 1][v] = ldnt(pi + pofs(nx) + (PAIR ? 128 * (v >> 1) + 2 * lane + (v & 1) : 64 * v + lane));
#pragma unroll
        for (int v = 0; v < V; ++v) b[h ^ 1][v] = ldnt(pq + pofs(nx) + (PAIR ? 128 * (v >> 1) + 2 * lane + (v & 1) : 64 * v + lane));
      }
      // one frame = 2048 floats = 2048 / PIECE pieces; output float4 j of the frame = sum of what came in
      constexpr int PPF = 2048 / PIECE;  // pieces per frame (PIECE <= 2048)
      const int q = (p + h) % PPF;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const float4 x = a[h][v], y = b[h][v];
        acc[(q * V + v) & 7] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
      }
      if (q == PPF - 1) {
        const int f = (p + h) / PPF;
#pragma unroll
        for (int j = 0; j < 8; ++j) stnt(acc[j], po + (FMAJOR ? (size_t)f * nchan * 512 : (size_t)f * 512) + 64 * j + lane);
      }
    }
  }
}

template <int PIECE, int WG_WAVES, bool LINEAR, bool PAIR = false, bool FMAJOR = false>
static void run(const char *name, float *I, float *Q, float *O, int T, int nchan) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int grid = (nchan + WG_WAVES - 1) / WG_WAVES;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PIECE, WG_WAVES, LINEAR, PAIR, FMAJOR>), dim3(grid), dim3(WG_WAVES * 64), 0, 0, I, Q, O, T, nchan);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<PIECE, WG_WAVES, LINEAR, PAIR, FMAJOR>), dim3(grid), dim3(WG_WAVES * 64), 0, 0, I, Q, O, T, nchan);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = 12.0 * nchan * T * 2048;
  printf("%-44s %8.1f us per launch  %6.2f us per 4096 x 2048 frame  %5.2f TB/s\n", name, ms * 1e3, ms * 1e3 / T * (4096.0 / nchan), bytes / ms / 1e9);
}

int main() {
  const int T = 32, nchan = 4096;
  const size_t n = (size_t)nchan * T * 2048;
  float *I, *Q, *O;
  if (hipMalloc(&I, n * 4) != hipSuccess || hipMalloc(&Q, n * 4) != hipSuccess || hipMalloc(&O, n * 4) != hipSuccess) return 1;
  (void)hipMemset(I, 0, n * 4);
  (void)hipMemset(Q, 0, n * 4);
  run<512, 16, true>("linear sweep, 16-wave workgroups", I, Q, O, T, nchan);
  run<512, 4, true>("linear sweep, 4-wave workgroups", I, Q, O, T, nchan);
  run<256, 16, false>("rows, 1 KiB pieces, 16-wave workgroups", I, Q, O, T, nchan);
  run<512, 16, false>("rows, 2 KiB pieces, 16-wave workgroups", I, Q, O, T, nchan);
  run<1024, 16, false>("rows, 4 KiB pieces, 16-wave workgroups", I, Q, O, T, nchan);
  run<2048, 16, false>("rows, 8 KiB pieces, 16-wave workgroups", I, Q, O, T, nchan);
  run<512, 16, false, true>("rows, 2 KiB pieces, 32 B per lane, 16-wave", I, Q, O, T, nchan);
  run<512, 16, false, false, true>("frame-major, 2 KiB pieces, 16-wave", I, Q, O, T, nchan);
  run<512, 16, false, true, true>("frame-major, 2 KiB pieces, 32 B per lane", I, Q, O, T, nchan);
  run<2048, 16, false, false, true>("frame-major, 8 KiB pieces, 16-wave", I, Q, O, T, nchan);
  run<512, 4, false>("rows, 2 KiB pieces, 4-wave workgroups", I, Q, O, T, nchan);
  run<2048, 4, false>("rows, 8 KiB pieces, 4-wave workgroups", I, Q, O, T, nchan);
  return 0;
}

#include "t41_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
int main(void) {
  for (int agc = 0; agc <= 4; agc++) for (int mode = 0; mode < 4; mode++) for (int N = 512; N <= 4096; N *= 8) {
    t41o_params p; t41o_default_params(&p); p.mode = mode; p.AGCMode = agc; p.fft_length = N;
    if (mode == 1) { p.FLoCut = -3000; p.FHiCut = -200; } if (mode == 2) { p.FLoCut = -3000; p.FHiCut = 3000; }
    t41o_coeffs *c = malloc(sizeof *c);
    if (t41o_design(&p, c)) { printf("design failed\n"); return 1; }
    const int L = 4 * N, nfr = 5;
    t41o_channel *ch = t41o_channel_create(N);
    float *I = malloc(sizeof(float) * L), *Q = malloc(sizeof(float) * L), *o = malloc(sizeof(float) * L);
    int16_t *a = malloc(2 * L), *b = malloc(2 * L), *d = malloc(2 * L);
    double acc = 0;
    for (int f = 0; f < nfr; f++) {
      for (int i = 0; i < L; i++) { I[i] = 0.3f * sinf(0.01f * (i + f * L)) * (f == 2 ? 0.05f : 1.0f); Q[i] = 0.3f * cosf(0.013f * (i + f * L)); a[i] = (int16_t)(I[i] * 32767); b[i] = (int16_t)(Q[i] * 32767); }
      if (t41o_process_frame(ch, &p, c, 5000, I, Q, o)) return 2;
      if (t41o_process_frame_q15(ch, &p, c, 5000, a, b, d)) return 3;
      for (int i = 0; i < L; i++) acc += o[i] + d[i];
    }
    float tap[8192]; t41o_channel_tap(ch, T41O_TAP_IFFT, tap, 8192); t41o_channel_tap(ch, T41O_TAP_AGC_VOLTS, tap, 8192);
    t41o_channel_destroy(ch); free(I); free(Q); free(o); free(a); free(b); free(d); free(c);
    if (!(acc == acc)) { printf("nan\n"); return 4; }
  }
  printf("oracle sanitizer harness ok\n");
  return 0;
}

// t41_sdr_amd/csrc/rx_experiments.hpp -- the ONE switch behind which every build option lives that makes the kernels
// compute something else than the product does, or write diagnostics next to it.
//
// The product builds with none of them (t41_sdr_amd/csrc/Makefile passes no -DT41RX_*).  A timing experiment or a
// diagnostic build must say so: -DT41RX_EXPERIMENT=1 next to its own switch, otherwise this header stops the
// compilation.  Such a library reports itself: t41::kernel_build_flags() != 0, and t41rx_create() refuses to make a
// context on it unless the environment says T41RX_ALLOW_EXPERIMENT=1 (rx_host.cpp), so that a library with WRONG RESULTS
// BY CONSTRUCTION cannot stand in for the product by accident.
//
//   wrong results (timing experiments)
//     T41RX_ABLATE = n   cuts stages from the END of the chain (1 interpolators, 2 FFTs, 3 /2 decimator, 4 /4 decimator,
//                        5 NCO, 6 DC high-pass, 7 1-KiB store instructions, 8 the fused kernel's 16 x 64 B store
//                        instructions); 9 keeps all arithmetic but makes every wave use the same 16 channels' buffers
//                        (cache-resident I/O)                                     (tools/ablation_table.py)
//     T41RX_LOO = n      (leave one out) cuts exactly stage n of that list and keeps every other one; 13 / 14: the /2 / the
//                        /4 decimator's window taken from registers instead of LDS
//     T41RX_AGC_X        bit mask: 1 no back-averages, 2 no bookkeeping, 4 no bracket in the pipelined AGC chain's block
//     T41RX_AGC_R04CHECK 1: round 4's per-block fast-decay check of the pipelined AGC chain (unsound when min_volts was raised
//                        under a decaying lane) -- to show that tests/ and tools/pipe_soak.py catch it
//     T41RX_FCABL        bit mask: 1 no output stores, 2 no 512-point FFTs, 4 no input loads, 8 no x4 arithmetic, 16 no x2
//                        arithmetic in the long-FFT kernels
//   diagnostics (results unchanged, extra stores / counters)
//     T41RX_STAMP, T41RX_PIPE_STAT, T41RX_CLK
// Every other T41RX_* macro of the kernel sources selects among forms that compute the product's values (A/B builds:
// T41RX_PF, T41RX_AGC_PHASED, T41RX_FIR_PLAIN ...); they need no guard.
#pragma once

#ifndef T41RX_EXPERIMENT
#define T41RX_EXPERIMENT 0
#endif
#ifndef T41RX_ABLATE
#define T41RX_ABLATE 0
#endif
#ifndef T41RX_LOO
#define T41RX_LOO 0
#endif
#ifndef T41RX_AGC_X
#define T41RX_AGC_X 0
#endif
#ifndef T41RX_FCABL
#define T41RX_FCABL 0
#endif
#ifndef T41RX_AGC_R04CHECK
#define T41RX_AGC_R04CHECK 0
#endif
#define T41RX_CUT(n) ((T41RX_ABLATE >= (n) && T41RX_ABLATE <= 8) || T41RX_LOO == (n))

#define T41RX_WRONG_RESULTS (T41RX_ABLATE != 0 || T41RX_LOO != 0 || T41RX_AGC_X != 0 || T41RX_FCABL != 0 || T41RX_AGC_R04CHECK != 0)
#if defined(T41RX_STAMP) || defined(T41RX_PIPE_STAT) || defined(T41RX_CLK)
#define T41RX_DIAGNOSTICS 1
#else
#define T41RX_DIAGNOSTICS 0
#endif
#if (T41RX_WRONG_RESULTS || T41RX_DIAGNOSTICS) && !T41RX_EXPERIMENT
#error "T41RX_ABLATE / _LOO / _AGC_X / _AGC_R04CHECK / _FCABL / _STAMP / _PIPE_STAT / _CLK are experiment builds: pass -DT41RX_EXPERIMENT=1 with them (rx_experiments.hpp)"
#endif

namespace t41 {
// bit 0: wrong results by construction; bit 1: diagnostic stores / counters
constexpr int kKernelBuildFlags = (T41RX_WRONG_RESULTS ? 1 : 0) | (T41RX_DIAGNOSTICS ? 2 : 0);
}  // namespace t41

"""Per-phase cycle shares of the RX kernel from the -DT41RX_STAMP diagnostic build (GPU box).
usage: T41RX_LIB=.../libt41rx_stamp.so python tools/phase_stamps.py [nchan]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import t41_sdr_amd as T  # noqa: E402

NAMES = ["wait loads + gain", "DC high-pass", "NCO + mix", "LDS stage + /4 FIR", "history rolls", "/2 FIR",
         "state save + assemble", "twiddle loads", "forward FFT", "mask + inverse FFT", "demod + x2 staging",
         "x2 interpolator", "x4 interp + transpose writes", "transposed reads + stores",
         "first sub-block: wait loads", "prologue d: uniformise state + DC prepass",
         "prologue a: issue loads + SMEM gains", "prologue b: table staging + barrier (first vmcnt wait)",
         "prologue c: delay lines -> LDS"]


def main():
    nch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    L, D = 2048, 256
    rx = T.RxChain(nch, T.default_params(), NCOFreq=np.full(nch, 5000, np.int32))
    g = torch.Generator(device="cuda").manual_seed(0)
    I = 0.2 * torch.randn(nch, L, generator=g, device="cuda")
    Q = 0.2 * torch.randn(nch, L, generator=g, device="cuda")
    buf = torch.zeros(nch * D + 2 * nch * 64, device="cuda")  # demod tap | uint64 stamps
    rx.set_debug_taps(None, None, buf)
    for _ in range(3):  # steady state (no start-up transient), warm caches
        rx.ProcessIQData(I, Q)
    torch.cuda.synchronize()
    st = buf[nch * D:].view(torch.int64).view(nch, 64).cpu().numpy().astype(np.float64)
    tot = st[:, :19].sum(axis=1)
    print("channels %d: mean wave cycles %.0f (min %.0f max %.0f)" % (nch, tot.mean(), tot.min(), tot.max()))
    for p, name in enumerate(NAMES):
        print("  %2d %-30s %8.0f cycles  %5.1f %%" % (p, name, st[:, p].mean(), 100 * st[:, p].mean() / tot.mean()))


if __name__ == "__main__":
    main()

"""Kernel time of one launch for different (channels x frames-per-launch) shapes of the same work
(GPU box).  usage: python tools/shape_sweep.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import t41_sdr_amd as T  # noqa: E402

L = 2048


def run(nch, nfr, reps=200, ring=6):
    rx = T.RxChain(nch, T.default_params(), NCOFreq=np.full(nch, 5000, np.int32))
    g = torch.Generator(device="cuda").manual_seed(0)
    Is = [0.2 * torch.randn(nch, nfr * L, generator=g, device="cuda") for _ in range(ring)]
    Qs = [0.2 * torch.randn(nch, nfr * L, generator=g, device="cuda") for _ in range(ring)]
    out = [torch.empty(nch, nfr * L, device="cuda") for _ in range(ring)]
    for k in range(max(10, reps // 4)):
        rx.ProcessIQData(Is[k % ring], Qs[k % ring], out=out[k % ring])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        rx.ProcessIQData(Is[k % ring], Qs[k % ring], out=out[k % ring])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("%5d channels x %d frames per launch: %7.2f us  (%.1f GS/s)" % (nch, nfr, us, nch * nfr * L / us / 1e3))


if __name__ == "__main__":
    for nch, nfr in ((4096, 1), (2048, 2), (1024, 4), (4096, 2), (2048, 4), (4096, 4), (8192, 1)):
        run(nch, nfr)

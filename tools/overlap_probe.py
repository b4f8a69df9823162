"""Do the three kernels of the long-FFT pipeline overlap when two independent batches run on two
streams?  (GPU box.)  An upper bound for what chunked pipelining inside one call could gain."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import t41_sdr_amd as T  # noqa: E402


def main(nch=1024, nfr=8, fft=4096, reps=40):
    Lf = fft * 4
    kw = dict(fft_length=fft, mode=0, FLoCut=400, FHiCut=600)
    g = torch.Generator(device="cuda").manual_seed(0)
    ctx = []
    for k in range(2):
        rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=np.full(nch, 5000, np.int32))
        x = 0.2 * torch.randn(nch, nfr * Lf, generator=g, device="cuda")
        y = 0.2 * torch.randn(nch, nfr * Lf, generator=g, device="cuda")
        o = torch.empty(nch, nfr * Lf, device="cuda")
        ctx.append((rx, x, y, o, torch.cuda.Stream()))
    torch.cuda.synchronize()

    def timed(which, concurrent):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        main_s = torch.cuda.current_stream()
        if concurrent:
            for (_, _, _, _, s) in ctx:
                s.wait_stream(main_s)
        for _ in range(reps):
            for k in which:
                rx, x, y, o, s = ctx[k]
                if concurrent:
                    with torch.cuda.stream(s):
                        rx.ProcessIQData(x, y, out=o)
                else:
                    rx.ProcessIQData(x, y, out=o)
        if concurrent:
            for (_, _, _, _, s) in ctx:
                main_s.wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for _ in range(2):
        a = timed([0], False)
        b = timed([0, 1], False)
        c = timed([0, 1], True)
        print("one batch %.1f us | two batches, one stream %.1f us | two streams %.1f us  (per %dx%d frame: %.1f / %.1f / %.1f)"
              % (a, b, c, nch, Lf, a / nfr, b / nfr / 2, c / nfr / 2), flush=True)


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:]])

"""Does a caller gain by splitting BASELINE config 2's batch into independent channel groups on
separate HIP streams (each group its own context and state, so launch k+1 of one group could
load while launch k of another computes)?  Measured on MI355X: no -- round 1, one frame per launch:
33.8 us per 4096 x 2048 samples on one stream, 33.1 us as two groups of 2048 on two streams, 51 us as
four groups (host-launch bound); round 2, 32 frames per launch: 23.4 / 23.4 / 23.1 us per frame (wall
clock) -- the drain of one launch is not what the resident kernel loses.  Not the bench.py figure.
usage (GPU box): python tools/two_stream_throughput.py [frames per launch]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import t41_sdr_amd as T  # noqa: E402

L = 2048
TOTAL = 4096


def run(groups, steps=300, ring=6, nfr=1):
    nch = TOTAL // groups
    rng = np.random.default_rng(1)
    streams = [torch.cuda.Stream() for _ in range(groups)]
    ctx, Is, Qs, outs = [], [], [], []
    g = torch.Generator(device="cuda").manual_seed(0)
    for k in range(groups):
        nco = (rng.integers(-860, 801, nch) * 50).astype(np.int32)
        ctx.append(T.RxChain(nch, T.default_params(), NCOFreq=nco))
        Is.append([0.2 * torch.randn(nch, nfr * L, generator=g, device="cuda") for _ in range(ring)])
        Qs.append([0.2 * torch.randn(nch, nfr * L, generator=g, device="cuda") for _ in range(ring)])
        outs.append([torch.empty(nch, nfr * L, device="cuda") for _ in range(ring)])
    torch.cuda.synchronize()

    def step(i):
        for k in range(groups):
            with torch.cuda.stream(streams[k]):
                ctx[k].ProcessIQData(Is[k][i % ring], Qs[k][i % ring], out=outs[k][i % ring])

    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / steps * 1e6 / nfr
    print("%d channel group(s) of %4d on %d stream(s): %6.2f us per frame of %d x %d samples  (%.1f GS/s, %.3f of 8 TB/s at 12 B/sample)"
          % (groups, nch, groups, us, TOTAL, L, TOTAL * L / us / 1e3, 12.0 * TOTAL * L / us / 1e6 / 8.0))


if __name__ == "__main__":
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # frames per launch
    for groups in (1, 2, 4):
        run(groups, steps=max(20, 300 // nfr), ring=2 if nfr > 4 else 6, nfr=nfr)

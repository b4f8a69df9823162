#!/usr/bin/env python3
"""Time-major against channel-major buffers (t41rx_set_buffer_layout): same results bit for bit, and what each costs.

  python tools/layout_probe.py [--frames 32] [--reps 60] [--mode 0] [--agc 0]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import t41_sdr_amd as T  # noqa: E402

L = 2048


def opt(flag, default):
    a = sys.argv[1:]
    return type(default)(a[a.index(flag) + 1]) if flag in a else default


def timed(rx, Is, Qs, outs, reps):
    ring = len(Is)
    for k in range(max(6, reps // 4)):
        rx.ProcessIQData(Is[k % ring], Qs[k % ring], out=outs[k % ring])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        rx.ProcessIQData(Is[k % ring], Qs[k % ring], out=outs[k % ring])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    nch, frames, reps = opt("--nch", 4096), opt("--frames", 32), opt("--reps", 60)
    p = T.default_params(mode=opt("--mode", 0), AGCMode=opt("--agc", 0))
    rng = np.random.default_rng(1000)
    nco = (rng.integers(-860, 801, nch) * 50).astype(np.int32)
    g = torch.Generator(device="cuda").manual_seed(0)
    # parity: the same samples in both layouts through two fresh chains, two calls each (state carried)
    a = T.RxChain(nch, p, NCOFreq=nco)
    b = T.RxChain(nch, p, NCOFreq=nco)
    b.set_buffer_layout("time")
    same = True
    for call in range(2):
        I = (0.2 * torch.randn(nch, 4 * L, generator=g, device="cuda")).clamp_(-0.999, 0.999)
        Q = (0.2 * torch.randn(nch, 4 * L, generator=g, device="cuda")).clamp_(-0.999, 0.999)
        oa = a.ProcessIQData(I, Q)
        It = I.view(nch, 4, L).transpose(0, 1).contiguous()
        Qt = Q.view(nch, 4, L).transpose(0, 1).contiguous()
        ob = b.ProcessIQData(It, Qt)
        same = same and bool(torch.equal(oa.view(nch, 4, L).transpose(0, 1), ob))
    same = same and bool(np.array_equal(a.get_state(), b.get_state()))
    del a, b
    res = {"bit_identical": same}
    ring = max(2, -(-(768 << 20) // (3 * nch * frames * L * 4)))
    for layout in ("channel", "time", "channel", "time"):
        rx = T.RxChain(nch, p, NCOFreq=nco)
        rx.set_buffer_layout(layout)
        shape = (nch, frames * L) if layout == "channel" else (frames, nch, L)
        Is = [(0.2 * torch.randn(*shape, generator=g, device="cuda")).clamp_(-0.999, 0.999) for _ in range(ring)]
        Qs = [(0.2 * torch.randn(*shape, generator=g, device="cuda")).clamp_(-0.999, 0.999) for _ in range(ring)]
        outs = [torch.empty(*shape, device="cuda") for _ in range(ring)]
        us = timed(rx, Is, Qs, outs, reps)
        res.setdefault(layout, []).append(round(us / frames, 3))
        del rx, Is, Qs, outs
        torch.cuda.empty_cache()
    res["frac_channel"] = round(12 * nch * L / min(res["channel"]) / 1e3 / 8000, 4)
    res["frac_time"] = round(12 * nch * L / min(res["time"]) / 1e3 / 8000, 4)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()

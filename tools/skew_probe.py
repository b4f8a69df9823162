"""GPU box: does the relative placement of the I, Q and audio buffers matter?  Config 2's shape (4096 channels x 32
frames), the three arrays carved out of ONE allocation with a byte skew between them (0 = what separate 1-GiB-sized
torch allocations give: equal offsets modulo every power of two up to the allocator's 2 MiB granule).
usage: python tools/skew_probe.py [skew_bytes ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = 2048


def one(skew, frames=32, reps=40, nch=4096):
    import torch
    import t41_sdr_amd as T
    rng = np.random.default_rng(1000)
    nco = (rng.integers(-860, 801, nch) * 50).astype(np.int32)
    rx = T.RxChain(nch, T.default_params(), NCOFreq=nco)
    n = nch * frames * L
    ring = 3
    g = torch.Generator(device="cuda").manual_seed(0)
    pool = torch.empty(ring * 3 * (n + (1 << 20)) + (1 << 22), device="cuda")  # floats
    Is, Qs, outs = [], [], []
    pos = 0
    for r in range(ring):
        bufs = []
        for k in range(3):
            start = pos + (k * skew) // 4
            bufs.append(pool[start:start + n].view(nch, frames * L))
            pos += n + (1 << 20)
        bufs[0].copy_((0.2 * torch.randn(nch, frames * L, generator=g, device="cuda")).clamp_(-0.999, 0.999))
        bufs[1].copy_((0.2 * torch.randn(nch, frames * L, generator=g, device="cuda")).clamp_(-0.999, 0.999))
        Is.append(bufs[0]); Qs.append(bufs[1]); outs.append(bufs[2])
    for k in range(10):
        rx.ProcessIQData(Is[k % ring], Qs[k % ring], out=outs[k % ring])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        rx.ProcessIQData(Is[k % ring], Qs[k % ring], out=outs[k % ring])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3 / frames
    return {"skew_bytes": skew, "us_per_frame": round(us, 3), "frac": round(12 * nch * L / us / 1e3 / 8000.0, 4),
            "addr_mod_2MiB": [int(x.data_ptr() % (1 << 21)) for x in (Is[0], Qs[0], outs[0])]}


if __name__ == "__main__":
    skews = [int(a) for a in sys.argv[1:]] or [0, 256, 1024, 4096, 4096 + 256, 65536 + 4096 + 256, 0]
    for s in skews:
        print(json.dumps(one(s)), flush=True)

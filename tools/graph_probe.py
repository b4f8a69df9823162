#!/usr/bin/env python3
"""GPU box: the firmware's calling shape -- ONE ProcessIQData() per 10.67 ms frame (Display.cpp:339) -- as N single-frame
launches replayed from a HIP graph, against the same launches issued one by one and against one N-frame launch
(VERDICT r04 item 8).  The frames lie time-major ([frame][channel][2048]): each single-frame call gets the contiguous
[channel][2048] buffers a caller would hand over every frame period.

  python tools/graph_probe.py [--frames 32] [--reps 20] [--agc 0]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import t41_sdr_amd as T  # noqa: E402


def opt(flag, default):
    a = sys.argv[1:]
    return type(default)(a[a.index(flag) + 1]) if flag in a else default


def main():
    nch, frames, reps, agc = 4096, opt("--frames", 32), opt("--reps", 20), opt("--agc", 0)
    L = 2048
    nco = (np.random.default_rng(1000).integers(-860, 801, nch) * 50).astype(np.int32)
    g = torch.Generator(device="cuda").manual_seed(0)
    ring = 3
    Is = [(0.2 * torch.randn(frames, nch, L, generator=g, device="cuda")).clamp_(-0.999, 0.999) for _ in range(ring)]
    Qs = [(0.2 * torch.randn(frames, nch, L, generator=g, device="cuda")).clamp_(-0.999, 0.999) for _ in range(ring)]
    outs = [torch.empty(frames, nch, L, device="cuda") for _ in range(ring)]

    def timed(fn, n):
        for k in range(3):
            fn(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(n):
            fn(k)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3 / frames  # us per 4096-channel frame

    res = {}
    # (a) one launch of `frames` frames, time-major
    rx = T.RxChain(nch, T.default_params(AGCMode=agc), NCOFreq=nco)
    rx.set_buffer_layout("time")
    res["one_launch_of_%d_frames" % frames] = timed(lambda k: rx.ProcessIQData(Is[k % ring], Qs[k % ring], out=outs[k % ring]), reps)
    ref = outs[(reps - 1) % ring].clone()
    # (b) `frames` single-frame launches, issued one by one
    rx1 = T.RxChain(nch, T.default_params(AGCMode=agc), NCOFreq=nco)

    def singles(k):
        r = k % ring
        for f in range(frames):
            rx1.ProcessIQData(Is[r][f], Qs[r][f], out=outs[r][f])
    res["single_frame_launches_eager"] = timed(singles, reps)
    # (c) the same launches captured once per ring buffer and replayed
    rx2 = T.RxChain(nch, T.default_params(AGCMode=agc), NCOFreq=nco)
    rx2.ProcessIQData(Is[0][0], Qs[0][0], out=outs[0][0])  # first call: lazy allocations outside the capture
    rx2.reset()
    graphs = []
    for r in range(ring):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for f in range(frames):
                rx2.ProcessIQData(Is[r][f], Qs[r][f], out=outs[r][f])
        graphs.append(gr)
    rx2.reset()
    res["single_frame_launches_graph_replay"] = timed(lambda k: graphs[k % ring].replay(), reps)
    # the three streams processed the same samples in the same order (3 + reps passes each): same audio
    same = bool(torch.equal(ref, outs[(reps - 1) % ring]))
    print(json.dumps({"frames": frames, "agc": agc, "us_per_frame": {k: round(v, 3) for k, v in res.items()},
                      "frac_of_8TBs": {k: round(12 * nch * L / v / 1e3 / 8000, 4) for k, v in res.items()},
                      "graph_audio_equals_multi_frame_launch": same}), flush=True)


if __name__ == "__main__":
    main()

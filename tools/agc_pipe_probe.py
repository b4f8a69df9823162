"""GPU box: the pipelined AGC kernel (rx_kernels.hip: agc_prep_pipe) against the barrier form it replaces
(T41RX_AGC_PIPE=0, a second process: the switch is read once) -- bit-identity on ragged batches, and the time per
frame of both on BASELINE config 2's shape with AGCMode = 1.
usage: python tools/agc_pipe_probe.py [check|time] ..."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
L = 2048


def outputs(tag):
    import siggen
    import t41_sdr_amd as T
    import torch
    res = {}
    for name, nch, nfr, kw in (("usb21", 21, 9, dict(AGCMode=1)), ("usb64", 64, 36, dict(AGCMode=2)), ("am5", 5, 12, dict(AGCMode=3, mode=2, FLoCut=-3000, FHiCut=3000)),
                               ("nfm37", 37, 8, dict(AGCMode=4, mode=3, FLoCut=-4000, FHiCut=4000)), ("gains16", 16, 6, dict(AGCMode=1, RFgain=3, IQPhaseCorrectionFactor=0.05)),
                               ("sam21", 21, 20, dict(mode=8, FLoCut=-3000, FHiCut=3000)), ("sam64", 64, 9, dict(mode=8, FLoCut=-3000, FHiCut=3000))):
        nco = siggen.nco_grid(nch, seed=3)
        I, Q = siggen.make_iq(nch, nfr * L, nco, mode=2 if kw.get("mode") == 8 else kw.get("mode", 0), seed=7)
        # level steps so that the gain law walks through its states
        env = np.ones(nfr * L, np.float32)
        env[(nfr * L) // 3:(nfr * L) // 2] = 0.05
        I, Q = I * env, Q * env
        rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
        dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
        whole = rx.ProcessIQData(dI, dQ).cpu().numpy()
        st = rx.get_state() if hasattr(rx, "get_state") else None
        res[name] = whole
        if st is not None:
            res[name + "_state"] = np.asarray(st)
        # split calls (1-3 frames: the barrier form; the rest pipelined) must give the same samples
        rx2 = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
        parts, pos = [], 0
        for n in (1, 4, nfr - 5):
            parts.append(rx2.ProcessIQData(dI[:, pos * L:(pos + n) * L].contiguous(), dQ[:, pos * L:(pos + n) * L].contiguous()))
            pos += n
        res[name + "_split"] = torch.cat(parts, dim=1).cpu().numpy()
    np.savez(os.path.join(ROOT, "gpurun_out", "agc_pipe_%s.npz" % tag), **res)


def timing(frames=32, reps=int(os.environ.get("T41RX_PROBE_REPS", "30"))):
    import t41_sdr_amd as T
    import torch
    nch = int(os.environ.get("T41RX_PROBE_NCH", "4096"))
    rng = np.random.default_rng(1000)
    nco = (rng.integers(-860, 801, nch) * 50).astype(np.int32)
    rx = T.RxChain(nch, T.default_params(AGCMode=1), NCOFreq=nco)
    g = torch.Generator(device="cuda").manual_seed(0)
    ring = 3
    Is = [(0.2 * torch.randn(nch, frames * L, generator=g, device="cuda")).clamp_(-0.999, 0.999) for _ in range(ring)]
    Qs = [(0.2 * torch.randn(nch, frames * L, generator=g, device="cuda")).clamp_(-0.999, 0.999) for _ in range(ring)]
    out = [torch.empty(nch, frames * L, device="cuda") for _ in range(ring)]
    for k in range(6):
        rx.ProcessIQData(Is[k % ring], Qs[k % ring], out=out[k % ring])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        rx.ProcessIQData(Is[k % ring], Qs[k % ring], out=out[k % ring])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3 / frames
    print(json.dumps({"pipe": os.environ.get("T41RX_AGC_PIPE", "1"), "frames": frames, "us_per_frame": round(us, 2),
                      "frac": round(12 * nch * L / us / 1e3 / 8000.0, 4), "finite": bool(torch.isfinite(out[0]).all())}), flush=True)


def main():
    cmd = sys.argv[1] if len(sys.argv) > 1 else "all"
    if cmd == "out":
        outputs(sys.argv[2])
    elif cmd == "time":
        timing(int(sys.argv[2]) if len(sys.argv) > 2 else 32)
    else:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        for tag, val in (("pipe", "1"), ("barrier", "0")):
            env = dict(os.environ, T41RX_AGC_PIPE=val)
            subprocess.run([sys.executable, os.path.abspath(__file__), "out", tag], env=env, check=True, timeout=600)
        a = np.load(os.path.join(ROOT, "gpurun_out", "agc_pipe_pipe.npz"))
        b = np.load(os.path.join(ROOT, "gpurun_out", "agc_pipe_barrier.npz"))
        for k in a.files:
            same = np.array_equal(a[k], b[k])
            print("%-16s pipe == barrier: %s   finite %s   max|diff| %.3g" % (k, same, bool(np.isfinite(a[k]).all()), float(np.abs(a[k].astype(np.float64) - b[k]).max())), flush=True)
        for k in a.files:
            if k.endswith("_split"):
                print("%-16s split == whole (pipe): %s" % (k, np.array_equal(a[k], a[k[:-6]])), flush=True)
        for tag in ("pipe", "barrier"):
            os.remove(os.path.join(ROOT, "gpurun_out", "agc_pipe_%s.npz" % tag))
        if cmd != "check":
            for val in ("1", "0"):
                for fr in (32, 8):
                    subprocess.run([sys.executable, os.path.abspath(__file__), "time", str(fr)], env=dict(os.environ, T41RX_AGC_PIPE=val), timeout=600)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Round 4: the AGC chain's speculative fast block (rx_kernels.hip, agc_fast_block_s: both candidates computed, state 3's
decay step bracketed by two FMAs) against round 3's fast block, which decides first and takes the decay step through
double precision (-DT41RX_AGC_SPEC=0): outputs and checkpoints must be identical bit for bit over long streams of every
AGC mode, envelope and kernel form.

  tools/build_variant.sh spec0 -DT41RX_AGC_SPEC=0 -DT41RX_AGC_PHASED=0
  python tools/agc_decay_check.py [--streams 24] [--frames 96]         (GPU box: runs itself once per library)
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = 2048
REF = "spec0"  # the build to compare the product with (tools/build_variant.sh spec0 -DT41RX_AGC_SPEC=0 -DT41RX_AGC_PHASED=0: round 3's fast block only, in both chains)


def opt(flag, default):
    a = sys.argv[1:]
    return type(default)(a[a.index(flag) + 1]) if flag in a else default


def worker():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import t41_sdr_amd as T
    streams, frames = opt("--streams", 24), opt("--frames", 96)
    out = []
    for s in range(streams):
        rng = np.random.default_rng(4000 + s)
        nch = int(rng.choice([16, 37, 64, 130]))
        mode = int(rng.choice([0, 0, 1, 2, 3]))
        agc = int(rng.integers(1, 5))
        kw = dict(mode=mode, AGCMode=agc)
        if mode == 1:
            kw.update(FLoCut=-3000, FHiCut=-200)
        if mode == 2:
            kw.update(FLoCut=-3000, FHiCut=3000)
        nco = (rng.integers(-860, 801, nch) * 50).astype(np.int32)
        g = torch.Generator(device="cuda").manual_seed(s)
        # an envelope that rises and falls over the stream: attacks, hangs, slow and fast decays
        n = frames * L
        t = torch.arange(n, device="cuda", dtype=torch.float32) / n
        env = (0.02 + 0.4 * torch.abs(torch.sin(3.0 * np.pi * (1 + s % 5) * t + s))) * (1.0 - 0.9 * (torch.rand(1, generator=g, device="cuda") < 0.5) * (t > 0.6))
        I = (env[None, :] * torch.randn(nch, n, generator=g, device="cuda")).clamp_(-0.999, 0.999)
        Q = (env[None, :] * torch.randn(nch, n, generator=g, device="cuda")).clamp_(-0.999, 0.999)
        rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
        h = hashlib.sha256()
        pos = 0
        for nf in (3, 1, 8, 32, frames - 44):  # barrier form (short calls) and pipelined form
            a = rx.ProcessIQData(I[:, pos * L:(pos + nf) * L].contiguous(), Q[:, pos * L:(pos + nf) * L].contiguous())
            h.update(a.cpu().numpy().tobytes())
            pos += nf
        h.update(rx.get_state().tobytes())
        out.append(h.hexdigest()[:16])
    print(json.dumps(out), flush=True)


def main():
    if "--worker" in sys.argv:
        return worker()
    res = {}
    for name in ("product", REF):
        env = dict(os.environ)
        env.pop("T41RX_LIB", None)
        if name != "product":
            env["T41RX_LIB"] = os.path.join(ROOT, "t41_sdr_amd", "abl", "libt41rx_%s.so" % name)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"] + sys.argv[1:], env=env, capture_output=True, text=True, timeout=1500)
        lines = [l for l in p.stdout.splitlines() if l.startswith("[")]
        if p.returncode != 0 or not lines:
            raise SystemExit("%s failed: %s" % (name, p.stderr[-500:]))
        res[name] = json.loads(lines[0])
    same = [a == b for a, b in zip(res["product"], res[REF])]
    print(json.dumps({"streams": len(same), "identical": sum(same), "mismatching_streams": [i for i, s in enumerate(same) if not s]}))
    if not all(same):
        raise SystemExit(1)


if __name__ == "__main__":
    main()

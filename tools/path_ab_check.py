#!/usr/bin/env python3
"""GPU box: the receive path of the product against another build of the library (t41_sdr_amd/abl/libt41rx_NAME.so), bit for
bit: outputs and checkpoints of the same streams, two calls each (1 frame, then the rest), one process per library --
SSB, LSB with IQ correction and band gain (the non-PLAIN kernels), NFM both ways, AM, SAM, each with the AGC off and on,
q15 samples, FFT_LENGTH 4096 / 1024, the time-major layout.  Round 5 uses it to show that the same-result switches of
rx_device.hpp (addresses, phases, start values, lane writes, register tails) really are that.
usage: python tools/path_ab_check.py NAME"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import hashlib, json, sys
sys.path[:0] = [%r, %r]
import numpy as np, torch
import t41_sdr_amd as T
import siggen
out = {}
CASES = [("usb", dict(mode=0, FLoCut=200, FHiCut=3000), 70, 9),
         ("lsb-iqcorr", dict(mode=1, FLoCut=-3000, FHiCut=-200, IQAmpCorrectionFactor=1.07, IQPhaseCorrectionFactor=0.03, RFgain=6), 33, 7),
         ("usb-agc", dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=2), 70, 9),
         ("nfm", dict(mode=3, FLoCut=-5000, FHiCut=5000), 40, 6),
         ("nfm-atan", dict(mode=3, FLoCut=-5000, FHiCut=5000, nfm_demod=1), 40, 6),
         ("am", dict(mode=2, FLoCut=-3000, FHiCut=3000), 40, 6),
         ("am-agc", dict(mode=2, FLoCut=-3000, FHiCut=3000, AGCMode=4), 40, 6),
         ("sam", dict(mode=8, FLoCut=-3000, FHiCut=3000), 20, 14),
         ("sam-agc", dict(mode=8, FLoCut=-3000, FHiCut=3000, AGCMode=1), 20, 14),
         ("fft4096", dict(mode=0, FLoCut=200, FHiCut=3000, fft_length=4096), 12, 16),
         ("fft1024-agc", dict(mode=0, FLoCut=200, FHiCut=3000, fft_length=1024, AGCMode=3), 12, 8)]
for name, kw, nch, nfr in CASES:
    try:
        p = T.default_params(**kw)
    except Exception as e:
        out[name] = "params: %%s" %% e
        continue
    nco = siggen.nco_grid(nch, seed=7)
    I, Q = siggen.make_iq(nch, nfr * 2048, nco, mode=min(kw["mode"], 3) if kw["mode"] != 8 else 2, seed=31)
    rx = T.RxChain(nch, p, NCOFreq=nco)
    h = hashlib.sha256()
    seg = p.fft_length // 512
    for sl in (slice(0, seg * 2048), slice(seg * 2048, nfr * 2048)):
        o = rx.ProcessIQData(torch.from_numpy(I[:, sl].copy()).cuda(), torch.from_numpy(Q[:, sl].copy()).cuda())
        h.update(o.cpu().numpy().tobytes())
    h.update(np.asarray(rx.get_state()).tobytes())
    out[name] = h.hexdigest()[:16]
print(json.dumps(out))
'''


def run(lib):
    env = dict(os.environ)
    env.pop("T41RX_LIB", None)
    if lib:
        env["T41RX_LIB"] = os.path.join(ROOT, "t41_sdr_amd", "abl", "libt41rx_%s.so" % lib)
    p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "tests"))], env=env, capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        raise SystemExit(p.stderr[-3000:])
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def main():
    other = sys.argv[1]
    a, b = run(None), run(other)
    diff = sorted(k for k in a if a[k] != b.get(k))
    print(json.dumps({"product": a, other: b, "identical": not diff, "differ": diff}))
    raise SystemExit(0 if not diff else 1)


if __name__ == "__main__":
    main()

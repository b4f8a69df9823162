#!/usr/bin/env python3
"""GPU box: the noise-reduction / notch kernels of the product against another build of the library (t41_sdr_amd/abl/libt41rx_NAME.so,
e.g. an experimental nr_kernels.hip linked with the product's other objects), bit for bit: outputs and checkpoints of the
same streams (notch, LMS + notch, LMS, notch behind AM + AGC, Kim, spectral, spectral + notch behind AM + AGC; two calls each), one process per library.
usage: python tools/anr_ab_check.py NAME"""
import hashlib
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
for name, kw, nch, nfr in [("notch", dict(ANR_notchOn=1), 37, 9), ("lms+notch", dict(nrOptionSelect=3, ANR_notchOn=1), 16, 6),
                           ("lms", dict(nrOptionSelect=3), 50, 5), ("notch-am-agc", dict(ANR_notchOn=1, mode=2, FLoCut=-3000, FHiCut=3000, AGCMode=2), 33, 8),
                           ("kim", dict(nrOptionSelect=1), 21, 12), ("spectral", dict(nrOptionSelect=2), 40, 40),
                           ("spectral-wide", dict(nrOptionSelect=2, FLoCut=200, FHiCut=9000), 24, 30),
                           ("spectral-am-agc-notch", dict(nrOptionSelect=2, ANR_notchOn=1, mode=2, FLoCut=-3000, FHiCut=3000, AGCMode=3), 17, 30)]:
    p = dict(mode=0, FLoCut=200, FHiCut=3000); p.update(kw)
    nco = siggen.nco_grid(nch, seed=7)
    I, Q = siggen.make_iq(nch, 2 * nfr * 2048, nco, mode=min(p["mode"], 3), seed=31)
    rx = T.RxChain(nch, T.default_params(**p), NCOFreq=nco)
    h = hashlib.sha256()
    for call in range(2):
        sl = slice(call * nfr * 2048, (call + 1) * nfr * 2048)
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
    p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "tests"))], env=env, capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        raise SystemExit(p.stderr[-2000:])
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def main():
    other = sys.argv[1] if len(sys.argv) > 1 else "anr0"
    a, b = run(None), run(other)
    print(json.dumps({"product": a, other: b, "identical": a == b}))
    raise SystemExit(0 if a == b else 1)


if __name__ == "__main__":
    main()

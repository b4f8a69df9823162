#!/usr/bin/env python3
"""GPU box: soak of the receive path of the product against another build of the library
(t41_sdr_amd/abl/libt41rx_NAME.so), bit for bit, on random shapes and parameters: each library runs the same seeded
sequence of cases in its own process for SECONDS, the outputs' and checkpoints' hashes are compared case by case.
Cases: USB / LSB / AM / NFM (both discriminators) / SAM, AGCMode 0..4, unit and non-unit gains with and without the IQ
correction (the PLAIN and the general kernels), FFT_LENGTH 512 / 1024 / 2048 / 4096, 1..300 channels, two calls of 1..40
frames (short calls take the barrier forms, long ones the pipelined ones).  Round 5: the product against the build with
every same-result switch of rx_device.hpp set back (abl/libt41rx_r5base.so).
usage: python tools/path_soak.py NAME [SECONDS]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import hashlib, json, sys, time
sys.path[:0] = [%r, %r]
import numpy as np, torch
import t41_sdr_amd as T
import siggen
budget, ncases = %f, %d
rng = np.random.default_rng(20261005)
out, t0 = [], time.time()
for case in range(ncases):
    mode = int(rng.choice([0, 1, 2, 3, 8]))
    fft = int(rng.choice([512, 512, 512, 1024, 2048, 4096]))
    if mode == 0: lo, hi = 200, int(rng.choice([2400, 3000, 4000]))
    elif mode == 1: lo, hi = -int(rng.choice([2400, 3000, 4000])), -200
    else: lo, hi = -int(rng.choice([3000, 5000])), int(rng.choice([3000, 5000]))
    kw = dict(mode=mode, FLoCut=lo, FHiCut=hi, fft_length=fft, AGCMode=int(rng.integers(0, 5)))
    if mode == 3: kw["nfm_demod"] = int(rng.integers(0, 2))
    if rng.integers(0, 2):
        kw.update(RFgain=int(rng.choice([-6, 3, 10])), IQAmpCorrectionFactor=float(rng.choice([1.0, 1.04, 0.97])), IQPhaseCorrectionFactor=float(rng.choice([0.0, 0.02, -0.03])))
    seg = kw['fft_length'] // 512
    nch, n1, n2 = int(rng.choice([1, 2, 3, 5, 15, 16, 17, 31, 33, 64, 100, 255, 300])), seg * int(rng.integers(1, 41 // seg + 1)), seg * int(rng.integers(1, 41 // seg + 1))
    seed = int(rng.integers(1 << 30))
    if time.time() - t0 > budget:
        break
    try:
        p = T.default_params(**kw)
        nco = siggen.nco_grid(nch, seed=seed & 0xffff)
        I, Q = siggen.make_iq(nch, (n1 + n2) * 2048, nco, mode=(2 if mode == 8 else mode), seed=seed)
        rx = T.RxChain(nch, p, NCOFreq=nco)
        h = hashlib.sha256()
        for sl in (slice(0, n1 * 2048), slice(n1 * 2048, (n1 + n2) * 2048)):
            o = rx.ProcessIQData(torch.from_numpy(I[:, sl].copy()).cuda(), torch.from_numpy(Q[:, sl].copy()).cuda())
            h.update(o.cpu().numpy().tobytes())
        h.update(np.asarray(rx.get_state()).tobytes())
        out.append([case, kw, nch, n1, n2, h.hexdigest()[:16]])
    except Exception as e:
        out.append([case, kw, nch, n1, n2, "error: %%s" %% str(e)[:80]])
print(json.dumps(out))
'''


def run(lib, budget, ncases):
    env = dict(os.environ)
    env.pop("T41RX_LIB", None)
    if lib:
        env["T41RX_LIB"] = os.path.join(ROOT, "t41_sdr_amd", "abl", "libt41rx_%s.so" % lib)
    p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "tests"), budget, ncases)], env=env, capture_output=True, text=True,
                       timeout=budget + 300)
    if p.returncode != 0:
        raise SystemExit(p.stderr[-3000:])
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("[")][-1])


def main():
    other = sys.argv[1]
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    a = run(None, budget, 100000)
    b = run(other, 10 * budget, len(a))  # (the same cases: the second run is bounded by their number, not by the clock)
    n = min(len(a), len(b))
    bad = [(x, y) for x, y in zip(a[:n], b[:n]) if x != y]
    errs = sum(1 for x in a[:n] if str(x[-1]).startswith("error"))
    kinds = {}
    for x in a[:n]:
        if str(x[-1]).startswith("error"):
            kinds[x[-1]] = kinds.get(x[-1], 0) + 1
    print(json.dumps({"cases": n, "cases_that_raised": errs, "what_they_raised": kinds, "mismatches": len(bad), "first_mismatches": bad[:3]}))
    raise SystemExit(0 if not bad else 1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU box: soak of the noise-reduction / notch kernels of the product against another build of the library
(t41_sdr_amd/abl/libt41rx_NAME.so), bit for bit, on random shapes and parameters: each library runs the same seeded
sequence of cases in its own process for SECONDS, the outputs' and checkpoints' hashes are compared case by case.
Cases: nrOptionSelect 0..3 x ANR_notchOn, filters from 2.4 to 9.5 kHz wide (above 6 kHz the spectral function's bin loop
runs over both of a lane's bins), USB / LSB / AM, AGC off or on, 1..70 channels, two calls of 1..12 frames.
usage: python tools/nr_soak.py NAME [SECONDS]"""
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
    nr = int(rng.integers(0, 4)); notch = int(rng.integers(0, 2))
    if nr == 0 and notch == 0: notch = 1
    width = int(rng.choice([2400, 2800, 3000, 4000, 5500, 6500, 8000, 9500]))
    mode = int(rng.choice([0, 1, 2]))
    lo, hi = (200, 200 + width) if mode == 0 else ((-200 - width, -200) if mode == 1 else (-width // 2 - 1500, width // 2 + 1500))
    if mode == 2: lo, hi = -min(width, 5000), min(width, 5000)
    kw = dict(mode=mode, FLoCut=lo, FHiCut=hi, nrOptionSelect=nr, ANR_notchOn=notch, AGCMode=int(rng.choice([0, 0, 1, 3])))
    nch, n1, n2 = int(rng.integers(1, 71)), int(rng.integers(1, 13)), int(rng.integers(1, 13))
    seed = int(rng.integers(1 << 30))
    if time.time() - t0 > budget:
        break
    try:
        p = T.default_params(**kw)
        nco = siggen.nco_grid(nch, seed=seed & 0xffff)
        I, Q = siggen.make_iq(nch, (n1 + n2) * 2048, nco, mode=mode, seed=seed)
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
    print(json.dumps({"cases": n, "cases_that_raised": errs, "mismatches": len(bad), "first_mismatches": bad[:3]}))
    raise SystemExit(0 if not bad else 1)


if __name__ == "__main__":
    main()

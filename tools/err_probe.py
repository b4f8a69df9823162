"""error of the random-parameter parity cases, per case (GPU box): python tools/err_probe.py [case indices]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as G
import siggen
import t41_sdr_amd as T
cases = G._random_cases(16, seed=2026)
sel = [int(x) for x in sys.argv[1:]] or range(len(cases))
L = 2048
for i in sel:
    kw = cases[i]
    nch, nfr = 7, 6
    nco = siggen.nco_grid(nch, seed=kw["FHiCut"] & 1023)
    mode = kw["mode"]
    side = 0
    if kw["xmtMode"] == 1:
        side = kw["CWFreqShift"] if mode == 1 else (-kw["CWFreqShift"] if mode == 0 else 0)
    if mode == 3:
        I, Q = siggen.make_fm(nch, nfr * L, nco + side, seed=kw["FLoCut"] & 255)
    else:
        lo, hi = (abs(kw["FHiCut"]), abs(kw["FLoCut"])) if mode == 1 else (max(kw["FLoCut"], 100), kw["FHiCut"])
        band = (lo + 0.15 * (hi - lo), lo + 0.85 * (hi - lo))
        I, Q = siggen.make_iq(nch, nfr * L, nco + side, mode=mode, seed=kw["audioVolume"], audio_hz=band)
    I, Q = siggen.fade(I, Q, [(0.5, 1.0), (0.5, 0.2)])
    got, _ = G.gpu_run(T, kw, nco, I, Q)
    ref = G.oracle_run(kw, nco, I, Q)
    err = siggen.block_rel_err(got, ref, L)
    print("case %2d mode %d agc %d: max %.3e  per-frame max over channels: %s" % (i, mode, kw["AGCMode"], err.max(), " ".join("%.1e" % e for e in err.max(axis=0))), flush=True)

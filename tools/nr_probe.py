"""GPU box: noise reduction / notch, HIP path vs oracle, per-frame block-relative error (sets the stated tolerances of
tests/test_noise_reduction.py).  usage: python tools/nr_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O  # noqa: E402
import siggen  # noqa: E402
import t41_sdr_amd as T  # noqa: E402
import torch  # noqa: E402

L = 2048
for kw in (dict(ANR_notchOn=1), dict(nrOptionSelect=3), dict(nrOptionSelect=3, ANR_notchOn=1), dict(nrOptionSelect=1),
           dict(nrOptionSelect=2), dict(nrOptionSelect=1, ANR_notchOn=1), dict(nrOptionSelect=2, mode=2, FLoCut=-3000, FHiCut=3000),
           dict(nrOptionSelect=1, AGCMode=2)):
    nch, nfr = 70, 40
    nco = siggen.nco_grid(nch, seed=5)
    I, Q = siggen.make_iq(nch, nfr * L, nco, mode=kw.get("mode", 0), seed=50)
    try:
        rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
        got = rx.ProcessIQData(torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()).cpu().numpy()
    except Exception as e:  # noqa: BLE001
        print(kw, "FAILED:", e)
        continue
    ref = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32)).process(I, Q, nthreads=8)
    e = siggen.block_rel_err(got, ref, L)
    print(kw, "finite", bool(np.isfinite(got).all()), "max over channels per frame:")
    print("   ", " ".join("%.1e" % v for v in e.max(axis=0)))
    print("    median per frame:", " ".join("%.1e" % v for v in np.median(e, axis=0)))
    # split == whole
    rx2 = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    parts, pos = [], 0
    for n in (1, 7, 12, 20):
        parts.append(rx2.ProcessIQData(dI[:, pos * L:(pos + n) * L].contiguous(), dQ[:, pos * L:(pos + n) * L].contiguous()))
        pos += n
    b = torch.cat(parts, dim=1).cpu().numpy()
    print("    split == whole:", bool(np.array_equal(b, got)), np.abs(b - got).max())

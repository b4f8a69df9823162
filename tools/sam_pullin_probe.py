"""GPU box: the synchronous detector's pull-in, HIP path vs oracle frame by frame (sets the stated bound of
tests/test_sam.py::test_gpu_sam_pull_in_bound).  usage: python tools/sam_pullin_probe.py"""
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
for agc in (0, 2):
    kw = dict(mode=8, FLoCut=-3000, FHiCut=3000, AGCMode=agc)
    nch, nfr = 64, 16
    nco = siggen.nco_grid(nch, seed=21)
    I, Q = siggen.make_am_carrier(nch, nfr * L, nco, seed=40 + agc)
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    got = rx.ProcessIQData(torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()).cpu().numpy()
    ref = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32)).process(I, Q, nthreads=8)
    d = np.abs(got.astype(np.float64) - ref).reshape(nch, nfr, L).max(axis=2)
    lvl = np.abs(ref[:, 12 * L:]).max(axis=1, keepdims=True)  # the locked audio level of the channel
    e = d / lvl
    print("AGCMode", agc, "per-frame max over channels of max|gpu - ref| / locked level:")
    print(" ".join("%.1e" % v for v in e.max(axis=0)))
    print("median over channels:")
    print(" ".join("%.1e" % v for v in np.median(e, axis=0)))

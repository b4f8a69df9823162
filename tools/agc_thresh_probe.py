#!/usr/bin/env python3
"""GPU box: where does the HIP path leave the oracle when AGC_thresh is not the default / changes mid-stream?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O, siggen
import t41_sdr_amd as T
L = 2048
nch, nfr, cut = 37, 16, 6
nco = siggen.nco_grid(nch, seed=91)
I, Q = siggen.make_iq(nch, nfr * L, nco, mode=0, seed=92)
I, Q = siggen.fade(I, Q, [(0.25, 2.5), (0.45, 0.004), (0.3, 1.5)])
np.set_printoptions(linewidth=250, precision=2)
for agcmode in (1, 3):
    for name, t0, t1 in (("90 throughout", 90, 90), ("-20 throughout", -20, -20), ("20 throughout", 20, 20), ("90 -> -20", 90, -20), ("60 -> 0", 60, 0)):
        kw = dict(mode=0, AGCMode=agcmode, AGC_thresh=t0)
        rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
        dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
        c = cut * L
        a = rx.ProcessIQData(dI[:, :c].contiguous(), dQ[:, :c].contiguous())
        if t1 != t0:
            rx.CalcFilters(AGC_thresh=t1)
        b = rx.ProcessIQData(dI[:, c:].contiguous(), dQ[:, c:].contiguous())
        got = torch.cat([a, b], 1).cpu().numpy()
        ob = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32))
        r1 = ob.process(np.ascontiguousarray(I[:, :c]), np.ascontiguousarray(Q[:, :c]))
        if t1 != t0:
            ob.p.AGC_thresh = t1
            ob.redesign()
        r2 = ob.process(np.ascontiguousarray(I[:, c:]), np.ascontiguousarray(Q[:, c:]))
        ref = np.concatenate([r1, r2], 1)
        err = siggen.block_rel_err(got, ref, L)
        lvl = np.abs(ref).reshape(nch, nfr, L).max(axis=2)
        ch, fr = np.unravel_index(err.argmax(), err.shape)
        print("AGCMode %d, AGC_thresh %s: max err %.2e at channel %d frame %d (ref level there %.2e); per-frame max over channels:" % (agcmode, name, err.max(), ch, fr, lvl[ch, fr]))
        print("   ", err.max(axis=0))
        print("    ref level, channel %d:" % ch, lvl[ch])
        d = np.abs(got[ch, fr * L:(fr + 1) * L] - ref[ch, fr * L:(fr + 1) * L])
        k = int(d.argmax())
        print("    worst sample %d of the frame: got %.6e ref %.6e" % (k, got[ch, fr * L + k], ref[ch, fr * L + k]))

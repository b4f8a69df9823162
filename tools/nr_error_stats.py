import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import numpy as np, torch
import siggen, oracle_lib as O
import test_noise_reduction as TN
import t41_sdr_amd as T
L, D = 2048, 256
for name in ("spectral", "spectral-am-agc", "kim"):
    kw, tol, how = TN.NR_CASES[name]
    nch, nfr = 70, 36
    nco = siggen.nco_grid(nch, seed=5)
    I, Q = siggen.make_iq(nch, nfr * L, nco, mode=kw.get("mode", 0), seed=50)
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    got, pre, pos = [], [], 0
    for n in (1, 15, 20):
        tap = torch.zeros(nch, n * D, device="cuda")
        rx.set_debug_taps(demod=tap)
        got.append(rx.ProcessIQData(dI[:, pos * L:(pos + n) * L].contiguous(), dQ[:, pos * L:(pos + n) * L].contiguous()))
        pre.append(tap)
        pos += n
    got, pre = torch.cat(got, dim=1).cpu().numpy(), torch.cat(pre, dim=1).cpu().numpy()
    want = TN._oracle_stage_and_interpolators(pre, kw)
    e = siggen.block_rel_err(got, want, L)
    print(name, "frac<1e-3 %.4f  frac<1e-4 %.4f frac<1e-5 %.4f median %.2e max %.2e  p99 %.2e" % ((e < 1e-3).mean(), (e < 1e-4).mean(), (e < 1e-5).mean(), np.median(e), e.max(), np.percentile(e, 99)))
for name in ("kim", "spectral", "lms", "notch-late"):
    kw = dict(TN.NR_CASES[name.replace("-late", "")][0])
    late = name.endswith("-late")
    nch, nfr = 12, 28
    nco = siggen.nco_grid(nch, seed=6)
    I, Q = siggen.make_iq(nch, nfr * L, nco, mode=0, seed=60)
    start = dict(kw, ANR_notchOn=0) if late else kw
    rx = T.RxChain(nch, T.default_params(**start), NCOFreq=nco)
    ob = O.OracleBatch(O.default_params(**start), np.asarray(nco, np.int32))
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    a = rx.ProcessIQData(dI[:, :4 * L].contiguous(), dQ[:, :4 * L].contiguous()).cpu().numpy()
    ra = ob.process(I[:, :4 * L], Q[:, :4 * L])
    if late:
        rx.CalcFilters(ANR_notchOn=1)
        ob.p.ANR_notchOn = 1
    b = rx.ProcessIQData(dI[:, 4 * L:].contiguous(), dQ[:, 4 * L:].contiguous()).cpu().numpy()
    rb = ob.process(I[:, 4 * L:], Q[:, 4 * L:])
    e = siggen.block_rel_err(np.concatenate([a, b], axis=1), np.concatenate([ra, rb], axis=1), L)
    print("whole path", name, "frac<1e-3 %.4f frac<1e-4 %.4f frac<1e-5 %.4f median %.2e max %.2e" % ((e < 1e-3).mean(), (e < 1e-4).mean(), (e < 1e-5).mean(), np.median(e), e.max()))

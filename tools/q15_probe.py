"""where the q15 entry and the f32 entry on the same samples differ (GPU box): python tools/q15_probe.py mode [agc]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import siggen
import t41_sdr_amd as T
L = 2048
mode = int(sys.argv[1]); agc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kw = dict(mode=mode, FLoCut=-3000 if mode == 2 else 200, FHiCut=3000, audioVolume=100, AGCMode=agc)
nch, nfr = 10, 5
nco = siggen.nco_grid(nch, seed=9)
I, Q = siggen.make_iq(nch, nfr * L, nco, mode=mode, seed=6)
to_q15 = lambda x: np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)
qI, qQ = to_q15(I), to_q15(Q)
fI, fQ = qI.astype(np.float32) / np.float32(32768), qQ.astype(np.float32) / np.float32(32768)
def run_q(fresh=True):
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    g = rx.ProcessIQData_q15(torch.from_numpy(qQ).cuda(), torch.from_numpy(qI).cuda()); torch.cuda.synchronize()
    return g.cpu().numpy()
def run_f():
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    f = rx.ProcessIQData(torch.from_numpy(fI).cuda(), torch.from_numpy(fQ).cuda()).cpu().numpy()
    return f
g1, g2 = run_q(), run_q()
f1, f2 = run_f(), run_f()
print("q15 run-to-run identical:", np.array_equal(g1, g2), " f32 run-to-run identical:", np.array_equal(f1, f2))
want = np.clip(np.trunc(f1.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)
d = (g1 != want).reshape(nch, nfr, L)
print("differing samples per (channel, frame):"); print(d.sum(axis=2))
idx = np.argwhere(g1 != want)[:10]
for c, n in idx:
    print("ch %d n %d (frame %d, off %d): q15 %d  f32*32768 %.4f" % (c, n, n // L, n % L, g1[c, n], f1[c, n] * 32768.0))

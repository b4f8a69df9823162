"""Per-phase cycles of the long-FFT fast-convolution kernel from the -DT41RX_STAMP build (GPU box).
usage: T41RX_LIB=.../libt41rx_stamp.so python tools/fc_stamps.py [nchan] [frames4k] [waves per channel]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import t41_sdr_amd as T  # noqa: E402

NAMES = ["tail of the previous frame", "barrier (array free)", "assemble: loads -> LDS", "mask loads issued", "barrier",
         "pass 1", "barrier", "pass 2 (2 x fft512 pairs)", "barrier", "pass 3", "barrier", "x2 interpolator", "barrier",
         "x4 + transposition + stores (last frame)"]


def main():
    nch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    nwv = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    fft = 4096
    L, D = 4 * fft * nfr, 8 * 256 * nfr
    rx = T.RxChain(nch, T.default_params(fft_length=fft, FLoCut=400, FHiCut=600), NCOFreq=np.full(nch, 5000, np.int32))
    g = torch.Generator(device="cuda").manual_seed(0)
    I = 0.2 * torch.randn(nch, L, generator=g, device="cuda")
    Q = 0.2 * torch.randn(nch, L, generator=g, device="cuda")
    buf = torch.zeros(nch * D + 2 * nch * 64 * (1 + nwv), device="cuda")
    rx.set_debug_taps(None, None, buf)
    for _ in range(3):
        rx.ProcessIQData(I, Q)
    torch.cuda.synchronize()
    st = buf[nch * D:].view(torch.int64)[nch * 64:].view(nch, nwv, 64).cpu().numpy().astype(np.float64)
    ph = st[:, :, :len(NAMES)] / nfr
    tot = ph.sum(axis=2)
    print("%d channels x %d frames: cycles per wave-frame mean %.0f (min %.0f max %.0f)" % (nch, nfr, tot.mean(), tot.min(), tot.max()))
    for p, name in enumerate(NAMES):
        print("  %2d %-42s %8.0f cycles %5.1f %%" % (p, name, ph[:, :, p].mean(), 100 * ph[:, :, p].mean() / tot.mean()))
    t0, t1 = st[:, :, 28], st[:, :, 29]
    base = t0.min()
    q = lambda x: " ".join("%.1f" % v for v in np.percentile(x, [0, 10, 50, 90, 100]))
    print("workgroup start [us, percentiles 0 10 50 90 100]: " + q((t0[:, 0] - base) / 100.0))
    print("workgroup end                                    : " + q((t1[:, 0] - base) / 100.0))
    print("workgroup life                                   : " + q((t1[:, 0] - t0[:, 0]) / 100.0))
    hw = st[:, 0, 27].astype(np.int64)
    xcc, cu, se, sh = (hw >> 32) & 15, (hw >> 8) & 15, (hw >> 13) & 7, (hw >> 12) & 1
    life = (t1[:, 0] - t0[:, 0]) / 100.0
    for x in np.unique(xcc):
        m = xcc == x
        cus = np.unique(((se << 5) | (sh << 4) | cu)[m])
        per = [int((m & (((se << 5) | (sh << 4) | cu) == c)).sum()) for c in cus]
        print("  XCC %d: %4d workgroups on %2d CUs (per CU min %d max %d), life mean %.1f min %.1f max %.1f" % (x, m.sum(), len(cus), min(per), max(per), life[m].mean(), life[m].min(), life[m].max()))
    key = (xcc << 8) | (se << 5) | (sh << 4) | cu
    cnt = {k: int((key == k).sum()) for k in np.unique(key)}
    for n in sorted(set(cnt.values())):
        ks = [k for k in cnt if cnt[k] == n]
        m = np.isin(key, ks)
        print("  CUs holding %d workgroups: %3d, life of their workgroups mean %.1f max %.1f" % (n, len(ks), life[m].mean(), life[m].max()))
    simd = (st[:, :, 27].astype(np.int64) >> 4) & 3
    same = (simd == simd[:, :1]).all(axis=1)
    print("  workgroups with all waves on one SIMD: %d; life %.1f vs %.1f" % (same.sum(), life[same].mean() if same.any() else 0, life[~same].mean() if (~same).any() else 0))
    late = (t0[:, 0] - base) / 100.0 > 5.0
    if late.any():
        print("second-round workgroups: %d, life %s" % (late.sum(), q((t1[late, 0] - t0[late, 0]) / 100.0)))
        print("first-round  workgroups: %d, life %s" % ((~late).sum(), q((t1[~late, 0] - t0[~late, 0]) / 100.0)))


if __name__ == "__main__":
    main()

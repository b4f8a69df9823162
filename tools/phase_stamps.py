"""Per-phase cycle shares of the RX kernel from the -DT41RX_STAMP diagnostic build (GPU box).
usage: T41RX_LIB=.../libt41rx_stamp.so python tools/phase_stamps.py [nchan] [AGCMode]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import t41_sdr_amd as T  # noqa: E402

NAMES = ["wait loads + gain", "DC high-pass", "NCO + mix", "LDS stage + /4 FIR", "history rolls", "/2 FIR",
         "state save + assemble", "twiddle loads", "forward FFT", "mask + inverse FFT", "demod + x2 staging",
         "x2 interpolator", "x4 interp + transpose writes", "transposed reads + stores",
         "first sub-block: wait loads", "prologue d: uniformise state + DC prepass",
         "prologue a: issue loads + SMEM gains", "prologue b: table staging + barrier (first vmcnt wait)",
         "prologue c: delay lines -> LDS",
         "AGC: magnitudes + look-ahead max", "AGC: barrier 1", "AGC: serial chain (chain wave)",
         "AGC: barrier 2 (others wait for the chain)", "AGC: gain, scaling, record store",
         "AGC chain: straight-line blocks", "AGC chain: slow blocks", "AGC chain: NUMBER of slow blocks (not cycles)"]


def main():
    nch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    agc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    fft = int(sys.argv[3]) if len(sys.argv) > 3 else 512  # 4096: stamps of the front kernel (8 segments)
    nfr = int(sys.argv[4]) if len(sys.argv) > 4 else 1    # frames per launch (cycles are then summed over them)
    L, D = 4 * fft * nfr, (256 if fft == 512 else 8 * 256) * nfr
    kw = dict(AGCMode=agc) if fft == 512 else dict(fft_length=4096, FLoCut=400, FHiCut=600)
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=np.full(nch, 5000, np.int32))
    g = torch.Generator(device="cuda").manual_seed(0)
    I = 0.2 * torch.randn(nch, L, generator=g, device="cuda")
    Q = 0.2 * torch.randn(nch, L, generator=g, device="cuda")
    if os.environ.get("T41RX_STAMP_SIGNAL") == "bench" and fft == 512:  # bench.py's tones + noise instead of noise
        import bench
        Is, Qs = bench.synth_ring(nch, np.full(nch, 5000), 4, torch.device("cuda"), seed=1)
        for k in range(3):
            rx.ProcessIQData(Is[k], Qs[k])
        I, Q = Is[3], Qs[3]
    buf = torch.zeros(nch * D + 2 * nch * 64, device="cuda")  # demod tap | uint64 stamps
    rx.set_debug_taps(None, None, buf)
    for _ in range(3):  # steady state (no start-up transient), warm caches
        rx.ProcessIQData(I, Q)
    torch.cuda.synchronize()
    st = buf[nch * D:].view(torch.int64).view(nch, 64).cpu().numpy().astype(np.float64)
    tot = st[:, :len(NAMES)].sum(axis=1)
    st[:, :len(NAMES)] /= nfr
    tot /= nfr
    print("channels %d x %d frames per launch: mean wave cycles per frame %.0f (min %.0f max %.0f)" % (nch, nfr, tot.mean(), tot.min(), tot.max()))
    for p, name in enumerate(NAMES):
        print("  %2d %-30s %8.0f cycles  %5.1f %%" % (p, name, st[:, p].mean(), 100 * st[:, p].mean() / tot.mean()))


    t0, t1 = st[:, 28], st[:, 29]  # s_memrealtime (100 MHz) at wave start / end
    if t0.any():
        base = t0.min()
        q = lambda x: " ".join("%.2f" % v for v in np.percentile((x - base) / 100.0, [0, 10, 50, 90, 100]))
        print("wave start  [us after the first wave, percentiles 0 10 50 90 100]: " + q(t0))
        print("wave end    [us after the first wave, percentiles 0 10 50 90 100]: " + q(t1))
        print("wave life   [us, percentiles]: " + " ".join("%.2f" % v for v in np.percentile((t1 - t0) / 100.0, [0, 10, 50, 90, 100])))
    hw = st[:, 27].astype(np.int64)
    if hw.any() and t0.any():
        xcc = (hw >> 32) & 15
        cuid = ((hw >> 8) & 0xff) | (xcc << 8)  # CU_ID, SH, SE + XCC
        ids = np.unique(cuid)
        ends = np.array([(t1[cuid == i].max() - base) / 100.0 for i in ids])
        firsts = np.array([(t1[cuid == i].min() - base) / 100.0 for i in ids])
        nw = np.array([(cuid == i).sum() for i in ids])
        print("CUs seen: %d (waves per CU min %d max %d)" % (len(ids), nw.min(), nw.max()))
        print("per-CU LAST wave end [us]: percentiles 0 10 50 90 100: " + " ".join("%.2f" % v for v in np.percentile(ends, [0, 10, 50, 90, 100])))
        print("per-CU FIRST wave end [us]: percentiles 0 10 50 90 100: " + " ".join("%.2f" % v for v in np.percentile(firsts, [0, 10, 50, 90, 100])))
        for x in range(8):
            m = xcc[np.searchsorted(ids, ids)] if False else None
        for x in np.unique(xcc):
            sel = (ids >> 8) == x
            print("  XCC %d: %3d CUs, last-wave end mean %.2f max %.2f us" % (x, sel.sum(), ends[sel].mean(), ends[sel].max()))
    if hw.any():  # HW_ID: WAVE_ID [3:0], SIMD_ID [5:4], PIPE [7:6], CU_ID [11:8], SH [12], SE [15:13] (gfx9 layout)
        wave, simd, cu, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7
        xcc = (hw >> 16) & 0xffff
        print("placement of the first 32 waves (channel: simd/wave_id/cu/se/hi16):")
        print("  " + " ".join("%d:%d/%d/%d/%d/%x" % (c, simd[c], wave[c], cu[c], se[c], xcc[c]) for c in range(32)))
        for wg in (0, 1, 256, 257, 512):
            c = 4 * wg
            if c + 3 < nch:
                print("  workgroup %4d: simd of waves 0..3 = %s, cu %s, se %s, hi16 %s" % (wg, simd[c:c + 4], cu[c:c + 4], se[c:c + 4], xcc[c:c+4]))
        key = (hw >> 8)  # everything above simd/wave = the CU identity
        from collections import Counter
        per_cu = Counter(key.tolist())
        print("  distinct CU ids %d, waves per CU id: min %d max %d" % (len(per_cu), min(per_cu.values()), max(per_cu.values())))
        # chain waves (wave 0 of each workgroup): how many share a (cu, simd)?
        k0 = Counter(((hw[0::4] >> 8) * 4 + simd[0::4]).tolist())
        print("  wave 0 of every workgroup: (cu, simd) slots used %d, max sharing %d" % (len(k0), max(k0.values())))


if __name__ == "__main__":
    main()

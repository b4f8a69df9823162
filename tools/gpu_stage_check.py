"""Stage-by-stage GPU-vs-oracle comparison (debug aid; run on the GPU box).
usage: python tools/gpu_stage_check.py [nchan] [nframes] [mode]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import siggen  # noqa: E402
import t41_sdr_amd as T  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def main():
    nchan = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    flo, fhi = (-3000, -200) if mode == 1 else (200, 3000)
    N, L, D = 512, 2048, 256
    nco = siggen.nco_grid(nchan)
    I, Q = siggen.make_iq(nchan, nfr * L, nco, mode=mode)
    p = T.default_params(mode=mode, FLoCut=flo, FHiCut=fhi)
    rx = T.RxChain(nchan, p, NCOFreq=nco)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    t_nco = torch.zeros(nchan, 2 * L, device="cuda")  # one frame per call below
    t_dec = torch.zeros(nchan, N, device="cuda")
    t_dem = torch.zeros(nchan, D, device="cuda")
    rx.set_debug_taps(t_nco, t_dec, t_dem)
    out = torch.empty_like(dI)
    # frame by frame so the oracle taps (last frame only) line up
    po = O.default_params(mode=mode, FLoCut=flo, FHiCut=fhi)
    ob = O.OracleBatch(po, nco)
    for f in range(nfr):
        sl = slice(f * L, (f + 1) * L)
        o1 = rx.ProcessIQData(dI[:, sl].contiguous(), dQ[:, sl].contiguous())
        torch.cuda.synchronize()
        out[:, sl] = o1
        ref = ob.process(I[:, sl], Q[:, sl])
        g_nco = t_nco.cpu().numpy()[:, :2 * L]
        g_dec = t_dec.cpu().numpy()[:, :N]
        g_dem = t_dem.cpu().numpy()[:, :D]
        worst = {"nco": 0, "dec": 0, "demod": 0, "out": 0}
        for c in range(nchan):
            rn = np.concatenate([ob.tap(c, O.TAP_POST_NCO_I, L), ob.tap(c, O.TAP_POST_NCO_Q, L)])
            rd = np.concatenate([ob.tap(c, O.TAP_DEC_I, D), ob.tap(c, O.TAP_DEC_Q, D)])
            rm = ob.tap(c, O.TAP_DEMOD, D)
            worst["nco"] = max(worst["nco"], rel(g_nco[c], rn))
            worst["dec"] = max(worst["dec"], rel(g_dec[c], rd))
            worst["demod"] = max(worst["demod"], rel(g_dem[c], rm))
            worst["out"] = max(worst["out"], rel(o1.cpu().numpy()[c], ref[c]))
        print("frame %d  worst block-rel err: " % f + "  ".join("%s=%.3e" % kv for kv in worst.items()), flush=True)
        if f == 0:
            c = 0
            rn = ob.tap(c, O.TAP_POST_NCO_I, L)
            print("  ch0 post-nco I ref[:6]", rn[:6], "gpu", g_nco[c][:6])
            rd = ob.tap(c, O.TAP_DEC_I, D)
            print("  ch0 dec I ref[:6]", rd[:6], "gpu", g_dec[c][:6])
            rm = ob.tap(c, O.TAP_DEMOD, D)
            print("  ch0 demod ref[:6]", rm[:6], "gpu", g_dem[c][:6])
            print("  ch0 out ref[:6]", ref[c][:6], "gpu", o1.cpu().numpy()[c][:6])
    # multi-frame single call from reset must equal frame-by-frame
    rx.reset()
    rx.set_debug_taps(None, None, None)
    out2 = rx.ProcessIQData(dI, dQ)
    torch.cuda.synchronize()
    print("multi-frame call vs frame-by-frame: max abs diff", float((out2 - out).abs().max()))


if __name__ == "__main__":
    main()

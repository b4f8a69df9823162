import sys, os, glob
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import numpy as np, torch
import siggen
import t41_sdr_amd as T
for path in sorted(glob.glob("tests/golden/*.npz")):
    g = np.load(path, allow_pickle=False)
    kw = {k: (float(v) if "." in v else int(v)) for k, v in g["params"]}
    if not (kw.get("ANR_notchOn", 0) or kw.get("nrOptionSelect", 0)):
        continue
    Lf = 2048
    nch, nfr = g["I"].shape[0], g["I"].shape[1] // Lf
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=g["nco"])
    got = rx.ProcessIQData(torch.from_numpy(g["I"]).cuda(), torch.from_numpy(g["Q"]).cuda()).cpu().numpy()
    err = siggen.block_rel_err(got, g["audio"], Lf)
    print(os.path.basename(path), kw, "nch", nch, "nfr", nfr)
    print("  per-frame max over channels:", " ".join("%.1e" % v for v in err.max(axis=0)))

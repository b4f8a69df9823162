"""where a 2-frame call and two 1-frame calls of the long-FFT pipeline differ (GPU box)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import t41_sdr_amd as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nch, nfr, L4 = int(sys.argv[2]) if len(sys.argv) > 2 else 16, 2, 4 * N
kw = dict(fft_length=N, mode=0, FLoCut=400, FHiCut=600)
nco = (np.random.default_rng(444).integers(-860, 801, nch) * 50).astype(np.int32)
g = torch.Generator(device="cuda").manual_seed(8)
x = 0.2 * torch.randn(nch, nfr * L4, generator=g, device="cuda")
y = 0.2 * torch.randn(nch, nfr * L4, generator=g, device="cuda")
out = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco).ProcessIQData(x, y)
rx2 = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
parts = torch.cat([rx2.ProcessIQData(x[:, k * L4:(k + 1) * L4].contiguous(), y[:, k * L4:(k + 1) * L4].contiguous()) for k in range(nfr)], dim=1)
d = (out - parts).abs().cpu().numpy()
print("max abs diff", d.max(), "max |out|", float(out.abs().max()))
seg = d.reshape(nch, -1, 2048).max(axis=2)
print("per segment max diff (channel 0):", " ".join("%.1e" % v for v in seg[0]))
print("per segment max diff (max over channels):", " ".join("%.1e" % v for v in seg.max(axis=0)))
i = np.argwhere(d > 0)
print("first differing positions:", i[:5].tolist(), " count", len(i))
bad = np.argwhere(seg > 0)
print("differing (channel, segment) pairs:", len(bad), bad[:20].tolist())
out2 = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco).ProcessIQData(x, y)
print("same single call twice identical:", bool(torch.equal(out, out2)))

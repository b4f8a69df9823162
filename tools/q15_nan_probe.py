"""GPU box: what tools/pipe_soak.py's first q15 runs flagged (round 4) -- NFM with the AGC on, q15 samples, pipelined against
barrier form: the AUDIO is identical; the checkpoints differ in the AGC's two back-average words of channels whose stream has
gone NaN (exact zeros on the q15 format make NFM's discriminator divide 0 by 0), and only in the NaNs' payload / sign bits.
Prints the differing words.  usage: python tools/q15_nan_probe.py"""
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import t41_sdr_amd as T
L = 2048
rng = np.random.default_rng(5)
g = torch.Generator(device="cuda").manual_seed(123)
found = 0
for trial in range(120):
    nch = int(rng.choice([255, 256, 300, 1000, 4096]))
    nfr = int(rng.integers(4, 41))
    agc = int(rng.integers(1, 5))
    kw = dict(mode=3, AGCMode=agc, FLoCut=200, FHiCut=3000)
    nco = (rng.integers(-860, 801, nch) * 50).astype(np.int32)
    env = torch.rand(nch, nfr * 8, generator=g, device="cuda").repeat_interleave(L // 8, dim=1) ** 3
    I = (0.3 * env * torch.randn(nch, nfr * L, generator=g, device="cuda")).clamp_(-0.999, 0.999)
    Q = (0.3 * env * torch.randn(nch, nfr * L, generator=g, device="cuda")).clamp_(-0.999, 0.999)
    Iq = (I * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
    Qq = (Q * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
    # dirty the device memory the next contexts will be given: another mode's context of the same size, used and freed
    rx0 = T.RxChain(nch, T.default_params(mode=8, AGCMode=2, FLoCut=-3000, FHiCut=3000), NCOFreq=nco)
    rx0.ProcessIQData(I, Q)
    del rx0
    rx1 = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    whole = rx1.ProcessIQData_q15(Qq, Iq)
    rx2 = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    parts, pos = [], 0
    while pos < nfr:
        n = min(int(rng.integers(1, 4)), nfr - pos)
        parts.append(rx2.ProcessIQData_q15(Qq[:, pos * L:(pos + n) * L].contiguous(), Iq[:, pos * L:(pos + n) * L].contiguous()))
        pos += n
    short = torch.cat(parts, dim=1)
    # a third time: the whole call again in a fresh context (is the pipelined form reproducible?)
    rx4 = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    again = rx4.ProcessIQData_q15(Qq, Iq)
    st1, st2 = np.asarray(rx1.get_state()), np.asarray(rx2.get_state())
    if torch.equal(whole, short) and torch.equal(whole, again) and np.array_equal(st1, st2):
        continue
    if not np.array_equal(st1, st2):
        a32, b32 = st1.view(np.uint8).tobytes(), st2.view(np.uint8).tobytes()
        a32 = np.frombuffer(a32[32:32 + (len(a32) - 32) // 4 * 4], np.uint32); b32 = np.frombuffer(b32[32:32 + (len(b32) - 32) // 4 * 4], np.uint32)
        hdr = np.frombuffer(st1.view(np.uint8).tobytes()[:32], np.uint32)
        per = int(hdr[4])
        idx = np.nonzero(a32 != b32)[0]
        print("state differs in", len(idx), "words; floats per channel record", per, "; (channel, word) of the first:", [(int(i // per), int(i % per)) for i in idx[:8]],
              "values", a32[idx[:4]].view(np.float32).tolist(), b32[idx[:4]].view(np.float32).tolist())
    print("whole == again:", bool(torch.equal(whole, again)), " whole == short:", bool(torch.equal(whole, short)), " again == short:", bool(torch.equal(again, short)))
    found += 1
    # the same samples through the f32 entry point: where are its NaNs?
    rx3 = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    f32 = rx3.ProcessIQData(Iq.float() / 32768.0, Qq.float() / 32768.0)
    diff = whole != short
    nan = ~torch.isfinite(f32)
    ch_bad = torch.nonzero(diff.any(dim=1)).flatten()[:4].tolist()
    fr_bad = sorted(set((torch.nonzero(diff)[:, 1] // L).tolist()))[:8]
    print("trial", trial, "nch", nch, "nfr", nfr, "channels", ch_bad, "frames", fr_bad, "agc", agc, "differing samples", int(diff.sum()), "of which where the f32 audio is not finite", int((diff & nan).sum()),
          "non-finite f32 samples", int(nan.sum()), "values", whole[diff][:6].tolist(), short[diff][:6].tolist(),
          "state equal", bool(np.array_equal(rx1.get_state(), rx2.get_state())))
    if found >= 4:
        break
print("found", found)

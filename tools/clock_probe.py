#!/usr/bin/env python3
"""The shader clock the fused kernel's waves see under their own load (-DT41RX_CLK diagnostic build, GPU box).

  tools/build_variant.sh clk -DT41RX_CLK          (here)
  T41RX_LIB=$PWD/t41_sdr_amd/abl/libt41rx_clk.so python tools/clock_probe.py [--agc 0] [--mode 0] [--layout channel]
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import t41_sdr_amd as T  # noqa: E402
from t41_sdr_amd import _lib  # noqa: E402

def opt(flag, default):
    a = sys.argv[1:]
    return type(default)(a[a.index(flag) + 1]) if flag in a else default


def main():
    fft = opt("--fft", 512)
    nch, frames, reps = opt("--nch", 4096 if fft == 512 else 1024), opt("--frames", 32), opt("--reps", 40)
    layout = opt("--layout", "channel")
    L = 4 * fft
    kw = dict(mode=opt("--mode", 0), AGCMode=opt("--agc", 0), fft_length=fft)
    if fft != 512:
        kw.update(FLoCut=400, FHiCut=600)
    if kw["mode"] in (2, 8):
        kw.update(FLoCut=-3000, FHiCut=3000)
    rx = T.RxChain(nch, T.default_params(**kw),
                   NCOFreq=(np.random.default_rng(1000).integers(-860, 801, nch) * 50).astype(np.int32))
    rx.set_buffer_layout(layout)
    shape = (nch, frames * L) if layout == "channel" else (frames, nch, L)
    g = torch.Generator(device="cuda").manual_seed(0)
    ring = 3
    Is = [(0.2 * torch.randn(*shape, generator=g, device="cuda")).clamp_(-0.999, 0.999) for _ in range(ring)]
    Qs = [(0.2 * torch.randn(*shape, generator=g, device="cuda")).clamp_(-0.999, 0.999) for _ in range(ring)]
    out = [torch.empty(*shape, device="cuda") for _ in range(ring)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(10):
        rx.ProcessIQData(Is[k % ring], Qs[k % ring], out=out[k % ring])
    e0.record()
    for k in range(reps):
        rx.ProcessIQData(Is[k % ring], Qs[k % ring], out=out[k % ring])
    e1.record()
    torch.cuda.synchronize()
    lib = _lib.load()
    nw = nch if fft == 512 else 4 * nch  # waves that report (the fused long-FFT kernel runs four per channel)
    buf = (C.c_ulonglong * (4 * nw))()
    # one reader per kernel translation unit (rx_device.hpp: T41RX_CLK_READER)
    reader = "t41rx_debug_read_clk" + ("_fc" if fft != 512 else {0: "", 1: "", 2: "_am", 3: "_nfm", 8: "_sam"}[kw["mode"]])
    rc = getattr(lib, reader)(buf, 4 * nw)
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 4)
    a = raw[:, :2].astype(np.float64)
    if "--dump" in sys.argv:
        np.save(opt("--dump", "gpurun_out/clk_dump.npy"), raw)
        # where the launch's last tenth goes: wave lifetimes / end times (100 MHz ticks -> us) by XCD, CU, SIMD, wave slot
        life = raw[:, 1].astype(np.float64) / 100.0
        start = (raw[:, 2] - raw[:, 2].min()).astype(np.float64) / 100.0
        end = start + life
        hw = raw[:, 3]
        xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(int)
        simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(int)
        cu = (((hw >> np.uint64(8)) & np.uint64(0xff)).astype(int)) + 256 * xcc   # CU_ID, SH_ID, SE_ID within the XCD
        slot = np.arange(nw) % 16
        def by(key):
            return {int(k): [round(float(life[key == k].mean()), 1), round(float(end[key == k].max()), 1)] for k in np.unique(key)}
        cu_end = np.array([end[cu == k].max() for k in np.unique(cu)])
        cu_first = np.array([end[cu == k].min() for k in np.unique(cu)])
        print(json.dumps({"launch_span_us": round(float(end.max()), 1), "start_spread_us": round(float(start.max()), 2),
                          "life_us_pct_0_10_50_90_100": [round(float(np.percentile(life, q)), 1) for q in (0, 10, 50, 90, 100)],
                          "end_us_pct_0_10_50_90_100": [round(float(np.percentile(end, q)), 1) for q in (0, 10, 50, 90, 100)],
                          "n_cus_seen": int(len(cu_end)),
                          "cu_last_end_pct_0_50_100": [round(float(np.percentile(cu_end, q)), 1) for q in (0, 50, 100)],
                          "cu_first_end_pct_0_50_100": [round(float(np.percentile(cu_first, q)), 1) for q in (0, 50, 100)],
                          "by_xcc_[mean_life,last_end]": by(xcc), "by_simd": by(simd), "by_wave_slot": by(slot)}), flush=True)
    ghz = a[:, 0] / a[:, 1] * 0.1
    print(json.dumps({"args": " ".join(sys.argv[1:]), "rc": rc, "us_per_frame": round(e0.elapsed_time(e1) / reps * 1e3 / frames, 3),
                      "clock_GHz_median": round(float(np.median(ghz)), 3), "min": round(float(ghz.min()), 3), "max": round(float(ghz.max()), 3),
                      "wave_life_us_median": round(float(np.median(a[:, 1])) / 100.0, 1),
                      "cycles_per_wave_frame": round(float(np.median(a[:, 0])) / frames)}), flush=True)


if __name__ == "__main__":
    main()

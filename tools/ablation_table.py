#!/usr/bin/env python3
"""Ablation table of the fused FFT_LENGTH 512 kernel (BASELINE config 2): where the frame's time goes.

Each row is a build of rx_kernels.hip with -DT41RX_ABLATE=n (see that file): stages are cut from the END
of the chain (1 interpolators, 2 + FFTs, 3 + /2 decimator, 4 + /4 decimator, 5 + oscillator, 6 + DC
high-pass, 7 = every arithmetic stage gone, the loads / LDS staging / 1-KiB stores kept: "memory
only"), 9 = all arithmetic but every wave works on the same 16 channels' buffers (cache-resident
I/O: "arithmetic only").  Outputs of n > 0 are wrong by construction; only the product row is a kernel.

  python tools/ablation_table.py build            (here: hipcc cross-compiles, ~45 s per variant)
  python tools/ablation_table.py run  [--rounds 3] [--frames 32] [--out gpurun_out/r03_ablation_ssb.md]   (GPU box)
  python tools/ablation_table.py one NAME         (one timing of one variant; what `run` starts per cell and what
                                                   tools/ablation_round.sh puts under rocprofv3 --kernel-trace --stats)

`run` interleaves the variants over several rounds (one process per cell: a process binds one
libt41rx build) and reports the median and the minimum per variant, so drift of the box's clock
between rows is visible instead of being read as a kernel property.
"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [
    ("product", None, "the kernel as shipped"),
    ("abl1", 1, "without the x2 / x4 interpolators (stores kept)"),
    ("abl2", 2, "... and without the two 512-point FFTs + mask"),
    ("abl3", 3, "... and without the /2 decimator"),
    ("abl4", 4, "... and without the /4 decimator"),
    ("abl5", 5, "... and without the oscillator / mixer"),
    ("abl6", 6, "... and without the DC high-pass"),
    ("abl7", 7, "memory only: loads, LDS staging, 1-KiB stores; no arithmetic"),
    ("abl9", 9, "arithmetic only: every wave on the same 16 channels' buffers (cache-resident I/O)"),
    # a full kernel (correct outputs), not an ablation: the 512-point transforms' first exchange through LDS, as in
    # round 2, instead of inside the VALU (v_permlane32_swap / v_permlane16_swap / DPP; wave_fft.hpp: T41RX_FFT_X1_PERM)
    ("x1lds", "-DT41RX_FFT_X1_PERM=0", "product with the FFTs' first exchange through LDS (round 2's form) instead of v_permlane*_swap + DPP"),
    # leave one out (round 4): the product minus exactly ONE stage, every other stage and the stores kept
    ("loo1", "-DT41RX_LOO=1", "product without the x2 / x4 interpolators only"),
    ("loo2", "-DT41RX_LOO=2", "product without the two 512-point FFTs + mask only"),
    ("loo3", "-DT41RX_LOO=3", "product without the /2 decimator only"),
    ("loo4", "-DT41RX_LOO=4", "product without the /4 decimator only"),
    ("loo5", "-DT41RX_LOO=5", "product without the oscillator / mixer only"),
    ("loo6", "-DT41RX_LOO=6", "product without the DC high-pass only"),
    ("pf1", "-DT41RX_PF=1", "product with ONE sub-block of the next frame requested across the back end (16 registers fewer held)"),
    ("pf0", "-DT41RX_PF=0", "product with none"),
]
L = 2048
ALG_BYTES_PER_FRAME = 12 * 4096 * L


def lib_path(name):
    return None if name == "product" else os.path.join(ROOT, "t41_sdr_amd", "abl", "libt41rx_%s.so" % name)


def build():
    for name, n, _ in VARIANTS:
        if n is None:
            continue
        print("building", name, flush=True)
        flag = n if isinstance(n, str) else "-DT41RX_ABLATE=%d" % n
        subprocess.check_call([os.path.join(ROOT, "tools", "build_variant.sh"), name, flag])


def one(frames, reps, layout="channel"):
    """HIP-event time per frame of the library this process bound (T41RX_LIB), config 2's shape"""
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import t41_sdr_amd as T
    nch = 4096
    rng = np.random.default_rng(1000)
    nco = (rng.integers(-860, 801, nch) * 50).astype(np.int32)
    a = sys.argv[1:]
    kw = {}
    if "--mode" in a:
        kw["mode"] = int(a[a.index("--mode") + 1])
    if "--agc" in a:
        kw["AGCMode"] = int(a[a.index("--agc") + 1])
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    rx.set_buffer_layout(layout)
    shape = (nch, frames * L) if layout == "channel" else (frames, nch, L)
    ring = max(2, -(-(768 << 20) // (3 * nch * frames * L * 4)))
    g = torch.Generator(device="cuda").manual_seed(0)
    Is = [(0.2 * torch.randn(*shape, generator=g, device="cuda")).clamp_(-0.999, 0.999) for _ in range(ring)]
    Qs = [(0.2 * torch.randn(*shape, generator=g, device="cuda")).clamp_(-0.999, 0.999) for _ in range(ring)]
    out = [torch.empty(*shape, device="cuda") for _ in range(ring)]
    for k in range(max(6, reps // 4)):
        rx.ProcessIQData(Is[k % ring], Qs[k % ring], out=out[k % ring])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        rx.ProcessIQData(Is[k % ring], Qs[k % ring], out=out[k % ring])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3 / frames


def run(rounds, frames, reps, out_path, layout="channel"):
    res = {name: [] for name, _, _ in VARIANTS}
    for r in range(rounds):
        for name, _, _ in VARIANTS:
            lp = lib_path(name)
            if lp is not None and not os.path.exists(lp):
                continue
            env = dict(os.environ)
            env.pop("T41RX_LIB", None)
            if lp:
                env["T41RX_LIB"] = lp
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "one", name, "--frames", str(frames), "--reps", str(reps), "--layout", layout],
                               env=env, capture_output=True, text=True, timeout=600)
            cells = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not cells:
                print("variant %s failed: %s" % (name, p.stderr[-400:]), flush=True)
                continue
            res[name].append(cells[0]["us_per_frame"])
            print("round %d %-8s %.2f us per frame" % (r, name, cells[0]["us_per_frame"]), flush=True)
    lines = ["# Ablation of `rx512_kernel<SSB, PLAIN>` (config 2: 4096 channels x %d frames x 2048 samples per launch, %s-major buffers)" % (frames, layout), "",
             "HIP-event time per 4096-channel frame, %d launches per cell, %d interleaved rounds, one process per cell "
             "(`tools/ablation_table.py run`).  GB/s = 12 B x 4096 x 2048 per frame over that time." % (reps, rounds), "",
             "| build | what is left | median us | min us | GB/s at the median | of 8 TB/s |", "|---|---|---|---|---|---|"]
    for name, _, what in VARIANTS:
        v = res[name]
        if not v:
            continue
        med, mn = statistics.median(v), min(v)
        gbs = ALG_BYTES_PER_FRAME / med / 1e3
        lines.append("| %s | %s | %.2f | %.2f | %.0f | %.3f |" % (name, what, med, mn, gbs, gbs / 8000.0))
    if res["product"]:
        base = statistics.median(res["product"])
        prev = base
        lines += ["", "Cost of each stage = the step between two consecutive cuts (medians):", "",
                  "| stage | us per frame | share of the product's time |", "|---|---|---|"]
        for name, label in (("abl1", "x2 + x4 interpolators"), ("abl2", "512-point FFT, mask, inverse FFT"), ("abl3", "/2 decimator"),
                            ("abl4", "/4 decimator"), ("abl5", "oscillator + mixer"), ("abl6", "DC high-pass"),
                            ("abl7", "gain / interleave, demodulator, transposition arithmetic")):
            if not res[name]:
                continue
            cur = statistics.median(res[name])
            lines.append("| %s | %.2f | %.1f %% |" % (label, prev - cur, 100.0 * (prev - cur) / base))
            prev = cur
        lines.append("| memory-only skeleton (abl7) | %.2f | %.1f %% |" % (prev, 100.0 * prev / base))
        if any(res.get("loo%d" % k) for k in range(1, 7)):
            lines += ["", "Leave one out: the product minus exactly one stage (medians), next to that stage's cumulative figure:", "",
                      "| stage left out | us per frame | saves | cumulative table says |", "|---|---|---|---|"]
            cum = {}
            prev = base
            for name in ("abl1", "abl2", "abl3", "abl4", "abl5", "abl6"):
                if res[name]:
                    cur = statistics.median(res[name])
                    cum[name[-1]] = prev - cur
                    prev = cur
            for k, label in ((1, "x2 + x4 interpolators"), (2, "512-point FFT, mask, inverse FFT"), (3, "/2 decimator"), (4, "/4 decimator"),
                             (5, "oscillator + mixer"), (6, "DC high-pass")):
                v = res.get("loo%d" % k)
                if v:
                    m = statistics.median(v)
                    lines.append("| %s | %.2f | %.2f (%.1f %%) | %s |" % (label, m, base - m, 100.0 * (base - m) / base,
                                                                     ("%.2f" % cum[str(k)]) if str(k) in cum else "-"))
    text = "\n".join(lines) + "\n"
    print(text)
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        with open(out_path, "w") as f:
            f.write(text)
        with open(os.path.splitext(out_path)[0] + ".json", "w") as f:
            json.dump({"frames_per_launch": frames, "launches_per_cell": reps, "us_per_frame": res}, f, indent=1)


def main():
    args = sys.argv[1:]
    if not args or args[0] not in ("build", "run", "one"):
        raise SystemExit(__doc__)

    def opt(flag, default):
        return type(default)(args[args.index(flag) + 1]) if flag in args else default
    if args[0] == "build":
        build()
    elif args[0] == "one":
        us = one(opt("--frames", 32), opt("--reps", 60), opt("--layout", "channel"))
        print(json.dumps({"variant": args[1] if len(args) > 1 else "product", "us_per_frame": round(us, 3)}), flush=True)
    else:
        run(opt("--rounds", 3), opt("--frames", 32), opt("--reps", 60), opt("--out", os.path.join(ROOT, "gpurun_out", "r04_ablation_ssb.md")),
            opt("--layout", "channel"))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A/B timing of kernel builds on config 2's shape: interleaved rounds, one process per cell (GPU box).

  python tools/ab_probe.py product uncond other ... [--rounds 3] [--frames 32] [--reps 60] [--layout channel]
    NAME = "product" (t41_sdr_amd/libt41rx.so) or a build of tools/build_variant.sh NAME (t41_sdr_amd/abl/libt41rx_NAME.so)
Prints the median / minimum us per 4096-channel frame per build and the spread, as JSON lines.
"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]

    def opt(flag, default):
        if flag in args:
            i = args.index(flag)
            v = type(default)(args[i + 1])
            del args[i:i + 2]
            return v
        return default
    rounds, frames, reps, layout = opt("--rounds", 3), opt("--frames", 32), opt("--reps", 60), opt("--layout", "channel")
    extra = []
    for f in ("--mode", "--agc"):
        if f in args:
            i = args.index(f)
            extra += args[i:i + 2]
            del args[i:i + 2]
    names = args
    res = {n: [] for n in names}
    for r in range(rounds):
        for n in names:
            env = dict(os.environ)
            env.pop("T41RX_LIB", None)
            if n != "product":
                env["T41RX_LIB"] = os.path.join(ROOT, "t41_sdr_amd", "abl", "libt41rx_%s.so" % n)
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ablation_table.py"), "one", n, "--frames", str(frames),
                                "--reps", str(reps), "--layout", layout] + extra, env=env, capture_output=True, text=True, timeout=600)
            cells = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not cells:
                print("build %s failed: %s" % (n, p.stderr[-300:]), flush=True)
                continue
            res[n].append(cells[0]["us_per_frame"])
    for n in names:
        v = res[n]
        if v:
            print(json.dumps({"build": n, "median_us": round(statistics.median(v), 3), "min_us": round(min(v), 3), "all": v,
                              "frac_of_8TBs": round(12 * 4096 * 2048 / statistics.median(v) / 1e3 / 8000, 4)}), flush=True)


if __name__ == "__main__":
    main()

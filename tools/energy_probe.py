#!/usr/bin/env python3
"""Energy per frame of kernel builds on config 2's shape (GPU box): board power sampled while a build loops.

VERDICT r04 item 1: under the power cap the kernel's time goes as cycles / clock with the clock set by the power its
instruction stream draws, so variants are to be ranked by JOULES per frame next to microseconds per frame.

  python tools/energy_probe.py product abl7 abl9 firplain ... [--seconds 14] [--frames 32] [--out gpurun_out/r05_energy_ssb.md]
    NAME = "product" (t41_sdr_amd/libt41rx.so) or a build of tools/build_variant.sh NAME (t41_sdr_amd/abl/libt41rx_NAME.so)

Per build: one child process loops the launch for `--seconds` (tools/ablation_table.py one NAME), this process samples
the board's power sensor at 20 Hz (sysfs hwmon power1_average / power1_input of the card that draws the most while the
load runs -- the box shows one GPU; `rocm-smi --showpower` as a cross-check when it is there), drops the first 4 s
(clock ramp) and the last second, and reports W, us per frame, mJ per 4096-channel frame.
"""
import glob
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sensors():
    out = []
    for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input",
                "/sys/class/hwmon/hwmon*/power1_average", "/sys/class/hwmon/hwmon*/power1_input"):
        for f in glob.glob(pat):
            try:
                with open(f) as fh:
                    int(fh.read().strip())
                real = os.path.realpath(f)
                if real not in [os.path.realpath(x) for x in out]:
                    out.append(f)
            except (OSError, ValueError):
                pass
    return out


def read_w(f):
    try:
        with open(f) as fh:
            return int(fh.read().strip()) / 1e6  # microwatts
    except (OSError, ValueError):
        return None


def smi_power():
    try:
        t = subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(t)
        vals = []
        for card, kv in d.items():
            for k, v in kv.items():
                if "ower" in k:
                    try:
                        vals.append(float(v))
                    except ValueError:
                        pass
        return max(vals) if vals else None
    except Exception:
        return None


def main():
    args = sys.argv[1:]

    def opt(flag, default):
        if flag in args:
            i = args.index(flag)
            v = type(default)(args[i + 1])
            del args[i:i + 2]
            return v
        return default
    seconds, frames, out_path = opt("--seconds", 14.0), opt("--frames", 32), opt("--out", "")
    extra = []
    for f in ("--mode", "--agc"):
        if f in args:
            i = args.index(f)
            extra += args[i:i + 2]
            del args[i:i + 2]
    names = args
    sens = sensors()
    print("power sensors:", sens, flush=True)
    rows = []
    for n in names:
        env = dict(os.environ)
        env.pop("T41RX_LIB", None)
        if n != "product":
            env["T41RX_LIB"] = os.path.join(ROOT, "t41_sdr_amd", "abl", "libt41rx_%s.so" % n)
            if not os.path.exists(env["T41RX_LIB"]):
                print("no build %s" % n, flush=True)
                continue
        reps = int(seconds / (frames * 22e-6))  # ~22 us per frame; the child warms up reps / 4 launches on top
        t0 = time.time()
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "ablation_table.py"), "one", n, "--frames", str(frames), "--reps", str(reps)] + extra,
                             env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        samples = {f: [] for f in sens}
        smi = []
        last_smi = 0.0
        while p.poll() is None:
            t = time.time() - t0
            for f in sens:
                w = read_w(f)
                if w is not None:
                    samples[f].append((t, w))
            if t - last_smi > 3.0 and t > 8.0:
                last_smi = t
                w = smi_power()
                if w is not None:
                    smi.append(w)
            time.sleep(0.05)
            if t > seconds * 3 + 120:
                p.kill()
                break
        so, se = p.communicate()
        t_end = time.time() - t0
        cells = [json.loads(l) for l in so.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not cells:
            print("build %s failed: %s" % (n, se[-300:]), flush=True)
            continue
        us = cells[0]["us_per_frame"]
        # the load's window: the timed loop is the last reps * frames * us of the run
        t_load0 = t_end - reps * frames * us * 1e-6 - 0.5
        best = None
        for f, v in samples.items():
            w = [x for (t, x) in v if t_load0 + 4.0 <= t <= t_end - 1.5]
            if len(w) >= 10 and (best is None or statistics.mean(w) > best[1]):
                best = (f, statistics.mean(w), statistics.pstdev(w), len(w))
        row = {"build": n, "us_per_frame": round(us, 3), "frac_of_8TBs": round(12 * 4096 * 2048 / us / 1e3 / 8000, 4),
               "watts": round(best[1], 1) if best else None, "watts_sd": round(best[2], 1) if best else None, "samples": best[3] if best else 0,
               "mJ_per_frame": round(best[1] * us * 1e-3, 3) if best else None, "rocm_smi_watts": round(statistics.mean(smi), 1) if smi else None,
               "sensor": best[0] if best else None}
        rows.append(row)
        print(json.dumps(row), flush=True)
    if out_path and rows:
        base = next((r for r in rows if r["build"] == "product"), rows[0])
        lines = ["# Energy per frame of `rx512_kernel<SSB, PLAIN>` builds (config 2: 4096 channels x %d frames per launch)" % frames, "",
                 "Board power (hwmon `power1_average`, 20 Hz, first 4 s of each load dropped) x HIP-event time per 4096-channel frame; one "
                 "process per build, %.0f s of back-to-back launches each (`tools/energy_probe.py`).  Builds other than `product` give WRONG "
                 "results by construction (timing experiments)." % seconds, "",
                 "| build | us per frame | of 8 TB/s | W | mJ per frame | time vs product | energy vs product |", "|---|---|---|---|---|---|---|"]
        for r in rows:
            if r["watts"] is None:
                lines.append("| %s | %.2f | %.3f | n/a | n/a | %+.1f %% | n/a |" % (r["build"], r["us_per_frame"], r["frac_of_8TBs"],
                                                                                 100.0 * (r["us_per_frame"] / base["us_per_frame"] - 1)))
            else:
                lines.append("| %s | %.2f | %.3f | %.0f (sd %.0f) | %.2f | %+.1f %% | %+.1f %% |" % (
                    r["build"], r["us_per_frame"], r["frac_of_8TBs"], r["watts"], r["watts_sd"], r["mJ_per_frame"],
                    100.0 * (r["us_per_frame"] / base["us_per_frame"] - 1),
                    100.0 * (r["mJ_per_frame"] / base["mJ_per_frame"] - 1) if base.get("mJ_per_frame") else float("nan")))
        with open(out_path, "w") as f:
            f.write("\n".join(lines) + "\n")
        print("\n".join(lines))


if __name__ == "__main__":
    main()

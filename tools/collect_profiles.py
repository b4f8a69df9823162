#!/usr/bin/env python3
"""Copy the summaries of tools/profile_round.sh into profiles/ and rewrite profiles/hbm_traffic.json.
usage: python tools/collect_profiles.py TAG     (reads gpurun_out/profile_TAG/)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (WORKLOADS, kernel_source_hash; importing it touches no GPU)


def counter_mean(d, name, kern="t41::"):
    vals = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if kern in row.get("Kernel_Name", "") and row["Counter_Name"] == name:
                    vals.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "profile_" + tag)
    dst = os.path.join(ROOT, "profiles")
    traffic = {"round": tag, "source_hash": bench.kernel_source_hash(),
               "correction": "gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies the 128-B requests of 16-B/lane streaming loads at "
                             "half size (MI355X_MICROARCH.md, HBM): reads doubled; WRITE_SIZE exact.  One rocprofv3 --pmc process "
                             "per counter; KiB per dispatch, mean over the timed dispatches, summed over the kernels of a step.",
               "workloads": {}}
    for w, cfg in bench.WORKLOADS.items():
        wd = os.path.join(src, w)
        if not os.path.isdir(wd):
            continue
        # kernel-trace stats: our kernels only
        for path in glob.glob(os.path.join(wd, "trace", "**", "*kernel_stats.csv"), recursive=True):
            with open(path) as f:
                rows = [r for r in csv.reader(f)]
            keep = [rows[0]] + [r for r in rows[1:] if r and "t41::" in r[0]]
            with open(os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, w)), "w", newline="") as f:
                csv.writer(f).writerows(keep)
        fetch = counter_mean(os.path.join(wd, "FETCH_SIZE"), "FETCH_SIZE")
        write = counter_mean(os.path.join(wd, "WRITE_SIZE"), "WRITE_SIZE")
        if not fetch or not write:
            continue
        frames = cfg.get("frames", bench.DEFAULT_FRAMES)
        fft = cfg["fft"]
        bps = 6.0 if cfg.get("q15") else 12.0
        alg = int(bps * cfg["batch"] * frames * 4 * fft)
        rd = sum(fetch.values()) * 1024 * 2
        wr = sum(write.values()) * 1024
        traffic["workloads"][w] = {
            "frames_per_launch": frames, "source_hash": traffic["source_hash"],
            "FETCH_SIZE_KiB": {k.split("(")[0][-60:]: round(v, 1) for k, v in fetch.items()},
            "WRITE_SIZE_KiB": {k.split("(")[0][-60:]: round(v, 1) for k, v in write.items()},
            "read_bytes": int(rd), "write_bytes": int(wr), "bytes_per_launch": int(rd + wr),
            "algorithmic_bytes_per_launch": alg, "ratio": round((rd + wr) / alg, 4),
        }
        with open(os.path.join(dst, "%s_pmc_hbm_%s.txt" % (tag, w)), "w") as f:
            for name, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
                for k, v in d.items():
                    f.write("%-12s mean %.1f KiB per dispatch  %s\n" % (name, v, k))
            f.write("HBM bytes per step (reads x2 + writes): %d = %.3f x algorithmic (%d)\n" % (rd + wr, (rd + wr) / alg, alg))
    # SQ passes of the headline workload
    lines = []
    for i in (1, 2, 3):
        d = os.path.join(src, "ssb", "sq%d" % i)
        names = set()
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if "t41::" in row.get("Kernel_Name", ""):
                        names.add(row["Counter_Name"])
        for n in sorted(names):
            for k, v in counter_mean(d, n).items():
                lines.append("%-24s mean %.6g per dispatch   %s" % (n, v, k.split("(")[0][-70:]))
    if lines:
        with open(os.path.join(dst, "%s_pmc_sq_ssb.txt" % tag), "w") as f:
            f.write("\n".join(lines) + "\n")
    with open(os.path.join(dst, "hbm_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps({w: (v["ratio"], v["bytes_per_launch"]) for w, v in traffic["workloads"].items()}))


if __name__ == "__main__":
    main()

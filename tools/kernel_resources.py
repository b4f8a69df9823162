#!/usr/bin/env python3
"""Register / LDS / spill table of every gfx950 kernel in a .hip file (no GPU needed).

usage: python tools/kernel_resources.py [file.hip ...] [-D...] [--grep REGEX]   (default: every kernel translation unit of t41_sdr_amd/csrc)
Compiles device-only with -Rpass-analysis=kernel-resource-usage and prints one line per kernel.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    pat = None
    if "--grep" in args:
        i = args.index("--grep")
        pat = re.compile(args[i + 1])
        del args[i:i + 2]
    defs = [a for a in args if a.startswith("-")]
    files = [a for a in args if not a.startswith("-")]
    csrc = os.path.join(ROOT, "t41_sdr_amd", "csrc")
    # default: every translation unit of the RX kernels (one per kernel family)
    srcs = [os.path.abspath(f) for f in files] if files else [os.path.join(csrc, f) for f in ("rx512_ssb.hip", "rx512_am.hip", "rx512_nfm.hip", "rx512_sam.hip", "rx_long.hip",
                                                               "fastconv.hip", "display_kernel.hip", "nr_kernels.hip")]
    procs = [subprocess.Popen(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
                               "--offload-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + defs,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=os.path.dirname(os.path.abspath(src))) for src in srcs]
    err = ""
    for p in procs:
        _, e = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(e)
            raise SystemExit(p.returncode)
        err += e
    cur = None
    rows = []
    for line in err.splitlines():
        m = re.search(r"remark:\s*([A-Za-z /\[\]]+?): (.+?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = name.replace("t41::", "").replace("(t41::RxArgs)", "").replace("void ", "")
        if pat and not pat.search(name):
            continue
        print("%-58s vgpr %3s agpr %3s sgpr %3s  spill v %3s s %3s  scratch %5s  occ %s  lds %s" % (
            name, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
            r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))


if __name__ == "__main__":
    main()

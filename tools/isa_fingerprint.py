#!/usr/bin/env python3
"""Per-kernel fingerprint of a built libt41rx.so's gfx950 code objects: sha256 of every kernel's disassembly (addresses and
encodings stripped, branch targets as offsets from the kernel's start), so a refactor of the sources can be shown to leave the
product's instruction streams untouched.

  python tools/isa_fingerprint.py [LIB] > fingerprints.json        python tools/isa_fingerprint.py --diff A.json B.json
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def fingerprints(lib):
    out = {}
    with tempfile.TemporaryDirectory() as t:
        dst = os.path.join(t, "lib.so")
        with open(lib, "rb") as f, open(dst, "wb") as g:
            g.write(f.read())
        subprocess.check_call([OBJDUMP, "--offloading", "lib.so"], cwd=t, stdout=subprocess.DEVNULL)
        for co in sorted(os.listdir(t)):
            if "hipv4-amdgcn" not in co or os.path.getsize(os.path.join(t, co)) == 0:
                continue
            dis = subprocess.check_output([OBJDUMP, "-d", os.path.join(t, co)], text=True)
            name, body, base = None, [], None
            def flush():
                if name is not None and body:
                    out[name] = {"sha": hashlib.sha256("\n".join(body).encode()).hexdigest()[:20], "instructions": len(body)}
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    flush()
                    name, body, base = m.group(1), [], None
                    continue
                m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
                if m and name is not None:
                    addr = int(m.group(3), 16)
                    base = addr if base is None else base
                    ops = re.sub(r"<[^>]*\+0x([0-9a-f]+)>", lambda k: "<+%s>" % k.group(1), m.group(2))
                    ops = re.sub(r"<[^>+]*>", "<+0>", ops)
                    body.append("%s %s" % (m.group(1), ops))
            flush()
    return out


def main():
    a = sys.argv[1:]
    if a and a[0] == "--diff":
        x, y = json.load(open(a[1])), json.load(open(a[2]))
        bad = [k for k in sorted(set(x) | set(y)) if x.get(k) != y.get(k)]
        print("%d kernels in A, %d in B, %d differ" % (len(x), len(y), len(bad)))
        for k in bad[:40]:
            print("  ", k[:150], x.get(k), y.get(k))
        sys.exit(1 if bad else 0)
    lib = a[0] if a else os.path.join(ROOT, "t41_sdr_amd", "libt41rx.so")
    json.dump(fingerprints(lib), sys.stdout, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()

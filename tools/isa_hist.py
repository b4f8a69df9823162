#!/usr/bin/env python3
"""Instruction histogram of a kernel's hottest loop from llvm-objdump -d output (no GPU needed).

usage: python tools/isa_hist.py kernel.s [--all]
Finds the backward branch that spans the most instructions (the frame loop) and counts the
mnemonics inside it, grouped by unit (VALU packed / plain / DPP, LDS, VMEM, SMEM, SALU, waits).
"""
import collections
import re
import sys


def main():
    path = sys.argv[1]
    ins = []
    for line in open(path):
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):(.*)", line)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2) + " " + m.group(4)))
    addr_idx = {a: i for i, (a, _, _) in enumerate(ins)}
    base = ins[0][0]
    loops = []
    for i, (a, mn, ops) in enumerate(ins):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            m = re.search(r"<.*\+0x([0-9a-f]+)>", ops)
            if m:
                tgt = base + int(m.group(1), 16)
                if tgt in addr_idx and addr_idx[tgt] < i:
                    loops.append((i - addr_idx[tgt], addr_idx[tgt], i))
    loops.sort(reverse=True)
    print("backward branches (span, from, to):", loops[:6])
    if "--all" in sys.argv or not loops:
        lo, hi = 0, len(ins) - 1
    else:
        _, lo, hi = loops[0]
    body = ins[lo:hi + 1]
    groups = collections.Counter()
    detail = collections.Counter()
    for a, mn, ops in body:
        if mn.startswith("v_pk_"):
            g = "VALU packed"
        elif mn.startswith("v_") and ("dpp" in mn or "row_" in ops or "quad_perm" in ops or "wave_" in ops):
            g = "VALU dpp"
        elif mn.startswith("v_permlane") or mn.startswith("v_readlane") or mn.startswith("v_readfirstlane") or mn.startswith("v_writelane"):
            g = "VALU lane"
        elif mn.startswith("v_"):
            g = "VALU plain"
        elif mn.startswith("ds_"):
            g = "LDS"
        elif mn.startswith("global_") or mn.startswith("buffer_") or mn.startswith("flat_") or mn.startswith("scratch_"):
            g = "VMEM"
        elif mn.startswith("s_load") or mn.startswith("s_buffer_load"):
            g = "SMEM"
        elif mn.startswith("s_waitcnt"):
            g = "wait"
        elif mn.startswith("s_"):
            g = "SALU"
        else:
            g = "other"
        groups[g] += 1
        detail[(g, mn)] += 1
    print("loop body: %d instructions" % len(body))
    for g, n in groups.most_common():
        print("  %-12s %5d" % (g, n))
        for (gg, mn), k in sorted(detail.items(), key=lambda t: -t[1]):
            if gg == g and k >= 3:
                print("      %-28s %4d" % (mn, k))


if __name__ == "__main__":
    main()

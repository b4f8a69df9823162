"""Per-frame kernel time against frames per launch (GPU box).  usage: python tools/shape_sweep2.py [nfr ...]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from shape_sweep import run  # noqa: E402

if __name__ == "__main__":
    shapes = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]
    for nfr in shapes:
        run(4096, nfr, reps=max(20, 12000 // nfr), ring=max(2, min(6, 48 // nfr)))  # ~0.3 s per shape

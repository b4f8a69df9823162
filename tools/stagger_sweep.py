"""frames-per-launch timing for several T41RX_STAGGER values (GPU box): each in its own process."""
import os, subprocess, sys
vals = sys.argv[1:] or ["0", "8", "16", "32", "48", "64"]
for v in vals:
    env = dict(os.environ, T41RX_STAGGER=v)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "shape_sweep2.py"), "16", "32"], env=env, capture_output=True, text=True).stdout
    print("stagger", v, "|", " | ".join(l.strip() for l in out.splitlines() if "channels" in l), flush=True)

"""frames-per-launch timing of several experiment builds (GPU box): python tools/variant_sweep.py NAME[:ENV=VAL,...] ..."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = os.environ.get("SWEEP_SHAPES", "16 32").split()
for spec in sys.argv[1:]:
    name, _, envs = spec.partition(":")
    env = dict(os.environ)
    if name != "product":
        env["T41RX_LIB"] = os.path.join(root, "t41_sdr_amd", "abl", "libt41rx_%s.so" % name)
    for kv in filter(None, envs.split(",")):
        k, v = kv.split("=")
        env[k] = v
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "shape_sweep2.py")] + shapes, env=env, capture_output=True, text=True)
    rows = [l.strip().replace("channels x ", "x").replace(" frames per launch:", ":") for l in out.stdout.splitlines() if "channels" in l]
    print("%-28s %s" % (spec, " | ".join(rows) if rows else out.stderr[-300:]), flush=True)

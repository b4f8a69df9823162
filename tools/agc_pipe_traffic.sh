#!/bin/bash
# GPU box: fabric traffic of the pipelined AGC kernel per channel-frame at two batch sizes (does the slots' footprint
# decide whether they are served from L2?)  usage: tools/agc_pipe_traffic.sh [nch ...]
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/pipe_traffic"
mkdir -p "$OUT"
cd /tmp
export T41RX_PROBE_REPS=6
for N in ${@:-4096 2048}; do
  export T41RX_PROBE_NCH=$N
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/$N/$C -o p -- python3 "$ROOT/tools/agc_pipe_probe.py" time 32 > $OUT/$N.$C.log 2>&1 || echo "pmc $C $N failed"
    F=$(find $OUT/$N/$C -name "*counter_collection.csv" | head -1)
    python3 - "$F" $C $N <<'P'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "rx512" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
n = int(sys.argv[3])
print(sys.argv[2], "nch", n, "dispatches", len(v), "KiB per channel-frame (raw): %.2f" % (sum(v) / len(v) / n / 32))
P
  done
done
find $OUT -name "*.csv" -size +2M -delete

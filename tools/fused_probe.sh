#!/bin/bash
# GPU box: FFT_LENGTH 4096 single-kernel form against the two-kernel pipeline: kernel-trace stats and HBM counters
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/fusedprof"
mkdir -p "$OUT"
cd /tmp
for V in fused old; do
  if [ $V = old ]; then export T41RX_FUSE_FRONT=0; else unset T41RX_FUSE_FRONT; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$V/trace" -o t -- python3 "$ROOT/bench.py" --workload fft4096 --steps 12 --warmup 4 --no-cpu-baseline > "$OUT/$V.trace.log" 2>&1 || echo "trace $V failed"
  for C in FETCH_SIZE WRITE_SIZE; do
    T41RX_BENCH_NOCHECK=1 timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d "$OUT/$V/$C" -o p -- python3 "$ROOT/bench.py" --workload fft4096 --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/$V.$C.log" 2>&1 || echo "pmc $C $V failed"
  done
  find "$OUT/$V" -name "*kernel_trace.csv" -delete
  echo "== $V"
  for F in $(find "$OUT/$V/trace" -name "*kernel_stats.csv"); do grep "t41::" "$F" | cut -c1-200; done
  python3 "$ROOT/tools/pmc_summary.py" "$OUT/$V/FETCH_SIZE" t41:: | cut -c1-160
  python3 "$ROOT/tools/pmc_summary.py" "$OUT/$V/WRITE_SIZE" t41:: | cut -c1-160
done
find "$OUT" -name "*.csv" -size +4M -delete

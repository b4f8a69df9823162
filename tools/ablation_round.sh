#!/bin/bash
# GPU box: the ablation table of the fused 512 kernel with its evidence.
#   1. tools/ablation_table.py run  -> gpurun_out/<TAG>_ablation_ssb.md / .json (HIP events, interleaved rounds:
#      cumulative cuts, leave-one-out builds, prefetch depths)
#   2. (unless NOPROF=1) one rocprofv3 --kernel-trace --stats run per build -> gpurun_out/<TAG>_ablation_stats.csv
# The builds come from `python tools/ablation_table.py build` (run where hipcc is; they travel under t41_sdr_amd/abl/).
# usage: tools/ablation_round.sh TAG [layout]
TAG=${1:-r04}
LAYOUT=${2:-channel}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
python3 "$ROOT/tools/ablation_table.py" run --rounds 3 --frames 32 --reps 60 --layout $LAYOUT --out "$OUT/${TAG}_ablation_ssb.md" > "$OUT/${TAG}_ablation_run.log" 2>&1 || { echo "ablation run failed"; tail -5 "$OUT/${TAG}_ablation_run.log"; exit 1; }
cat "$OUT/${TAG}_ablation_ssb.md"
[ "$NOPROF" = 1 ] && exit 0
CSV="$OUT/${TAG}_ablation_stats.csv"
echo '"build","Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"' > "$CSV"
mkdir -p "$OUT/ablprof_$TAG"
cd /tmp
for V in product abl1 abl2 abl3 abl4 abl5 abl6 abl7 abl9 loo1 loo2 loo3 loo4 loo5 loo6 pf1 pf0; do
  if [ "$V" = product ]; then unset T41RX_LIB; else export T41RX_LIB="$ROOT/t41_sdr_amd/abl/libt41rx_$V.so"; [ -f "$T41RX_LIB" ] || continue; fi
  D="$OUT/ablprof_$TAG/$V"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o t -- python3 "$ROOT/tools/ablation_table.py" one $V --frames 32 --reps 30 --layout $LAYOUT > "$D.log" 2>&1 || { echo "trace $V failed"; continue; }
  for F in $(find "$D" -name "*kernel_stats.csv"); do grep "t41::rx512_kernel" "$F" | sed "s/^/\"$V\",/" >> "$CSV"; done
  find "$D" -name "*kernel_trace.csv" -delete
done
unset T41RX_LIB
cat "$CSV" | cut -c1-220

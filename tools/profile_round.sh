#!/bin/bash
# Round profiles (GPU box): for each bench workload a rocprofv3 kernel-trace summary and the two HBM
# PMC passes (FETCH_SIZE, WRITE_SIZE: separate processes, counters only), plus the SQ utilisation
# passes for the headline workload.  Output under gpurun_out/profile_<tag>/; tools/collect_profiles.py
# copies the summaries into profiles/ and rewrites profiles/hbm_traffic.json.
# usage: tools/profile_round.sh TAG [workload ...]
TAG=$1; shift
WL=${@:-ssb ssb_32fpl ssb_time_major ssb_1fpl ssb_4fpl nfm nfm_atan am sam sam_agc ssb_agc ssb_q15 ssb_agc_q15 fft4096 ssb_notch ssb_kim ssb_spectral}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/profile_$TAG"
mkdir -p "$OUT"
export T41RX_BENCH_NOCHECK=1   # (the parity replay is bench.py's own business; here only the timed launches' kernels count)
cd /tmp
# one un-profiled run first: a fresh box's first seconds of load (clock / power management settling, first-touch of the
# allocator) otherwise land in the first workload's trace (round 5: 782 us +- 82 per launch there, 693 in the bench minutes later)
python3 "$ROOT/bench.py" --no-other-workloads --workload ssb --steps 200 --warmup 20 --no-cpu-baseline > $OUT/settle.log 2>&1
for W in $WL; do
  echo "== $W"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$W/trace -o t -- python3 "$ROOT/bench.py" --no-other-workloads --workload $W --steps 30 --warmup 5 --no-cpu-baseline > $OUT/$W.trace.log 2>&1 || echo "trace $W failed"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $OUT/$W/$C -o p -- python3 "$ROOT/bench.py" --no-other-workloads --workload $W --steps 6 --warmup 2 --no-cpu-baseline > $OUT/$W.$C.log 2>&1 || echo "pmc $C $W failed"
  done
  tail -1 $OUT/$W.trace.log | cut -c1-200
done
# SQ passes for the headline workload
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
 "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
 "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
)
i=0
for CNT in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $CNT --output-format csv -d $OUT/ssb/sq$i -o p -- python3 "$ROOT/bench.py" --no-other-workloads --workload ssb --steps 6 --warmup 2 --no-cpu-baseline > $OUT/ssb.sq$i.log 2>&1 || echo "sq pass $i failed"
done
# keep the merge small: per-dispatch traces are not needed, and of the counter CSVs only our kernels' rows
find $OUT -name "*kernel_trace.csv" -delete
for F in $(find $OUT -name "*counter_collection.csv"); do
  (head -1 $F; grep "t41::" $F) > $F.tmp && mv $F.tmp $F
done
find $OUT -name "*.csv" -size +8M -delete
echo done

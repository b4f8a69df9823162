#!/bin/bash
# SQ counters of the noise-reduction stage kernels ("nrspec_kernel<2> 9 x Kim's with no counter evidence why")
# usage (GPU box): tools/sq_nr.sh TAG
TAG=${1:-r05}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/sq_nr_$TAG"
mkdir -p "$OUT"
export T41RX_BENCH_NOCHECK=1
cd /tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
 "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
)
for W in ssb_kim ssb_spectral ssb_notch; do
  i=0
  for CNT in "${PASSES[@]}"; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --pmc $CNT --output-format csv -d $OUT/$W/sq$i -o p -- python3 "$ROOT/bench.py" --no-other-workloads --workload $W --steps 4 --warmup 2 --no-cpu-baseline > $OUT/$W.sq$i.log 2>&1 || echo "sq pass $i $W failed"
  done
done
for F in $(find $OUT -name "*counter_collection.csv"); do
  (head -1 $F; grep "t41::" $F) > $F.tmp && mv $F.tmp $F
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for w in ("ssb_kim", "ssb_spectral", "ssb_notch"):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(out, w, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].replace("void t41::", "")[:48]
            vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in vals.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        waves = m.get("SQ_WAVES", 1)
        print("%-14s %-48s waves %6d | per wave: VALU %9.0f SALU %8.0f LDS %8.0f VMEM %6.0f | wave-cycles %10.0f | VALU-active %.2f of wave-cycles, waiting on an instruction's issue %.2f, parked %.2f" % (
            w, k, waves, m.get("SQ_INSTS_VALU", 0) / waves, m.get("SQ_INSTS_SALU", 0) / waves, m.get("SQ_INSTS_LDS", 0) / waves, m.get("SQ_INSTS_VMEM", 0) / waves,
            4 * m.get("SQ_WAVE_CYCLES", 0) / waves, m.get("SQ_ACTIVE_INST_VALU", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
            m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
PY

#!/bin/bash
# GPU box: the two SQ utilisation passes of tools/profile_round.sh for the ssb_agc workload (the pipelined AGC kernel);
# prints the per-dispatch means (profiles/r03_pmc_sq_ssb_agc.txt)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$ROOT
OUT=$ROOT/gpurun_out/sq_agc
mkdir -p $OUT
cd /tmp
export T41RX_BENCH_NOCHECK=1
i=0
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $CNT --output-format csv -d $OUT/sq$i -o p -- python3 "$ROOT/bench.py" --no-other-workloads --workload ssb_agc --steps 6 --warmup 2 --no-cpu-baseline > $OUT/sq$i.log 2>&1 || echo "sq pass $i failed"
done
python3 - <<'P'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/sq_agc"
for p in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    acc = {}
    for r in csv.DictReader(open(p)):
        if "rx512" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-24s mean %.6g per dispatch (%d)" % (k, sum(v) / len(v), len(v)))
P
find $OUT -name "*.csv" -size +1M -delete

#!/bin/bash
# rocprofv3 PMC passes over a short bench run (GPU box): SQ utilisation / stall / fetch counters and HBM bytes.
# usage: tools/prof_pmc2.sh <tag> [bench args...]     (one process per pass, counters only)
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM"
 "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_BUSY_CU_CYCLES"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
)
i=0
for CNT in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $CNT --output-format csv -d $OUT/pass$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; }
done
for K in ${PMC_KERNELS:-rx512_kernel}; do echo "== $K"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT $K | tee $OUT/summary_$K.txt; done
find $OUT -name "*.csv" -size +2M -delete

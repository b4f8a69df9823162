#!/bin/bash
# one PMC pass (instruction / LDS counters) per experiment build: tools/prof_pmc_variants.sh TAG name1 name2 ...
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcv_$TAG
mkdir -p $OUT
cd /tmp
for V in "$@"; do
  if [ "$V" = product ]; then unset T41RX_LIB; else export T41RX_LIB=$GRAFT_REPO_ROOT/t41_sdr_amd/abl/libt41rx_$V.so; fi
  T41RX_BENCH_NOCHECK=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/$V -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --frames-per-launch 16 > $OUT/$V.log 2>&1 || { echo "$V failed"; tail -3 $OUT/$V.log; }
  echo "== $V"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/$V
done
find $OUT -name "*.csv" -size +1M -delete

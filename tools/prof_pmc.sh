#!/bin/bash
# rocprofv3 PMC passes over a short bench run (GPU box).  Each pass = its own process, counters
# only (no tracing flags besides the implicit kernel dispatch records).
# usage: tools/prof_pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
i=0
for CNT in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $CNT --output-format csv -d $OUT/pass$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline ${T41RX_PROF_ARGS} > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; }
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT | tee $OUT/summary.txt

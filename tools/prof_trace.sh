#!/bin/bash
# rocprofv3 kernel-trace summary of a bench run (GPU box): tools/prof_trace.sh TAG [bench args...]
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-300
find $OUT -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-260
find $OUT -name "*kernel_trace.csv" -size +3M -delete

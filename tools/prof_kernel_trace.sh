#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command (run on the GPU box via gpurun).
# usage: tools/prof_kernel_trace.sh <tag>
set -e
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline ${T41RX_PROF_ARGS} > $OUT/bench_under_prof.log 2>&1
ls -R $OUT | head -30

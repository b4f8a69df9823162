#!/bin/bash
# round 5: the notch's window requested during the previous sample's chain (product) against at the top of its own step (abl/libt41rx_anrold.so):
# the stage's tests, both builds bit for bit, interleaved timing
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OTHER=${1:-anrold}
timeout -k 10 600 python -m pytest tests/test_noise_reduction.py -m gpu -x -q > gpurun_out/r05_pytest_anr.log 2>&1
rc=$?; echo "nr tests rc $rc"; tail -n 2 gpurun_out/r05_pytest_anr.log | cut -c1-200
[ $rc = 0 ] || exit 3
timeout -k 10 300 python tools/anr_ab_check.py $OTHER > gpurun_out/r05_anr_bits.txt 2>&1; echo "bit check rc $?"; tail -c 300 gpurun_out/r05_anr_bits.txt
: > gpurun_out/r05_ab_anr.txt
for r in 1 2 3; do
  for v in product $OTHER; do
    if [ $v = product ]; then unset T41RX_LIB; else export T41RX_LIB=$PWD/t41_sdr_amd/abl/libt41rx_$v.so; fi
    T41RX_BENCH_NOCHECK=1 timeout -k 10 120 python bench.py --workload ssb_notch --steps 10 --warmup 3 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v ssb_notch', d['roofline']['us_per_frame'], d['roofline']['frac'])" >> gpurun_out/r05_ab_anr.txt || exit 3
  done
done
unset T41RX_LIB
cat gpurun_out/r05_ab_anr.txt

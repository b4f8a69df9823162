#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fft4096 or long_fft or segment_run" > gpurun_out/r05_pytest_gpu7.log 2>&1
rc=$?; tail -n 4 gpurun_out/r05_pytest_gpu7.log | cut -c1-300
[ $rc -ge 124 ] && exit $rc
: > gpurun_out/r05_ab_freshwv.txt
for r in 1 2 3 4; do
  for v in product nofreshwv; do
    if [ $v = product ]; then unset T41RX_LIB; else export T41RX_LIB=$PWD/t41_sdr_amd/abl/libt41rx_$v.so; fi
    T41RX_BENCH_NOCHECK=1 timeout -k 10 120 python bench.py --workload fft4096 --steps 30 --warmup 8 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v', d['roofline']['us_per_frame'], d['roofline']['frac'])" >> gpurun_out/r05_ab_freshwv.txt || exit 3
  done
done
unset T41RX_LIB
cat gpurun_out/r05_ab_freshwv.txt
timeout -k 10 400 python tools/ab_probe.py product tapsum --agc 1 --rounds 3 --reps 60 > gpurun_out/r05_ab_taps_agc.txt 2>&1
cat gpurun_out/r05_ab_taps_agc.txt
echo "== the round-4 check (experiment build): do the new test and the soak catch it?"
T41RX_LIB=$PWD/t41_sdr_amd/abl/libt41rx_r04check.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "min_volts" > gpurun_out/r05_r04check_test.log 2>&1
echo "pytest with the round-4 check: rc $? (non-zero expected)"; tail -n 5 gpurun_out/r05_r04check_test.log | cut -c1-200
T41RX_LIB=$PWD/t41_sdr_amd/abl/libt41rx_r04check.so timeout -k 10 300 python tools/pipe_soak.py 150 > gpurun_out/r05_r04check_soak.log 2>&1
echo "soak with the round-4 check: rc $? (non-zero expected)"; grep -c MISMATCH gpurun_out/r05_r04check_soak.log; tail -n 1 gpurun_out/r05_r04check_soak.log | cut -c1-300

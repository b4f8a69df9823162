#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_shapes_and_boundary.py -m gpu -q -k "fft4096 or long_fft or segment_run or checkpoint or state or shapes" > gpurun_out/r05_pytest_gpu11.log 2>&1
echo "rc $?"; tail -n 3 gpurun_out/r05_pytest_gpu11.log | cut -c1-200
: > gpurun_out/r05_ab_prevreg2.txt
for r in 1 2 3; do
  for v in product prevreg_nox2 prevglob; do
    if [ $v = product ]; then unset T41RX_LIB; else export T41RX_LIB=$PWD/t41_sdr_amd/abl/libt41rx_$v.so; fi
    T41RX_BENCH_NOCHECK=1 timeout -k 10 120 python bench.py --workload fft4096 --steps 30 --warmup 8 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v', d['roofline']['us_per_frame'], d['roofline']['frac'])" >> gpurun_out/r05_ab_prevreg2.txt || exit 3
  done
done
unset T41RX_LIB
cat gpurun_out/r05_ab_prevreg2.txt
timeout -k 10 200 python bench.py --workload fft4096 --steps 30 --warmup 8 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('with parity:', d['roofline']['us_per_frame'], d['roofline']['frac'], d['parity_check']['ok'], d['parity_check']['max_block_rel_err'], d['parity_check']['replay_bit_identical_to_timed_run'])"

#!/bin/bash
# round 5: the spectral noise reduction's bin loop with the gains in registers (product) against round 4's LDS loop
# (abl/libt41rx_nrold.so): the stage's tests, both builds bit for bit, then interleaved timing
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_noise_reduction.py tests/test_buffer_layout.py -m gpu -x -q > gpurun_out/r05_pytest_nrspec.log 2>&1
rc=$?; echo "nr tests rc $rc"; tail -n 3 gpurun_out/r05_pytest_nrspec.log | cut -c1-200
[ $rc = 0 ] || exit 3
timeout -k 10 300 python tools/anr_ab_check.py nrold > gpurun_out/r05_nrspec_bits.txt 2>&1; echo "bit check rc $?"; tail -n 12 gpurun_out/r05_nrspec_bits.txt
: > gpurun_out/r05_ab_nrspec.txt
for r in 1 2 3; do
  for v in product nrold; do
    if [ $v = product ]; then unset T41RX_LIB; else export T41RX_LIB=$PWD/t41_sdr_amd/abl/libt41rx_$v.so; fi
    for w in ssb_spectral ssb_kim; do
      T41RX_BENCH_NOCHECK=1 timeout -k 10 120 python bench.py --workload $w --steps 12 --warmup 3 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v $w', d['roofline']['us_per_frame'], d['roofline']['frac'], d['ms_per_step'])" >> gpurun_out/r05_ab_nrspec.txt || exit 3
    done
  done
done
unset T41RX_LIB
cat gpurun_out/r05_ab_nrspec.txt

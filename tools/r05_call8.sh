#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "min_volts" > gpurun_out/r05_pytest_gpu6.log 2>&1
rc=$?; tail -n 6 gpurun_out/r05_pytest_gpu6.log | cut -c1-300
[ $rc -ge 124 ] && exit $rc
timeout -k 10 700 python tools/ab_probe.py product maxilp maxmem itilp --rounds 4 --reps 80 > gpurun_out/r05_ab_sched.txt 2>&1
rc=$?; cat gpurun_out/r05_ab_sched.txt

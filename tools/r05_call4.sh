#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 300 python tools/agc_thresh_probe.py > gpurun_out/r05_agc_thresh_probe.txt 2>&1
rc=$?; tail -n 60 gpurun_out/r05_agc_thresh_probe.txt | cut -c1-260
[ $rc -ge 124 ] && exit $rc
timeout -k 10 400 python tools/ab_probe.py product tapsum --rounds 4 --reps 80 > gpurun_out/r05_ab_taps.txt 2>&1
rc=$?; cat gpurun_out/r05_ab_taps.txt
[ $rc -ge 124 ] && exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -q -s --deselect "tests/test_gpu_parity.py::test_agc_min_volts_raised_mid_stream" > gpurun_out/r05_pytest_gpu2.log 2>&1
rc=$?; tail -n 25 gpurun_out/r05_pytest_gpu2.log | cut -c1-300; grep -n "HIP-vs-exact\|per frame --" gpurun_out/r05_pytest_gpu2.log | cut -c1-400
[ $rc -ge 124 ] && exit $rc
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench2.json 2> gpurun_out/r05_bench2.err
echo "bench rc $?"; tail -n 3 gpurun_out/r05_bench2.err

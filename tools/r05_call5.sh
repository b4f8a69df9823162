#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "min_volts or am_hip" > gpurun_out/r05_pytest_gpu3.log 2>&1
rc=$?; tail -n 12 gpurun_out/r05_pytest_gpu3.log | cut -c1-300
[ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python tools/graph_probe.py > gpurun_out/r05_graph_probe.txt 2>&1
rc=$?; tail -n 3 gpurun_out/r05_graph_probe.txt
[ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python tools/graph_probe.py --agc 1 >> gpurun_out/r05_graph_probe.txt 2>&1
rc=$?; tail -n 1 gpurun_out/r05_graph_probe.txt
[ $rc -ge 124 ] && exit $rc
timeout -k 10 400 python tools/nr_error_stats.py > gpurun_out/r05_nr_error_stats.txt 2>&1
tail -n 9 gpurun_out/r05_nr_error_stats.txt

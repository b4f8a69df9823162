#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_noise_reduction.py tests/test_buffer_layout.py tests/test_multi_gpu_host.py -m gpu -q -s -k "min_volts or stage_in_isolation or whole_path_with_nr or exact_model or time_major or rccl or cpp_multi" > gpurun_out/r05_pytest_gpu4.log 2>&1
rc=$?; tail -n 12 gpurun_out/r05_pytest_gpu4.log | cut -c1-300
[ $rc -ge 124 ] && exit $rc
timeout -k 10 500 python tools/ab_probe.py product prioage1 prioage2 --rounds 4 --reps 80 > gpurun_out/r05_ab_prio.txt 2>&1
rc=$?; cat gpurun_out/r05_ab_prio.txt

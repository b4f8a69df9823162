#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "restored_below" > gpurun_out/r05_pytest_gpu8.log 2>&1
echo "product: rc $? (0 expected)"; tail -n 3 gpurun_out/r05_pytest_gpu8.log | cut -c1-200
T41RX_LIB=$PWD/t41_sdr_amd/abl/libt41rx_r04check.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "restored_below" > gpurun_out/r05_r04check_test2.log 2>&1
echo "round-4 check: rc $? (non-zero expected)"; tail -n 6 gpurun_out/r05_r04check_test2.log | cut -c1-200

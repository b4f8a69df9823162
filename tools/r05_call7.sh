#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 900 python -m pytest tests -m gpu -q -s > gpurun_out/r05_pytest_gpu5.log 2>&1
rc=$?; tail -n 8 gpurun_out/r05_pytest_gpu5.log | cut -c1-300
[ $rc -ge 124 ] && exit $rc
[ $rc -ne 0 ] && echo "PYTEST FAILED rc $rc"
echo "profile round start $(date +%T)"
timeout -k 10 2400 bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
rc=$?; tail -n 6 gpurun_out/r05_profile_round.log | cut -c1-200; echo "profile round end $(date +%T) rc $rc"
[ $rc -ge 124 ] && exit $rc
t0=$(date +%s)
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench3.json 2> gpurun_out/r05_bench3.err
echo "bench rc $? in $(( $(date +%s) - t0 )) s"; tail -n 2 gpurun_out/r05_bench3.err | cut -c1-200

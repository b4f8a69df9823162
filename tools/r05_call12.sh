#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 400 python -m pytest tests/test_nfm_atan_variant.py tests/test_gpu_parity.py -m gpu -q -k "nfm or restored_below" > gpurun_out/r05_pytest_gpu9.log 2>&1
echo "rc $?"; tail -n 3 gpurun_out/r05_pytest_gpu9.log | cut -c1-200
for r in 1 2 3; do
timeout -k 10 200 python bench.py --workload nfm_atan --steps 30 --warmup 8 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); pc = d['parity_check']; print('nfm_atan', d['roofline']['us_per_frame'], d['roofline']['frac'], pc['ok'], pc['frames_over_tolerance'], pc['frames_checked'], pc['median_block_rel_err'])"
done

#!/bin/bash
# the round's closing GPU call: every -m gpu test, smoke(), the profile round (kernel stats + HBM PMC per workload, SQ passes), the default bench line
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r05_pytest_gpu_final.log 2>&1
rc=$?; tail -n 4 gpurun_out/r05_pytest_gpu_final.log | cut -c1-300
[ $rc -ge 124 ] && exit $rc
[ $rc -ne 0 ] && echo "PYTEST FAILED rc $rc"
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/r05_smoke.log 2>&1; echo "smoke rc $?"; tail -n 2 gpurun_out/r05_smoke.log | cut -c1-200
timeout -k 10 900 bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
rc=$?; tail -n 2 gpurun_out/r05_profile_round.log | cut -c1-200; echo "profile round rc $rc"
[ $rc -ge 124 ] && exit $rc
t0=$(date +%s)
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
echo "bench rc $? in $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_final.json').readline())
print(d['value'], d['roofline']['frac'], d['roofline']['us_per_frame'], d['roofline']['traffic'])
PY

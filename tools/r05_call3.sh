#!/bin/bash
# round 5, GPU call: full -m gpu suite, the energy table (VERDICT r04 item 1's stop-rule deliverable), the default bench line
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 900 python -m pytest tests -m gpu -q -x -s > gpurun_out/r05_pytest_gpu1.log 2>&1
rc=$?
tail -n 15 gpurun_out/r05_pytest_gpu1.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 420 python tools/energy_probe.py product abl7 abl9 firplain loo13 loo14 loo2 loo3 loo4 product --seconds 12 --out gpurun_out/r05_energy_ssb.md > gpurun_out/r05_energy.log 2>&1
rc=$?
tail -n 25 gpurun_out/r05_energy.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench1.json 2> gpurun_out/r05_bench1.err
echo "bench rc $?"; cut -c1-600 gpurun_out/r05_bench1.json; tail -n 5 gpurun_out/r05_bench1.err

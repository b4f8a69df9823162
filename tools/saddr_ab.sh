#!/bin/bash
# round 5: scalar-base addressing of the streaming and table requests, the oscillator phases split into a scalar and a
# lane part, no zero start values for conditionally loaded registers, DPP-fused scan steps (product) against round 4's
# forms (r5base) and the product without the scan change (noscan): parity first, then interleaved timing per workload
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r05_pytest_saddr.log 2>&1
rc=$?; echo "gpu tests rc $rc"; tail -n 3 gpurun_out/r05_pytest_saddr.log | cut -c1-200
[ $rc = 0 ] || exit 3
timeout -k 10 400 python tools/ab_probe.py product noregtail r5base --rounds 4 --reps 60 > gpurun_out/r05_ab_saddr.txt 2>&1 || exit 3
cat gpurun_out/r05_ab_saddr.txt
: > gpurun_out/r05_ab_saddr_modes.txt
for r in 1 2 3; do
  for v in product r5base; do
    if [ $v = product ]; then unset T41RX_LIB; else export T41RX_LIB=$PWD/t41_sdr_amd/abl/libt41rx_$v.so; fi
    for w in ${WORKLOADS:-fft4096 nfm am sam sam_agc ssb_agc ssb_agc_q15}; do
      T41RX_BENCH_NOCHECK=1 timeout -k 10 120 python bench.py --workload $w --steps 30 --warmup 8 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v $w', d['roofline']['us_per_frame'], d['roofline']['frac'])" >> gpurun_out/r05_ab_saddr_modes.txt || exit 3
    done
  done
done
unset T41RX_LIB
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r05_ab_saddr_modes.txt"):
    v, w, us, fr = l.split()
    d[(w, v)].append(float(us))
for (w, v), x in sorted(d.items()):
    print(w, v, "median us/frame", statistics.median(x), x)
PY

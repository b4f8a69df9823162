#!/bin/bash
# frames-per-launch sweep of configs[3] (4096-point) and of the AGC-on workload on ONE box: 32 / 64 / 128, twice
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r05_fpl_sweep_others.txt
: > $out
for r in 1 2; do
  for w in fft4096 ssb_agc; do
    for f in 32 64 128; do
      steps=$(( 1600 / f ))
      T41RX_BENCH_NOCHECK=1 timeout -k 10 200 python bench.py --workload $w --frames-per-launch $f --steps $steps --warmup 8 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$w fpl $f steps $steps', d['roofline']['us_per_frame'], d['roofline']['frac'])" >> $out || exit 3
    done
  done
done
cat $out

#!/bin/bash
# frames-per-launch sweep of the headline workload on ONE box: 32 / 64 / 128 / 32 again
set -e
out=gpurun_out/r05_fpl_sweep.txt
: > $out
for f in 32 64 128 32 64 128; do
  steps=$(( 3200 / f ))
  echo "fpl $f steps $steps" >> $out
  python bench.py --frames-per-launch $f --steps $steps --warmup 10 --no-other-workloads --no-cpu-baseline 2>>gpurun_out/r05_fpl_sweep.err | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(json.dumps({k: d[k] for k in ('value', 'ms_per_step')} | {'frac': d['roofline']['frac'], 'us_per_frame': d['roofline']['us_per_frame'], 'parity': d.get('parity_check', {}).get('max_block_rel_err'), 'ok': d.get('parity_check', {}).get('ok')}))" >> $out
done
cat $out

#!/bin/bash
# scheduler strategies per kernel family (the split makes per-TU flags possible): fft4096, AGC on, SAM
cd "${GRAFT_REPO_ROOT:-.}"
: > gpurun_out/r05_ab_sched2.txt
for r in 1 2 3; do
  for v in product maxilp maxmem itilp itminreg; do
    if [ $v = product ]; then unset T41RX_LIB; else export T41RX_LIB=$PWD/t41_sdr_amd/abl/libt41rx_$v.so; fi
    for w in fft4096 ssb_agc sam; do
      T41RX_BENCH_NOCHECK=1 timeout -k 10 120 python bench.py --workload $w --steps 24 --warmup 6 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v $w', d['roofline']['us_per_frame'])" >> gpurun_out/r05_ab_sched2.txt || exit 3
    done
  done
done
unset T41RX_LIB
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open('gpurun_out/r05_ab_sched2.txt'):
    v, w, t = l.split(); d[(w, v)].append(float(t))
for k in sorted(d): print(k, round(statistics.median(d[k]), 3), d[k])
PY

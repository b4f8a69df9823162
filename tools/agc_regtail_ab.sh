#!/bin/bash
# round 5: the /4 window's register tail in the AGC / SAM kernels too (regtailagc) against the product: AGC tests, interleaved timing
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
: > gpurun_out/r05_ab_regtail_agc.txt
for r in 1 2 3; do
  for v in product regtailagc; do
    if [ $v = product ]; then unset T41RX_LIB; else export T41RX_LIB=$PWD/t41_sdr_amd/abl/libt41rx_$v.so; fi
    for w in ssb_agc ssb_agc_q15 sam sam_agc; do
      T41RX_BENCH_NOCHECK=1 timeout -k 10 200 python bench.py --workload $w --steps 16 --warmup 4 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v $w', d['roofline']['us_per_frame'], d['roofline']['frac'])" >> gpurun_out/r05_ab_regtail_agc.txt || exit 3
    done
  done
done
unset T41RX_LIB
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/r05_ab_regtail_agc.txt"):
    v, w, us, fr = l.split()
    d[(w, v)].append(float(us))
for (w, v), x in sorted(d.items()):
    print(w, v, "median us/frame", statistics.median(x), x)
PY

#!/bin/bash
# AGC on, pipelined kernel: popped samples in registers of their owner (product) against through the slot (nokeepre): time, traffic, bit identity
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "agc" > gpurun_out/r05_pytest_keepre.log 2>&1
echo "agc tests rc $?"; tail -n 3 gpurun_out/r05_pytest_keepre.log | cut -c1-200
: > gpurun_out/r05_ab_keepre.txt
for r in 1 2 3; do
  for v in product nokeepre; do
    if [ $v = product ]; then unset T41RX_LIB; else export T41RX_LIB=$PWD/t41_sdr_amd/abl/libt41rx_$v.so; fi
    for w in ssb_agc ssb_agc_q15; do
      T41RX_BENCH_NOCHECK=1 timeout -k 10 120 python bench.py --workload $w --steps 30 --warmup 8 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v $w', d['roofline']['us_per_frame'], d['roofline']['frac'])" >> gpurun_out/r05_ab_keepre.txt || exit 3
    done
  done
done
cat gpurun_out/r05_ab_keepre.txt
ROOT=$PWD
cd /tmp
for v in product nokeepre; do
  if [ $v = product ]; then unset T41RX_LIB; else export T41RX_LIB=$ROOT/t41_sdr_amd/abl/libt41rx_$v.so; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    T41RX_BENCH_NOCHECK=1 timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/pmc_keepre/$v/$C -o p -- python3 $ROOT/bench.py --no-other-workloads --workload ssb_agc --steps 6 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/pmc_keepre_$v.$C.log 2>&1 || echo "pmc failed"
  done
done
unset T41RX_LIB
python3 - <<'PY'
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for v in ("product", "nokeepre"):
    tot = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for p in glob.glob(os.path.join(root, "gpurun_out/pmc_keepre", v, C, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(p)):
                if "rx512_kernel" in row["Kernel_Name"] and row["Counter_Name"] == C:
                    vals.append(float(row["Counter_Value"]))
        tot[C] = sum(vals) / max(len(vals), 1)
    rd, wr = tot["FETCH_SIZE"] * 1024 * 2, tot["WRITE_SIZE"] * 1024
    alg = 12.0 * 4096 * 32 * 2048
    print(v, "read %.0f MB write %.0f MB ratio %.4f" % (rd / 1e6, wr / 1e6, (rd + wr) / alg))
PY
find $ROOT/gpurun_out/pmc_keepre -name "*.csv" -size +2M -delete

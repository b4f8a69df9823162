#!/bin/bash
# GPU box: the pipelined AGC kernel vs the barrier form -- bit identity (tools/agc_pipe_probe.py), bench lines of the
# ssb_agc workload with either, and the diagnostic build's counters (tools/build_variant.sh pstat -DT41RX_PIPE_STAT)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT" && mkdir -p gpurun_out
B="python bench.py --workload ssb_agc --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads"
timeout -k 10 300 python tools/agc_pipe_probe.py check > gpurun_out/pipe_check.log 2>&1 || { tail -5 gpurun_out/pipe_check.log; exit 1; }
grep -c "True" gpurun_out/pipe_check.log; grep "False" gpurun_out/pipe_check.log
timeout -k 10 200 $B > gpurun_out/pipe_b1.log 2>&1 || exit 1
T41RX_AGC_PIPE=0 timeout -k 10 200 $B > gpurun_out/pipe_b0.log 2>&1 || exit 1
if [ -f t41_sdr_amd/abl/libt41rx_pstat.so ]; then
  T41RX_PIPE_STAT=1 T41RX_LIB=$PWD/t41_sdr_amd/abl/libt41rx_pstat.so timeout -k 10 200 $B > gpurun_out/pipe_bs.log 2>&1 || exit 1
fi
for f in gpurun_out/pipe_b1.log gpurun_out/pipe_b0.log gpurun_out/pipe_bs.log; do
  [ -f $f ] && grep -o "\"ms_per_step\": [0-9.]*\|pipe_stat.*\|\"ok\": [a-z]*\|max_block_rel_err\": [0-9.e-]*\|\"traffic\": [0-9]*" $f | tr '\n' ' '; echo
done

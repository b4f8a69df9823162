"""GPU box: the pipelined kernels' bounded waits report instead of hanging -- with a build whose spin cap is 1
(tools/build_variant.sh pcap1 -DT41RX_PIPE_SPINCAP=1; T41RX_LIB=.../libt41rx_pcap1.so) t41rx_get_state() must refuse
with T41RX_ERR_STATE after a pipelined call and accept again after t41rx_reset(); the product build must not count any.
usage: [T41RX_LIB=...] python tools/pipe_timeout_probe.py"""
import sys, os
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests")]
import numpy as np, torch
import t41_sdr_amd as T
import siggen
nch, nfr, L = 64, 12, 2048
nco = siggen.nco_grid(nch, seed=1)
I, Q = siggen.make_iq(nch, nfr * L, nco, seed=2)
rx = T.RxChain(nch, T.default_params(AGCMode=1), NCOFreq=nco)
rx.ProcessIQData(torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda())
try:
    rx.get_state()
    print("get_state: ok (no time-out counted)")
except T.T41RxError as e:
    print("get_state refused:", e.status, str(e)[:120])
rx.reset()
rx.get_state()
print("after reset: get_state ok")

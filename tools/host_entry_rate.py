"""PCIe-inclusive rate of the host-pointer entry points (t41rx_process_host / _host_q15): the
reference's own calling convention (caller-owned host arrays).  Never the bench.py figure.
usage (GPU box): python tools/host_entry_rate.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import t41_sdr_amd as T  # noqa: E402

L = 2048


def main():
    nch = 4096
    rng = np.random.default_rng(0)
    nco = (rng.integers(-860, 801, nch) * 50).astype(np.int32)
    rx = T.RxChain(nch, T.default_params(), NCOFreq=nco)
    I = (0.2 * rng.standard_normal((nch, L))).astype(np.float32)
    Q = (0.2 * rng.standard_normal((nch, L))).astype(np.float32)
    out = np.empty_like(I)
    for _ in range(3):
        rx.ProcessIQData(I, Q, out=out)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        rx.ProcessIQData(I, Q, out=out)
    dt = (time.perf_counter() - t0) / reps
    print("f32 host arrays (pageable): %.2f ms per 4096 x 2048 samples = %.0f MSamples/s (%.1f GB/s over PCIe both ways)"
          % (dt * 1e3, nch * L / dt / 1e6, 12.0 * nch * L / dt / 1e9))
    qI = np.clip(np.round(I * 32768), -32768, 32767).astype(np.int16)
    qQ = np.clip(np.round(Q * 32768), -32768, 32767).astype(np.int16)
    qo = np.empty_like(qI)
    for _ in range(3):
        rx.ProcessIQData_q15(qQ, qI, out=qo)
    t0 = time.perf_counter()
    for _ in range(reps):
        rx.ProcessIQData_q15(qQ, qI, out=qo)
    dt = (time.perf_counter() - t0) / reps
    print("q15 host arrays (pageable): %.2f ms per 4096 x 2048 samples = %.0f MSamples/s (%.1f GB/s)"
          % (dt * 1e3, nch * L / dt / 1e6, 6.0 * nch * L / dt / 1e9))


if __name__ == "__main__":
    main()

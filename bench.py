#!/usr/bin/env python3
"""bench.py -- throughput of the T41 RX hot path (batched ProcessIQData) on MI355X.

One "step" = one launch of the hot path over one batch: 4096 independent channels x
`--frames-per-launch` consecutive 2048-sample I/Q frames each (BASELINE.json configs[1]: batched
SSB RX chain, decimate-by-8 + 512-pt fast convolution + demod + interpolate-by-8).  Consecutive
steps are the following frames of the same channels: the streaming state of a channel stays on
chip inside a launch and is carried in HBM between launches.  Inputs are synthetic, generated on
the GPU and resident in HBM before the timed region; one launch touches more bytes than the
256 MiB Infinity Cache holds and a ring of distinct buffers is cycled, so the traffic is HBM's.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames-per-launch T] [--workload NAME]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes this process a launcher: it
starts N rank processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before
anything touches a GPU, waits for them and fails if any of them fails.  Under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks come from
the environment instead.  Either way `--gpus` must equal the world size RCCL reports.

Rank 0 prints ONE JSON line (the contract in the task description): whole-job MSamples/s (input
complex samples, all ranks), `roofline` (algorithmic 12 B/sample over the kernel's average launch
duration from HIP events, vs 8 TB/s; `traffic` from the committed rocprofv3 PMC passes if they
were taken from this very kernel source) and, at N = 1, `cpu_baseline` (the CPU oracle timed on
this host's cores on a bounded sample).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 192000.0
# --workload selects which BASELINE.json config is timed; the default (configs[1]) is the one
# the metric is quoted on, the others are reported in DESIGN.md
WORKLOADS = {
    "ssb": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000),
                name="configs[1]: batched SSB (USB 200-3000 Hz) RX chain, decimate-by-8 + 512-pt fast-conv + demod + "
                     "interpolate-by-8, 4096 channels x 2048 complex f32 samples per frame per GPU, per-channel NCO, AGC off"),
    "nfm": dict(batch=4096, fft=512, kw=dict(mode=3, FLoCut=200, FHiCut=3000, nfmFilterBW=12000),
                name="configs[2]: NFM path as the firmware runs it (quadri-correlator + limiter + real overlap-save audio "
                     "filter), 4096 channels x 2048 samples per frame"),
    "nfm_atan": dict(batch=4096, fft=512, kw=dict(mode=3, FLoCut=200, FHiCut=3000, nfmFilterBW=12000, nfm_demod=1),
                     name="configs[2] as BASELINE words it: the atan2 discriminator + de-emphasis the reference's source keeps "
                          "commented out (Demod.cpp:148-197, 324-392; t41rx_params.nfm_demod = 1), 4096 channels x 2048 samples per frame"),
    "am": dict(batch=4096, fft=512, kw=dict(mode=2, FLoCut=-3000, FHiCut=3000),
               name="AM path (AlphaBetaMag envelope, DC block, biquad low-pass; Process.cpp:697-707), 4096 channels x 2048 "
                    "samples per frame"),
    "sam": dict(batch=4096, fft=512, kw=dict(mode=8, FLoCut=-3000, FHiCut=3000), frames=8,
                name="synchronous AM (AMDecodeSAM, Demod.cpp:40-139: a per-sample PLL, serial in time; SURVEY 8f rank 4), "
                     "4096 channels x 2048 samples per frame"),
    "ssb_agc": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=1),
                    name="configs[1] with the firmware's default AGCMode = 1 (look-ahead AGC, DSP_Fn.cpp:504-631) instead of "
                         "the fixed gain: 4096 channels x 2048 samples per frame (SURVEY 8f rank 1)"),
    "ssb_q15": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000), q15=True,
                    name="configs[1] on the firmware's own sample format either side (q15 record-queue blocks in, "
                         "arm_float_to_q15 out; Process.cpp:102-111, 936): 6 B per input complex sample (SURVEY 8f rank 3)"),
    "fft4096": dict(batch=1024, fft=4096, kw=dict(mode=0, FLoCut=400, FHiCut=600), frames=32,
                    name="configs[3]: PSK31-like narrow USB filter, 4096-pt fast-conv (synthetic generalisation), "
                         "1024 channels x 16384 samples per frame"),
}
BYTES_PER_SAMPLE = 12.0     # 2 x f32 in + 1 x f32 out per input complex sample
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DEFAULT_FRAMES = 32         # consecutive frames per channel per launch (>= 100 consecutive frames per run)
KERNEL_SOURCES = ("t41_sdr_amd/csrc/rx_kernels.hip", "t41_sdr_amd/csrc/rx_kernels.hpp", "t41_sdr_amd/csrc/rx_internal.hpp")


def kernel_source_hash():
    """sha256 over the kernel sources: what a PMC measurement in profiles/ is valid for"""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_traffic(workload, frames):
    """HBM bytes per launch of this workload's kernel from the committed rocprofv3 PMC passes
    (profiles/hbm_traffic.json) -- only if they were measured on the kernel source this run uses
    and at this launch shape; otherwise None, with the reason on stderr."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            e = json.load(f)["workloads"][workload]
    except Exception:
        return None
    if e.get("source_hash") != kernel_source_hash():
        print("note: profiles/hbm_traffic.json[%s] was measured on another kernel source (hash %s, now %s): "
              "roofline.traffic = null until tools/prof_pmc.sh is re-run" % (workload, e.get("source_hash"), kernel_source_hash()),
              file=sys.stderr)
        return None
    if e.get("frames_per_launch", 1) != frames:
        print("note: profiles/hbm_traffic.json[%s] was measured at %s frames per launch, this run uses %d: "
              "roofline.traffic = null" % (workload, e.get("frames_per_launch", 1), frames), file=sys.stderr)
        return None
    return e["bytes_per_launch"]


# ------------------------------------------------------------------------------------------
# launcher: `--gpus N` without a torch.distributed environment
# ------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """Start n rank processes of this script and wait.  Nothing in this process has touched a GPU
    (no HIP call, no torch.cuda.*): the children are plain child processes, not exec replacements."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    deadline = time.time() + float(os.environ.get("T41RX_BENCH_TIMEOUT", "1500"))
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:  # one rank failed: the others would wait at a collective forever
                    q.terminate()
        if time.time() > deadline:
            for q in pending:
                q.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


# ------------------------------------------------------------------------------------------
# the workload
# ------------------------------------------------------------------------------------------
def synth_ring(torch, n_channels, nco_hz, ring, frames, frame_len, device, seed, mode=0):
    """RING launch buffers of `frames` consecutive frames of SURVEY 8d's synthetic signal, built on
    the GPU.  Returns lists of [n_channels, frames * frame_len] float32 tensors (I, Q)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    nco = torch.as_tensor(np.asarray(nco_hz, dtype=np.float64), device=device)
    amps = 0.05 + 0.25 * torch.rand(n_channels, 3, generator=g, device=device, dtype=torch.float64)
    freqs = -90000.0 + 180000.0 * torch.rand(n_channels, 3, generator=g, device=device, dtype=torch.float64)
    phases = 2 * np.pi * torch.rand(n_channels, 3, generator=g, device=device, dtype=torch.float64)
    audio = 400.0 + 2100.0 * torch.rand(n_channels, generator=g, device=device, dtype=torch.float64)
    freqs[:, 0] = 48000.0 - nco - audio  # lands at +audio Hz in the USB pass band (I sign flip, +Fs/4, -NCO)
    if mode == 8:  # SAM: the first tone is the carrier, 100 Hz off the tuned frequency
        freqs[:, 0] = 48000.0 - nco - 100.0
        amps[:, 0] = 0.3
    if mode == 3:  # NFM: no I sign flip; put a carrier on the tuned frequency
        freqs[:, 0] = -48000.0 + nco
        amps[:, 0] = 0.3
    Is, Qs = [], []
    # a few large chunks rather than many small ones: fewer dispatches (rocprofv3 --pmc has segfaulted inside torch's
    # element-wise launches when a profiled run made tens of thousands of them), ~256 MiB per float64 temporary
    chunk = frame_len
    while 2 * chunk * n_channels <= (1 << 25) and (frames * frame_len) % (2 * chunk) == 0:
        chunk *= 2
    for r in range(ring):
        bI = torch.empty(n_channels, frames * frame_len, device=device, dtype=torch.float32)
        bQ = torch.empty_like(bI)
        for c0 in range(0, frames * frame_len, chunk):
            n = torch.arange(r * frames * frame_len + c0, r * frames * frame_len + c0 + chunk, device=device, dtype=torch.float64)
            re = torch.zeros(n_channels, chunk, device=device, dtype=torch.float64)
            im = torch.zeros_like(re)
            for k in range(3):
                ph = (2 * np.pi / FS) * freqs[:, k:k + 1] * n[None, :] + phases[:, k:k + 1]
                re += amps[:, k:k + 1] * torch.cos(ph)
                im += amps[:, k:k + 1] * torch.sin(ph)
            noise = torch.randn(2, n_channels, chunk, generator=g, device=device, dtype=torch.float32)
            bI[:, c0:c0 + chunk] = (re.float() + 0.01 / np.sqrt(2) * noise[0]).clamp_(-0.999, 0.999)
            bQ[:, c0:c0 + chunk] = (im.float() + 0.01 / np.sqrt(2) * noise[1]).clamp_(-0.999, 0.999)
        Is.append(bI)
        Qs.append(bQ)
    return Is, Qs


def cpu_baseline(torch, Is, Qs, nco_hz, params_kw, n_channels, frame_len):
    """The CPU oracle (oracle/t41_oracle.c, a port: the reference itself is Teensy firmware)
    on a bounded sample of the same workload, all host threads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    nch = min(1024, n_channels)
    nfr = min(8, Is[0].shape[1] // frame_len)
    passes = 6
    threads = max(1, min(os.cpu_count() or 1, 64))
    I = Is[0][:nch, :nfr * frame_len].cpu().numpy()
    Q = Qs[0][:nch, :nfr * frame_len].cpu().numpy()
    ob = O.OracleBatch(O.default_params(**params_kw), np.asarray(nco_hz[:nch], dtype=np.int32), native=True)
    ob.process(I[:, :frame_len], Q[:, :frame_len], nthreads=threads)  # warm caches / twiddle tables
    t0 = time.perf_counter()
    for _ in range(passes):
        ob.process(I, Q, nthreads=threads)
    dt = time.perf_counter() - t0
    nsamp = passes * nch * I.shape[1]
    # single-thread figure on a smaller slice
    ob1 = O.OracleBatch(O.default_params(**params_kw), np.asarray(nco_hz[:64], dtype=np.int32), native=True)
    t1 = time.perf_counter()
    ob1.process(I[:64], Q[:64], nthreads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": round(nsamp / dt / 1e6, 3),
        "unit": "MSamples/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d of the %d channels x %d consecutive frames, %d passes, oracle/t41_oracle.c -O3 -march=native, "
                  "one channel range per thread; 1-thread rate on 64 channels x %d frames: %.3f MSamples/s"
                  % (nch, n_channels, nfr, passes, nfr, 64 * I.shape[1] / dt1 / 1e6),
    }


def dry_run(args, world, rank):
    """--dry-run: the multi-rank plumbing of this script without a GPU (gloo): rendezvous, channel
    sharding, the one-shot coefficient broadcast, max-over-ranks timing, one JSON line on rank 0.
    Used by tests/test_dist_cpu.py; measures nothing."""
    import torch.distributed as dist
    import t41_sdr_amd as T
    from t41_sdr_amd.dist import broadcast_coeffs, max_over_ranks, shard_channels
    if os.environ.get("T41RX_BENCH_FAIL_RANK") == str(rank):  # test hook: one rank dies before the rendezvous
        raise SystemExit(3)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        dist.init_process_group(backend="gloo", world_size=world, rank=rank, timeout=datetime.timedelta(seconds=60))
    observed = dist.get_world_size() if world > 1 else 1
    if args.gpus != observed:
        raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, observed))
    wl = WORKLOADS[args.workload]
    lo, hi = shard_channels(world * wl["batch"], rank, world)
    params = T.default_params(fft_length=wl["fft"], **wl["kw"]) if rank == 0 else T.default_params(fft_length=wl["fft"])
    blob = T.design_coeffs(params)
    want = hashlib.sha256(T.design_coeffs(T.default_params(fft_length=wl["fft"], **wl["kw"])).tobytes()).hexdigest()
    if world > 1:
        blob = broadcast_coeffs(blob, src=0)
    if hashlib.sha256(blob.tobytes()).hexdigest() != want:
        raise SystemExit("rank %d: broadcast coefficient blob differs from rank 0's design" % rank)
    wall = 1.0 + rank
    if world > 1:
        wall = max_over_ranks(wall)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": observed, "world_size_observed": observed,
                          "channels": [lo, hi], "wall_max": wall, "coeff_sha256": want[:16]}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="consecutive frames per channel per launch (default %d; 1 = one ProcessIQData() per launch)" % DEFAULT_FRAMES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="ssb")
    ap.add_argument("--dry-run", action="store_true", help="CPU-only rehearsal of the multi-rank plumbing (gloo)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(args, world, rank)

    import torch
    import t41_sdr_amd as T
    from t41_sdr_amd.dist import broadcast_coeffs, max_over_ranks, shard_channels
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", world_size=world, rank=rank,
                                device_id=torch.device("cuda", local_rank))
    observed = dist.get_world_size() if dist is not None else 1
    if args.gpus != observed:
        raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, observed))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the RX path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    wl = WORKLOADS[args.workload]
    n_channels = int(os.environ.get("T41RX_BENCH_NCH", wl["batch"]))  # env override: scaling experiments only
    fft_length = wl["fft"]
    frame_len = 4 * fft_length
    frames = args.frames_per_launch or wl.get("frames", DEFAULT_FRAMES)
    params_kw = dict(fft_length=fft_length, rfGainAllBands=1, RFgain=1, AGCMode=0, audioVolume=30)
    params_kw.update(wl["kw"])
    # the global batch is world x n_channels channels; this rank owns a contiguous shard of it
    lo, hi = shard_channels(world * n_channels, rank, world)
    rng = np.random.default_rng(1000)
    nco_all = (rng.integers(-860, 801, world * n_channels) * 50).astype(np.int32)  # [-43000, 40000] Hz, 50 Hz steps
    nco = nco_all[lo:hi]

    # One-shot coefficient broadcast: rank 0 designs the filters, everyone installs them (RCCL over xGMI).
    params = T.default_params(**params_kw)
    rx = T.RxChain(hi - lo, params if rank == 0 else T.default_params(fft_length=fft_length), device=local_rank, NCOFreq=nco)
    if dist is not None:
        rx.set_coeffs(broadcast_coeffs(rx.coeffs(), src=0, device=dev))
    got = rx.get_params()
    for k, v in params_kw.items():
        if getattr(got, k) != v:
            raise SystemExit("rank %d: parameter %s = %r after the coefficient broadcast, expected %r" % (rank, k, getattr(got, k), v))

    # ring: more than the 256 MiB Infinity Cache between two uses of the same buffer
    bytes_per_buf = 3 * (hi - lo) * frames * frame_len * 4
    ring = max(2, -(-(768 << 20) // bytes_per_buf))
    Is, Qs = synth_ring(torch, hi - lo, nco, ring, frames, frame_len, dev, seed=0x5441315F + rank, mode=params.mode)
    q15 = bool(wl.get("q15"))
    bytes_per_sample = 6.0 if q15 else BYTES_PER_SAMPLE  # 2 x int16 in + int16 out
    if q15:  # what the codec would deliver for these waveforms
        Is = [(x * 32768.0).round_().clamp_(-32768, 32767).to(torch.int16) for x in Is]
        Qs = [(x * 32768.0).round_().clamp_(-32768, 32767).to(torch.int16) for x in Qs]
    outs = [torch.empty_like(x) for x in Is]

    def step(k):
        r = k % ring
        if q15:
            rx.ProcessIQData_q15(Qs[r], Is[r], out=outs[r])  # L queue carries Q, R queue carries I
        else:
            rx.ProcessIQData(Is[r], Qs[r], out=outs[r])

    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # same stream the kernels are enqueued on (torch's current stream)
    for k in range(args.steps):
        step(args.warmup + k)
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # average launch duration, HIP events
    if dist is not None:
        wall = max_over_ranks(wall, device=dev)

    if not q15 and not os.environ.get("T41RX_BENCH_NOCHECK") and not torch.isfinite(outs[(args.warmup + args.steps - 1) % ring]).all():
        raise SystemExit("non-finite audio output")

    samples_per_step = (hi - lo) * frames * frame_len
    value = world * samples_per_step * args.steps / wall / 1e6
    achieved = bytes_per_sample * samples_per_step / (kernel_ms * 1e-3) / 1e9
    line = {
        "metric": "MSamples/s I/Q through full RX chain, batch=4096; achieved HBM GB/s vs roofline",
        "value": round(value, 1),
        "unit": "MSamples/s",
        "n_gpus": observed,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if not q15 else "f32 (q15 samples in and out)",
        "data": "synthetic",
        "config": {
            "workload": wl["name"],
            "batch": hi - lo, "frame_len": frame_len, "fft_length": fft_length,
            "frames_per_launch": frames,
            "world_size_observed": observed,
            "parallelism": "channels sharded per GPU (one process per GPU), one-shot RCCL coefficient broadcast, no data-path collective",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": load_traffic(args.workload, frames),
            "kernel": "rx512_kernel" if fft_length == 512 else "rx512_kernel<front> + fastconv_kernel + rx512_kernel<back>",
            "kernel_ms": round(kernel_ms, 5),
            "us_per_frame": round(kernel_ms * 1e3 / frames, 3),
            "algorithmic_bytes_per_launch": int(bytes_per_sample * samples_per_step),
            "kernel_source_hash": kernel_source_hash(),
        },
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.workload == "ssb":
            line["cpu_baseline"] = cpu_baseline(torch, Is, Qs, nco, params_kw, hi - lo, frame_len)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

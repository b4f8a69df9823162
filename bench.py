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
    # 128 frames per launch since the end of round 5 (rounds 1-4 and most of round 5: 32, still timed as `ssb_32fpl`): the
    # previous review asked for the sweep and "if it pays, make it the default and say so in config"; on the round's final
    # kernels it pays 0.5-0.9 % (profiles/r05_fpl_sweep2.txt: 20.83 / 20.60 us per frame at 32, 20.67 / 20.68 at 64,
    # 20.52 / 20.53 at 128 on one box) in 100-launch runs and 2-4 % in the 25 launches of `--steps 20 --warmup 5`, where 32-frame
    # launches spend their 13 ms inside the clock's ramp.  A channel's filter memories stay on chip for the whole call either way.
    "ssb": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000), frames=128,
                name="configs[1]: batched SSB (USB 200-3000 Hz) RX chain, decimate-by-8 + 512-pt fast-conv + demod + "
                     "interpolate-by-8, 4096 channels x 2048 complex f32 samples per frame per GPU, per-channel NCO, AGC off, "
                     "128 consecutive frames per channel per launch"),
    "ssb_32fpl": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000), frames=32,
                      name="configs[1] at 32 frames per launch: the headline's shape in rounds 1-4 and in round 5's A/B runs"),
    "ssb_time_major": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000), layout="time",
                           name="configs[1] with the call's frames stacked as [frame][channel][2048] (t41rx_set_buffer_layout: the buffers "
                                "of consecutive single-frame calls as they arrive) instead of [channel][frames x 2048]"),
    "nfm": dict(batch=4096, fft=512, kw=dict(mode=3, FLoCut=200, FHiCut=3000, nfmFilterBW=12000),
                name="configs[2]: NFM path as the firmware runs it (quadri-correlator + limiter + real overlap-save audio "
                     "filter), 4096 channels x 2048 samples per frame"),
    "nfm_atan": dict(batch=4096, fft=512, kw=dict(mode=3, FLoCut=200, FHiCut=3000, nfmFilterBW=12000, nfm_demod=1),
                     name="configs[2] as BASELINE words it: the atan2 discriminator + de-emphasis the reference's source keeps "
                          "commented out (Demod.cpp:148-197, 324-392; t41rx_params.nfm_demod = 1), 4096 channels x 2048 samples per frame"),
    "am": dict(batch=4096, fft=512, kw=dict(mode=2, FLoCut=-3000, FHiCut=3000),
               name="AM path (AlphaBetaMag envelope, DC block, biquad low-pass; Process.cpp:697-707), 4096 channels x 2048 "
                    "samples per frame"),
    "sam": dict(batch=4096, fft=512, kw=dict(mode=8, FLoCut=-3000, FHiCut=3000), frames=128,
                name="synchronous AM (AMDecodeSAM, Demod.cpp:40-139: a per-sample PLL, serial in time; SURVEY 8f rank 4), "
                     "4096 channels x 2048 samples per frame"),
    "sam_agc": dict(batch=4096, fft=512, kw=dict(mode=8, FLoCut=-3000, FHiCut=3000, AGCMode=1), frames=128,
                    name="synchronous AM behind the firmware's default AGCMode = 1 (Demod.cpp:40-139 behind DSP_Fn.cpp:504-631): two serial "
                         "chains per frame, each on a duty wave of its own (round 4), 4096 channels x 2048 samples per frame"),
    "ssb_agc": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=1), frames=128,
                    name="configs[1] with the firmware's default AGCMode = 1 (look-ahead AGC, DSP_Fn.cpp:504-631) instead of "
                         "the fixed gain: 4096 channels x 2048 samples per frame (SURVEY 8f rank 1)"),
    "ssb_q15": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000), q15=True,
                    name="configs[1] on the firmware's own sample format either side (q15 record-queue blocks in, "
                         "arm_float_to_q15 out; Process.cpp:102-111, 936): 6 B per input complex sample (SURVEY 8f rank 3)"),
    "ssb_agc_q15": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=1), q15=True, frames=128,
                        name="configs[1] exactly as the firmware ships: AGCMode = 1 (gwv.cpp:15) and q15 samples either side "
                             "(Process.cpp:102-111, 936): 6 B per input complex sample"),
    "ssb_notch": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000, ANR_notchOn=1), frames=32,
                      name="configs[1] with the automatic notch on (Xanr(), Noise.cpp:322-370, Process.cpp:862-866; SURVEY 8f rank 4): "
                           "fused kernel up to the demodulator, lane-per-channel LMS kernel, interpolator kernel"),
    "ssb_kim": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000, nrOptionSelect=1), frames=32,
                    name="configs[1] with Kim1_NR() on (Noise.cpp:108-313, Process.cpp:844-848)"),
    "ssb_spectral": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000, nrOptionSelect=2), frames=32,
                         name="configs[1] with SpectralNoiseReduction() on (Noise.cpp:379-655, Process.cpp:849-851)"),
    "ssb_1fpl": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000), frames=1,
                     name="configs[1] in the firmware's own calling shape: ONE ProcessIQData() per launch -- one 10.67 ms frame of every "
                          "channel per call, as ShowSpectrum() calls it (Display.cpp:339) -- streaming state through HBM every call"),
    "ssb_4fpl": dict(batch=4096, fft=512, kw=dict(mode=0, FLoCut=200, FHiCut=3000), frames=4,
                     name="configs[1] with 4 frames (42.7 ms of signal) buffered per call"),
    "fft4096": dict(batch=1024, fft=4096, kw=dict(mode=0, FLoCut=400, FHiCut=600), frames=64,
                    name="configs[3]: PSK31-like narrow USB filter, 4096-pt fast-conv (synthetic generalisation), "
                         "1024 channels x 16384 samples per frame"),
}
BYTES_PER_SAMPLE = 12.0     # 2 x f32 in + 1 x f32 out per input complex sample
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DEFAULT_FRAMES = 32         # consecutive frames per channel per launch (>= 100 consecutive frames per run)
KERNEL_SOURCES = tuple("t41_sdr_amd/csrc/" + f for f in (
    "rx_experiments.hpp", "rx_internal.hpp", "rx_kernels.hpp", "wave_fft.hpp", "rx_device.hpp", "rx_chains.hpp", "rx512_kernel.hpp",
    "rx512_launch.hpp", "rx_launch.hpp", "fastconv_kernels.hpp", "rx512_ssb.hip", "rx512_am.hip", "rx512_nfm.hip", "rx512_sam.hip",
    "rx_long.hip", "fastconv.hip", "rx_dispatch.hip", "nr_kernels.hip", "nr_kernels.hpp"))
# what the default run times behind the headline (N = 1): every mode and sample format the README claims, each replayed
# against the oracle, and the headline workload in the firmware's own calling shape (1 frame per call) and at 4 frames
# (frames per launch: 128 for the headline and for the workloads whose serial chains run as a pipeline across the frames of a
# launch -- AGC on, SAM: the pipeline's fill and drain are per launch --, 64 long frames for configs[3], 32 for the rest;
# profiles/r05_fpl_sweep2.txt, r05_fpl_sweep_others.txt)
OTHER_WORKLOADS = ("ssb_32fpl", "nfm", "nfm_atan", "am", "sam", "fft4096", "ssb_agc", "ssb_q15", "ssb_agc_q15", "sam_agc", "ssb_time_major",
                   "ssb_1fpl", "ssb_4fpl", "ssb_kim", "ssb_spectral", "ssb_notch")


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
              "roofline.traffic = null until tools/profile_round.sh + tools/collect_profiles.py are re-run" % (workload, e.get("source_hash"), kernel_source_hash()),
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
def synth_ring(torch, n_channels, nco_hz, ring, frames, frame_len, device, seed, mode=0, flo=200, fhi=3000):
    """RING launch buffers of `frames` consecutive frames of SURVEY 8d's synthetic signal, built on
    the GPU.  Returns lists of [n_channels, frames * frame_len] float32 tensors (I, Q).
    Three tones + noise per channel; the first tone is what the channel is tuned to: a tone inside the
    filter's pass band (SSB / AM), a carrier 100 Hz off tune (SAM), or -- NFM -- a sinusoidally
    frequency-modulated carrier on the tuned frequency with the other two tones kept at least 20 kHz
    away from it (a discriminator fed two carriers of similar strength divides by their beat nulls:
    its output is then ill-conditioned in ANY arithmetic, which says nothing about the kernel)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    nco = torch.as_tensor(np.asarray(nco_hz, dtype=np.float64), device=device)
    amps = 0.05 + 0.25 * torch.rand(n_channels, 3, generator=g, device=device, dtype=torch.float64)
    freqs = -90000.0 + 180000.0 * torch.rand(n_channels, 3, generator=g, device=device, dtype=torch.float64)
    phases = 2 * np.pi * torch.rand(n_channels, 3, generator=g, device=device, dtype=torch.float64)
    lo, hi = (max(flo, 0), fhi) if fhi > 0 else (-fhi, -flo)      # the audio band the filter passes
    lo, hi = lo + 0.1 * (hi - lo), hi - 0.1 * (hi - lo)            # (inside its edges: 420..580 Hz for the 400..600 Hz filter)
    lo, hi = max(lo, min(400.0, hi)), min(hi, max(2500.0, lo))     # SURVEY 8d's 400..2500 Hz where the filter is wider
    audio = lo + (hi - lo) * torch.rand(n_channels, generator=g, device=device, dtype=torch.float64)
    freqs[:, 0] = 48000.0 - nco - audio  # lands at +audio Hz in the USB pass band (I sign flip, +Fs/4, -NCO)
    fm_dev = fm_rate = fm_phase = None
    if mode == 8:  # SAM: the first tone is the carrier, 100 Hz off the tuned frequency
        freqs[:, 0] = 48000.0 - nco - 100.0
        amps[:, 0] = 0.3
    am_rate = am_depth = am_phase = None
    if mode == 2:
        # AM: a MODULATED carrier near the tuned frequency (300..2500 Hz, depth 0.3..0.8).  The envelope detector's DC
        # remover (Process.cpp:698-704) turns an unmodulated carrier into audio ~1e-3 of the carrier's level, of which
        # ANY f32 evaluation -- the reference's, the oracle's -- holds three digits (measured: the oracle itself sits
        # 5.6e-4 from the float64 model on such a signal): that input tests nothing
        freqs[:, 0] = 48000.0 - nco - (-120.0 + 240.0 * torch.rand(n_channels, generator=g, device=device, dtype=torch.float64))
        amps[:, 0] = 0.15 + 0.2 * torch.rand(n_channels, generator=g, device=device, dtype=torch.float64)
        am_rate = 300.0 + 2200.0 * torch.rand(n_channels, 1, generator=g, device=device, dtype=torch.float64)
        am_depth = 0.3 + 0.5 * torch.rand(n_channels, 1, generator=g, device=device, dtype=torch.float64)
        am_phase = 2 * np.pi * torch.rand(n_channels, 1, generator=g, device=device, dtype=torch.float64)
    if mode == 3:  # NFM: no I sign flip; an FM carrier on the tuned frequency
        freqs[:, 0] = -48000.0 + nco
        amps[:, 0] = 0.3
        fm_rate = 300.0 + 2200.0 * torch.rand(n_channels, 1, generator=g, device=device, dtype=torch.float64)
        fm_dev = 0.5 + 2.0 * torch.rand(n_channels, 1, generator=g, device=device, dtype=torch.float64)  # modulation index
        fm_phase = 2 * np.pi * torch.rand(n_channels, 1, generator=g, device=device, dtype=torch.float64)
        for k in (1, 2):  # interferers: outside +-20 kHz of the carrier (folded back into +-96 kHz)
            off = 20000.0 + 56000.0 * torch.rand(n_channels, generator=g, device=device, dtype=torch.float64)
            sign = torch.where(torch.rand(n_channels, generator=g, device=device, dtype=torch.float64) < 0.5, -1.0, 1.0)
            f = freqs[:, 0] + sign * off
            freqs[:, k] = torch.remainder(f + 96000.0, 192000.0) - 96000.0
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
                if k == 0 and fm_dev is not None:
                    ph = ph + fm_dev * torch.sin((2 * np.pi / FS) * fm_rate * n[None, :] + fm_phase)
                a_k = amps[:, k:k + 1]
                if k == 0 and am_depth is not None:
                    a_k = a_k * (1.0 + am_depth * torch.sin((2 * np.pi / FS) * am_rate * n[None, :] + am_phase))
                re += a_k * torch.cos(ph)
                im += a_k * torch.sin(ph)
            noise = torch.randn(2, n_channels, chunk, generator=g, device=device, dtype=torch.float32)
            bI[:, c0:c0 + chunk] = (re.float() + 0.01 / np.sqrt(2) * noise[0]).clamp_(-0.999, 0.999)
            bQ[:, c0:c0 + chunk] = (im.float() + 0.01 / np.sqrt(2) * noise[1]).clamp_(-0.999, 0.999)
        Is.append(bI)
        Qs.append(bQ)
    return Is, Qs


def host_cpu():
    """(model string, logical CPUs the OS reports, CPUs this process may actually use: affinity mask and cgroup quota)"""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
    try:  # cgroup v2 CPU quota ("max" or "<quota> <period>")
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            usable = max(1, min(usable, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return model, logical, usable


def cpu_baseline(torch, Is, Qs, nco_hz, params_kw, n_channels, frame_len):
    """The CPU oracle (oracle/t41_oracle.c, a port: the reference itself is Teensy firmware) on a
    bounded sample of the same workload: >= 1.5 s of wall time with every usable host thread and
    >= 1.5 s on one thread (BASELINE.md section 3 asks for both and for the CPU model)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    model, logical, usable = host_cpu()
    threads = max(1, min(usable, 64))
    nch = min(1024, n_channels)
    nfr = min(8, Is[0].shape[1] // frame_len)
    I = Is[0][:nch, :nfr * frame_len].cpu().numpy()
    Q = Qs[0][:nch, :nfr * frame_len].cpu().numpy()

    def rate(nchan, nthreads, min_s):
        ob = O.OracleBatch(O.default_params(**params_kw), np.asarray(nco_hz[:nchan], dtype=np.int32), native=True)
        ob.process(I[:nchan, :frame_len], Q[:nchan, :frame_len], nthreads=nthreads)  # warm caches / twiddle tables
        passes, t0 = 0, time.perf_counter()
        while passes < 3 or time.perf_counter() - t0 < min_s:
            ob.process(I[:nchan], Q[:nchan], nthreads=nthreads)
            passes += 1
        dt = time.perf_counter() - t0
        ob.close()
        return passes * nchan * I.shape[1] / dt / 1e6, passes, dt

    v_all, p_all, t_all = rate(nch, threads, 1.5)
    v_one, p_one, t_one = rate(min(64, nch), 1, 1.5)
    return {
        "value": round(v_all, 1),
        "unit": "MSamples/s",
        "cores": threads,
        "kind": "port",
        "single_thread_value": round(v_one, 2),
        "host_cpu": model,
        "host_logical_cpus": logical,
        "host_usable_cpus": usable,
        "sample": "oracle/t41_oracle.c -O3 -march=native on %d of the %d channels x %d consecutive frames, one channel range per "
                  "thread: %d passes in %.2f s on %d threads; 1 thread: %d channels, %d passes in %.2f s"
                  % (nch, n_channels, nfr, p_all, t_all, threads, min(64, nch), p_one, t_one),
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
    # the device every rank would select (torch.cuda.set_device(LOCAL_RANK) in the real run): one GPU each
    local_rank = int(os.environ.get("T41RX_BENCH_FORCE_LOCAL_RANK", os.environ.get("LOCAL_RANK", "0")))  # (test hook: a broken launcher)
    local_ranks = [local_rank]
    if world > 1:
        wall = max_over_ranks(wall)
        local_ranks = [None] * world
        dist.all_gather_object(local_ranks, local_rank)
        dist.barrier()
    if len(set(local_ranks)) != len(local_ranks):
        raise SystemExit("two ranks would select the same device: LOCAL_RANK values %r" % (local_ranks,))
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": observed, "world_size_observed": observed,
                          "channels": [lo, hi], "wall_max": wall, "coeff_sha256": want[:16], "local_ranks": local_ranks}), flush=True)
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
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the other modes' timings + parity checks behind the headline (N = 1, ssb)")
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

    # which physical device every rank runs on: N ranks must sit on N distinct GPUs (rank 0 checks, the line reports)
    props = torch.cuda.get_device_properties(local_rank)
    my_dev = "%s %s" % (getattr(props, "uuid", "no-uuid-%d" % local_rank), props.name)
    devices = [my_dev]
    if dist is not None:
        devices = [None] * world
        dist.all_gather_object(devices, my_dev)
    print("rank %d of %d: LOCAL_RANK %d -> %s" % (rank, world, local_rank, my_dev), file=sys.stderr, flush=True)
    if len(set(devices)) != len(devices):
        raise SystemExit("two ranks run on the same device: %r" % (devices,))

    head = Workload(torch, T, args.workload, world, rank, local_rank, dev, dist, args.frames_per_launch)
    wall, kernel_ms = head.time(args.steps, args.warmup)
    if not head.q15 and not os.environ.get("T41RX_BENCH_NOCHECK") and not torch.isfinite(head.outs[(args.warmup + args.steps - 1) % head.ring]).all():
        raise SystemExit("non-finite audio output")

    wl = head.wl
    value = world * head.samples_per_step * args.steps / wall / 1e6
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
        "dtype": "f32" if not head.q15 else "f32 (q15 samples in and out)",
        "data": "synthetic",
        "config": {
            "workload": wl["name"],
            "batch": head.n, "frame_len": head.frame_len, "fft_length": head.fft_length,
            "frames_per_launch": head.frames,
            "buffer_layout": head.layout + "-major",
            "world_size_observed": observed,
            "devices": devices,
            "parallelism": "channels sharded per GPU (one process per GPU), one-shot RCCL coefficient broadcast, no data-path collective",
        },
        "roofline": head.roofline(kernel_ms),
    }
    if rank == 0:
        # the timed stream against the CPU oracle, in the timed launch shape (replayed: see parity_check)
        # (the noise-reduction workloads start their adaptive stages on the path's start-up transient: their parity is
        # the business of tests/test_noise_reduction.py, which feeds both sides identical audio)
        if not os.environ.get("T41RX_BENCH_NOCHECK") and not head.params_kw.get("ANR_notchOn", 0):
            line["parity_check"] = parity_check(torch, head, min(args.warmup + args.steps, 160), args.warmup + args.steps)
        if world == 1 and not args.no_cpu_baseline and args.workload == "ssb":
            line["cpu_baseline"] = cpu_baseline(torch, head.Is, head.Qs, head.nco, head.params_kw, head.n, head.frame_len)
    head.free()
    if rank == 0 and world == 1 and args.workload == "ssb" and not args.no_other_workloads:
        # BASELINE configs[2] / [3] and the firmware's default AGC mode, timed in this same run
        others = {}
        for name in OTHER_WORKLOADS:
            w = Workload(torch, T, name, 1, 0, local_rank, dev, None, 0)
            # >= 1280 frames of every channel in the timed region: 40 launches of 32 frames, 320 of 4, 1280 of 1 (until the end
            # of round 5: 12 / 96 / 384 launches behind 4 of warm-up -- 8 ms of timed region for the fast workloads, most of it
            # inside the clock's ramp: those entries read 2-3 % below the same workload's 30-launch A/B runs)
            wsteps, wwarm = max(40, 1280 // w.frames), 8
            _, kms = w.time(wsteps, wwarm)
            r = w.roofline(kms)
            entry = {"workload": w.wl["name"], "batch": w.n, "frames_per_launch": w.frames, "steps": wsteps, "warmup": wwarm,
                     "kernel_ms": r["kernel_ms"], "us_per_frame": r["us_per_frame"], "frac": r["frac"], "achieved_GBs": r["achieved"],
                     "traffic": r["traffic"], "dtype": "f32" if not w.q15 else "f32 (q15 samples in and out)"}
            if w.params_kw.get("ANR_notchOn", 0):
                # the automatic notch from power-on adapts on the front end's start-up transient -- samples of 1e-7 whose power it
                # divides by: two f32 front ends that agree to 1e-7 leave it 1e-2 apart for tens of frames, the oracle against its
                # own perturbed self included (tests/test_noise_reduction.py::test_oracle_stage_conditioning); its parity is held
                # in isolation (3e-6) and from a running stream (1e-5) by the tests, not by a replay from power-on
                entry["parity_check"] = None
                entry["parity_note"] = "timed only: the notch from power-on is ill-conditioned in any arithmetic (DESIGN.md 4.10); tests hold it in isolation"
            elif not os.environ.get("T41RX_BENCH_NOCHECK"):
                # every launch of this workload's run (warm-up + timed), replayed and compared with the oracle; the
                # replay must reproduce the timed pass bit for bit (round 3 checked 2 of the 16 and could not assert that)
                entry["parity_check"] = parity_check(torch, w, wwarm + wsteps, wwarm + wsteps, sample=8 if w.fft_length == 4096 else 16)
            others[name] = entry
            w.free()
        line["other_workloads"] = others
    if rank == 0:
        print(json.dumps(line), flush=True)
        bad = [k for k, v in [("headline", line.get("parity_check"))] + [(n, e.get("parity_check")) for n, e in line.get("other_workloads", {}).items()]
               if v is not None and not v["ok"]]
        if bad:
            raise SystemExit("parity check failed for: %s" % ", ".join(bad))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


class Workload:
    """One bench workload on this rank: context, synthetic ring of launch buffers, the step."""

    def __init__(self, torch, T, name, world, rank, local_rank, dev, dist, frames_override):
        from t41_sdr_amd.dist import broadcast_coeffs, shard_channels
        self.torch, self.dist, self.dev, self.name = torch, dist, dev, name
        self.wl = wl = WORKLOADS[name]
        n_channels = int(os.environ.get("T41RX_BENCH_NCH", wl["batch"]))  # env override: scaling experiments only
        self.fft_length = wl["fft"]
        self.frame_len = 4 * self.fft_length
        self.frames = frames_override or wl.get("frames", DEFAULT_FRAMES)
        self.params_kw = dict(fft_length=self.fft_length, rfGainAllBands=1, RFgain=1, AGCMode=0, audioVolume=30)
        self.params_kw.update(wl["kw"])
        # the global batch is world x n_channels channels; this rank owns a contiguous shard of it
        lo, hi = shard_channels(world * n_channels, rank, world)
        self.n = hi - lo
        rng = np.random.default_rng(1000)
        nco_all = (rng.integers(-860, 801, world * n_channels) * 50).astype(np.int32)  # [-43000, 40000] Hz, 50 Hz steps
        self.nco = nco_all[lo:hi]
        # One-shot coefficient broadcast: rank 0 designs the filters, everyone installs them (RCCL over xGMI).
        params = T.default_params(**self.params_kw)
        self.rx = T.RxChain(self.n, params if rank == 0 else T.default_params(fft_length=self.fft_length), device=local_rank, NCOFreq=self.nco)
        if dist is not None:
            self.rx.set_coeffs(broadcast_coeffs(self.rx.coeffs(), src=0, device=dev))
        got = self.rx.get_params()
        for k, v in self.params_kw.items():
            if getattr(got, k) != v:
                raise SystemExit("rank %d: parameter %s = %r after the coefficient broadcast, expected %r" % (rank, k, getattr(got, k), v))
        # ring: more than the 256 MiB Infinity Cache between two uses of the same buffer
        bytes_per_buf = 3 * self.n * self.frames * self.frame_len * 4
        self.ring = max(2, -(-(768 << 20) // bytes_per_buf))
        self.Is, self.Qs = synth_ring(torch, self.n, self.nco, self.ring, self.frames, self.frame_len, dev, seed=0x5441315F + rank,
                                      mode=params.mode, flo=params.FLoCut, fhi=params.FHiCut)
        self.layout = wl.get("layout", "channel")
        if self.layout == "time":  # [frame][channel][frame_len]
            self.rx.set_buffer_layout("time")
            self.Is = [x.view(self.n, self.frames, self.frame_len).transpose(0, 1).contiguous() for x in self.Is]
            self.Qs = [x.view(self.n, self.frames, self.frame_len).transpose(0, 1).contiguous() for x in self.Qs]
        self.q15 = bool(wl.get("q15"))
        self.bytes_per_sample = 6.0 if self.q15 else BYTES_PER_SAMPLE  # 2 x int16 in + int16 out
        if self.q15:  # what the codec would deliver for these waveforms
            self.Is = [(x * 32768.0).round_().clamp_(-32768, 32767).to(torch.int16) for x in self.Is]
            self.Qs = [(x * 32768.0).round_().clamp_(-32768, 32767).to(torch.int16) for x in self.Qs]
        self.outs = [torch.empty_like(x) for x in self.Is]
        self.samples_per_step = self.n * self.frames * self.frame_len

    def channels_of(self, x, idx_t):
        """the sampled channels of a launch buffer as [len(idx), frames * frame_len], whatever the layout"""
        if self.layout == "time":
            return x.index_select(1, idx_t).transpose(0, 1).reshape(idx_t.numel(), -1)
        return x.index_select(0, idx_t)

    def step(self, k):
        r = k % self.ring
        if self.q15:
            self.rx.ProcessIQData_q15(self.Qs[r], self.Is[r], out=self.outs[r])  # L queue carries Q, R queue carries I
        else:
            self.rx.ProcessIQData(self.Is[r], self.Qs[r], out=self.outs[r])

    def time(self, steps, warmup):
        """W untimed launches, then exactly K timed ones between barrier + synchronize on both sides.
        Returns (wall seconds, max over ranks; average launch duration in ms from HIP events on the launch stream)."""
        torch, dist = self.torch, self.dist
        for k in range(warmup):
            self.step(k)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()  # same stream the kernels are enqueued on (torch's current stream)
        for k in range(steps):
            self.step(warmup + k)
        ev1.record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        kernel_ms = ev0.elapsed_time(ev1) / steps
        if dist is not None:
            from t41_sdr_amd.dist import max_over_ranks
            wall = max_over_ranks(wall, device=self.dev)
        return wall, kernel_ms

    def roofline(self, kernel_ms):
        achieved = self.bytes_per_sample * self.samples_per_step / (kernel_ms * 1e-3) / 1e9
        return {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": load_traffic(self.name, self.frames),
            "kernel": ("rx512_kernel" + (" + nr kernels + rx512_kernel<back>" if (self.params_kw.get("nrOptionSelect", 0) or self.params_kw.get("ANR_notchOn", 0)) else ""))
                      if self.fft_length == 512 else
                      ("fastconv_fused_kernel" if self.fft_length == 4096 and self.params_kw.get("mode", 0) in (0, 1) and not self.params_kw.get("AGCMode", 0) and not self.q15
                       else "rx512_kernel<front> + fastconv_kernel (+ rx512_kernel<back>)"),
            "kernel_ms": round(kernel_ms, 5),
            "us_per_frame": round(kernel_ms * 1e3 / self.frames, 3),
            "algorithmic_bytes_per_launch": int(self.bytes_per_sample * self.samples_per_step),
            "kernel_source_hash": kernel_source_hash(),
        }

    def free(self):
        self.rx.close()
        self.Is = self.Qs = self.outs = None
        self.torch.cuda.empty_cache()


def parity_check(torch, w, launches, launches_timed, sample=16):
    """The launches of the timed run once more, from the power-on state, in the timed launch shape
    (same buffers, same ring order, same frames per launch), with the audio of `sample` channels
    copied out after every launch and compared with the CPU oracle run over the same stream:
    per-frame max|gpu - ref| / max|ref|, the bar 1e-5.  The copies are why this is a
    replay and not the timed pass itself; the kernel is deterministic, and that the replay
    reproduces the timed pass is checked bit for bit on every buffer the timed pass left behind."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    L = w.frame_len
    idx = np.unique(np.linspace(0, w.n - 1, sample).round().astype(np.int64))
    idx_t = torch.as_tensor(idx, device=w.dev)
    left = [o.clone() for o in w.outs] if launches == launches_timed else None
    w.rx.reset()
    got = []
    for k in range(launches):
        w.step(k)
        got.append(w.channels_of(w.outs[k % w.ring], idx_t))
    torch.cuda.synchronize()
    replay_identical = None
    if left is not None:
        replay_identical = all(torch.equal(a, b) for a, b in zip(left, w.outs))
    ob = O.OracleBatch(O.default_params(**w.params_kw), np.asarray(w.nco[idx], dtype=np.int32), native=False)
    threads = max(1, min(host_cpu()[2], len(idx)))
    hI = [w.channels_of(x, idx_t).cpu().numpy() for x in w.Is]
    hQ = [w.channels_of(x, idx_t).cpu().numpy() for x in w.Qs]
    worst, worst_at, sq_err, sq_ref = 0.0, None, 0.0, 0.0
    rels = []  # per launch: [channel][frame] block-relative errors
    for k in range(launches):
        r = k % w.ring
        if w.q15:
            ref = ob.process_q15(hQ[r], hI[r]).astype(np.float64)
        else:
            ref = ob.process(hI[r], hQ[r], nthreads=threads).astype(np.float64)
        g = got[k].cpu().numpy().astype(np.float64)
        if w.q15:  # q15 samples: +-1 LSB at truncation boundaries is the documented bar (tests/test_q15_boundary.py)
            e = np.abs(g - ref).max() / 32768.0
            if e > worst:
                worst, worst_at = float(e), [int(k), -1]
            continue
        d = np.abs(g - ref).reshape(len(idx), -1, L).max(axis=2)
        m = np.abs(ref).reshape(len(idx), -1, L).max(axis=2)
        rel = np.where(m >= 1e-6, d / np.maximum(m, 1e-30), d)
        if w.params_kw.get("mode") == 8 and k * w.frames < 12:
            rel[:, :12 - k * w.frames] = 0.0  # SAM: the PLL's pull-in (the stream's first 12 frames) is compared once locked (tests/test_sam.py)
        rels.append(rel)
        if rel.max() > worst:
            c, f = np.unravel_index(rel.argmax(), rel.shape)
            worst, worst_at = float(rel.max()), [int(k), int(f), int(idx[c])]
        sq_err += float(((g - ref) ** 2).sum())
        sq_ref += float((ref ** 2).sum())
    ob.close()
    # the stated bars: 1e-5 (north_star); AM 5e-5 (the reference's f32 DC blocker, tests/test_gpu_parity.py::test_parity_am); q15 +-1 LSB
    # The noise-reduction stages, WHOLE PATH from power-on (the two sides' stage inputs differ by the front ends' 1e-7): the bars
    # are the stages' own conditioning, measured on the oracle against its own 1e-7-perturbed self
    # (tests/test_noise_reduction.py::test_oracle_stage_conditioning) -- Kim's gain 1 - M / E cancels where the
    # noise is stationary, as on this workload's tones: 1e-4; the spectral function: 5e-5, and its integer smoothing width flips
    # in isolated frames (allowed: 1 frame in 1000).  On IDENTICAL stage input the tests hold them to 1e-5 / 5e-5.
    nr_opt = w.params_kw.get("nrOptionSelect", 0)
    tol = 1.0 / 32768.0 + 1e-9 if w.q15 else (1e-4 if nr_opt == 1 else 5e-5 if (w.params_kw.get("mode") == 2 or nr_opt == 2) else 1e-5)
    over = int(sum(int((r > tol).sum()) for r in rels))
    frames = int(sum(r.size for r in rels))
    # nfm_demod = 1: ApproxAtan2 as written (Demod.cpp:176-193: 2 pi where pi / 2 is meant) JUMPS by 3 pi / 2 where |x| = |y|,
    # so two f32 evaluations whose (x, y) differ in the last bit take different branches about once per million samples
    # (this check replays ~2 million): such a frame differs grossly in ANY pair of implementations.  Allowed: 1 frame in
    # 1000, counted and reported; every other frame meets the bar.
    allowed = frames // 1000 if (w.params_kw.get("nfm_demod") or nr_opt == 2) else 0
    ok = (bool(worst <= tol) or over <= allowed) and replay_identical is not False
    return {"ok": ok, "max_block_rel_err": worst, "tolerance": tol, "at_launch_frame_channel": worst_at,
            "frames_checked": frames, "frames_over_tolerance": over, "frames_over_tolerance_allowed": allowed,
            "median_block_rel_err": float(np.median(np.concatenate([r.ravel() for r in rels]))) if rels else None,
            "rms_rel_err": (sq_err / sq_ref) ** 0.5 if sq_ref > 0 else None,
            "channels": [int(c) for c in idx], "launches_checked": launches, "launches_timed": launches_timed,
            "frames_per_launch": w.frames, "replay_bit_identical_to_timed_run": replay_identical,
            "how": "replay of the timed launches from the power-on state in the timed shape, sampled channels vs oracle/t41_oracle.c"}


if __name__ == "__main__":
    main()

"""Parity of the HIP path (through the C ABI) with the CPU oracle.  Run with -m gpu on MI355X.

Tolerance (BASELINE north_star / SURVEY 8d): per frame max|gpu - ref| / max|ref| <= 1e-5,
frames with max|ref| < 1e-6 compared absolutely.  The GPU evaluates the same f32 chain with a
different summation tree in the FFT (radix-8 vs the oracle's radix-2), FMA contraction, a
parallel scan for the DC high-pass and an f32 fixed-point-phase oscillator instead of the
reference's f64 recurrence, so ~1e-6 is expected; 1e-5 is the bar.
"""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O
import siggen

pytestmark = pytest.mark.gpu

L = 2048
TOL = 1e-5
AM_TOL = 5e-5  # see the note above test_parity_am
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def T(built):
    import torch
    import t41_sdr_amd
    assert torch.cuda.is_available()
    t41_sdr_amd.load()
    return t41_sdr_amd


def gpu_run(T, kw, nco, I, Q, split=None):
    import torch
    kw = dict(kw)
    rx = T.RxChain(I.shape[0], T.default_params(**kw), NCOFreq=nco)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    if split is None:
        out = rx.ProcessIQData(dI, dQ)
    else:
        out = torch.cat([rx.ProcessIQData(dI[:, a:b].contiguous(), dQ[:, a:b].contiguous())
                         for a, b in zip(split[:-1], split[1:])], dim=1)
    torch.cuda.synchronize()
    return out.cpu().numpy(), rx


def oracle_run(kw, nco, I, Q):
    return O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32)).process(I, Q, nthreads=8)


@pytest.mark.parametrize("kw", [
    dict(mode=0, FLoCut=200, FHiCut=3000),
    dict(mode=1, FLoCut=-3000, FHiCut=-200),
    dict(mode=0, FLoCut=400, FHiCut=600),
    dict(mode=0, FLoCut=100, FHiCut=11000),
    dict(mode=0, FLoCut=300, FHiCut=2700, rfGainAllBands=6, RFgain=3, audioVolume=55,
         IQAmpCorrectionFactor=1.02, IQPhaseCorrectionFactor=-0.013),
    dict(mode=1, FLoCut=-2700, FHiCut=-300, IQAmpCorrectionFactor=0.97, IQPhaseCorrectionFactor=0.021,
         xmtMode=1, CWFreqShift=750),
], ids=["usb", "lsb", "usb-narrow", "usb-wide", "usb-gains-iqcorr", "lsb-cw-sidetone"])
def test_parity_vs_oracle(T, kw):
    nch, nfr = 48, 6
    nco = siggen.nco_grid(nch, seed=3)
    nco[:6] = [0, 50, -50, 40000, -43000, 96000 // 4]  # edge tunings incl. 0 Hz and Fs/8
    band = (420.0, 580.0) if kw.get("FHiCut") == 600 else (400.0, 2500.0)
    side = 0  # CW side-tone offset moves the tuning (Freq_Shift.cpp:108-120): keep the test tone in band
    if kw.get("xmtMode") == 1:
        side = kw["CWFreqShift"] if kw["mode"] == 1 else -kw["CWFreqShift"]
    I, Q = siggen.make_iq(nch, nfr * L, nco + side, mode=kw["mode"], seed=21, audio_hz=band)
    got, _ = gpu_run(T, kw, nco, I, Q)
    ref = oracle_run(kw, nco, I, Q)
    err = siggen.block_rel_err(got, ref, L)
    assert np.isfinite(got).all()
    assert err.max() <= TOL, "worst block-relative error %.3e at %s" % (err.max(), np.unravel_index(err.argmax(), err.shape))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))))
def test_parity_vs_golden_fixture(T, path):
    g = np.load(path, allow_pickle=False)
    kw = {k: (float(v) if "." in v else int(v)) for k, v in g["params"]}
    import torch
    Lf = 4 * kw.get("fft_length", 512)
    nch, nfr = g["I"].shape[0], g["I"].shape[1] // Lf
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=g["nco"])
    if "spect" in g:
        sp = torch.zeros(nch, nfr, 1024, device="cuda")
        mx = torch.zeros(nch, nfr, 3, device="cuda")
        rx.set_audio_spectrum(sp, mx)
    got = rx.ProcessIQData(torch.from_numpy(g["I"]).cuda(), torch.from_numpy(g["Q"]).cuda()).cpu().numpy()
    err = siggen.block_rel_err(got, g["audio"], Lf)
    if "fft_length" in kw:  # frame 0 is mostly filter start-up: compared absolutely (see test_parity_fft4096)
        assert np.abs(got[:, :Lf] - g["audio"][:, :Lf]).max() <= 1e-5 * np.abs(g["audio"]).max()
        err = err[:, 1:]
    if kw["mode"] == 8:  # the synchronous detector's pull-in is not comparable sample by sample (tests/test_sam.py)
        assert np.isfinite(got).all()
        err = err[:, 12:]
    if kw.get("nrOptionSelect", 0) or kw.get("ANR_notchOn", 0):
        # the noise-reduction stages' own conditioning (tests/test_noise_reduction.py: measured on the oracle)
        assert np.isfinite(got).all()
        if kw.get("ANR_notchOn", 0):
            # The notch from power-on adapts on the front end's start-up transient, where it divides by the power of
            # rounding-level samples: two f32 front ends that agree to 1e-7 leave it apart by what the ORACLE ITSELF
            # moves by when its input is perturbed by one part in 1e7 (test_oracle_stage_conditioning).  That measured
            # envelope is the tolerance here (round 3 stopped at isfinite): the HIP path must stay inside it.
            rng = np.random.default_rng(2)
            pI = (g["I"].astype(np.float64) * (1 + 1e-7 * rng.standard_normal(g["I"].shape))).astype(np.float32)
            pQ = (g["Q"].astype(np.float64) * (1 + 1e-7 * rng.standard_normal(g["Q"].shape))).astype(np.float32)
            env = siggen.block_rel_err(oracle_run(kw, g["nco"], pI, pQ), g["audio"], Lf)
            assert env.max() > 1e-4  # (ill-conditioned indeed; otherwise this case belongs with the others)
            assert err.max() <= 3.0 * env.max(), (err.max(), env.max())
            return
        # round 4: the spectral function's smoothing width is decided exactly as the scalar code decides it
        # (nr_kernels.hip) -- its isolated 1e-3 .. 1e-2 frames are gone; Kim's measured error stays below 1e-5 here
        assert err.max() <= (1e-4 if kw["nrOptionSelect"] == 2 else 2e-5), err
        return
    assert err.max() <= (AM_TOL if kw["mode"] == 2 else TOL), err
    if "spect" in g:
        sp, mx = sp.cpu().numpy(), mx.cpu().numpy()
        assert (np.abs(sp - g["spect"]).max(axis=2) <= 3e-5 * g["spect"].max(axis=2)).all()
        assert np.allclose(mx[:, :, [0, 2]], g["spect_max"][:, :, [0, 2]], rtol=3e-5, atol=0)
    if "Q_out_L" in g:
        rx.reset()
        q = rx.ProcessIQData_q15(torch.from_numpy(g["Q_in_L"]).cuda(), torch.from_numpy(g["Q_in_R"]).cuda()).cpu().numpy()
        d = np.abs(q.astype(np.int32) - g["Q_out_L"].astype(np.int32))
        assert d.max() <= 1 + np.ceil(1e-5 * np.abs(g["Q_out_L"].astype(np.int32)).max()) and (d > 0).mean() < 0.02


def test_parity_nfm(T):
    """BASELINE config 3's demodulator as the firmware actually runs it: quadri-
    correlator + limiter + real overlap-save audio filter, dec filters redesigned for 12 kHz"""
    nch, nfr = 32, 6
    nco = siggen.nco_grid(nch, seed=13)
    nco[:3] = [0, 40000, -43000]
    kw = dict(mode=3, FLoCut=200, FHiCut=3000, nfmFilterBW=12000)
    I, Q = siggen.make_fm(nch, nfr * L, nco)
    got, _ = gpu_run(T, kw, nco, I, Q)
    ref = oracle_run(kw, nco, I, Q)
    err = siggen.block_rel_err(got, ref, L)
    assert err.max() <= TOL, (err.max(), np.unravel_index(err.argmax(), err.shape))
    # frame-by-frame == one call (the discriminator's odd "last sample" state included)
    split, _ = gpu_run(T, kw, nco, I, Q, split=[k * L for k in range(nfr + 1)])
    assert np.array_equal(got, split)


# ---- AGC on (DSP_Fn.cpp:504-631).  Fading signals that drive the gain law through
# attack, fast decay, hang, decay and hang-decay; 10 channels = two full workgroups + a ragged one.
AGC_CASES = [
    (2, 0, 140, [(0.15, 2.5), (0.75, 0.05), (0.1, 2.5)]),   # slow: all five states (hang expires)
    (1, 0, 40, [(0.5, 2.5), (0.3, 0.05), (0.2, 1.5)]),      # long: attack, fast decay, hang
    (3, 1, 40, [(0.4, 2.5), (0.3, 0.05), (0.3, 1.6)]),      # med, LSB: attack, fast decay, decay
    (4, 2, 30, [(0.4, 2.0), (0.3, 0.05), (0.3, 1.6)]),      # fast, AM
    (1, 3, 12, [(0.4, 1.0), (0.3, 0.3), (0.3, 1.0)]),       # NFM: AGC acts on the filtered audio
    # bursts: fast decay ending in hang-decay (1>4), decay (1>3), hang (1>2); with slow-usb these walk all
    # eleven transitions of the gain law (tests/test_oracle_agc.py: test_agc_scenarios_walk_every_transition)
    (2, 0, 160, [(0.1, 1.5), (0.05, 0.3), (0.003, 2.5), (0.1, 0.3), (0.003, 2.5), (0.55, 0.3), (0.003, 2.5), (0.191, 0.3)]),
    (2, 0, 60, [(0.25, 0.6), (0.15, 0.2), (0.006, 2.5), (0.594, 0.06)]),
    (1, 0, 100, [(0.2, 1.5), (0.4, 0.2), (0.003, 2.5), (0.397, 0.2)]),
]


@pytest.mark.parametrize("agcmode,mode,nfr,segs", AGC_CASES, ids=["slow-usb", "long-usb", "med-lsb", "fast-am", "long-nfm", "bursts-1to4", "bursts-1to3", "bursts-1to2"])
def test_parity_agc(T, agcmode, mode, nfr, segs):
    nch = 10
    nco = siggen.nco_grid(nch, seed=5 + agcmode)
    if mode == 3:
        I, Q = siggen.make_fm(nch, nfr * L, nco, seed=3)
    else:
        I, Q = siggen.make_iq(nch, nfr * L, nco, mode=mode, seed=5, audio_hz=(400.0, 2500.0))
    I, Q = siggen.fade(I, Q, segs)
    flo, fhi = {0: (200, 3000), 1: (-3000, -200), 2: (-3000, 3000), 3: (200, 3000)}[mode]
    kw = dict(mode=mode, AGCMode=agcmode, FLoCut=flo, FHiCut=fhi)
    got, rx = gpu_run(T, kw, nco, I, Q)
    ob = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32))
    ref = ob.process(I, Q, nthreads=8)
    err = siggen.block_rel_err(got, ref, L)
    assert np.isfinite(got).all()
    assert err.max() <= (AM_TOL if mode == 2 else TOL), (err.max(), np.unravel_index(err.argmax(), err.shape))
    # the gain-law state itself: `volts` after the last sample (state record: 768 + 200 history floats, word 2)
    st = rx.state_records()
    volts_ref = np.array([ob.tap(c, O.TAP_AGC_VOLTS, 256)[-1] for c in range(nch)])
    # what this case is here for: the state changes its signals drive the gain law through
    edges = sum(ob.tap(c, O.TAP_AGC_EDGES, 25).reshape(5, 5) for c in range(nch))
    seen = {(a, b) for a in range(5) for b in range(5) if a != b and edges[a, b] > 0}
    must = {140: {(0, 1), (0, 2), (0, 3), (1, 0), (2, 0), (2, 4), (3, 0), (4, 0)}, 160: {(1, 4), (1, 2), (1, 3)},
            60: {(1, 3)}, 100: {(1, 2)}}.get(nfr, set())
    assert must <= seen, (must - seen)
    assert np.abs(st[:, 768 + 200 + 2] - volts_ref).max() <= 1e-5 * volts_ref.max()
    # frame by frame == one call, bit for bit
    split, _ = gpu_run(T, kw, nco, I, Q, split=[0, L, 5 * L, nfr * L])
    assert np.array_equal(got, split)


@pytest.mark.parametrize("nch,mode,agcmode", [(1, 0, 1), (5, 0, 1), (17, 1, 2), (21, 2, 3), (37, 3, 4), (64, 0, 2)],
                         ids=["1ch-usb", "5ch-usb", "17ch-lsb", "21ch-am", "37ch-nfm", "64ch-usb"])
def test_agc_pipelined_equals_barrier_form(T, nch, mode, agcmode):
    """Calls of four frames or more run the pipelined AGC kernel (rx_kernels.hip: agc_prep_pipe -- the chain of one
    frame on a duty wave -- the one furthest ahead -- while the others work on the neighbouring frames), shorter ones the barrier form:
    the same samples and the same checkpoint, bit for bit, on ragged batches (last workgroup 5 of 16 channels)"""
    nfr = 14
    nco = siggen.nco_grid(nch, seed=40 + nch)
    if mode == 3:
        I, Q = siggen.make_fm(nch, nfr * L, nco, seed=4)
    else:
        I, Q = siggen.make_iq(nch, nfr * L, nco, mode=mode, seed=6)
    I, Q = siggen.fade(I, Q, [(0.3, 2.0), (0.3, 0.05), (0.4, 1.5)])
    flo, fhi = {0: (200, 3000), 1: (-3000, -200), 2: (-3000, 3000), 3: (200, 3000)}[mode]
    kw = dict(mode=mode, AGCMode=agcmode, FLoCut=flo, FHiCut=fhi)
    whole, rx_w = gpu_run(T, kw, nco, I, Q)                                             # one pipelined call
    short, rx_s = gpu_run(T, kw, nco, I, Q, split=[0, 3 * L, 6 * L, 9 * L, 12 * L, 14 * L])  # barrier form only
    mixed, rx_m = gpu_run(T, kw, nco, I, Q, split=[0, 4 * L, 5 * L, 14 * L])              # pipelined, barrier, pipelined
    assert np.isfinite(whole).all()
    assert np.array_equal(whole, short) and np.array_equal(whole, mixed)
    assert np.array_equal(rx_w.get_state(), rx_s.get_state()) and np.array_equal(rx_w.get_state(), rx_m.get_state())
    ref = oracle_run(kw, nco, I, Q)
    err = siggen.block_rel_err(whole, ref, L)
    assert err.max() <= (AM_TOL if mode == 2 else TOL), err.max()


@pytest.mark.parametrize("agcmode", [1, 3, 4])
def test_agc_min_volts_raised_mid_stream(T, agcmode):
    """a live CalcFilters() that lowers AGC_thresh raises min_volts (DSP_Fn.cpp:408-415) while lanes sit in
    their decay states with volts below the new floor.  The clamp of DSP_Fn.cpp:629 then RAISES volts at the next step
    -- the one case in which volts rises without an attack -- and the pipelined chain's per-block short-cut (one
    comparison with the fast decay's threshold per four steps, rx_kernels.hip: agc_block_phased) must still leave the
    fast decay where the step-by-step forms do: pipelined == barrier form bit for bit, both within 1e-5 of the oracle
    that makes the same switch."""
    import torch
    nch, nfr, cut = 37, 20, 6
    nco = siggen.nco_grid(nch, seed=91)
    I, Q = siggen.make_iq(nch, nfr * L, nco, mode=0, seed=92)
    # loud, then a fade (the gain law decays towards the quiet level: states 1..4), then loud again; the fade starts at a
    # different instant in every channel -- between 2.6 and 0.1 frames before the parameters change -- so that at the
    # change the channels sit in every phase of the decay, the fast one (state 1, a few ms long) included.
    # (A 32 dB step, like the other AGC cases: behind a 50 dB step the quiet frames next to it hold the loud samples'
    # FFT rounding noise at 4e-5 of their own level IN BOTH implementations -- tools/agc_thresh_probe.py, any AGC_thresh.)
    steady = np.ones((nch, nfr), bool)  # frames at least one frame away from a step of the channel's envelope
    steady[:, 0] = False  # power-on: with max_gain 10^4.5 the gain law multiplies the filters' start-up ROUNDING NOISE (the
    # signal is still inside the 97-sample look-ahead) by up to 3e4 -- 1.4e-7 absolute in both implementations, uncorrelated
    for c in range(nch):
        start = (cut - 2.6 + 2.5 * c / (nch - 1)) / nfr
        e = siggen.envelope_steps(nfr * L, [(start, 2.0), (0.7 - start, 0.05), (0.3, 1.5)])  # quiet until frame 14
        I[c] = np.clip(I[c] * e, -0.999, 0.999)
        Q[c] = np.clip(Q[c] * e, -0.999, 0.999)
        for step in np.flatnonzero(np.diff(e)) // L:  # (the path delays the step by about one frame)
            steady[c, max(step - 1, 0):step + 3] = False
    kw = dict(mode=0, AGCMode=agcmode, AGC_thresh=90)  # max_gain 10^4.5: min_volts far below the faded level

    def run(split_a, split_b):
        rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
        dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
        outs = [rx.ProcessIQData(dI[:, a:b].contiguous(), dQ[:, a:b].contiguous()) for a, b in zip(split_a[:-1], split_a[1:])]
        rx.CalcFilters(AGC_thresh=-20)  # max_gain 0.1: min_volts now ABOVE what the decay has reached
        outs += [rx.ProcessIQData(dI[:, a:b].contiguous(), dQ[:, a:b].contiguous()) for a, b in zip(split_b[:-1], split_b[1:])]
        torch.cuda.synchronize()
        return torch.cat(outs, dim=1).cpu().numpy(), rx.get_state()

    c = cut * L
    pipe, st_p = run([0, c], [c, nfr * L])                                              # two pipelined calls
    barr, st_b = run([0, 3 * L, c], [c + 3 * k * L for k in range(5)] + [nfr * L])      # barrier form only (<= 3 frames per call)
    assert np.isfinite(pipe).all()
    assert np.array_equal(pipe, barr)
    assert np.array_equal(st_p, st_b)
    ob = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32))
    r1 = ob.process(np.ascontiguousarray(I[:, :c]), np.ascontiguousarray(Q[:, :c]))
    ob.p.AGC_thresh = -20
    ob.redesign()
    r2 = ob.process(np.ascontiguousarray(I[:, c:]), np.ascontiguousarray(Q[:, c:]))
    err = siggen.block_rel_err(pipe, np.concatenate([r1, r2], 1), L)
    # Held to 1e-5 in the frames away from the 32 dB steps.  The frames a step runs through hold, next to samples 40 x
    # smaller, the FFT rounding noise of the loud ones -- in the oracle as in the kernel, at 1e-7 of the LOUD level --
    # so their block-relative figure says how deep the fade is, not how the two agree: 5e-5 there.  The frames after the
    # parameter change in which a wrongly kept fast decay would show are steady ones (the quiet stretch behind the fade).
    assert steady[:, cut + 3:cut + 6].all()
    where = np.unravel_index(np.where(steady, err, 0).argmax(), err.shape)
    assert err[steady].max() <= TOL, (err[steady].max(), "channel, frame", where)
    assert err.max() <= 6e-5, (err.max(), np.unravel_index(err.argmax(), err.shape))


@pytest.mark.parametrize("agcmode", [1, 2, 4])
def test_agc_restored_below_min_volts_in_fast_decay(T, agcmode):
    """ADVICE r04's case made deterministic: a restored checkpoint puts every channel in the fast decay (state 1) with
    volts <= save_volts < min_volts.  DSP_Fn.cpp:579-594 leaves the fast decay at the very first step (volts is not above
    save_volts) -- to the hang, the slow or the hang decay, by hang counter and decay type -- and :629 then lifts volts to
    min_volts.  A short-cut that looks at the last step of a four-step block only sees min_volts > save_volts and keeps
    the lane in state 1 (round 4's did: tools/build_variant.sh r04check -DT41RX_AGC_R04CHECK=1 fails this test, and
    tools/pipe_soak.py finds it within seconds).  Pipelined form == barrier form, outputs and checkpoints."""
    import torch
    nch, nfr = 37, 12
    nco = siggen.nco_grid(nch, seed=191)
    I, Q = siggen.make_iq(nch, nfr * L, nco, mode=0, seed=192)
    I[:, :4 * L] = 0.0  # four frames of silence: no attack while the restored state plays out, then signal
    Q[:, :4 * L] = 0.0
    kw = dict(mode=0, AGCMode=agcmode)  # AGC_thresh 20: min_volts 0.0654

    def run(split):
        rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
        ck = rx.get_state()
        rec = rx.state_records(ck)            # a view into ck
        w = rec[:, -8:]                       # rx_internal.hpp kAgcSt*: fast / hang back-average, volts, save_volts, state, decay type, hang counter
        w[:, 2] = 0.01
        w[:, 3] = 0.02
        wi = w.view(np.int32)
        wi[:, 4] = 1
        wi[:, 5] = np.arange(nch) % 2
        wi[:, 6] = np.where(np.arange(nch) % 3 == 0, 40, 0)
        rx.set_state(ck)
        dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
        out = torch.cat([rx.ProcessIQData(dI[:, a:b].contiguous(), dQ[:, a:b].contiguous()) for a, b in zip(split[:-1], split[1:])], dim=1)
        torch.cuda.synchronize()
        return out.cpu().numpy(), rx.get_state(), rx.state_records()[:, -8:].view(np.int32)[:, 4].copy()

    pipe, st_p, states_p = run([0, 6 * L, nfr * L])                  # two pipelined calls
    barr, st_b, states_b = run([k * L for k in range(nfr + 1)])      # one frame per call: the barrier form
    mid_p = run([0, 6 * L])[2]                                       # the state words after the silence + two frames
    mid_b = run([k * L for k in range(7)])[2]
    assert np.isfinite(pipe).all() and np.abs(pipe[:, 6 * L:]).max() > 0
    assert np.array_equal(mid_p, mid_b), (mid_p, mid_b)
    assert np.array_equal(pipe, barr)
    assert np.array_equal(st_p, st_b)


def test_agc_mode_change_and_reset(T):
    """AGCMode is a parameter like the filter edges: switching it mid-stream keeps the delay line and
    the gain state (the firmware only re-runs AGCLoadValues()), reset() returns to power-on"""
    nch, nfr = 6, 8
    nco = siggen.nco_grid(nch, seed=77)
    I, Q = siggen.make_iq(nch, nfr * L, nco, mode=0, seed=78)
    I, Q = siggen.fade(I, Q, [(0.5, 2.0), (0.5, 0.1)])
    kw = dict(mode=0, AGCMode=3)
    first, rx = gpu_run(T, kw, nco, I, Q)
    rx.reset()
    import torch
    again = rx.ProcessIQData(torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()).cpu().numpy()
    assert np.array_equal(first, again)
    # oracle with the same mid-stream switch 3 -> 4 after frame 4
    ob = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32))
    r1 = ob.process(np.ascontiguousarray(I[:, :4 * L]), np.ascontiguousarray(Q[:, :4 * L]))
    ob.p.AGCMode = 4
    ob.redesign()
    r2 = ob.process(np.ascontiguousarray(I[:, 4 * L:]), np.ascontiguousarray(Q[:, 4 * L:]))
    rx.reset()
    g1 = rx.ProcessIQData(torch.from_numpy(I[:, :4 * L].copy()).cuda(), torch.from_numpy(Q[:, :4 * L].copy()).cuda()).cpu().numpy()
    rx.CalcFilters(AGCMode=4)
    g2 = rx.ProcessIQData(torch.from_numpy(I[:, 4 * L:].copy()).cuda(), torch.from_numpy(Q[:, 4 * L:].copy()).cuda()).cpu().numpy()
    err = siggen.block_rel_err(np.concatenate([g1, g2], 1), np.concatenate([r1, r2], 1), L)
    assert err.max() <= TOL, err.max()


def _random_cases(n, seed):
    """valid points of the t41rx_params space (design.cpp: params_valid), seeded"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        mode = int(rng.integers(0, 4))
        if mode == 1:
            hi = -int(rng.integers(50, 800))
            lo = hi - int(rng.integers(300, 5000))
        elif mode == 2:
            lo, hi = -int(rng.integers(1500, 6000)), int(rng.integers(1500, 6000))
        else:
            lo = int(rng.integers(50, 800))
            hi = lo + int(rng.integers(300, 5000))
        kw = dict(mode=mode, FLoCut=lo, FHiCut=hi,
                  rfGainAllBands=int(rng.integers(-10, 11)), RFgain=int(rng.integers(1, 5)),
                  audioVolume=int(rng.integers(5, 101)),
                  IQAmpCorrectionFactor=float(np.float32(rng.uniform(0.9, 1.1))),
                  IQPhaseCorrectionFactor=float(np.float32(rng.uniform(-0.05, 0.05))),
                  xmtMode=int(rng.integers(0, 3)), CWFreqShift=int(rng.choice([562, 656, 750, 843])),
                  AGCMode=int(rng.integers(0, 5)), AGC_thresh=int(rng.integers(10, 40)))
        out.append(kw)
    return out


@pytest.mark.parametrize("kw", _random_cases(16, seed=2026), ids=lambda kw: "m%d-agc%d" % (kw["mode"], kw["AGCMode"]))
def test_parity_random_parameter_points(T, kw):
    """seeded random points of the parameter space, 7 channels (ragged workgroup) x 6 frames, a
    level step in the middle so the AGC (when on) leaves its attack state"""
    nch, nfr = 7, 6
    nco = siggen.nco_grid(nch, seed=kw["FHiCut"] & 1023)
    mode = kw["mode"]
    side = 0
    if kw["xmtMode"] == 1:
        side = kw["CWFreqShift"] if mode == 1 else (-kw["CWFreqShift"] if mode == 0 else 0)
    if mode == 3:
        I, Q = siggen.make_fm(nch, nfr * L, nco + side, seed=kw["FLoCut"] & 255)
    else:
        lo, hi = (abs(kw["FHiCut"]), abs(kw["FLoCut"])) if mode == 1 else (max(kw["FLoCut"], 100), kw["FHiCut"])
        band = (lo + 0.15 * (hi - lo), lo + 0.85 * (hi - lo))
        I, Q = siggen.make_iq(nch, nfr * L, nco + side, mode=mode, seed=kw["audioVolume"], audio_hz=band)
    I, Q = siggen.fade(I, Q, [(0.5, 1.0), (0.5, 0.2)])
    got, _ = gpu_run(T, kw, nco, I, Q)
    ref = oracle_run(kw, nco, I, Q)
    err = siggen.block_rel_err(got, ref, L)
    assert np.isfinite(got).all()
    assert err.max() <= (AM_TOL if mode == 2 else TOL), (kw, err.max(), np.unravel_index(err.argmax(), err.shape))


@pytest.mark.parametrize("N", [1024, 2048])
def test_parity_fft1024_2048(T, N):
    """the other synthetic FFT lengths of the oracle/designer (FFT_LENGTH is a compile-time
    constant of the firmware, 512): same three-kernel pipeline as 4096 with 2 / 4 segments per frame"""
    Lf = 4 * N
    nch, nfr = 9, 5
    nco = siggen.nco_grid(nch, seed=N)
    kw = dict(fft_length=N, mode=1, FLoCut=-2800, FHiCut=-300)
    I, Q = siggen.make_iq(nch, nfr * Lf, nco, mode=1, seed=N + 1, audio_hz=(500.0, 2400.0))
    got, _ = gpu_run(T, kw, nco, I, Q)
    ref = oracle_run(kw, nco, I, Q)
    err = siggen.block_rel_err(got, ref, Lf)
    assert err.max() <= TOL, (err.max(), np.unravel_index(err.argmax(), err.shape))
    split, _ = gpu_run(T, kw, nco, I, Q, split=[0, Lf, 3 * Lf, nfr * Lf])
    assert np.array_equal(got, split)


@pytest.mark.parametrize("N,kw", [
    (1024, dict(mode=2, FLoCut=-3000, FHiCut=3000)),
    (4096, dict(mode=2, FLoCut=-3000, FHiCut=3000)),
    (1024, dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=2)),
    (2048, dict(mode=1, FLoCut=-3000, FHiCut=-200, AGCMode=3)),
    (4096, dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=1)),
    (2048, dict(mode=2, FLoCut=-3000, FHiCut=3000, AGCMode=4)),
    (1024, dict(mode=3, FLoCut=200, FHiCut=3000)),
    (4096, dict(mode=3, FLoCut=200, FHiCut=3000)),
    (2048, dict(mode=3, FLoCut=200, FHiCut=3000, AGCMode=2)),
], ids=["am-1024", "am-4096", "usb-agc-1024", "lsb-agc-2048", "usb-agc-4096", "am-agc-2048", "nfm-1024", "nfm-4096", "nfm-agc-2048"])
def test_parity_long_fft_am_nfm_agc(T, N, kw):
    """AM, NFM and the AGC at the synthetic FFT lengths: the fast convolution hands the complex valid half
    to the back kernel, which runs the AGC / demodulator per 256-sample segment"""
    Lf = 4 * N
    nch, nfr = 7, 6
    nco = siggen.nco_grid(nch, seed=N + 3)
    kw = dict(kw, fft_length=N)
    if kw["mode"] == 3:
        I, Q = siggen.make_fm(nch, nfr * Lf, nco, seed=N + 4)
    else:
        I, Q = siggen.make_iq(nch, nfr * Lf, nco, mode=kw["mode"], seed=N + 4, audio_hz=(500.0, 2400.0))
    I, Q = siggen.fade(I, Q, [(0.5, 2.0), (0.2, 0.1), (0.3, 1.0)])
    got, _ = gpu_run(T, kw, nco, I, Q)
    ref = oracle_run(kw, nco, I, Q)
    err = siggen.block_rel_err(got, ref, Lf)
    assert np.isfinite(got).all()
    assert err.max() <= (AM_TOL if kw["mode"] == 2 else TOL), (err.max(), np.unravel_index(err.argmax(), err.shape))
    split, _ = gpu_run(T, kw, nco, I, Q, split=[0, Lf, 4 * Lf, nfr * Lf])
    assert np.array_equal(got, split)


def test_parity_fft4096(T):
    """BASELINE config 4 (synthetic generalisation): FFT_LENGTH 4096, 16384-sample
    frames, 2049-tap narrow USB filter (400..600 Hz), three-kernel pipeline"""
    Lf = 16384
    nch, nfr = 12, 4
    nco = siggen.nco_grid(nch, seed=23)
    kw = dict(fft_length=4096, mode=0, FLoCut=400, FHiCut=600)
    I, Q = siggen.make_iq(nch, nfr * Lf, nco, mode=0, seed=41, audio_hz=(450.0, 550.0))
    got, _ = gpu_run(T, kw, nco, I, Q)
    ref = oracle_run(kw, nco, I, Q)
    err = siggen.block_rel_err(got, ref, Lf)
    # frame 0 is almost pure filter start-up (2049-tap filter): compared absolutely by block_rel_err's rule
    assert err[:, 1:].max() <= TOL, (err.max(), np.unravel_index(err.argmax(), err.shape))
    assert np.abs(got[:, :Lf] - ref[:, :Lf]).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30)
    split, _ = gpu_run(T, kw, nco, I, Q, split=[0, Lf, 3 * Lf, nfr * Lf])
    assert np.array_equal(got, split)


# AM: the reference's DC remover w = |z| + 0.99 w_old accumulates ~100x the signal in f32, so the
# ORACLE itself sits ~1e-5 (block-relative) away from an exact evaluation of the same formula
# (tests/test_oracle_vs_f64.py measures that floor).  Two f32 evaluations with inputs that differ
# in the last bit decorrelate at that level, so 1e-5 vs the oracle is not attainable by any
# implementation; the GPU runs this scan in f64 and is held to 5e-5.


def test_parity_am(T):
    nch, nfr = 32, 6
    nco = siggen.nco_grid(nch, seed=17)
    kw = dict(mode=2, FLoCut=-3000, FHiCut=3000)
    I, Q = siggen.make_iq(nch, nfr * L, nco, mode=2, seed=31)
    got, _ = gpu_run(T, kw, nco, I, Q)
    ref = oracle_run(kw, nco, I, Q)
    err = siggen.block_rel_err(got, ref, L)
    assert err.max() <= AM_TOL, (err.max(), np.unravel_index(err.argmax(), err.shape))
    split, _ = gpu_run(T, kw, nco, I, Q, split=[0, L, 3 * L, nfr * L])
    assert np.array_equal(got, split)


def test_am_hip_is_closer_to_the_exact_formula_than_the_oracle(T):
    """The evidence behind AM_TOL: Process.cpp:698-704's DC remover w = m + 0.99 w_old, y = w - w_old
    accumulates ~100 x the signal level in f32, so an f32 evaluation (the reference's, the oracle's) sits ~1e-5 away from
    the formula evaluated exactly, and two f32-fed evaluations decorrelate at that level.  The HIP path runs that scan in
    f64.  Shown here on the same input, against the independent float64 stream model (tests/f64_model.py): per block,
    the HIP output is CLOSER to the exact evaluation than the oracle is, HIP-vs-exact meets north_star's 1e-5, and the
    HIP-vs-oracle distance the 5e-5 bar covers is the oracle's own f32 error (triangle inequality, measured)."""
    import f64_model as M
    nch, nfr = 24, 8
    nco = siggen.nco_grid(nch, seed=117)
    kw = dict(mode=2, FLoCut=-3000, FHiCut=3000)
    I, Q = siggen.make_iq(nch, nfr * L, nco, mode=2, seed=131)
    got, _ = gpu_run(T, kw, nco, I, Q)
    ob = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32))
    orc = ob.process(I, Q, nthreads=8)
    coeffs = O.coeff_arrays(ob.c, 512)
    exact = np.stack([M.run(I[c], Q[c], int(nco[c]), coeffs, mode=2, FLoCut=-3000, FHiCut=3000) for c in range(nch)])
    d_hip = siggen.block_rel_err(got, exact, L)   # HIP vs the exact formula
    d_orc = siggen.block_rel_err(orc, exact, L)   # oracle (f32 scan, as the reference) vs the exact formula
    d_go = siggen.block_rel_err(got, orc, L)      # what test_parity_am holds to AM_TOL
    msg = "HIP-vs-exact max %.2e median %.2e; oracle-vs-exact max %.2e median %.2e; HIP-vs-oracle max %.2e" % (
        d_hip.max(), np.median(d_hip), d_orc.max(), np.median(d_orc), d_go.max())
    print(msg)
    assert d_hip.max() <= TOL, msg                                   # 1e-5 against the exact evaluation
    assert d_hip.max() <= d_orc.max() and np.median(d_hip) <= np.median(d_orc), msg
    assert (d_hip <= d_orc + 1e-6).mean() >= 0.95, msg               # block by block, not just in the maximum
    assert d_go.max() <= d_hip.max() + d_orc.max() + 1e-9 and d_go.max() <= AM_TOL, msg


def test_streaming_split_is_bit_identical(T):
    """frame-by-frame launches == one multi-frame launch == uneven splits (state carried in HBM)"""
    nch, nfr = 16, 6
    nco = siggen.nco_grid(nch, seed=5)
    I, Q = siggen.make_iq(nch, nfr * L, nco, seed=8)
    kw = dict(mode=0, FLoCut=200, FHiCut=3000)
    a, _ = gpu_run(T, kw, nco, I, Q)
    b, _ = gpu_run(T, kw, nco, I, Q, split=[k * L for k in range(nfr + 1)])
    c, _ = gpu_run(T, kw, nco, I, Q, split=[0, L, 4 * L, 6 * L])
    assert np.array_equal(a, b) and np.array_equal(a, c)


def test_long_stream_phase_and_state_do_not_drift(T):
    """200 consecutive frames (2 s of signal): the fixed-point NCO phase and every delay line
    must track the oracle's f64 recurrence for the whole stream, not just the first frames"""
    nch, nfr = 4, 200
    nco = np.array([39950, -42950, 12350, 50], np.int32)
    I, Q = siggen.make_iq(nch, nfr * L, nco, seed=77)
    kw = dict(mode=0, FLoCut=200, FHiCut=3000)
    got, _ = gpu_run(T, kw, nco, I, Q)
    ref = oracle_run(kw, nco, I, Q)
    err = siggen.block_rel_err(got, ref, L)
    assert err.max() <= TOL, (err.max(), err[:, -5:])
    assert err[:, -20:].max() <= 3 * max(err[:, 5:25].max(), 2e-7) + 1e-6  # no growth over time


@pytest.mark.parametrize("nfr", [4, 5, 6, 7, 8])
def test_agc_pipelined_short_calls(T, nfr):
    """the shortest calls the pipelined kernel takes (its pipeline is two frames deep, its slot ring three): every
    length from 4 to 8 frames against the barrier form, 19 channels, AM (both slot halves of the popped samples in use)"""
    nch = 19
    nco = siggen.nco_grid(nch, seed=60 + nfr)
    I, Q = siggen.make_iq(nch, (nfr + 2) * L, nco, mode=2, seed=61)
    I, Q = siggen.fade(I, Q, [(0.5, 1.5), (0.5, 0.1)])
    kw = dict(mode=2, AGCMode=4, FLoCut=-3000, FHiCut=3000)
    a, rxa = gpu_run(T, kw, nco, I, Q, split=[0, 2 * L, (nfr + 2) * L])            # barrier (2 frames), then pipelined (nfr)
    b, rxb = gpu_run(T, kw, nco, I, Q, split=[k * L for k in range(0, nfr + 3)])   # frame by frame: barrier form only
    assert np.array_equal(a, b) and np.array_equal(rxa.get_state(), rxb.get_state())


def test_long_stream_agc_pipelined(T):
    """200 frames in ONE call with the AGC on: the pipelined kernel's three-slot ring turns 66 times and the chain's
    duty goes round the workgroup's waves 10 times (19 channels: a full workgroup and a ragged one of 3); fading drives
    the gain law through its states the whole way.  Against the oracle, and against the same stream in short calls."""
    nch, nfr = 19, 200
    nco = siggen.nco_grid(nch, seed=91)
    I, Q = siggen.make_iq(nch, nfr * L, nco, seed=92)
    I, Q = siggen.fade(I, Q, [(0.1, 2.0), (0.15, 0.05), (0.1, 1.0), (0.2, 0.02), (0.15, 2.5), (0.3, 0.3)])
    kw = dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=3)
    got, rx = gpu_run(T, kw, nco, I, Q)
    ref = oracle_run(kw, nco, I, Q)
    err = siggen.block_rel_err(got, ref, L)
    assert np.isfinite(got).all() and err.max() <= TOL, (err.max(), np.unravel_index(err.argmax(), err.shape))
    short, rx2 = gpu_run(T, kw, nco, I, Q, split=[k * L for k in range(0, nfr + 1, 2)])  # 2-frame calls: the barrier form
    assert np.array_equal(got, short) and np.array_equal(rx.get_state(), rx2.get_state())


def test_filter_and_tuning_change_mid_stream(T):
    """CalcFilters()/NCOFreq change between two ProcessIQData() calls: coefficients change, state
    (delay lines, oscillator phase) is kept -- exactly what the reference does"""
    import torch
    nch = 8
    nco1 = siggen.nco_grid(nch, seed=1)
    nco2 = nco1 + 300  # audio tone moves down by 300 Hz: stays inside both filters
    I, Q = siggen.make_iq(nch, 6 * L, nco1, seed=4, audio_hz=(1000.0, 1500.0))
    kw1 = dict(mode=0, FLoCut=200, FHiCut=3000)
    kw2 = dict(mode=0, FLoCut=300, FHiCut=1800, audioVolume=45)
    rx = T.RxChain(nch, T.default_params(**kw1), NCOFreq=nco1)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    o1 = rx.ProcessIQData(dI[:, :2 * L].contiguous(), dQ[:, :2 * L].contiguous())
    rx.CalcFilters(**kw2)
    o2 = rx.ProcessIQData(dI[:, 2 * L:4 * L].contiguous(), dQ[:, 2 * L:4 * L].contiguous())
    rx.SetNCOFreq(nco2)
    o3 = rx.ProcessIQData(dI[:, 4 * L:].contiguous(), dQ[:, 4 * L:].contiguous())
    torch.cuda.synchronize()
    got = torch.cat([o1, o2, o3], dim=1).cpu().numpy()
    ob = O.OracleBatch(O.default_params(**kw1), nco1)
    r1 = ob.process(I[:, :2 * L], Q[:, :2 * L])
    for k, v in kw2.items():
        setattr(ob.p, k, v)
    ob.redesign()
    r2 = ob.process(I[:, 2 * L:4 * L], Q[:, 2 * L:4 * L])
    ob.nco[:] = nco2
    r3 = ob.process(I[:, 4 * L:], Q[:, 4 * L:])
    ref = np.concatenate([r1, r2, r3], axis=1)
    err = siggen.block_rel_err(got, ref, L)
    assert err.max() <= TOL, (err.max(), np.unravel_index(err.argmax(), err.shape))


@pytest.mark.parametrize("kw", [dict(), dict(AGCMode=2), dict(mode=2, FLoCut=-3000, FHiCut=3000, AGCMode=1), dict(mode=3)],
                         ids=["usb", "usb-agc", "am-agc", "nfm"])
def test_checkpoint_and_reset(T, kw):
    import torch
    nch = 8
    nco = siggen.nco_grid(nch, seed=9)
    I, Q = siggen.make_iq(nch, 4 * L, nco, seed=10, mode=kw.get("mode", 0))
    I, Q = siggen.fade(I, Q, [(0.4, 1.5), (0.6, 0.3)])
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    a1 = rx.ProcessIQData(dI[:, :2 * L].contiguous(), dQ[:, :2 * L].contiguous())
    snap = rx.get_state()
    a2 = rx.ProcessIQData(dI[:, 2 * L:].contiguous(), dQ[:, 2 * L:].contiguous()).clone()
    rx.set_state(snap)  # resume from the checkpoint: identical continuation
    a2b = rx.ProcessIQData(dI[:, 2 * L:].contiguous(), dQ[:, 2 * L:].contiguous())
    assert torch.equal(a2, a2b)
    rx.reset()          # power-on state: identical to a fresh context
    b1 = rx.ProcessIQData(dI[:, :2 * L].contiguous(), dQ[:, :2 * L].contiguous())
    assert torch.equal(a1, b1)
    # a restored snapshot also works in a different context
    rx2 = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    rx2.set_state(snap)
    a2c = rx2.ProcessIQData(dI[:, 2 * L:].contiguous(), dQ[:, 2 * L:].contiguous())
    assert torch.equal(a2, a2c)


def test_host_pointer_entry_point_matches_device_one(T):
    nch = 5  # ragged: not a multiple of the 4 channels a workgroup carries
    nco = siggen.nco_grid(nch, seed=2)
    I, Q = siggen.make_iq(nch, 2 * L, nco, seed=6)
    dev, _ = gpu_run(T, dict(), nco, I, Q)
    rx = T.RxChain(nch, T.default_params(), NCOFreq=nco)
    host = rx.ProcessIQData(I, Q)
    assert isinstance(host, np.ndarray) and np.array_equal(host, dev)
    assert siggen.block_rel_err(host, oracle_run(dict(), nco, I, Q), L).max() <= TOL


def test_silence_and_full_scale_inputs(T):
    """edge inputs: all-zero frames give exactly zero audio (no NaN from 0/0 anywhere); +-full
    scale square-ish input stays finite and matches"""
    nch = 4
    nco = np.array([0, 1000, -30000, 40000], np.int32)
    Z = np.zeros((nch, 2 * L), np.float32)
    got, _ = gpu_run(T, dict(), nco, Z, Z)
    assert np.array_equal(got, np.zeros_like(got))
    rng = np.random.default_rng(0)
    I = np.sign(rng.standard_normal((nch, 3 * L))).astype(np.float32) * 0.999
    Q = np.sign(rng.standard_normal((nch, 3 * L))).astype(np.float32) * 0.999
    got, _ = gpu_run(T, dict(), nco, I, Q)
    assert siggen.block_rel_err(got, oracle_run(dict(), nco, I, Q), L).max() <= TOL


@pytest.mark.parametrize("agcmode", [1, 4])
def test_agc_edge_inputs(T, agcmode):
    """AGC on with silence (volts rests on min_volts, the gain stays finite, audio exactly 0), silence
    followed by a full-scale burst (the 97-sample look-ahead meets it) and back to silence"""
    nch = 5
    nco = np.array([0, 1000, -30000, 40000, 5000], np.int32)
    kw = dict(AGCMode=agcmode)
    Z = np.zeros((nch, 2 * L), np.float32)
    got, rx = gpu_run(T, kw, nco, Z, Z)
    assert np.array_equal(got, np.zeros_like(got))
    st = rx.state_records()
    assert np.all(st[:, 768 + 200 + 2] == np.float32(O.coeff_arrays(O.design(O.default_params(**kw)), 512)["agc"][9]))  # min_volts
    rng = np.random.default_rng(1)
    I = np.sign(rng.standard_normal((nch, 6 * L))).astype(np.float32) * 0.999
    Q = np.sign(rng.standard_normal((nch, 6 * L))).astype(np.float32) * 0.999
    I[:, :2 * L] = 0
    Q[:, :2 * L] = 0
    I[:, 4 * L + 300:] = 0
    Q[:, 4 * L + 300:] = 0
    got, _ = gpu_run(T, kw, nco, I, Q)
    ref = oracle_run(kw, nco, I, Q)
    assert np.isfinite(got).all()
    assert siggen.block_rel_err(got, ref, L).max() <= TOL


def test_argument_errors(T):
    import torch
    from t41_sdr_amd import _lib
    rx = T.RxChain(4, T.default_params())
    x = torch.zeros(4, L, device="cuda")
    with pytest.raises(ValueError):
        rx.ProcessIQData(x[:, :1000].contiguous(), x[:, :1000].contiguous())  # not a whole frame
    with pytest.raises(ValueError):
        rx.ProcessIQData(torch.zeros(3, L, device="cuda"), torch.zeros(3, L, device="cuda"))  # wrong batch
    lib = T.load()
    assert lib.t41rx_process_device(rx._ctx, None, None, None, 1, None) == _lib.ERR_ARG
    assert lib.t41rx_process_device(rx._ctx, x.data_ptr(), x.data_ptr(), x.data_ptr(), 0, None) == _lib.ERR_ARG
    with pytest.raises(T.T41RxError) as e:
        rx.SetNCOFreq(np.full(4, 200000))
    assert e.value.status == _lib.ERR_ARG
    # what has no kernel answers UNSUPPORTED: the side output and the q15 samples at the long FFT lengths
    rx4k = T.RxChain(2, T.default_params(fft_length=4096, FLoCut=400, FHiCut=600))
    with pytest.raises(T.T41RxError) as e:
        rx4k.set_audio_spectrum(torch.zeros(2, 1, 1024, device="cuda"), torch.zeros(2, 1, 3, device="cuda"))
    assert e.value.status == _lib.ERR_UNSUPPORTED
    with pytest.raises(T.T41RxError) as e:
        rx.CalcFilters(fft_length=1024)
    assert e.value.status in (_lib.ERR_ARG, _lib.ERR_UNSUPPORTED)


# ---- BASELINE.json full size: 4096 channels x 2048 samples, checked through properties that
# ---- do not need the (slow) oracle on the whole batch
def test_full_batch_properties(T):
    import torch
    nch, nfr = 4096, 3
    rng = np.random.default_rng(4096)
    nco = (rng.integers(-860, 801, nch) * 50).astype(np.int32)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = 0.2 * torch.randn(nch, nfr * L, generator=g, device="cuda")
    y = 0.2 * torch.randn(nch, nfr * L, generator=g, device="cuda")
    xq = 0.2 * torch.randn(nch, nfr * L, generator=g, device="cuda")
    yq = 0.2 * torch.randn(nch, nfr * L, generator=g, device="cuda")

    def run(i, q):
        rx = T.RxChain(nch, T.default_params(), NCOFreq=nco)
        return rx.ProcessIQData(i.contiguous(), q.contiguous())

    fx, fy = run(x, xq), run(y, yq)
    # (1) linearity of the SSB chain with AGC off: F(a x + b y) = a F(x) + b F(y)
    fz = run(0.5 * x - 1.25 * y, 0.5 * xq - 1.25 * yq)
    want = 0.5 * fx - 1.25 * fy
    scale = want.abs().amax(dim=1, keepdim=True).clamp_min(1e-12)
    assert float(((fz - want).abs() / scale).max()) < 2e-5
    # (2) channels are independent: permuting channels (with their NCOs) permutes outputs exactly
    perm = torch.randperm(nch, generator=torch.Generator().manual_seed(3))
    rxp = T.RxChain(nch, T.default_params(), NCOFreq=nco[perm.numpy()])
    fp = rxp.ProcessIQData(x[perm.cuda()].contiguous(), xq[perm.cuda()].contiguous())
    assert torch.equal(fp, fx[perm.cuda()])
    # (3) a 64-channel sample of the big batch agrees with the oracle
    idx = np.sort(rng.choice(nch, 64, replace=False))
    ref = oracle_run(dict(), nco[idx], x[idx].cpu().numpy(), xq[idx].cpu().numpy())
    assert siggen.block_rel_err(fx[idx].cpu().numpy(), ref, L).max() <= TOL
    # (4) out-of-band rejection: energy far outside the 200-3000 Hz audio band is gone
    spec = torch.fft.rfft(fx[:64, L:].double() * torch.hann_window(2 * L, device="cuda", dtype=torch.float64))
    f = torch.fft.rfftfreq(2 * L, 1 / 192000.0).cuda()
    inband = spec[:, (f > 300) & (f < 2900)].abs().amax(dim=1)
    far = spec[:, f > 20000].abs().amax(dim=1)
    assert float((far / inband).max()) < 1e-3


# ---- stage-level attribution: the C ABI's taps against the oracle's, stage by stage ----
@pytest.mark.parametrize("mode,flo,fhi", [(0, 200, 3000), (1, -3000, -200), (2, -3000, 3000), (3, 200, 3000)],
                         ids=["usb", "lsb", "am", "nfm"])
def test_stage_taps_match_the_oracle(T, mode, flo, fhi):
    """post-NCO, post-decimation and demodulator taps (t41rx_set_debug_taps) vs T41O_TAP_POST_NCO_*,
    TAP_DEC_*, TAP_DEMOD, frame by frame from reset: the first frame carries the oscillator's
    start-up transient (Freq_Shift.cpp:128-139), so a regression there shows up at its stage."""
    import torch
    nch, nfr, N, D = 6, 3, 512, 256
    nco = siggen.nco_grid(nch, seed=21)
    if mode == 3:
        I, Q = siggen.make_fm(nch, nfr * L, nco, seed=22)
    else:
        I, Q = siggen.make_iq(nch, nfr * L, nco, mode=mode, seed=22)
    kw = dict(mode=mode, FLoCut=flo, FHiCut=fhi)
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    t_nco = torch.zeros(nch, 2 * L, device="cuda")
    t_dec = torch.zeros(nch, N, device="cuda")
    t_dem = torch.zeros(nch, D, device="cuda")
    rx.set_debug_taps(t_nco, t_dec, t_dem)
    ob = O.OracleBatch(O.default_params(**kw), nco)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))  # noqa: E731
    for f in range(nfr):
        sl = slice(f * L, (f + 1) * L)
        out = rx.ProcessIQData(dI[:, sl].contiguous(), dQ[:, sl].contiguous())
        torch.cuda.synchronize()
        ref = ob.process(I[:, sl], Q[:, sl])
        g_nco, g_dec, g_dem = t_nco.cpu().numpy(), t_dec.cpu().numpy(), t_dem.cpu().numpy()
        for c in range(nch):
            r_nco = np.concatenate([ob.tap(c, O.TAP_POST_NCO_I, L), ob.tap(c, O.TAP_POST_NCO_Q, L)])
            r_dec = np.concatenate([ob.tap(c, O.TAP_DEC_I, D), ob.tap(c, O.TAP_DEC_Q, D)])
            r_dem = ob.tap(c, O.TAP_DEMOD, D)
            assert rel(g_nco[c], r_nco) <= TOL, ("post_nco", f, c)
            if mode != 3:  # NFM: the dec tap of the kernel is the discriminator's output (audio, 0), checked as demod below
                assert rel(g_dec[c], r_dec) <= TOL, ("dec", f, c)
            assert rel(g_dem[c], r_dem) <= (AM_TOL if mode == 2 else TOL), ("demod", f, c)
        assert siggen.block_rel_err(out.cpu().numpy(), ref, L).max() <= (AM_TOL if mode == 2 else TOL)
    # more frames than the tap buffers hold: refused, nothing written out of bounds
    from t41_sdr_amd import _lib
    with pytest.raises(T.T41RxError) as e:
        rx.ProcessIQData(dI[:, :2 * L].contiguous(), dQ[:, :2 * L].contiguous())
    assert e.value.status == _lib.ERR_ARG
    rx.set_debug_taps(None, None, None)
    rx.ProcessIQData(dI[:, :2 * L].contiguous(), dQ[:, :2 * L].contiguous())


def test_broadcast_blob_carries_the_parameters(T):
    """t41rx_set_coeffs (the broadcast path): a context created with other parameters that
    installs rank 0's blob runs rank 0's configuration -- AGC on into an AGC-off context, and the
    other way round -- and reports its parameters"""
    import torch
    nch = 6
    nco = siggen.nco_grid(nch, seed=31)
    I, Q = siggen.make_iq(nch, 5 * L, nco, seed=32, mode=1)
    I, Q = siggen.fade(I, Q, [(0.5, 1.0), (0.5, 0.15)])
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    for src_kw, dst_kw in ((dict(mode=1, FLoCut=-2800, FHiCut=-250, AGCMode=3, audioVolume=40), dict()),
                           (dict(mode=1, FLoCut=-2800, FHiCut=-250, AGCMode=0, audioVolume=40), dict(AGCMode=2))):
        blob = T.design_coeffs(T.default_params(**src_kw))
        rx = T.RxChain(nch, T.default_params(**dst_kw), NCOFreq=nco)
        rx.set_coeffs(blob)
        got = rx.get_params()
        for k, v in src_kw.items():
            assert getattr(got, k) == v
        assert T.blob_params(blob).AGCMode == src_kw["AGCMode"]
        out = rx.ProcessIQData(dI, dQ).cpu().numpy()
        ref = oracle_run(src_kw, nco, I, Q)
        assert siggen.block_rel_err(out, ref, L).max() <= TOL
        # a later filter change on that context starts from the installed parameters
        rx.CalcFilters(**{k: getattr(got, k) for k in ("mode", "FLoCut", "FHiCut", "AGCMode", "audioVolume")})
        assert np.array_equal(rx.coeffs(), blob)
    # a blob with a damaged parameter section is refused
    from t41_sdr_amd import _lib
    bad = T.design_coeffs(T.default_params()).copy()
    bad[4 * 8 + 4 * 8] = 77  # AGCMode word of the parameter section
    with pytest.raises(T.T41RxError) as e:
        T.RxChain(2, T.default_params()).set_coeffs(bad)
    assert e.value.status == _lib.ERR_STATE


def test_iq_amplitude_correction_of_minus_one(T):
    """IQAmpCorrectionFactor = -1 makes the reference's I <- I * (-A) a multiplication by +1: the
    specialised kernel that folds the usual sign flip into the RF gain must not be chosen"""
    nch = 5
    nco = siggen.nco_grid(nch, seed=41)
    for mode, flo, fhi in ((0, 200, 3000), (2, -3000, 3000)):
        kw = dict(mode=mode, FLoCut=flo, FHiCut=fhi, IQAmpCorrectionFactor=-1.0)
        I, Q = siggen.make_iq(nch, 3 * L, nco, seed=42, mode=mode)
        got, _ = gpu_run(T, kw, nco, I, Q)
        assert siggen.block_rel_err(got, oracle_run(kw, nco, I, Q), L).max() <= (AM_TOL if mode == 2 else TOL)


def test_checkpoint_header_and_out_buffer_validation(T):
    import torch
    from t41_sdr_amd import _lib
    rx = T.RxChain(4, T.default_params())
    snap = rx.get_state()
    assert snap.size == rx.state_records(snap).size * 4 + 32
    with pytest.raises(T.T41RxError) as e:  # a checkpoint of another batch size / FFT length of the same total size class
        T.RxChain(3, T.default_params()).set_state(snap)
    assert e.value.status == _lib.ERR_STATE
    bad = snap.copy()
    bad[4] ^= 0xFF  # abi word
    with pytest.raises(T.T41RxError) as e:
        rx.set_state(bad)
    assert e.value.status == _lib.ERR_STATE
    bad = snap.copy()
    rec = bad[32:].view(np.float32).reshape(4, -1)
    rec[1, 768 + 200 + 4] = np.frombuffer(np.int32(9).tobytes(), np.float32)[0]  # AGC state word out of range
    with pytest.raises(T.T41RxError) as e:
        rx.set_state(bad)
    assert e.value.status == _lib.ERR_STATE
    rx.set_state(snap)
    x = torch.zeros(4, L, device="cuda")
    for wrong in (torch.zeros(4, L + 4, device="cuda"), torch.zeros(4, L, device="cuda", dtype=torch.float64),
                  torch.zeros(L, 4, device="cuda").t(), torch.zeros(4, L)):
        with pytest.raises(ValueError):
            rx.ProcessIQData(x, x, out=wrong)
    h = np.zeros((4, L), np.float32)
    with pytest.raises(ValueError):
        rx.ProcessIQData(h, h, out=np.zeros((4, L), np.float64))
    with pytest.raises(ValueError):
        rx.ProcessIQData(h, h, out=np.zeros((L, 4), np.float32).T)


# ---- BASELINE.json configs 3 and 4 at full size, through properties + an oracle sample ----
def test_full_batch_nfm(T):
    """config 3: NFM, 4096 channels: finite, channel-independent (a permutation of the channels
    permutes the outputs exactly), and a 64-channel sample agrees with the oracle"""
    import torch
    nch, nfr = 4096, 2
    rng = np.random.default_rng(333)
    nco = (rng.integers(-860, 801, nch) * 50).astype(np.int32)
    kw = dict(mode=3, FLoCut=200, FHiCut=3000, nfmFilterBW=12000)
    idx = np.sort(rng.choice(nch, 64, replace=False))
    g = torch.Generator(device="cuda").manual_seed(5)
    x = 0.2 * torch.randn(nch, nfr * L, generator=g, device="cuda")
    y = 0.2 * torch.randn(nch, nfr * L, generator=g, device="cuda")
    fi, fq = siggen.make_fm(64, nfr * L, nco[idx], seed=6)  # real FM carriers on the sampled channels
    x[torch.from_numpy(idx).cuda()] = torch.from_numpy(fi).cuda()
    y[torch.from_numpy(idx).cuda()] = torch.from_numpy(fq).cuda()
    out = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco).ProcessIQData(x, y)
    assert torch.isfinite(out).all()
    perm = torch.randperm(nch, generator=torch.Generator().manual_seed(7))
    outp = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco[perm.numpy()]).ProcessIQData(x[perm.cuda()].contiguous(), y[perm.cuda()].contiguous())
    assert torch.equal(outp, out[perm.cuda()])
    ref = oracle_run(kw, nco[idx], fi, fq)
    assert siggen.block_rel_err(out[torch.from_numpy(idx).cuda()].cpu().numpy(), ref, L).max() <= TOL


@pytest.mark.parametrize("fft_length", [1024, 4096])
def test_segment_run_length_invariance(T, fft_length, monkeypatch):
    """The long-FFT front / back kernels run segments in parallel and rebuild the filter memories of a
    wave that starts inside the call from the preceding input: the audio and the
    channel state must not depend on how many segments one wave runs."""
    import torch
    nch, nfr = 64, 6
    Lf = fft_length * 4
    rng = np.random.default_rng(fft_length)
    nco = (rng.integers(-860, 801, nch) * 50).astype(np.int32)
    kw = dict(fft_length=fft_length, mode=0, FLoCut=400, FHiCut=600)
    g = torch.Generator(device="cuda").manual_seed(fft_length + 1)
    x = 0.2 * torch.randn(nch, nfr * Lf, generator=g, device="cuda")
    y = 0.2 * torch.randn(nch, nfr * Lf, generator=g, device="cuda")
    outs, states = [], []
    for run in ("1", "3", "8", None):
        if run is None:
            monkeypatch.delenv("T41RX_SEG_RUN", raising=False)
        else:
            monkeypatch.setenv("T41RX_SEG_RUN", run)
        rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
        a = rx.ProcessIQData(x[:, :4 * Lf].contiguous(), y[:, :4 * Lf].contiguous())
        b = rx.ProcessIQData(x[:, 4 * Lf:].contiguous(), y[:, 4 * Lf:].contiguous())
        outs.append(torch.cat([a, b], dim=1))
        states.append(rx.get_state())
    monkeypatch.delenv("T41RX_SEG_RUN", raising=False)
    # FFT_LENGTH 4096 without T41RX_SEG_RUN runs the single-kernel form (front end inside the fast convolution's
    # workgroup, rx_kernels.hip: fastconv_fused_kernel): the same arithmetic compiled into another kernel, compared
    # to rounding level; the explicit run lengths select the two-kernel pipeline and must agree bit for bit
    same = len(outs) - (1 if fft_length == 4096 else 0)
    for o, s in zip(outs[1:same], states[1:same]):
        assert torch.equal(o, outs[0])
        assert np.array_equal(np.asarray(s), np.asarray(states[0]))
    if fft_length == 4096:
        e = siggen.block_rel_err(outs[-1].cpu().numpy(), outs[0].cpu().numpy(), Lf)
        assert e.max() <= 2e-6, e.max()
    ref = oracle_run(kw, nco[:8], x[:8].cpu().numpy(), y[:8].cpu().numpy())
    assert siggen.block_rel_err(outs[0][:8].cpu().numpy(), ref, Lf).max() <= TOL


def test_full_batch_fft4096(T):
    """config 4: the 4096-point fast convolution at 1024 channels x 16384 samples: finite,
    channel-independent, frame-split invariant, and a 32-channel sample agrees with the oracle"""
    import torch
    nch, nfr, L4 = 1024, 2, 16384
    rng = np.random.default_rng(444)
    nco = (rng.integers(-860, 801, nch) * 50).astype(np.int32)
    kw = dict(fft_length=4096, mode=0, FLoCut=400, FHiCut=600)
    g = torch.Generator(device="cuda").manual_seed(8)
    x = 0.2 * torch.randn(nch, nfr * L4, generator=g, device="cuda")
    y = 0.2 * torch.randn(nch, nfr * L4, generator=g, device="cuda")
    out = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco).ProcessIQData(x, y)
    assert torch.isfinite(out).all()
    rx2 = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    parts = [rx2.ProcessIQData(x[:, k * L4:(k + 1) * L4].contiguous(), y[:, k * L4:(k + 1) * L4].contiguous()) for k in range(nfr)]
    assert torch.equal(torch.cat(parts, dim=1), out)
    perm = torch.randperm(nch, generator=torch.Generator().manual_seed(9))
    outp = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco[perm.numpy()]).ProcessIQData(x[perm.cuda()].contiguous(), y[perm.cuda()].contiguous())
    assert torch.equal(outp, out[perm.cuda()])
    idx = np.sort(rng.choice(nch, 32, replace=False))
    sel = torch.from_numpy(idx).cuda()
    ref = oracle_run(kw, nco[idx], x[sel].cpu().numpy(), y[sel].cpu().numpy())
    assert siggen.block_rel_err(out[sel].cpu().numpy(), ref, L4).max() <= TOL

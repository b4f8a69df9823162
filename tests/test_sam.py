"""Synchronous AM detection (DEMOD_SAM, AMDecodeSAM Demod.cpp:40-139): the oracle's
restatement against an independent float64 model, and the HIP path against the oracle."""
import ctypes as C

import numpy as np
import pytest

import f64_model as M
import oracle_lib as O
import siggen

L = 2048
SAM = O.DEMOD_SAM
KW = dict(mode=SAM, FLoCut=-3000, FHiCut=3000)  # Filter.cpp:363-367


def _lib():
    lib = O.lib()
    lib.t41o_arm_sin_f32.restype = C.c_float
    lib.t41o_arm_sin_f32.argtypes = [C.c_float]
    lib.t41o_arm_cos_f32.restype = C.c_float
    lib.t41o_arm_cos_f32.argtypes = [C.c_float]
    lib.t41o_sin_table.restype = C.POINTER(C.c_float)
    lib.t41o_sam_constants.argtypes = [C.POINTER(C.c_float)]
    return lib


def test_fast_sine_restatement(built):
    """arm_sin_f32 / arm_cos_f32 as restated: 512-entry table + linear interpolation, within the
    method's (2 pi / 512)^2 / 8 = 1.9e-5 of the true functions, table = rounded exact sines"""
    lib = _lib()
    tab = np.ctypeslib.as_array(lib.t41o_sin_table(), shape=(513,))
    want = np.sin(2 * np.pi * np.arange(513) / 512.0).astype(np.float32)
    want[[0, 256, 512]] = 0.0
    assert np.array_equal(tab, want)
    xs = np.concatenate([np.linspace(-20.0, 20.0, 4001), [0.0, 6.2831855, 1e-8, -1e-8]]).astype(np.float32)
    s = np.array([lib.t41o_arm_sin_f32(float(x)) for x in xs])
    c = np.array([lib.t41o_arm_cos_f32(float(x)) for x in xs])
    assert np.abs(s - np.sin(xs.astype(np.float64))).max() < 2.5e-5
    assert np.abs(c - np.cos(xs.astype(np.float64))).max() < 2.5e-5
    # and against the float64 twin of the method used by the stream model
    s64 = np.array([M.fast_sin(float(x)) for x in xs])
    assert np.abs(s - s64).max() < 2e-6


def test_pll_table_index_short_form():
    """rx_kernels.hip: sam_table_index_pos() -- arm_sin_f32's index arithmetic for 0 <= x < 2, where the kernel's PLL keeps
    its arguments, in 5 instructions (x - floor(x), times 512, truncate) -- against the form as CMSIS writes it, on EVERY
    float32 of [0, 1.26] (what phase * 0.159154943092f and + 0.25f can give for a phase in [0, 2 pi]) and a sample up to 2"""
    def as_written(x):
        n = x.astype(np.int32)            # (int32_t) in: truncation
        n = np.where(x < 0, n - 1, n)
        fr = x - n.astype(np.float32)
        findex = np.float32(512.0) * fr
        idx = findex.astype(np.uint32) & np.uint32(0xffff)
        wrap = idx >= 512
        idx = np.where(wrap, np.uint32(0), idx)
        findex = np.where(wrap, findex - np.float32(512.0), findex)
        return idx, findex - idx.astype(np.float32)

    def short(x):
        findex = np.float32(512.0) * (x - np.floor(x))   # v_fract_f32
        idx = findex.astype(np.uint32)
        return idx, findex - idx.astype(np.float32)

    hi = np.float32(1.26).view(np.uint32)
    step = 1 << 24
    for lo in range(0, int(hi) + 1, step):
        x = np.arange(lo, min(lo + step, int(hi) + 1), dtype=np.uint32).view(np.float32)
        (ia, fa), (ib, fb) = as_written(x), short(x)
        assert np.array_equal(ia, ib) and np.array_equal(fa.view(np.uint32), fb.view(np.uint32)), lo
    x = np.random.default_rng(5).uniform(1.26, 2.0, 1 << 22).astype(np.float32)
    (ia, fa), (ib, fb) = as_written(x), short(x)
    assert np.array_equal(ia, ib) and np.array_equal(fa, fb)
    # the largest arguments the PLL can form: phase = 2 pi (a tiny negative phase + 2 pi rounds to it)
    top = np.float32(6.2831855) * np.float32(0.159154943092)
    assert top < 2 and top + np.float32(0.25) < 2


def test_pll_constants(built):
    out = (C.c_float * 4)()
    _lib().t41o_sam_constants(out)
    g1 = 1.0 - np.exp(-2.0 * 200.0 * 0.65 / 24000.0)
    g2 = -g1 + 2.0 * (1.0 - np.exp(-200.0 * 0.65 / 24000.0) * np.cos(200.0 / 24000.0 * np.sqrt(1.0 - 0.65 ** 2)))
    want = [-2 * np.pi * 4000 / 24000, 2 * np.pi * 4000 / 24000, g1, g2]
    assert np.allclose(list(out)[:3], want[:3], rtol=2e-5, atol=0)
    # g2 is a difference of nearly equal numbers evaluated through cosf() / float products (Demod.cpp:18): the f32
    # roundings leave ~1e-3 of it open (and make it depend on the C library's cosf -- one more unpinned convention)
    assert abs(out[3] / want[3] - 1.0) < 2e-3
    assert 0.010 < out[2] < 0.011 and 5e-5 < out[3] < 7e-5


LOCKED = 12  # frames after which the loop has forgotten how it was pulled in (its time constant is 0.7 frame)


def test_oracle_sam_matches_f64_stream_model(built):
    """The loop starts on the filter's start-up transient, where the phase detector divides rounding-level
    numbers: the first frames depend on the arithmetic's last bits (a decaying difference, factor ~3 per
    frame), so the two implementations are compared once the loop has locked."""
    nfr, nco = 16, 7350
    for seed in (5, 7):
        I, Q = siggen.make_am_carrier(1, nfr * L, [nco], seed=seed)
        ob = O.OracleBatch(O.default_params(**KW), [nco])
        out = ob.process(I, Q)[0]
        ref = M.run(I[0], Q[0], nco, O.coeff_arrays(ob.c, 512), **KW)
        err = siggen.block_rel_err(out[None], ref[None], L)
        assert err[:, LOCKED:].max() < 1e-5, err
        assert np.isfinite(out).all()


def test_oracle_sam_locks_and_recovers_the_modulation(built):
    """the detector does what its name says on a carrier it can lock to: after the pull-in the audio is
    (I + Q of the de-rotated signal =) the envelope's modulation at its frequency"""
    nfr, nco = 24, -1250
    rng = np.random.default_rng(3)
    n = np.arange(nfr * L)
    fm, depth, off = 1000.0, 0.5, 40.0
    fc = siggen.passband_tone_hz(2, nco, off)
    x = 0.3 * (1 + depth * np.sin(2 * np.pi * fm / 192000 * n)) * np.exp(1j * (2 * np.pi * fc / 192000 * n + 0.3))
    ob = O.OracleBatch(O.default_params(**KW), [nco])
    out = ob.process(x.real.astype(np.float32)[None], x.imag.astype(np.float32)[None])[0]
    tail = out[-8 * L:]
    spec = np.abs(np.fft.rfft(tail * np.hanning(tail.size)))
    f = np.fft.rfftfreq(tail.size, 1 / 192000.0)
    peak = f[np.argmax(spec[8:]) + 8]
    assert abs(peak - fm) < 30.0, peak


# ---- the HIP path ---------------------------------------------------------------------------
def _gpu_run(T, kw, nco, I, Q, splits):
    import torch
    rx = T.RxChain(len(nco), T.default_params(**kw), NCOFreq=nco)
    x, y = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    outs, pos = [], 0
    for n in splits:
        outs.append(rx.ProcessIQData(x[:, pos * L:(pos + n) * L].contiguous(), y[:, pos * L:(pos + n) * L].contiguous()))
        pos += n
    return torch.cat(outs, dim=1).cpu().numpy(), rx


@pytest.mark.gpu
@pytest.mark.parametrize("agc", [0, 2], ids=["agc-off", "agc-slow"])
def test_gpu_sam_parity(built, agc):
    """HIP path vs oracle once the loops have locked (see test_oracle_sam_matches_f64_stream_model for why the
    pull-in is not compared sample by sample), 6 channels = one full and one ragged workgroup, several calls"""
    import t41_sdr_amd as T
    nch, nfr = 6, 18
    nco = siggen.nco_grid(nch, seed=21)
    I, Q = siggen.make_am_carrier(nch, nfr * L, nco, seed=40 + agc)
    kw = dict(KW, AGCMode=agc)
    got, _ = _gpu_run(T, kw, nco, I, Q, [1, 5, 12])
    ref = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32)).process(I, Q)
    err = siggen.block_rel_err(got, ref, L)
    assert np.isfinite(got).all()
    assert err[:, LOCKED:].max() <= 1e-5, err
    assert np.abs(ref[:, LOCKED * L:]).max() > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("agc", [0, 2], ids=["agc-off", "agc-slow"])
def test_gpu_sam_pull_in_bound(built, agc):
    """The frames BEFORE lock, under a stated bound instead of being skipped.  The loop starts on the
    filter's start-up transient (the phase detector divides rounding-level numbers and jumps by 2 pi where
    pi / 2 is meant), so for about three frames two implementations follow different trajectories -- the
    difference is of the order of the signal -- and then converge geometrically (measured on MI355X, 64
    channels, difference normalised to the channel's locked audio level: 2.6, 1.1, 0.78, 0.09, 0.046, 6e-3,
    3e-3, 4e-4, 2e-4, 3e-5, 1.3e-5, then the rounding floor ~1.3e-6; tools/sam_pullin_probe.py).  Bar:
    frame f <= 20 x 2.5^-f of the locked level (a factor ~8 above the measured envelope), every frame
    finite and no larger than the detector's own output range, and 1e-5 from frame 12 on."""
    import t41_sdr_amd as T
    nch, nfr = 64, 16
    nco = siggen.nco_grid(nch, seed=21)
    I, Q = siggen.make_am_carrier(nch, nfr * L, nco, seed=40 + agc)
    kw = dict(KW, AGCMode=agc)
    got, _ = _gpu_run(T, kw, nco, I, Q, [nfr])
    ref = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32)).process(I, Q, nthreads=8)
    assert np.isfinite(got).all()
    lvl = np.abs(ref[:, LOCKED * L:]).max(axis=1, keepdims=True)
    d = np.abs(got.astype(np.float64) - ref).reshape(nch, nfr, L).max(axis=2) / lvl
    worst = d.max(axis=0)
    for f in range(LOCKED):
        assert worst[f] <= max(1e-5, 20.0 * 2.5 ** -f), (f, worst)
    assert worst[LOCKED:].max() <= 1e-5, worst
    # both are outputs of the same bounded detector: neither runs away during the pull-in
    assert np.abs(got[:, :LOCKED * L]).max() <= 6.0 * max(np.abs(ref[:, :LOCKED * L]).max(), lvl.max())


def _pull_in_envelope(a, b, ref_locked, nch, nfr):
    lvl = np.abs(ref_locked[:, LOCKED * L:]).max(axis=1, keepdims=True)
    return (np.abs(a.astype(np.float64) - b).reshape(nch, nfr, L).max(axis=2) / lvl).max(axis=0)


def _perturbed(I, Q, eps, seed):
    rng = np.random.default_rng(seed)
    return ((I * (1 + eps * rng.standard_normal(I.shape))).astype(np.float32),
            (Q * (1 + eps * rng.standard_normal(Q.shape))).astype(np.float32))


def test_oracle_sam_pull_in_is_ill_conditioned_in_any_arithmetic(built):
    """The evidence behind "compared once locked" (VERDICT r04 weak #2): the ORACLE against ITSELF on input perturbed by
    1e-7 (one f32 rounding) follows another trajectory for the first frames -- the loop starts on the filters' start-up
    transient, the phase detector divides rounding-level numbers and jumps by 2 pi where pi / 2 is meant -- and converges
    geometrically: difference / locked level 1.9, 0.9, 0.6, 0.08, 0.04, 6e-3, 3e-3, 4e-4, 2e-4, 3e-5, 1e-5, then the
    rounding floor.  That is the envelope the HIP path shows against the oracle (test_gpu_sam_pull_in_bound), so the pull-in
    says nothing about an implementation; from frame 12 on two evaluations agree to 1e-5."""
    nch, nfr = 24, 16
    nco = siggen.nco_grid(nch, seed=21)
    I, Q = siggen.make_am_carrier(nch, nfr * L, nco, seed=40)
    ref = O.OracleBatch(O.default_params(**KW), np.asarray(nco, np.int32)).process(I, Q, nthreads=8)
    Ip, Qp = _perturbed(I, Q, 1e-7, 1)
    per = O.OracleBatch(O.default_params(**KW), np.asarray(nco, np.int32)).process(Ip, Qp, nthreads=8)
    env = _pull_in_envelope(per, ref, ref, nch, nfr)
    assert env[0] > 0.3 and env[2] > 0.05          # order of the signal itself while the trajectories differ
    assert env[LOCKED:].max() <= 1e-5, env         # and the same answer once locked
    assert all(env[f] <= 20.0 * 2.5 ** -f or env[f] <= 1e-5 for f in range(LOCKED)), env   # the bound the GPU test uses


@pytest.mark.gpu
def test_gpu_sam_pull_in_is_within_the_oracles_own_envelope(built):
    """... and the comparison itself: frame by frame, the HIP path is no farther from the oracle during the pull-in than the
    oracle is from its own 1e-7-perturbed selves (3 x the largest of four perturbations, as for the notch from power-on)"""
    import t41_sdr_amd as T
    nch, nfr = 64, 16
    nco = siggen.nco_grid(nch, seed=21)
    I, Q = siggen.make_am_carrier(nch, nfr * L, nco, seed=40)
    got, _ = _gpu_run(T, KW, nco, I, Q, [nfr])
    ob = lambda a, b: O.OracleBatch(O.default_params(**KW), np.asarray(nco, np.int32)).process(a, b, nthreads=8)  # noqa: E731
    ref = ob(I, Q)
    own = np.max([_pull_in_envelope(ob(*_perturbed(I, Q, 1e-7, sd)), ref, ref, nch, nfr) for sd in (1, 2, 3, 4)], axis=0)
    hip = _pull_in_envelope(got, ref, ref, nch, nfr)
    msg = "pull-in, difference / locked level per frame -- HIP vs oracle: %s | oracle vs its perturbed selves: %s" % (
        np.array2string(hip, precision=1), np.array2string(own, precision=1))
    print(msg)
    assert all(hip[f] <= max(3.0 * own[f], 1e-5) for f in range(nfr)), msg


@pytest.mark.gpu
def test_gpu_sam_split_and_state(built):
    """frames in one call or in several: bit-identical audio and PLL state; reset returns to power-on"""
    import t41_sdr_amd as T
    nch, nfr = 9, 8
    nco = siggen.nco_grid(nch, seed=22)
    I, Q = siggen.make_am_carrier(nch, nfr * L, nco, seed=77)
    a, rxa = _gpu_run(T, KW, nco, I, Q, [8])          # one call: the pipelined kernel (sam_chain_pipe, four frames or more)
    b, rxb = _gpu_run(T, KW, nco, I, Q, [3, 1, 4])    # barrier form, barrier form, pipelined
    assert np.array_equal(a, b)
    assert np.array_equal(rxa.get_state(), rxb.get_state())  # (bytes: the oscillator's phase word is not a float)
    for n in (1, 21, 37):  # ragged workgroups of the 16-channel pipelined kernel against 3-frame calls (barrier form only)
        nc = siggen.nco_grid(n, seed=30 + n)
        Ii, Qi = siggen.make_am_carrier(n, 12 * L, nc, seed=70 + n)
        w, rxw = _gpu_run(T, KW, nc, Ii, Qi, [12])
        p, rxp = _gpu_run(T, KW, nc, Ii, Qi, [3, 3, 3, 3])
        assert np.array_equal(w, p) and np.array_equal(rxw.get_state(), rxp.get_state()), n
    rec = rxa.state_records()
    assert np.abs(rec[:, 184 + 13:184 + 16]).max() > 0  # phzerror / fil_out / omega2 live in the record
    rxa.reset()
    c = rxa.ProcessIQData(__import__("torch").from_numpy(I).cuda(), __import__("torch").from_numpy(Q).cuda()).cpu().numpy()
    assert np.array_equal(a, c)


@pytest.mark.gpu
def test_gpu_sam_refusals(built):
    import torch
    import t41_sdr_amd as T
    from t41_sdr_amd import _lib
    with pytest.raises(T.T41RxError) as e:
        T.RxChain(2, T.default_params(**dict(KW, fft_length=1024)))
    assert e.value.status == _lib.ERR_ARG


@pytest.mark.gpu
@pytest.mark.parametrize("agc", [0, 1], ids=["agc-off", "agc-long"])
def test_gpu_sam_q15(built, agc):
    """the firmware only ever feeds q15 samples (Process.cpp:107-108, 936): the synchronous detector on the q15 entry
    points = bit for bit its f32 entry point on the converted samples, then arm_float_to_q15; against the oracle's
    q15 path once locked"""
    import torch
    import t41_sdr_amd as T
    nch, nfr = 6, 18
    kw = dict(KW, AGCMode=agc, audioVolume=60)
    nco = siggen.nco_grid(nch, seed=23)
    I, Q = siggen.make_am_carrier(nch, nfr * L, nco, seed=91)
    qI = np.clip(np.round(I * 32768.0), -32768, 32767).astype(np.int16)
    qQ = np.clip(np.round(Q * 32768.0), -32768, 32767).astype(np.int16)
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    got = rx.ProcessIQData_q15(torch.from_numpy(qQ).cuda(), torch.from_numpy(qI).cuda()).cpu().numpy()  # L queue = Q, R queue = I
    rx.reset()
    f = rx.ProcessIQData(torch.from_numpy(qI.astype(np.float32) / np.float32(32768)).cuda(),
                         torch.from_numpy(qQ.astype(np.float32) / np.float32(32768)).cuda()).cpu().numpy()
    want = np.clip(np.trunc(f.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)
    assert np.array_equal(got, want) and np.abs(got[:, LOCKED * L:]).max() > 300
    ref = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32)).process_q15(qQ, qI)
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))[:, LOCKED * L:]
    assert d.max() <= 1 + np.ceil(1e-5 * np.abs(ref.astype(np.int32)).max()), d.max()


@pytest.mark.gpu
def test_gpu_sam_side_outputs(built):
    """the tap kernels exist for the mode too: the audio spectrum / S-meter by-product (Process.cpp:550-570,
    taken in front of the detector, so comparable from the first frame on) and the post-decimation tap"""
    import torch
    import t41_sdr_amd as T
    nch, nfr = 5, 2
    nco = siggen.nco_grid(nch, seed=31)
    I, Q = siggen.make_am_carrier(nch, nfr * L, nco, seed=9)
    rx = T.RxChain(nch, T.default_params(**KW), NCOFreq=nco)
    sp = torch.zeros(nch, nfr, 1024, device="cuda")
    mx = torch.zeros(nch, nfr, 3, device="cuda")
    dec = torch.zeros(nch, nfr * 512, device="cuda")
    rx.set_audio_spectrum(sp, mx)
    rx.set_debug_taps(None, dec, None)
    rx.ProcessIQData(torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda())
    sp, dec = sp.cpu().numpy(), dec.cpu().numpy()
    ob = O.OracleBatch(O.default_params(**KW), np.asarray(nco, np.int32))
    for f in range(nfr):
        ob.process(np.ascontiguousarray(I[:, f * L:(f + 1) * L]), np.ascontiguousarray(Q[:, f * L:(f + 1) * L]))
        for c in range(nch):
            rs = ob.tap(c, O.TAP_AUDIO_SPECT, 1024)
            assert np.abs(sp[c, f] - rs).max() <= 3e-5 * rs.max()
            ri, rq = ob.tap(c, O.TAP_DEC_I, 256), ob.tap(c, O.TAP_DEC_Q, 256)
            g = dec[c, f * 512:(f + 1) * 512]
            assert max(np.abs(g[:256] - ri).max(), np.abs(g[256:] - rq).max()) <= 1e-5 * max(np.abs(ri).max(), np.abs(rq).max())


def test_designer_carries_the_pll_constants_bit_for_bit(built):
    """host-side designer (no GPU needed): the blob's SAM scalars are the oracle's constants exactly, and the
    mode is refused at the long FFT lengths"""
    import t41_sdr_amd as T
    from t41_sdr_amd import _lib as tlib
    want = (C.c_float * 4)()
    _lib().t41o_sam_constants(want)
    s = T.blob_fields(T.design_coeffs(T.default_params(**KW)), 512)["scalars"]
    assert np.array_equal(np.asarray(s[10:14], np.float32), np.asarray(list(want), np.float32))
    assert s[7] == 1.0  # IQ correction applies in SAM (Process.cpp:165-169)
    with pytest.raises(T.T41RxError) as e:
        T.design_coeffs(T.default_params(**dict(KW, fft_length=1024)))
    assert e.value.status == tlib.ERR_ARG
    p = T.blob_params(T.design_coeffs(T.default_params(**KW)))
    assert p.mode == T.DEMOD_SAM == 8


@pytest.mark.gpu
def test_gpu_switch_to_sam_mid_stream(built):
    """SetupMode() from USB to SAM between two calls: another kernel geometry takes over the same channel records
    (delay lines, oscillator, overlap block); the PLL starts from rest and is compared once locked"""
    import torch
    import t41_sdr_amd as T
    nch, n0, n1 = 5, 2, 14
    nco = siggen.nco_grid(nch, seed=41)
    I, Q = siggen.make_am_carrier(nch, (n0 + n1) * L, nco, seed=11)
    rx = T.RxChain(nch, T.default_params(mode=0, FLoCut=200, FHiCut=3000), NCOFreq=nco)
    x, y = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    a = rx.ProcessIQData(x[:, :n0 * L].contiguous(), y[:, :n0 * L].contiguous()).cpu().numpy()
    rx.SetupMode(**KW)
    b = rx.ProcessIQData(x[:, n0 * L:].contiguous(), y[:, n0 * L:].contiguous()).cpu().numpy()
    ob = O.OracleBatch(O.default_params(mode=0, FLoCut=200, FHiCut=3000), np.asarray(nco, np.int32))
    ra = ob.process(np.ascontiguousarray(I[:, :n0 * L]), np.ascontiguousarray(Q[:, :n0 * L]))
    for k, v in KW.items():
        setattr(ob.p, k, v)
    ob.redesign()
    rb = ob.process(np.ascontiguousarray(I[:, n0 * L:]), np.ascontiguousarray(Q[:, n0 * L:]))
    assert siggen.block_rel_err(a, ra, L).max() <= 1e-5
    assert siggen.block_rel_err(b, rb, L)[:, LOCKED:].max() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("agc", [1, 2, 3, 4], ids=["agc-long", "agc-slow", "agc-med", "agc-fast"])
def test_gpu_sam_with_agc_pipelined_equals_barrier_form(built, agc):
    """Round 4: the synchronous detector behind the AGC (the firmware's default AGCMode = 1,
    gwv.cpp:15; Demod.cpp:40-139 behind DSP_Fn.cpp:504-631) on the pipelined kernel -- AGC chain and PLL each on a duty
    wave of their own, five frames in flight -- for calls of four frames or more; shorter calls run the barrier form.
    Same samples, same PLL / AGC state, bit for bit, in every AGC mode, on full and ragged workgroups and across
    mixed call lengths; against the oracle once locked."""
    import t41_sdr_amd as T
    kw = dict(KW, AGCMode=agc)
    for nch, nfr in ((16, 14), (21, 12), (1, 9), (37, 10)):
        nco = siggen.nco_grid(nch, seed=60 + nch)
        I, Q = siggen.make_am_carrier(nch, nfr * L, nco, seed=300 + 10 * agc + nch)
        I, Q = siggen.fade(I, Q, [(0.3, 1.5), (0.3, 0.08), (0.4, 1.2)])  # so that the gain law walks through its states
        whole, rx_w = _gpu_run(T, kw, nco, I, Q, [nfr])                                  # one call: pipelined
        short, rx_s = _gpu_run(T, kw, nco, I, Q, [3] * (nfr // 3) + ([nfr % 3] if nfr % 3 else []))  # barrier form only
        mixed, rx_m = _gpu_run(T, kw, nco, I, Q, [4, 1, nfr - 5])                      # pipelined, barrier, pipelined
        assert np.isfinite(whole).all()
        assert np.array_equal(whole, short) and np.array_equal(whole, mixed), (agc, nch)
        assert np.array_equal(rx_w.get_state(), rx_s.get_state()) and np.array_equal(rx_w.get_state(), rx_m.get_state()), (agc, nch)
    nch, nfr = 24, 20
    nco = siggen.nco_grid(nch, seed=61)
    I, Q = siggen.make_am_carrier(nch, nfr * L, nco, seed=340 + agc)
    got, _ = _gpu_run(T, kw, nco, I, Q, [nfr])
    ref = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32)).process(I, Q, nthreads=8)
    err = siggen.block_rel_err(got, ref, L)
    assert err[:, LOCKED:].max() <= 1e-5, err[:, LOCKED:].max()

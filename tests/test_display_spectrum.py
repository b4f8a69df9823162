"""The display FFT side output: CalcZoom1Magn() and ZoomFFTExe() (FFT.cpp:67-251) up to FFT_spec /
FFT_spec_old.  CPU: the oracle's restatement against an independent float64 numpy / scipy model;
GPU: the HIP path against the oracle, every zoom level, several consecutive frames (zoom filter
memories, ring pointer and the low-pass memory carry over)."""
import numpy as np
import pytest

import oracle_lib as O
import siggen

L, R = 2048, 512
# Tolerance per zoom level, relative to the spectrum's maximum.  Zoom 0 / 2x: the f32 FFT's 2e-5.
# From 4x on the zoom IIR's poles sit at radius > 0.98 and its f32 DF1 recursion amplifies any
# 1-ulp difference of its INPUT (another but equally valid f32 evaluation order upstream: the
# DC high-pass as a parallel scan) into 1e-5 .. 5e-4 of the output -- the oracle itself moves by
# that much (test_oracle_zoom_conditioning); the bar is 5 x that sensitivity.
ZOOM_TOL = {0: 2e-5, 1: 2e-5, 2: 1e-4, 3: 1e-3, 4: 3e-3}


def model(pre_i, pre_q, zoom, nframes, coeffs=None, fir=None):
    """float64 model working on the pre-shift I/Q (what CalcZoom1Magn sees); returns FFT_spec and
    FFT_spec_old of every frame"""
    from scipy.signal import lfilter
    win = 0.5 - 0.5 * np.cos(6.28 * np.arange(R) / R)
    old = np.zeros(R)
    out_s, out_o = [], []
    ring = np.zeros(R, complex)
    ptr = 0
    zi = [None] * 4
    hist = np.zeros(3, complex)
    jn = np.array([1, 1j, -1, -1j])
    for f in range(nframes):
        x = pre_i[f * L:(f + 1) * L].astype(np.float64) + 1j * pre_q[f * L:(f + 1) * L].astype(np.float64)
        if zoom == 0:
            blk = x[:R] * win
        else:
            y = x * jn[np.arange(L) & 3]  # FreqShift1
            for s in range(4):
                b = coeffs[5 * s:5 * s + 3].astype(np.float64)
                a = np.array([1.0, -coeffs[5 * s + 3], -coeffs[5 * s + 4]], np.float64)
                if zi[s] is None:
                    zi[s] = np.zeros(2, complex)
                y, zi[s] = lfilter(b, a, y, zi=zi[s])
            M = 1 << zoom
            ext = np.concatenate([hist, y])
            dec = np.array([np.dot(fir.astype(np.float64), ext[k * M:k * M + 4]) for k in range(L // M)])
            hist = y[-3:]
            n = min(L // M, R)
            for k in range(n):
                ring[(ptr + k) % R] = dec[k]
            ptr = (ptr + n) % R
            mult = float(1 << zoom) if zoom > 3 else float(zoom)
            blk = mult * ring[(ptr + np.arange(R)) % R] * win
        X = np.fft.fft(blk)
        spec = np.roll(np.abs(X) ** 2, R // 2)
        if zoom == 0:
            old = 0.7 * spec + (1.0 - np.float64(np.float32(0.7))) * old
            out_s.append(spec)
        else:
            old = np.float64(np.float32(0.7)) * spec + np.float64(np.float32(1.0 - np.float64(np.float32(0.7)))) * old
            out_s.append(old.copy())
        out_o.append(old.copy())
    return np.array(out_s), np.array(out_o)


def pre_shift_iq(I, Q, kw):
    """the display FFT's input, from the reference's order of operations in float64: RF gain, DC
    high-pass (one instance, I then Q per frame), band gain, IQ correction (USB: I <- -I)"""
    from scipy.signal import lfilter
    b = [0.927176191943378969, -0.927176191943378969]
    a = [1.0, -0.854352383886757938]
    nfr = I.shape[-1] // L
    g = float(np.float32(10.0 ** (np.float32(kw.get("rfGainAllBands", 1)) / 20.0))) * float(kw.get("RFgain", 1))  # Process.cpp:117, 133
    I, Q = I.astype(np.float64) * g, Q.astype(np.float64) * g
    oi, oq = np.empty(I.shape, np.float64), np.empty(Q.shape, np.float64)
    z = np.zeros(1)
    for f in range(nfr):
        sl = slice(f * L, (f + 1) * L)
        oi[sl], z = lfilter(b, a, I[sl].astype(np.float64), zi=z)
        oq[sl], z = lfilter(b, a, Q[sl].astype(np.float64), zi=z)
    return -oi, oq


@pytest.mark.parametrize("zoom", [0, 1, 3])
def test_oracle_display_matches_a_float64_model(zoom):
    nfr = 4
    nco = np.array([5000], np.int32)
    I, Q = siggen.make_iq(1, nfr * L, nco, seed=77)
    kw = dict()
    ob = O.OracleBatch(O.default_params(**kw), nco)
    ob.set_display(zoom)
    pi, pq = pre_shift_iq(I[0], Q[0], kw)
    coeffs = fir = None
    if zoom:
        import t41_sdr_amd  # noqa: F401  (only for the product's tables below when built; the oracle has its own)
        from test_display_spectrum_tables import MAG_COEFFS
        coeffs = np.array(MAG_COEFFS[zoom - 1], np.float32)
        fir = np.zeros(4, np.float32)
        O.lib().t41o_CalcFIRCoeffs(O.fptr(fir), 4, np.float32(0.5 * 192000 / (1 << zoom)), 60.0, 0, 0.0, 192000.0)
    ms, mo = model(pi, pq, zoom, nfr, coeffs, fir)
    for f in range(nfr):
        ob.process(I[:, f * L:(f + 1) * L], Q[:, f * L:(f + 1) * L])
        s, o = ob.tap(0, O.TAP_FFT_SPEC, R), ob.tap(0, O.TAP_FFT_SPEC_OLD, R)
        assert np.abs(s - ms[f]).max() <= 2e-5 * ms[f].max(), (zoom, f, "FFT_spec")
        assert np.abs(o - mo[f]).max() <= 2e-5 * mo[f].max(), (zoom, f, "FFT_spec_old")
    # the tone that lands in the pass band sits where it should: zoom 0 shows 192 kHz over 512 bins with DC at 256
    if zoom == 0:
        k = int(np.argmax(ms[-1]))
        assert abs((k - 256) * 375.0 - (-(48000.0 - 5000.0))) < 3000.0 or ms[-1].max() > 0


def test_oracle_zoom_conditioning():
    """what ZOOM_TOL rests on: perturbing every input sample by at most one ulp moves the oracle's own
    FFT_spec by far more than 1e-5 at the high zoom levels, and stays within a fifth of the tolerance"""
    nch, nfr = 5, 6
    nco = siggen.nco_grid(nch, seed=51)
    I, Q = siggen.make_iq(nch, nfr * L, nco, seed=52)
    rng = np.random.default_rng(0)
    I2 = (I * (1 + rng.uniform(-6e-8, 6e-8, I.shape))).astype(np.float32)
    Q2 = (Q * (1 + rng.uniform(-6e-8, 6e-8, Q.shape))).astype(np.float32)
    for zoom in (0, 2, 3, 4):
        a, b = O.OracleBatch(O.default_params(), nco), O.OracleBatch(O.default_params(), nco)
        a.set_display(zoom)
        b.set_display(zoom)
        worst = 0.0
        for f in range(nfr):
            sl = slice(f * L, (f + 1) * L)
            a.process(I[:, sl], Q[:, sl])
            b.process(I2[:, sl], Q2[:, sl])
            for ch in range(nch):
                s1, s2 = a.tap(ch, O.TAP_FFT_SPEC, R), b.tap(ch, O.TAP_FFT_SPEC, R)
                worst = max(worst, float(np.abs(s1 - s2).max() / s1.max()))
        assert worst <= ZOOM_TOL[zoom] / 4, (zoom, worst)
        if zoom >= 3:
            assert worst > 2e-5, (zoom, worst)  # i.e. 1e-5 parity is not a property this stage has


@pytest.mark.gpu
@pytest.mark.parametrize("zoom", [0, 1, 2, 3, 4])
def test_gpu_display_spectrum_parity(built, zoom):
    import torch
    import t41_sdr_amd as T
    from t41_sdr_amd import _lib
    nch, nfr = 5, 6
    nco = siggen.nco_grid(nch, seed=51)
    I, Q = siggen.make_iq(nch, nfr * L, nco, seed=52)
    ob = O.OracleBatch(O.default_params(), nco)
    ob.set_display(zoom)
    rx = T.RxChain(nch, T.default_params(), NCOFreq=nco)
    spec = torch.zeros(nch, 3, R, device="cuda")
    old = torch.zeros(nch, 3, R, device="cuda")
    rx.set_display_spectrum(spec, old, zoom)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    # two calls of three frames each (the display state carries over), oracle frame by frame
    for c in range(2):
        sl = slice(3 * c * L, 3 * (c + 1) * L)
        audio = rx.ProcessIQData(dI[:, sl].contiguous(), dQ[:, sl].contiguous()).cpu().numpy()
        gs, go = spec.cpu().numpy(), old.cpu().numpy()
        for f in range(3):
            fs = slice((3 * c + f) * L, (3 * c + f + 1) * L)
            ref = ob.process(I[:, fs], Q[:, fs])
            assert siggen.block_rel_err(audio[:, f * L:(f + 1) * L], ref, L).max() <= 1e-5  # the audio path is untouched
            for ch in range(nch):
                rs, ro = ob.tap(ch, O.TAP_FFT_SPEC, R), ob.tap(ch, O.TAP_FFT_SPEC_OLD, R)
                assert np.abs(gs[ch, f] - rs).max() <= ZOOM_TOL[zoom] * rs.max(), (zoom, c, f, ch, "FFT_spec")
                assert np.abs(go[ch, f] - ro).max() <= ZOOM_TOL[zoom] * ro.max(), (zoom, c, f, ch, "FFT_spec_old")
    with pytest.raises(T.T41RxError) as e:  # more frames than the buffers hold
        rx.ProcessIQData(dI[:, :4 * L].contiguous(), dQ[:, :4 * L].contiguous())
    assert e.value.status == _lib.ERR_ARG
    with pytest.raises(T.T41RxError) as e:
        rx.set_display_spectrum(spec, old, 5)
    assert e.value.status == _lib.ERR_ARG
    rx.set_display_spectrum(None, None)
    rx.ProcessIQData(dI[:, :4 * L].contiguous(), dQ[:, :4 * L].contiguous())


@pytest.mark.gpu
@pytest.mark.parametrize("zoom", [0, 3])
def test_gpu_checkpoint_carries_the_display_memories(built, zoom):
    """round 4: with the display spectrum on, a checkpoint carries the zoom filters' states, the sample ring
    and FFT_spec_old (FFT.cpp:14-26): a restored stream draws the same spectra bit for bit; a checkpoint of another
    spectrumZoom, or one with the section offered to a context whose display is off, is refused."""
    import torch
    import t41_sdr_amd as T
    from t41_sdr_amd import _lib
    nch, nfr = 4, 8
    nco = siggen.nco_grid(nch, seed=61)
    I, Q = siggen.make_iq(nch, nfr * L, nco, seed=62)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    rx = T.RxChain(nch, T.default_params(), NCOFreq=nco)
    spec, old = torch.zeros(nch, nfr, R, device="cuda"), torch.zeros(nch, nfr, R, device="cuda")
    rx.set_display_spectrum(spec, old, zoom)
    rx.ProcessIQData(dI, dQ)
    whole = (spec.clone(), old.clone())
    rx.reset()
    cut = 3
    rx.ProcessIQData(dI[:, :cut * L].contiguous(), dQ[:, :cut * L].contiguous())
    ck = rx.get_state()
    plain = T.RxChain(nch, T.default_params(), NCOFreq=nco)
    assert ck.size == plain.get_state().size + 4 * nch * (64 + 1024 + 512)  # rx_internal.hpp: kDispFloats
    rx.ProcessIQData(dI[:, :2 * L].contiguous(), dQ[:, :2 * L].contiguous())  # disturb
    rx.set_state(ck)
    rx.ProcessIQData(dI[:, cut * L:].contiguous(), dQ[:, cut * L:].contiguous())
    n = nfr - cut  # (a call of n frames fills the buffers as [n_channels][n][512])
    gs, go = spec.view(-1)[:nch * n * R].view(nch, n, R), old.view(-1)[:nch * n * R].view(nch, n, R)
    assert torch.equal(gs, whole[0][:, cut:]) and torch.equal(go, whole[1][:, cut:])
    with pytest.raises(T.T41RxError) as e:  # the display is off there: the section has no home
        plain.set_state(ck)
    assert e.value.status == _lib.ERR_STATE
    rx.set_display_spectrum(spec, old, 1 if zoom != 1 else 2)
    with pytest.raises(T.T41RxError) as e:
        rx.set_state(ck)
    assert e.value.status == _lib.ERR_STATE

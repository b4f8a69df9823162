"""Launch shapes nothing else runs, and boundary behaviour added in round 3 (-m gpu unless marked otherwise).

* the literal INTEGRATION.md shim: ONE channel, ONE frame per call, host pointers, f32 and q15
  (BASELINE configs[0]'s "single block" shape through the product path)
* the bench shape: ONE launch of 4096 channels x 32 frames against a 64-channel oracle sample,
  including the last frame and the state record the launch leaves behind
* set_coeffs() followed by a one-field CalcFilters() keeps the designer's parameters
* checkpoints with an out-of-range synchronous-detector phase are refused; the exciter accepts SAM
"""
import numpy as np
import pytest

import oracle_lib as O
import siggen

L = 2048
TOL = 1e-5


@pytest.fixture(scope="module")
def T(built):
    import torch
    import t41_sdr_amd
    assert torch.cuda.is_available()
    t41_sdr_amd.load()
    return t41_sdr_amd


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(mode=0, FLoCut=200, FHiCut=3000), dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=1),
                                dict(mode=3, FLoCut=200, FHiCut=3000)], ids=["usb", "usb-agc", "nfm"])
def test_single_channel_single_frame_host_calls(T, kw):
    """what INTEGRATION.md section 2 binds: n_channels = 1, one ProcessIQData() per call, caller-owned host arrays"""
    nfr = 10
    nco = np.array([7350], np.int32)
    I, Q = siggen.make_iq(1, nfr * L, nco, mode=kw["mode"], seed=77)
    rx = T.RxChain(1, T.default_params(**kw), NCOFreq=nco)
    got = np.concatenate([rx.ProcessIQData(I[:, f * L:(f + 1) * L].copy(), Q[:, f * L:(f + 1) * L].copy()) for f in range(nfr)], axis=1)
    ref = O.OracleBatch(O.default_params(**kw), nco).process(I, Q)
    err = siggen.block_rel_err(got, ref, L)
    assert np.isfinite(got).all() and err.max() <= TOL, err


@pytest.mark.gpu
def test_single_channel_single_frame_host_calls_q15(T):
    """the q15 queues of one radio, one call per 16 x 128-sample block pair"""
    kw = dict(mode=0, FLoCut=200, FHiCut=3000)
    nfr = 8
    nco = np.array([-12500], np.int32)
    I, Q = siggen.make_iq(1, nfr * L, nco, mode=0, seed=78)
    qi = np.clip(np.round(I * 32768.0), -32768, 32767).astype(np.int16)  # R queue -> float_buffer_L = I (Process.cpp:107)
    qq = np.clip(np.round(Q * 32768.0), -32768, 32767).astype(np.int16)  # L queue -> float_buffer_R = Q
    rx = T.RxChain(1, T.default_params(**kw), NCOFreq=nco)
    got = np.concatenate([rx.ProcessIQData_q15(qq[:, f * L:(f + 1) * L].copy(), qi[:, f * L:(f + 1) * L].copy()) for f in range(nfr)], axis=1)
    ref = O.OracleBatch(O.default_params(**kw), nco).process_q15(qq, qi)
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.02, (d.max(), (d > 0).mean())


@pytest.mark.gpu
def test_bench_shape_one_launch(T):
    """4096 channels x 32 frames in ONE launch (bench.py's step): 64 sampled channels against the oracle over all
    32 frames, and the record the launch leaves behind continues the stream exactly like a second oracle call"""
    import torch
    nch, nfr = 4096, 32
    kw = dict(mode=0, FLoCut=200, FHiCut=3000)
    nco = siggen.nco_grid(nch, seed=11)
    pick = np.unique(np.linspace(0, nch - 1, 64).round().astype(int))
    g = torch.Generator(device="cuda").manual_seed(5)
    dI = (0.25 * torch.randn(nch, (nfr + 2) * L, generator=g, device="cuda")).clamp_(-0.999, 0.999)
    dQ = (0.25 * torch.randn(nch, (nfr + 2) * L, generator=g, device="cuda")).clamp_(-0.999, 0.999)
    # a pass-band tone on the sampled channels so that the audio is not just filtered noise
    hI, hQ = siggen.make_iq(len(pick), (nfr + 2) * L, nco[pick], mode=0, seed=9)
    dI[torch.as_tensor(pick, device="cuda")] = torch.from_numpy(hI).cuda()
    dQ[torch.as_tensor(pick, device="cuda")] = torch.from_numpy(hQ).cuda()
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    a = rx.ProcessIQData(dI[:, :nfr * L].contiguous(), dQ[:, :nfr * L].contiguous())
    b = rx.ProcessIQData(dI[:, nfr * L:].contiguous(), dQ[:, nfr * L:].contiguous())  # continues from the stored record
    torch.cuda.synchronize()
    got = torch.cat([a, b], dim=1)[torch.as_tensor(pick, device="cuda")].cpu().numpy()
    ref = O.OracleBatch(O.default_params(**kw), nco[pick]).process(hI, hQ, nthreads=8)
    err = siggen.block_rel_err(got, ref, L)
    assert np.isfinite(got).all()
    assert err.max() <= TOL, "worst %.3e at %s" % (err.max(), np.unravel_index(err.argmax(), err.shape))
    assert err[:, nfr - 1].max() <= TOL and err[:, nfr:].max() <= TOL  # the launch's last frame, and what follows it
    assert torch.isfinite(a).all() and torch.isfinite(b).all()


@pytest.mark.gpu
def test_set_coeffs_then_one_field_change_keeps_the_designers_parameters(T):
    """the broadcast path: a rank created with default parameters installs rank 0's blob, then changes ONE field"""
    designer = T.default_params(mode=1, FLoCut=-2800, FHiCut=-250, AGCMode=2, rfGainAllBands=4, RFgain=2, audioVolume=41)
    a = T.RxChain(4, designer)
    b = T.RxChain(4, T.default_params())
    b.set_coeffs(a.coeffs())
    for f, _ in designer._fields_:
        assert getattr(b.params, f) == getattr(designer, f), f  # the Python mirror follows the context
    b.CalcFilters(audioVolume=55)
    got = b.get_params()
    for f, _ in designer._fields_:
        want = 55 if f == "audioVolume" else getattr(designer, f)
        assert getattr(got, f) == want, (f, getattr(got, f), want)
    a.CalcFilters(audioVolume=55)
    assert np.array_equal(a.coeffs(), b.coeffs())


@pytest.mark.gpu
def test_checkpoint_with_a_wild_pll_phase_is_refused(T):
    import t41_sdr_amd._lib as lib
    rx = T.RxChain(3, T.default_params(mode=8, FLoCut=-3000, FHiCut=3000))
    buf = rx.get_state()
    rec = rx.state_records(buf)
    rx.set_state(buf)  # as it came: accepted
    for bad in (7.0, -0.5, np.nan):
        rec[1, 184 + 13] = bad  # kStMisc + kMiscSamPhz
        with pytest.raises(T.T41RxError) as e:
            rx.set_state(buf)
        assert e.value.status == lib.ERR_STATE
    rec[1, 184 + 13] = 1.0
    rec[2, 184 + 15] = 3.0  # omega2 beyond omega_max
    with pytest.raises(T.T41RxError):
        rx.set_state(buf)


@pytest.mark.gpu
def test_exciter_accepts_every_receive_mode(T):
    """ExciterIQData() runs whatever bands[].mode is and corrects IQ only in LSB / USB (Exciter.cpp:117-140):
    a caller that shares one mode value between its RX and TX contexts must not be refused in SAM"""
    from t41_sdr_amd.tx import TxChain, default_tx_params
    rng = np.random.default_rng(4)
    mic = (8000 * rng.standard_normal((2, 2 * 2048))).astype(np.int16)
    outs = {}
    for mode in (T.DEMOD_AM, T.DEMOD_SAM):
        tx = TxChain(2, default_tx_params(mode=mode, IQXAmpCorrectionFactor=1.03, IQXPhaseCorrectionFactor=0.02))
        outs[mode] = tx.ExciterIQData(mic)
    for a, b in zip(outs[T.DEMOD_AM], outs[T.DEMOD_SAM]):
        assert np.array_equal(a, b)  # no IQ correction in either


def test_nfm_variant_needs_fft_512_at_design_time(built):
    """params_valid refuses what process_device would refuse later: no GPU needed"""
    import t41_sdr_amd as T
    import t41_sdr_amd._lib as lib
    T.design_coeffs(T.default_params(mode=3, nfm_demod=1))
    for n in (1024, 2048, 4096):
        with pytest.raises(T.T41RxError) as e:
            T.design_coeffs(T.default_params(mode=3, nfm_demod=1, fft_length=n))
        assert e.value.status == lib.ERR_ARG
        T.design_coeffs(T.default_params(mode=3, nfm_demod=0, fft_length=n))

"""The q15 sample-format boundary of ProcessIQData() (Process.cpp:102-111, 936-937):
oracle-side conversions on CPU, HIP path vs oracle and vs its own f32 entry point on the GPU."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import siggen

L = 2048


def to_q15(x):
    """what the codec would deliver for a float waveform in (-1, 1)"""
    return np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)


def test_conversions_follow_cmsis(built):
    lib = O.lib()
    src = np.array([-32768, -1, 0, 1, 12345, 32767], dtype=np.int16)
    dst = np.zeros(src.size, dtype=np.float32)
    lib.t41o_q15_to_float(src.ctypes.data_as(C.POINTER(C.c_int16)), O.fptr(dst), src.size)
    assert np.array_equal(dst, src.astype(np.float32) / np.float32(32768.0))
    f = np.array([0.0, 0.99999, 1.0, 1.5, -1.0, -1.00004, -7.0, 3.0517578125e-05, 5.9e-05, -5.9e-05, 1e9, -1e9,
                  0.5 + 1.6e-05], dtype=np.float32)
    q = np.zeros(f.size, dtype=np.int16)
    lib.t41o_float_to_q15(O.fptr(f), q.ctypes.data_as(C.POINTER(C.c_int16)), f.size)
    # truncation toward zero, saturation at both ends
    assert q.tolist() == [0, 32767, 32767, 32767, -32768, -32768, -32768, 1, 1, -1, 32767, -32768, 16384]


def test_oracle_q15_wrapper_is_swap_convert_process_convert(built):
    nco = [5000, -12350]
    I, Q = siggen.make_iq(2, 3 * L, nco, mode=0, seed=77)
    qI, qQ = to_q15(I), to_q15(Q)
    p = O.default_params(audioVolume=100)
    got = O.OracleBatch(p, nco).process_q15(qQ, qI)  # L queue carries Q, R queue carries I (Process.cpp:107-108)
    ref_f = O.OracleBatch(p, nco).process(qI.astype(np.float32) / np.float32(32768), qQ.astype(np.float32) / np.float32(32768))
    ref = np.clip(np.trunc(ref_f.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)
    assert np.array_equal(got, ref)
    assert np.abs(got).max() > 1000  # a real signal, not silence


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(mode=0, FLoCut=200, FHiCut=3000, audioVolume=100),
    dict(mode=1, FLoCut=-3000, FHiCut=-200, audioVolume=100, RFgain=3, IQAmpCorrectionFactor=1.02, IQPhaseCorrectionFactor=-0.013),
    dict(mode=2, FLoCut=-3000, FHiCut=3000, audioVolume=100),
    dict(mode=3, FLoCut=200, FHiCut=3000, audioVolume=100),
    dict(mode=0, FLoCut=200, FHiCut=3000, audioVolume=100, AGCMode=1),
    dict(mode=0, FLoCut=200, FHiCut=3000, audioVolume=100, rfGainAllBands=20),  # drives the output into saturation
], ids=["usb", "lsb-gains", "am", "nfm", "usb-agc", "usb-saturating"])
def test_gpu_q15_parity(built, kw):
    import torch
    import t41_sdr_amd as T
    nch, nfr = 10, 5
    nco = siggen.nco_grid(nch, seed=9)
    if kw["mode"] == 3:
        I, Q = siggen.make_fm(nch, nfr * L, nco, seed=4)
    else:
        I, Q = siggen.make_iq(nch, nfr * L, nco, mode=kw["mode"], seed=6)
    qI, qQ = to_q15(I), to_q15(Q)
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    dL, dR = torch.from_numpy(qQ).cuda(), torch.from_numpy(qI).cuda()
    got = rx.ProcessIQData_q15(dL, dR)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    # (1) against the oracle: the float chains differ by <= 1e-5 of the block maximum, so the
    # truncated integers may differ by that much + 1 LSB at a truncation boundary
    ref = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32)).process_q15(qQ, qI)
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    tol = 5e-5 if kw["mode"] == 2 else 1e-5  # AM: see AM_TOL in test_gpu_parity.py
    bound = 1 + np.ceil(tol * np.abs(ref.astype(np.int32)).reshape(nch, nfr, L).max(axis=2, keepdims=True))
    assert (d.reshape(nch, nfr, L) <= bound).all(), (d.max(), bound.max())
    # and almost everywhere they are identical (a float difference of e LSB flips a truncation in a fraction e of the samples)
    assert (d > 0).mean() < (0.2 if kw["mode"] == 2 else 0.02)
    if "rfGainAllBands" in kw:
        assert (np.abs(ref.astype(np.int32)) >= 32767).mean() > 0.01  # the saturating case does saturate
    # (2) against its own f32 entry point on the converted samples: bit for bit the same chain,
    # then arm_float_to_q15
    rx.reset()
    f = rx.ProcessIQData(torch.from_numpy(qI.astype(np.float32) / np.float32(32768)).cuda(),
                         torch.from_numpy(qQ.astype(np.float32) / np.float32(32768)).cuda()).cpu().numpy()
    want = np.clip(np.trunc(f.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)
    assert np.array_equal(got, want)
    # (3) host-pointer form, frame by frame == device form in one call
    rx.reset()
    parts = [rx.ProcessIQData_q15(np.ascontiguousarray(qQ[:, k * L:(k + 1) * L]), np.ascontiguousarray(qI[:, k * L:(k + 1) * L]))
             for k in range(nfr)]
    assert np.array_equal(np.concatenate(parts, axis=1), got)


@pytest.mark.gpu
@pytest.mark.parametrize("N,kw", [(1024, dict(mode=0, audioVolume=100)), (4096, dict(mode=1, FLoCut=-3000, FHiCut=-200, audioVolume=100)),
                                  (2048, dict(mode=2, FLoCut=-3000, FHiCut=3000, audioVolume=100, AGCMode=3)),
                                  (1024, dict(mode=3, audioVolume=100))], ids=["usb-1024", "lsb-4096", "am-agc-2048", "nfm-1024"])
def test_gpu_q15_long_fft(built, N, kw):
    """q15 samples through the three-kernel pipeline of the long FFT lengths: bit-identical to the f32
    entry point on the converted samples, then arm_float_to_q15"""
    import torch
    import t41_sdr_amd as T
    Lf = 4 * N
    nch, nfr = 5, 3
    nco = siggen.nco_grid(nch, seed=N)
    kw = dict(kw, fft_length=N)
    if kw["mode"] == 3:
        I, Q = siggen.make_fm(nch, nfr * Lf, nco, seed=2)
    else:
        I, Q = siggen.make_iq(nch, nfr * Lf, nco, mode=kw["mode"], seed=3)
    qI, qQ = to_q15(I), to_q15(Q)
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    got = rx.ProcessIQData_q15(torch.from_numpy(qQ).cuda(), torch.from_numpy(qI).cuda()).cpu().numpy()
    rx.reset()
    f = rx.ProcessIQData(torch.from_numpy(qI.astype(np.float32) / np.float32(32768)).cuda(),
                         torch.from_numpy(qQ.astype(np.float32) / np.float32(32768)).cuda()).cpu().numpy()
    want = np.clip(np.trunc(f.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)
    assert np.array_equal(got, want) and np.abs(got).max() > 500
    ref = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32)).process_q15(qQ, qI)
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    tol = 5e-5 if kw["mode"] == 2 else 1e-5
    assert d[:, Lf:].max() <= 1 + np.ceil(tol * np.abs(ref.astype(np.int32)).max())


@pytest.mark.gpu
def test_gpu_q15_argument_errors(built):
    import torch
    import t41_sdr_amd as T
    from t41_sdr_amd import _lib
    rx = T.RxChain(4, T.default_params())
    x = torch.zeros(4, L, dtype=torch.int16, device="cuda")
    with pytest.raises(ValueError):
        rx.ProcessIQData_q15(x.float(), x.float())
    # (round 4: the stage taps and side outputs work on the q15 entry points -- test_gpu_q15_side_outputs_and_taps;
    #  what stays refused is a call longer than the tap buffers were sized for)
    tap = torch.zeros(4 * 256, device="cuda")
    rx.set_debug_taps(None, None, tap)
    rx.ProcessIQData_q15(x, x)
    x2 = torch.zeros(4, 2 * L, dtype=torch.int16, device="cuda")
    with pytest.raises(T.T41RxError) as e:
        rx.ProcessIQData_q15(x2, x2)
    assert e.value.status == _lib.ERR_ARG


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(mode=0), dict(mode=0, AGCMode=1), dict(mode=2, FLoCut=-3000, FHiCut=3000), dict(mode=3),
                                dict(mode=8, FLoCut=-3000, FHiCut=3000, AGCMode=2)], ids=["usb", "usb-agc", "am", "nfm", "sam-agc"])
def test_gpu_q15_side_outputs_and_taps(built, kw):
    """the reference computes its display FFT and audio spectrum inside every ProcessIQData() call on
    the q15-fed buffers (Process.cpp:107-108 -> :184-186, :211-215, :550-570).  On the q15 entry points the display
    spectrum (zoom 0 and 2), the audio spectrum / S-meter words and the three stage taps must be bit for bit what the f32
    entry point gives on the converted samples (arm_q15_to_float is exact), over two calls (the side stages' memories
    carry over), and the audio must be that call's audio through arm_float_to_q15."""
    import torch
    import t41_sdr_amd as T
    nch, nfr = 7, 3
    nco = siggen.nco_grid(nch, seed=31)
    I, Q = siggen.make_iq(nch, 2 * nfr * L, nco, mode=min(kw["mode"], 2) if kw["mode"] != 3 else 3, seed=32)
    qI = np.clip(np.round(I * 32768.0), -32768, 32767).astype(np.int16)
    qQ = np.clip(np.round(Q * 32768.0), -32768, 32767).astype(np.int16)
    fI, fQ = qI.astype(np.float32) / np.float32(32768), qQ.astype(np.float32) / np.float32(32768)
    kw = dict(kw, audioVolume=60)
    for zoom in (0, 2):
        outs = []
        for q15 in (False, True):
            rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
            spec, old = torch.zeros(nch, nfr, 512, device="cuda"), torch.zeros(nch, nfr, 512, device="cuda")
            asp, amx = torch.zeros(nch, nfr, 1024, device="cuda"), torch.zeros(nch, nfr, 3, device="cuda")
            t_nco, t_dec, t_dem = (torch.zeros(nch, nfr * 2 * L, device="cuda"), torch.zeros(nch, nfr * 512, device="cuda"),
                                   torch.zeros(nch, nfr * 256, device="cuda"))
            rx.set_display_spectrum(spec, old, zoom)
            rx.set_audio_spectrum(asp, amx)
            rx.set_debug_taps(t_nco, t_dec, t_dem)
            got = []
            for c in range(2):
                sl = slice(c * nfr * L, (c + 1) * nfr * L)
                if q15:
                    a = rx.ProcessIQData_q15(torch.from_numpy(qQ[:, sl].copy()).cuda(), torch.from_numpy(qI[:, sl].copy()).cuda())
                else:
                    a = rx.ProcessIQData(torch.from_numpy(fI[:, sl].copy()).cuda(), torch.from_numpy(fQ[:, sl].copy()).cuda())
                got.append([t.clone() for t in (a, spec, old, asp, amx, t_nco, t_dec, t_dem)])
            outs.append(got)
        for c in range(2):
            f32, q = outs[0][c], outs[1][c]
            want = torch.clamp(torch.trunc(f32[0].double() * 32768.0), -32768, 32767).to(torch.int16)
            assert torch.equal(q[0], want), (zoom, c, "audio")
            for k, name in enumerate(("FFT_spec", "FFT_spec_old", "audioSpectBuffer", "audioMax", "post_nco", "dec", "demod"), start=1):
                assert torch.equal(q[k], f32[k]), (zoom, c, name)
            assert float(q[1].abs().max()) > 0 and float(q[3].abs().max()) > 0

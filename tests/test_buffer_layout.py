"""t41rx_set_buffer_layout (ABI 5): the frames of a call as [n_frames][n_channels][frame_len] -- the buffers of
consecutive single-frame ProcessIQData() calls (float_buffer_L/R[2048] per channel, T41_SDR.ino:375-376) stacked as
they arrive -- instead of [n_channels][n_frames * frame_len].  Same arithmetic on the same samples: outputs and
checkpoints must match the channel-major call bit for bit, in every mode, sample format and kernel form; and against
the oracle like any other call.
"""
import numpy as np
import pytest

import oracle_lib as O
import siggen

L = 2048


@pytest.fixture(scope="module")
def T(built):
    import torch
    import t41_sdr_amd
    assert torch.cuda.is_available()
    t41_sdr_amd.load()
    return t41_sdr_amd


def test_layout_symbols_and_constants(built):
    """CPU: the header's two constants and both entry points are there (no device needed to bind them)"""
    import re
    import os
    import t41_sdr_amd
    lib = t41_sdr_amd.load()
    assert hasattr(lib, "t41rx_set_buffer_layout") and hasattr(lib, "t41rx_get_buffer_layout")
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "t41rx.h")).read()
    assert re.search(r"#define T41RX_LAYOUT_CHANNEL_MAJOR 0", hdr) and re.search(r"#define T41RX_LAYOUT_TIME_MAJOR 1", hdr)
    assert lib.t41rx_set_buffer_layout(None, 1) == -1 and lib.t41rx_get_buffer_layout(None) == -1  # T41RX_ERR_ARG


CASES = [
    ("usb", dict(mode=0, FLoCut=200, FHiCut=3000), 5),
    ("usb-gains", dict(mode=0, FLoCut=200, FHiCut=3000, rfGainAllBands=3, RFgain=2, IQAmpCorrectionFactor=1.02, IQPhaseCorrectionFactor=0.01), 3),
    ("lsb", dict(mode=1, FLoCut=-3000, FHiCut=-200), 3),
    ("am", dict(mode=2, FLoCut=-4000, FHiCut=4000), 3),
    ("nfm", dict(mode=3, FLoCut=200, FHiCut=3000), 3),
    ("sam", dict(mode=8, FLoCut=-4000, FHiCut=4000), 4),
    ("usb-agc-short", dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=1), 3),   # barrier form (fewer than 4 frames)
    ("usb-agc-pipe", dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=2), 6),    # pipelined form
    ("am-agc-pipe", dict(mode=2, FLoCut=-4000, FHiCut=4000, AGCMode=3), 5),
    ("sam-pipe", dict(mode=8, FLoCut=-4000, FHiCut=4000), 6),
    ("notch", dict(mode=0, FLoCut=200, FHiCut=3000, ANR_notchOn=1), 3),
    # the kernel forms that compute their I / Q / out addresses from chan_stride / frame_stride and were not compared
    ("sam-agc-pipe", dict(mode=8, FLoCut=-4000, FHiCut=4000, AGCMode=1), 7),  # SAM behind the AGC: the PSA pipeline (kSkew 4)
    ("kim", dict(mode=0, FLoCut=200, FHiCut=3000, nrOptionSelect=1), 5),         # launch_back512 behind the stage kernels
    ("spectral", dict(mode=0, FLoCut=200, FHiCut=3000, nrOptionSelect=2), 12),   # (its first 19 half-blocks pass the audio through)
    ("spectral-agc", dict(mode=0, FLoCut=200, FHiCut=3000, nrOptionSelect=2, AGCMode=2), 12),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,nfr", CASES, ids=[c[0] for c in CASES])
def test_time_major_equals_channel_major(T, name, kw, nfr):
    import torch
    nch = 37  # ragged last workgroup
    nco = siggen.nco_grid(nch, seed=3)
    I, Q = siggen.make_iq(nch, 2 * nfr * L, nco, mode=min(kw["mode"], 3) if kw["mode"] != 8 else 2, seed=21)
    a = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    b = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    b.set_buffer_layout("time")
    for call in range(2):  # the second call continues from the state the first left
        sl = slice(call * nfr * L, (call + 1) * nfr * L)
        dI, dQ = torch.from_numpy(I[:, sl].copy()).cuda(), torch.from_numpy(Q[:, sl].copy()).cuda()
        oa = a.ProcessIQData(dI, dQ)
        tI = dI.view(nch, nfr, L).transpose(0, 1).contiguous()
        tQ = dQ.view(nch, nfr, L).transpose(0, 1).contiguous()
        ob = b.ProcessIQData(tI, tQ)
        assert tuple(ob.shape) == (nfr, nch, L)
        assert torch.equal(oa.view(nch, nfr, L).transpose(0, 1), ob), (name, call)
    assert np.array_equal(a.get_state(), b.get_state())


@pytest.mark.gpu
def test_time_major_q15_and_host_entry_points(T):
    import torch
    kw = dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=1)
    nch, nfr = 19, 5
    nco = siggen.nco_grid(nch, seed=4)
    I, Q = siggen.make_iq(nch, nfr * L, nco, mode=0, seed=22)
    qi = np.clip(np.round(I * 32768.0), -32768, 32767).astype(np.int16)
    qq = np.clip(np.round(Q * 32768.0), -32768, 32767).astype(np.int16)
    a = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    b = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    b.set_buffer_layout("time")
    oa = a.ProcessIQData_q15(torch.from_numpy(qq).cuda(), torch.from_numpy(qi).cuda())
    tq = torch.from_numpy(qq).cuda().view(nch, nfr, L).transpose(0, 1).contiguous()
    ti = torch.from_numpy(qi).cuda().view(nch, nfr, L).transpose(0, 1).contiguous()
    ob = b.ProcessIQData_q15(tq, ti)
    assert torch.equal(oa.view(nch, nfr, L).transpose(0, 1), ob)
    # host pointers, f32
    c = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    d = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    d.set_buffer_layout("time")
    oc = c.ProcessIQData(I, Q)
    od = d.ProcessIQData(np.ascontiguousarray(I.reshape(nch, nfr, L).transpose(1, 0, 2)), np.ascontiguousarray(Q.reshape(nch, nfr, L).transpose(1, 0, 2)))
    assert np.array_equal(oc.reshape(nch, nfr, L).transpose(1, 0, 2), od)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["f32", "q15"])
def test_time_major_side_outputs_and_taps(T, fmt):
    """the <DEBUG, WQ15> instantiations and the side-output kernels under the time-major layout -- the audio
    spectrum / S-meter words, the display FFT and the stage taps keep their [n_channels][n_frames][...] shapes whatever
    the call buffers' layout, and hold the same bits"""
    import torch
    kw = dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=1)
    nch, nfr = 21, 5
    nco = siggen.nco_grid(nch, seed=6)
    I, Q = siggen.make_iq(nch, nfr * L, nco, mode=0, seed=24)
    if fmt == "q15":
        I = np.clip(np.round(I * 32768.0), -32768, 32767).astype(np.int16)
        Q = np.clip(np.round(Q * 32768.0), -32768, 32767).astype(np.int16)
    outs = {}
    for layout in ("channel", "time"):
        rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
        rx.set_buffer_layout(layout)
        spect, mx = torch.zeros(nch, nfr, 1024, device="cuda"), torch.zeros(nch, nfr, 3, device="cuda")
        spec, old = torch.zeros(nch, nfr, 512, device="cuda"), torch.zeros(nch, nfr, 512, device="cuda")
        dec, dem = torch.zeros(nch, nfr * 512, device="cuda"), torch.zeros(nch, nfr * 256, device="cuda")
        rx.set_audio_spectrum(spect, mx)
        rx.set_display_spectrum(spec, old, spectrumZoom=2)
        rx.set_debug_taps(dec=dec, demod=dem)
        dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
        if layout == "time":
            dI = dI.view(nch, nfr, L).transpose(0, 1).contiguous()
            dQ = dQ.view(nch, nfr, L).transpose(0, 1).contiguous()
        o = rx.ProcessIQData_q15(dQ, dI) if fmt == "q15" else rx.ProcessIQData(dI, dQ)
        if layout == "time":
            o = o.transpose(0, 1).reshape(nch, nfr * L)
        torch.cuda.synchronize()
        outs[layout] = [t.cpu().numpy() for t in (o, spect, mx, spec, old, dec, dem)] + [rx.get_state()]
    for a, b in zip(outs["channel"], outs["time"]):
        assert np.array_equal(a, b)
    assert np.abs(outs["time"][1]).max() > 0 and np.abs(outs["time"][3]).max() > 0 and np.abs(outs["time"][6]).max() > 0


@pytest.mark.gpu
def test_time_major_against_the_oracle(T):
    """not only equal to the other layout: the time-major call is checked against the oracle directly"""
    import torch
    kw = dict(mode=0, FLoCut=200, FHiCut=3000)
    nch, nfr = 24, 6
    nco = siggen.nco_grid(nch, seed=5)
    I, Q = siggen.make_iq(nch, nfr * L, nco, mode=0, seed=23)
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    rx.set_buffer_layout("time")
    tI = torch.from_numpy(np.ascontiguousarray(I.reshape(nch, nfr, L).transpose(1, 0, 2))).cuda()
    tQ = torch.from_numpy(np.ascontiguousarray(Q.reshape(nch, nfr, L).transpose(1, 0, 2))).cuda()
    got = rx.ProcessIQData(tI, tQ).cpu().numpy().transpose(1, 0, 2).reshape(nch, nfr * L)
    ref = O.OracleBatch(O.default_params(**kw), nco).process(I, Q)
    err = siggen.block_rel_err(got, ref, L)
    assert err.max() <= 1e-5, err.max()


@pytest.mark.gpu
def test_time_major_refused_for_the_long_fft(T):
    rx = T.RxChain(4, T.default_params(fft_length=4096, FLoCut=400, FHiCut=600))
    with pytest.raises(T.T41RxError) as e:
        rx.set_buffer_layout("time")
    assert e.value.status == -2  # T41RX_ERR_UNSUPPORTED
    assert rx._lib.t41rx_get_buffer_layout(rx._ctx) == 0
    rx.set_buffer_layout("channel")
    assert rx._lib.t41rx_set_buffer_layout(rx._ctx, 7) == -1  # T41RX_ERR_ARG: unknown layout

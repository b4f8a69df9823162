"""The audio-spectrum / S-meter by-product of the path (Process.cpp:550-570):
HIP side output vs the oracle's audioSpectBuffer, audioMaxSquared, AudioMaxIndex, audioMaxSquaredAve."""
import numpy as np
import pytest

import oracle_lib as O
import siggen

L = 2048


def test_oracle_audio_spectrum_is_the_reversed_squared_masked_spectrum(built):
    nco = [5000]
    I, Q = siggen.make_iq(1, 2 * L, nco, mode=0, seed=3)
    ob = O.OracleBatch(O.default_params(), nco)
    ob.process(I, Q)
    sp = ob.tap(0, O.TAP_AUDIO_SPECT, 1024)
    mx = ob.tap(0, O.TAP_AUDIO_MAX, 3)
    assert sp.shape == (1024,) and np.all(sp >= 0)
    assert mx[0] == sp.max() and int(mx[1]) == int(np.argmax(sp))  # arm_max_f32: first occurrence
    # USB 200..3000 Hz at 24 kS/s / 512 bins: the pass band is bins 4..64 -> floats 8..129 -> reversed 894..1015
    assert 890 <= int(mx[1]) <= 1020
    assert sp[:700].max() < 1e-6 * sp.max()


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(mode=0), dict(mode=1, FLoCut=-3000, FHiCut=-200), dict(mode=2, FLoCut=-3000, FHiCut=3000),
                                dict(mode=3), dict(mode=0, AGCMode=2)], ids=["usb", "lsb", "am", "nfm", "usb-agc"])
def test_gpu_audio_spectrum(built, kw):
    import torch
    import t41_sdr_amd as T
    nch, nfr = 6, 4
    nco = siggen.nco_grid(nch, seed=12)
    if kw["mode"] == 3:
        I, Q = siggen.make_fm(nch, nfr * L, nco, seed=5)
    else:
        I, Q = siggen.make_iq(nch, nfr * L, nco, mode=kw["mode"], seed=7)
    rx = T.RxChain(nch, T.default_params(**kw), NCOFreq=nco)
    sp = torch.zeros(nch, nfr, 1024, device="cuda")
    mx = torch.zeros(nch, nfr, 3, device="cuda")
    rx.set_audio_spectrum(sp, mx)
    audio = rx.ProcessIQData(torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()).cpu().numpy()
    sp, mx = sp.cpu().numpy(), mx.cpu().numpy()
    ob = O.OracleBatch(O.default_params(**kw), np.asarray(nco, np.int32))
    for f in range(nfr):
        ref_a = ob.process(np.ascontiguousarray(I[:, f * L:(f + 1) * L]), np.ascontiguousarray(Q[:, f * L:(f + 1) * L]))
        assert siggen.block_rel_err(audio[:, f * L:(f + 1) * L], ref_a, L).max() <= (5e-5 if kw["mode"] == 2 else 1e-5)
        for c in range(nch):
            rs = ob.tap(c, O.TAP_AUDIO_SPECT, 1024)
            rm = ob.tap(c, O.TAP_AUDIO_MAX, 3)
            assert np.abs(sp[c, f] - rs).max() <= 3e-5 * rs.max()  # squares: twice the 1e-5 of the spectrum
            assert abs(mx[c, f, 0] - rm[0]) <= 3e-5 * rm[0]
            assert abs(mx[c, f, 2] - rm[2]) <= 3e-5 * rm[2]
            gi = int(mx[c, f, 1])
            assert gi == int(rm[1]) or rs[gi] >= (1 - 1e-4) * rm[0]  # same bin unless two bins tie within rounding
            assert mx[c, f, 0] == sp[c, f].max() and gi == int(np.argmax(sp[c, f]))  # self-consistent, first occurrence
    # switching it off returns to the plain kernels and leaves the running average alone
    rx.set_audio_spectrum(None, None)
    before = rx.state_records()[:, 184 + 12].copy()
    rx.ProcessIQData(torch.from_numpy(I[:, :L].copy()).cuda(), torch.from_numpy(Q[:, :L].copy()).cuda())
    after = rx.state_records()[:, 184 + 12]
    assert np.array_equal(before, after) and np.all(before > 0)

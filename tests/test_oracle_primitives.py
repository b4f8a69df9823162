"""CMSIS-DSP f32 primitive restatements in oracle/ vs independent numpy/scipy float64 models
(SURVEY App. B).  CPU only."""
import ctypes as C

import numpy as np
import pytest
from scipy import signal

import oracle_lib as O

fptr = O.fptr


@pytest.fixture(scope="module")
def L(built):
    return O.lib()


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096])
@pytest.mark.parametrize("inv", [0, 1])
def test_cfft_matches_numpy(L, n, inv):
    rng = np.random.default_rng(n + inv)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    buf = np.empty(2 * n, np.float32)
    buf[0::2], buf[1::2] = x.real, x.imag
    L.t41o_cfft_f32(fptr(buf), n, inv)
    got = buf[0::2] + 1j * buf[1::2]
    ref = np.fft.ifft(x.astype(np.complex128)) if inv else np.fft.fft(x.astype(np.complex128))
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6


def test_cfft_impulse_and_roundtrip(L):
    n = 512
    buf = np.zeros(2 * n, np.float32)
    buf[2 * 3] = 1.0  # delta at n=3 -> X[k] = exp(-j 2 pi 3 k / N)
    L.t41o_cfft_f32(fptr(buf), n, 0)
    k = np.arange(n)
    ref = np.exp(-2j * np.pi * 3 * k / n)
    assert np.abs((buf[0::2] + 1j * buf[1::2]) - ref).max() < 1e-6
    L.t41o_cfft_f32(fptr(buf), n, 1)
    exp = np.zeros(2 * n, np.float32)
    exp[6] = 1.0
    assert np.abs(buf - exp).max() < 1e-6


@pytest.mark.parametrize("M,ntaps,bs", [(4, 28, 2048), (2, 46, 512), (4, 28, 512)])
def test_fir_decimate_streaming(L, M, ntaps, bs):
    """two consecutive blocks == one long lfilter + downsample; tap order = reversed h"""
    rng = np.random.default_rng(M * 100 + ntaps)
    c = rng.standard_normal(ntaps).astype(np.float32)
    x = rng.standard_normal(2 * bs).astype(np.float32)
    state = np.zeros(ntaps - 1 + bs, np.float32)
    out = np.empty(2 * bs // M, np.float32)
    for b in range(2):
        src = x[b * bs:(b + 1) * bs].copy()
        dst = np.empty(bs // M, np.float32)
        L.t41o_fir_decimate_f32(fptr(c), ntaps, M, fptr(state), fptr(src), fptr(dst), bs)
        out[b * bs // M:(b + 1) * bs // M] = dst
    ref = signal.lfilter(c[::-1].astype(np.float64), [1.0], x.astype(np.float64))[::M]
    assert np.abs(out - ref).max() < 2e-5 * np.abs(ref).max()


def test_fir_decimate_in_place(L):
    """the reference calls it with pSrc == pDst (Process.cpp:474)"""
    rng = np.random.default_rng(5)
    c = rng.standard_normal(28).astype(np.float32)
    x = rng.standard_normal(2048).astype(np.float32)
    s1, s2 = np.zeros(27 + 2048, np.float32), np.zeros(27 + 2048, np.float32)
    a = x.copy()
    L.t41o_fir_decimate_f32(fptr(c), 28, 4, fptr(s1), fptr(a), fptr(a), 2048)
    b = np.empty(512, np.float32)
    L.t41o_fir_decimate_f32(fptr(c), 28, 4, fptr(s2), fptr(x), fptr(b), 2048)
    assert np.array_equal(a[:512], b)


@pytest.mark.parametrize("Lf,ntaps,bs", [(2, 48, 256), (4, 32, 512)])
def test_fir_interpolate_streaming(L, Lf, ntaps, bs):
    rng = np.random.default_rng(Lf * 10 + ntaps)
    c = rng.standard_normal(ntaps).astype(np.float32)
    x = rng.standard_normal(2 * bs).astype(np.float32)
    P = ntaps // Lf
    state = np.zeros(P - 1 + bs, np.float32)
    out = np.empty(2 * bs * Lf, np.float32)
    for b in range(2):
        src = x[b * bs:(b + 1) * bs].copy()
        dst = np.empty(bs * Lf, np.float32)
        L.t41o_fir_interpolate_f32(fptr(c), ntaps, Lf, fptr(state), fptr(src), fptr(dst), bs)
        out[b * bs * Lf:(b + 1) * bs * Lf] = dst
    ref = signal.upfirdn(c[::-1].astype(np.float64), x.astype(np.float64), up=Lf)[:out.size]
    assert np.abs(out - ref).max() < 2e-5 * np.abs(ref).max()


def test_biquads_match_lfilter(L):
    rng = np.random.default_rng(9)
    x = rng.standard_normal(1000).astype(np.float32)
    # DF2T with the DC high-pass coefficients (FIR.cpp:87-89): {b0,b1,b2,a1,a2}, a's pre-negated
    c = np.array([0.927176191943378969, -0.927176191943378969, 0.0, 0.854352383886757938, 0.0], np.float32)
    st = np.zeros(2, np.float32)
    y = np.empty_like(x)
    L.t41o_biquad_df2T_f32(fptr(c), fptr(st), fptr(x), fptr(y), x.size)
    ref = signal.lfilter(c[:3].astype(np.float64), [1.0, -float(c[3]), -float(c[4])], x.astype(np.float64))
    assert np.abs(y - ref).max() < 1e-5
    # DF1 with a real low-pass
    cs = np.zeros(5, np.float32)
    L.t41o_SetIIRCoeffs(fptr(cs), 3000.0, 1.3, 24000.0, 0)
    st4 = np.zeros(4, np.float32)
    y2 = np.empty_like(x)
    L.t41o_biquad_df1_f32(fptr(cs), fptr(st4), fptr(x), fptr(y2), x.size)
    ref2 = signal.lfilter(cs[:3].astype(np.float64), [1.0, -float(cs[3]), -float(cs[4])], x.astype(np.float64))
    assert np.abs(y2 - ref2).max() < 1e-5
    # RBJ low-pass has unity DC gain: (b0+b1+b2)/(1-a1-a2) == 1
    assert abs(cs[:3].sum() / (1 - cs[3] - cs[4]) - 1.0) < 1e-6


def test_kaiser_lowpass_properties(L):
    """CalcFIRCoeffs (FIR.cpp:908): taps are the first nc of an (nc+1)-point symmetric design"""
    for nc, fc, fs in ((28, 3000.0, 192000.0), (46, 3000.0, 48000.0), (48, 3000.0, 48000.0), (32, 3000.0, 192000.0)):
        h = np.zeros(nc, np.float32)
        L.t41o_CalcFIRCoeffs(fptr(h), nc, fc, 90.0, 0, 0.0, fs)
        assert np.all(np.isfinite(h))
        # symmetric about index nc/2: h[j] == h[nc - j] for j = 1..nc-1
        j = np.arange(1, nc)
        assert np.allclose(h[j], h[nc - j], rtol=0, atol=2e-7 * np.abs(h).max())
        # centre tap = 2 fc / fs (MSinc(0) = 1, window = 1)
        assert abs(h[nc // 2] - np.float32(np.float32(fc / fs) * 2.0)) < 1e-9
        # an independent float64 Kaiser design agrees
        beta = 0.1102 * (90.0 - 8.71)
        ii = np.arange(-nc, nc, 2)
        x = ii / nc
        w = np.i0(beta * np.sqrt(1 - x * x)) / np.i0(beta)
        fcf = 2 * fc / fs
        ref = fcf * np.sinc(ii * 0.5 * fcf) * w
        assert np.abs(h - ref).max() < 5e-6 * np.abs(ref).max()


def test_complex_bandpass_and_mask():
    """CalcCplxFIRCoeffs + InitFilterMask: pass band where the reference puts it, last Q tap zeroed"""
    p = O.default_params(mode=0, FLoCut=200, FHiCut=3000)
    c = O.coeff_arrays(O.design(p), 512)
    m = c["mask"].reshape(512, 2)
    H = np.abs(m[:, 0] + 1j * m[:, 1])
    f = np.fft.fftfreq(512, 1 / 24000.0)
    assert H[(f > 600) & (f < 2600)].min() > 0.99
    assert H[(f < -400) | (f > 3800)].max() < 1e-3
    # impulse response recovered from the mask has 257 taps, tap 256 purely real (Filter.cpp:276-278)
    h = np.fft.ifft(m[:, 0] + 1j * m[:, 1])
    assert np.abs(h[257:]).max() < 1e-6
    assert abs(h[256].imag) < 1e-7
    # LSB mirrors it
    pl = O.default_params(mode=1, FLoCut=-3000, FHiCut=-200)
    ml = O.coeff_arrays(O.design(pl), 512)["mask"].reshape(512, 2)
    Hl = np.abs(ml[:, 0] + 1j * ml[:, 1])
    assert Hl[(f < -600) & (f > -2600)].min() > 0.99
    assert Hl[(f > 400)].max() < 1e-3


def test_izero_matches_numpy(L):
    for x in (0.0, 0.5, 3.0, 8.958, 12.0):
        assert abs(L.t41o_Izero(x) / np.i0(x) - 1) < 1e-6

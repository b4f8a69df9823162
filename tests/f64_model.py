"""Independent float64 numpy/scipy model of the T41 RX chain.

Purpose: catch *semantic* errors in oracle/t41_oracle.c (tap order, phase of the decimators,
overlap-save indexing, state carry across frames, polyphase ordering).  It deliberately does
NOT share structure with the oracle: it works on the whole multi-frame stream at once with
scipy.signal.lfilter / upfirdn / np.convolve, where the oracle works frame by frame with CMSIS
style state buffers.  Coefficients are taken as given (float32 values promoted to float64).

Citations (relative to /root/reference/software/T41_SDR/): Process.cpp:117-134 (gains, DC-HP),
:165-173 (IQ correction), Freq_Shift.cpp:42-141, Process.cpp:474-492, :498-605 (overlap-save),
:615-761 (demod), :917-929 (interpolation, volume).
"""
import numpy as np
from scipy import signal

HP_DC = (0.927176191943378969, -0.927176191943378969, 0.854352383886757938)  # b0, b1, a1 (FIR.cpp:87-89)
K_FM = 0.340447550238101026565118445432744920253753662109375  # Demod.h:7
PI_F = float(np.float32(3.1415926535897932384626433832795))


def _alpha_beta_mag(i, q):
    a = float(np.float32(0.960433870103))
    b = float(np.float32(0.397824734759))
    ai, aq = np.abs(i), np.abs(q)
    return np.where(ai > aq, a * ai + b * aq, a * aq + b * ai)


def _log10_fast(x):
    """Utility.cpp:245-258: cubic in the mantissa, in float64 here"""
    f, e = np.frexp(abs(x))
    y = ((1.23149591368684 * f - 4.11852516267426) * f + 6.02197014179219) * f - 3.13396450166353 + e
    return y * 0.3010299956639812


def agc_f64(y, g, trace=None):
    """AGC() (DSP_Fn.cpp:504-631) on a whole stream of complex samples, float64.

    Written from the algorithm's meaning rather than its ring-buffer mechanics: the output is
    the input delayed by W = attack_buffsize samples, `ring_max` is the maximum of |y| over the W
    most recent inputs (the reference keeps it incrementally and rescans when the maximum
    leaves the window, which is the same thing), the rest is the five-state gain law.
    g: dict name -> value (tests/oracle_lib.AGC_NAMES)."""
    W = int(g["attack_buffsize"])
    n = y.size
    a = np.abs(y)
    ap = np.concatenate([np.zeros(W), a])
    yp = np.concatenate([np.zeros(W, dtype=complex), y])
    win = np.lib.stride_tricks.sliding_window_view(ap, W)  # win[k] = ap[k : k + W]
    rmax = win[1:n + 1].max(axis=1)                         # inputs i - W + 1 .. i
    out = np.empty(n, dtype=complex)
    fba = hba = volts = save = 0.0
    state, decay_type, hang = 0, 0, 0
    for i in range(n):
        ao = ap[i]  # |input i - W|
        fba = g["fast_backmult"] * ao + g["onemfast_backmult"] * fba
        hba = g["hang_backmult"] * ao + g["onemhang_backmult"] * hba
        r = rmax[i]
        if hang > 0:
            hang -= 1
        rising = r >= volts
        if rising:
            if state >= 2:
                save = volts
            state = 0
            volts += (r - volts) * g["attack_mult"]
        elif state == 0:
            if volts > g["pop_ratio"] * fba:
                state = 1
                volts += (r - volts) * g["fast_decay_mult"]
            elif hba > g["hang_level"]:
                state, decay_type, hang = 2, 1, int(g["hang_count"])
            else:
                state, decay_type = 3, 0
                volts += (r - volts) * g["decay_mult"]
        elif state == 1:
            if volts > save:
                volts += (r - volts) * g["fast_decay_mult"]
            elif hang > 0:
                state = 2
            elif decay_type == 0:
                state = 3
                volts += (r - volts) * g["decay_mult"]
            else:
                state = 4
                volts += (r - volts) * g["hang_decay_mult"]
        elif state == 2:
            if hang == 0:
                state = 4
                volts += (r - volts) * g["hang_decay_mult"]
        elif state == 3:
            volts += (r - volts) * g["decay_mult"] * 0.05
        else:
            volts += (r - volts) * g["hang_decay_mult"]
        volts = max(volts, g["min_volts"])
        mult = (g["out_target"] - g["slope_constant"] * min(0.0, _log10_fast(g["inv_max_input"] * volts))) / volts
        out[i] = yp[i] * mult
        if trace is not None:
            trace.append((state, volts))
    return out


def mask_taps(coeffs, N):
    """impulse response the mask represents: N/2+1 complex taps, Q of the last one zeroed"""
    m = np.asarray(coeffs["mask"], dtype=np.float64).reshape(N, 2)
    h = np.fft.ifft(m[:, 0] + 1j * m[:, 1])
    return h  # length N; taps beyond N/2 are ~0


def fast_sin(x, quarter=0.0):
    """arm_sin_f32 / arm_cos_f32's method in float64: 512-entry table (exact sines), linear interpolation"""
    t = x * (1.0 / (2.0 * np.pi)) + quarter
    t = t - np.floor(t)
    fi = 512.0 * t
    k = int(fi)
    fr = fi - k
    return (1.0 - fr) * np.sin(2.0 * np.pi * k / 512.0) + fr * np.sin(2.0 * np.pi * (k + 1) / 512.0)


def approx_atan2(y, x):
    """Demod.cpp:148-197 as written (2 pi where pi/2 is meant)"""
    at = lambda z: (0.97239411 + (-0.19194795) * z * z) * z
    if x != 0.0:
        if abs(x) > abs(y):
            z = y / x
            return at(z) if x > 0.0 else (at(z) + np.pi if y >= 0.0 else at(z) - np.pi)
        z = x / y
        return -at(z) + 2.0 * np.pi if y > 0.0 else -at(z) - 2.0 * np.pi
    return 2.0 * np.pi if y > 0.0 else (-2.0 * np.pi if y < 0.0 else 0.0)


def sam_pll(y):
    """AMDecodeSAM's loop over a whole stream in float64: table-interpolated sine / cosine, the phase detector
    with ApproxAtan2 as written, the fade leveler with its time constants degenerate (no effect), the PLL
    constants of Demod.cpp:13-18 with omegaN = 200, pll_fmax = 4000, zeta = 0.65"""
    omegaN, fmax, zeta = 200.0, 4000.0, 0.65
    wmin, wmax = -2.0 * np.pi * fmax / 24000.0, 2.0 * np.pi * fmax / 24000.0
    g1 = 1.0 - np.exp(-2.0 * omegaN * zeta / 24000.0)
    g2 = -g1 + 2.0 * (1.0 - np.exp(-omegaN * zeta / 24000.0) * np.cos(omegaN / 24000.0 * np.sqrt(1.0 - zeta * zeta)))
    ph = fil = om = 0.0
    out = np.empty(y.size)
    for i in range(y.size):
        s, c = fast_sin(ph), fast_sin(ph, 0.25)
        ai, bi, aq, bq = c * y[i].real, s * y[i].real, c * y[i].imag, s * y[i].imag
        out[i] = (ai - bi) + (aq + bq)
        det = approx_atan2(-bi + aq, ai + bq)
        dl = fil
        om = min(max(om + g2 * det, wmin), wmax)
        fil = g1 * det + om
        ph = ph + dl
        while ph >= 2.0 * np.pi:
            ph -= 2.0 * np.pi
        while ph < 0.0:
            ph += 2.0 * np.pi
    return out


def run(I, Q, nco_freq, coeffs, *, fft_length=512, mode=0, FLoCut=200, FHiCut=3000,
        rfGainAllBands=1, RFgain=1, iq_amp=1.0, iq_phase=0.0, audioVolume=30,
        xmtMode=0, CWFreqShift=750, agc=None, agc_trace=None):
    """I, Q: 1-D float arrays, a whole number of frames (4*fft_length each). Returns audio.
    agc: None = AGCMode 0 (fixed gain 20), else the dict of AGC constants for agc_f64()."""
    def gain(y):
        return y * 20.0 if agc is None else agc_f64(y, agc, agc_trace)

    N = fft_length
    D = N // 2
    L = 4 * N
    I = np.asarray(I, dtype=np.float64)
    Q = np.asarray(Q, dtype=np.float64)
    assert I.size % L == 0
    nfr = I.size // L

    g = float(np.float32(10.0 ** (float(np.float32(rfGainAllBands) / np.float32(20)))))
    I = I * g
    Q = Q * g

    # DC high-pass with ONE state shared: per frame I then Q (Process.cpp:127-128)
    b = [HP_DC[0], HP_DC[1]]
    a = [1.0, -HP_DC[2]]
    seq = np.empty(2 * I.size)
    for f in range(nfr):
        seq[2 * f * L:(2 * f + 1) * L] = I[f * L:(f + 1) * L]
        seq[(2 * f + 1) * L:(2 * f + 2) * L] = Q[f * L:(f + 1) * L]
    seq = signal.lfilter(b, a, seq)
    for f in range(nfr):
        I[f * L:(f + 1) * L] = seq[2 * f * L:(2 * f + 1) * L]
        Q[f * L:(f + 1) * L] = seq[(2 * f + 1) * L:(2 * f + 2) * L]

    I = I * float(RFgain)
    Q = Q * float(RFgain)
    if mode in (0, 1, 2, 8):
        I = I * (-float(np.float32(iq_amp)))
        ph = float(np.float32(iq_phase))
        if ph < 0.0:
            Q = Q + ph * I
        else:
            I = I + ph * Q

    n = np.arange(I.size)
    z = (I + 1j * Q) * (1j ** (n % 4))  # FreqShift1: x[n] * j^n

    side = 0
    if xmtMode == 1:
        side = CWFreqShift if mode == 1 else (-CWFreqShift if mode == 0 else 0)
    inc = float(np.float32(2.0 * PI_F * (nco_freq + side) / 192000.0))
    # amplitude recurrence of the quadrature oscillator (Freq_Shift.cpp:128-134)
    r = np.empty(I.size)
    rv = 1.0
    for k in range(I.size):
        r[k] = rv
        rv = rv * (1.95 - rv * rv)
        if abs(rv * rv - 0.95) < 1e-15:
            r[k + 1:] = rv
            break
    osc = r * np.exp(1j * inc * (n + 1))
    z = z * float(np.float32(1.1)) * np.conj(osc)

    # decimators: y[m] = sum_d h[d] x[M m - d], h = reversed pCoeffs (CMSIS convention)
    h1 = np.asarray(coeffs["dec1"], dtype=np.float64)[::-1]
    h2 = np.asarray(coeffs["dec2"], dtype=np.float64)[::-1]
    z = signal.lfilter(h1, [1.0], z)[::4]
    z = signal.lfilter(h2, [1.0], z)[::2]

    if mode == 3:  # NFM (Process.cpp:252-276, 716-727, 765-816)
        out = np.empty(z.size)
        li = lq = 0.0
        for f in range(nfr):
            blk = z[f * D:(f + 1) * D]
            i_, q_ = blk.real, blk.imag
            o = np.empty(D)
            o[0] = K_FM * (i_[0] * (q_[0] - lq) - q_[0] * (i_[0] - li)) / (i_[0] ** 2 + q_[0] ** 2)
            o[1:] = K_FM * (q_[1:] * i_[:-1] - i_[1:] * q_[:-1]) / (i_[1:] ** 2 + q_[1:] ** 2)
            o[1:] = np.clip(o[1:], -1.0, 1.0)
            # quirk: "last sample" is floats D-2, D-1 of the interleaved buffer = complex D/2-1
            li, lq = i_[D // 2 - 1], q_[D // 2 - 1]
            out[f * D:(f + 1) * D] = o
        h = mask_taps(coeffs, N)
        y = gain(np.convolve(out, h)[:out.size])
        aud = y.real
    else:
        fk = (-float(np.float32(FLoCut)) * 0.001) if mode == 1 else (float(np.float32(FHiCut)) * 0.001)
        fk = float(np.float32(fk))
        vs = float(np.float32(7.0874 * fk ** (-1.232)))
        z = z * vs
        h = mask_taps(coeffs, N)
        y = gain(np.convolve(z, h)[:z.size])
        if mode in (0, 1):
            aud = y.real.copy()
        elif mode == 8:  # SAM (Demod.cpp:40-139 as written, see sam_pll)
            aud = sam_pll(y)
        else:  # AM
            m = _alpha_beta_mag(y.real, y.imag)
            w = signal.lfilter([1.0, -1.0], [1.0, -float(np.float32(0.99))], m)
            c = np.asarray(coeffs["biquad_lowpass1"], dtype=np.float64)
            aud = signal.lfilter(c[:3], [1.0, -c[3], -c[4]], w)

    g1 = np.asarray(coeffs["int1"], dtype=np.float64)[::-1]
    g2 = np.asarray(coeffs["int2"], dtype=np.float64)[::-1]
    a1 = signal.upfirdn(g1, aud, up=2)[:2 * aud.size]
    a2 = signal.upfirdn(g2, a1, up=4)[:4 * a1.size]
    x = float(np.float32(audioVolume / 100.0))
    vol = float(np.float32(8.0) * np.float32(np.float32(5.0) * np.float32(x) ** 5))
    return a2 * vol

"""Synthetic I/Q generator shared by tests and bench (SURVEY 8d "Synthetic input").

Per channel: three complex tones (one placed in the demodulator's pass band after the Fs/4 and
NCO shifts) + complex white noise, amplitudes in [0.05, 0.3], clipped to (-1, 1), 192 kS/s.
"""
import numpy as np

FS = 192000.0


def passband_tone_hz(mode, nco_hz, audio_hz=1000.0):
    """input frequency that lands at +/-audio_hz after (optional I flip), +Fs/4 and -NCO"""
    if mode == 3:  # NFM: no I sign flip (Process.cpp:165-173 skips it)
        return -48000.0 + nco_hz + audio_hz
    if mode == 1:  # LSB pass band is negative
        return 48000.0 - nco_hz + audio_hz
    return 48000.0 - nco_hz - audio_hz


def make_iq(n_channels, n_samples, nco_hz, mode=0, seed=0x5441315F, sigma=0.01, audio_hz=(400.0, 2500.0)):
    nco_hz = np.broadcast_to(np.asarray(nco_hz, dtype=np.float64), (n_channels,))
    I = np.empty((n_channels, n_samples), dtype=np.float32)
    Q = np.empty((n_channels, n_samples), dtype=np.float32)
    n = np.arange(n_samples, dtype=np.float64)
    for c in range(n_channels):
        rng = np.random.default_rng(seed + c)
        amps = rng.uniform(0.05, 0.3, 3)
        freqs = rng.uniform(-90000.0, 90000.0, 3)
        phases = rng.uniform(0, 2 * np.pi, 3)
        freqs[0] = passband_tone_hz(mode, nco_hz[c], rng.uniform(*audio_hz))
        if mode == 2:  # AM: a second in-band tone gives the envelope something to demodulate
            freqs[1] = freqs[0] + rng.uniform(300.0, 900.0)
            amps[1] = 0.5 * amps[0]
        x = np.zeros(n_samples, dtype=np.complex128)
        for a, f, ph in zip(amps, freqs, phases):
            x += a * np.exp(1j * (2 * np.pi * f / FS * n + ph))
        x += sigma * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples)) / np.sqrt(2)
        I[c] = np.clip(x.real, -0.999, 0.999)
        Q[c] = np.clip(x.imag, -0.999, 0.999)
    return I, Q


def nco_grid(n_channels, seed=7):
    """NCOFreq per channel: uniform in [-43000, 40000] Hz in 50 Hz steps (Tune.cpp:164)"""
    rng = np.random.default_rng(seed)
    return (rng.integers(-860, 801, n_channels) * 50).astype(np.int32)


def block_rel_err(a, b, frame_len):
    """per-frame max|a-b| / max|b| (SURVEY 8d tolerance definition); returns array [chan, frame]"""
    a = np.asarray(a, dtype=np.float64).reshape(a.shape[0], -1, frame_len)
    b = np.asarray(b, dtype=np.float64).reshape(b.shape[0], -1, frame_len)
    den = np.abs(b).max(axis=2)
    num = np.abs(a - b).max(axis=2)
    return np.where(den < 1e-6, num, num / np.maximum(den, 1e-30))


def make_fm(n_channels, n_samples, nco_hz, seed=0x464D, noise=0.003):
    """narrow-band FM carriers (one per channel) that land at 0 Hz after +Fs/4 and -NCO
    (NFM path: no I sign flip), sinusoidal modulation 300..2500 Hz, index 0.5..2.5"""
    nco_hz = np.broadcast_to(np.asarray(nco_hz, dtype=np.float64), (n_channels,))
    I = np.empty((n_channels, n_samples), dtype=np.float32)
    Q = np.empty((n_channels, n_samples), dtype=np.float32)
    n = np.arange(n_samples, dtype=np.float64)
    for c in range(n_channels):
        rng = np.random.default_rng(seed + c)
        fm, dev, amp = rng.uniform(300, 2500), rng.uniform(0.5, 2.5), rng.uniform(0.1, 0.5)
        ph = 2 * np.pi * (-48000.0 + nco_hz[c]) / FS * n + dev * np.sin(2 * np.pi * fm / FS * n + rng.uniform(0, 6.28))
        x = amp * np.exp(1j * ph) + noise * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
        I[c], Q[c] = x.real, x.imag
    return I, Q


def envelope_steps(n_samples, segments):
    """piecewise-constant fading envelope: segments = [(fraction of the stream, amplitude), ...]"""
    e = np.empty(n_samples, dtype=np.float32)
    pos = 0
    for frac, amp in segments:
        m = int(round(frac * n_samples))
        e[pos:pos + m] = amp
        pos += m
    e[pos:] = segments[-1][1]
    return e


def fade(I, Q, segments):
    """apply envelope_steps to every channel, clipped to the (-1, 1) range of the q15 front end"""
    e = envelope_steps(I.shape[-1], segments)
    return (np.clip(I * e, -0.999, 0.999).astype(np.float32),
            np.clip(Q * e, -0.999, 0.999).astype(np.float32))


def make_am_carrier(n_channels, n_samples, nco_hz, seed=0x53414D, noise=0.003, offset_hz=(-120.0, 120.0)):
    """AM carriers for the synchronous detector (SAM): one per channel, landing `offset_hz` away from
    0 Hz after the I flip, +Fs/4 and -NCO (inside the PLL's +-4 kHz range), sinusoidal modulation
    300..2500 Hz with depth 0.3..0.8, random start phase"""
    nco_hz = np.broadcast_to(np.asarray(nco_hz, dtype=np.float64), (n_channels,))
    I = np.empty((n_channels, n_samples), dtype=np.float32)
    Q = np.empty((n_channels, n_samples), dtype=np.float32)
    n = np.arange(n_samples, dtype=np.float64)
    for c in range(n_channels):
        rng = np.random.default_rng(seed + c)
        fm, depth, amp = rng.uniform(300, 2500), rng.uniform(0.3, 0.8), rng.uniform(0.1, 0.4)
        fc = passband_tone_hz(2, nco_hz[c], rng.uniform(*offset_hz))
        env = amp * (1.0 + depth * np.sin(2 * np.pi * fm / FS * n + rng.uniform(0, 6.28)))
        x = env * np.exp(1j * (2 * np.pi * fc / FS * n + rng.uniform(0, 6.28)))
        x += noise * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
        I[c] = np.clip(x.real, -0.999, 0.999)
        Q[c] = np.clip(x.imag, -0.999, 0.999)
    return I, Q

"""The arithmetic facts the pipelined AGC chain's block rests on (t41_sdr_amd/csrc/rx_kernels.hip: agc_fast_block_s,
agc_block_phased; DSP_Fn.cpp:525-629), checked in exact rational arithmetic on the CPU -- no GPU, no oracle:

  1. the "sandwich": state 3's decay step is (float)((double)volts + (double)step * .05) in the reference (:614).  For the
     second float neighbours c_lo < .05 < c_hi, whenever fma(step, c_lo, volts) == fma(step, c_hi, volts) that common value
     IS the reference's; where they differ the kernel falls back to the double expression.
  2. an FMA by 1 is a sum: fma(ss, 1, volts) == fl(volts + ss) -- so one v_pk_fma_f32 with a per-lane constant pair serves
     the lanes in state 3 (c_lo, c_hi) and every other lane (1, 1).
  3. without an attack volts does not rise: for ring_max < volts and a multiplier in [0, 1) every form of the step gives
     a value <= volts, so "volts > threshold" before a block's last step implies it before the earlier ones (the
     per-block bookkeeping's one comparison).
"""
from fractions import Fraction

import numpy as np

C_LO = np.frombuffer(np.uint32(0x3D4CCCCB).tobytes(), np.float32)[0]
C_HI = np.frombuffer(np.uint32(0x3D4CCCCE).tobytes(), np.float32)[0]


def round_f32(q):
    """nearest-even float32 of an exact rational (via float64 candidates: the two float32 neighbours of float(q))"""
    if q == 0:
        return np.float32(0.0)
    x = np.float32(float(q))  # float(q) is correctly rounded to f64; its f32 rounding may be off by double rounding
    cands = {x, np.nextafter(x, np.float32(np.inf), dtype=np.float32), np.nextafter(x, np.float32(-np.inf), dtype=np.float32)}
    best = None
    for c in cands:
        err = abs(Fraction(float(c)) - q)
        key = (err, int(np.frombuffer(np.float32(c).tobytes(), np.uint32)[0]) & 1)  # ties: even mantissa
        if best is None or key < best[0]:
            best = (key, c)
    return np.float32(best[1])


def fma_f32(a, b, c):
    return round_f32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def draws(n, seed):
    rng = np.random.default_rng(seed)
    volts = np.exp(rng.uniform(np.log(1e-6), np.log(4.0), n)).astype(np.float32)
    ring = (volts * rng.uniform(0.0, 0.9999, n).astype(np.float32)).astype(np.float32)  # no attack: ring_max < volts
    mult = np.exp(rng.uniform(np.log(1e-6), np.log(0.5), n)).astype(np.float32)
    return volts, ring, mult


def test_neighbours_bracket_the_literal():
    """.05 lies between the floats 0x3d4ccccc and 0x3d4ccccd; the kernel's constants are the SECOND neighbours either side"""
    bits = lambda x: int(np.frombuffer(np.float32(x).tobytes(), np.uint32)[0])
    below = np.frombuffer(np.uint32(0x3D4CCCCC).tobytes(), np.float32)[0]
    above = np.frombuffer(np.uint32(0x3D4CCCCD).tobytes(), np.float32)[0]
    assert Fraction(float(below)) < Fraction(1, 20) < Fraction(float(above)) and np.float32(0.05) == above
    assert bits(C_LO) == bits(below) - 1 and bits(C_HI) == bits(above) + 1


def test_sandwich_agreement_is_the_reference_value():
    volts, ring, mult = draws(4000, 1)
    agree = 0
    for v, r, m in zip(volts, ring, mult):
        step = np.float32(np.float32(r - v) * m)  # (ring_max - volts) * decay_mult, two f32 roundings
        ref = np.float32(np.float64(v) + np.float64(step) * 0.05)  # DSP_Fn.cpp:614 as compiled: double arithmetic, one cast
        lo, hi = fma_f32(step, C_LO, v), fma_f32(step, C_HI, v)
        if lo == hi:
            agree += 1
            assert hi == ref, (v, r, m, lo, hi, ref)
        else:  # the kernel redoes such a block with the double expression; the bracket still holds
            assert min(lo, hi) <= ref <= max(lo, hi)
    assert agree > 0.98 * len(volts)  # the fallback is rare


def test_fma_by_one_is_the_plain_sum():
    volts, ring, mult = draws(4000, 2)
    for v, r, m in zip(volts, ring, mult):
        ss = np.float32(np.float32(r - v) * m)
        assert fma_f32(ss, np.float32(1.0), v) == np.float32(v + ss)
    # signed zeros and an exact cancellation
    z = np.float32(0.0)
    assert fma_f32(-z, np.float32(1.0), z) == np.float32(z + -z)
    assert fma_f32(np.float32(-1.5), np.float32(1.0), np.float32(1.5)) == np.float32(0.0)


def test_volts_does_not_rise_without_an_attack():
    volts, ring, mult = draws(4000, 3)
    min_volts = np.float32(1e-7)
    for v, r, m in zip(volts, ring, mult):
        diff = np.float32(r - v)
        assert diff < 0
        ss = np.float32(diff * m)
        for nxt in (np.float32(v + ss), fma_f32(ss, C_HI, v), fma_f32(ss, C_LO, v), np.float32(v + np.float32(diff * np.float32(0.0)))):
            assert max(nxt, min_volts) <= v or v < min_volts

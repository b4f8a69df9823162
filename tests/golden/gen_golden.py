"""Generates tests/golden/*.npz: seeded inputs + the oracle's outputs for every demod mode.

The reference ships no golden vectors for this path, and cannot run here, so
these fixtures pin the ORACLE's behaviour (oracle/t41_oracle.c at the commit that generated
them): they catch regressions of the oracle and give the GPU tests data that does not depend
on rebuilding anything.  Re-run only when the oracle's semantics change on purpose:
    python tests/golden/gen_golden.py [case ...]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import siggen  # noqa: E402

CASES = {
    # name: (params overrides, nco list)
    "usb": (dict(mode=0, FLoCut=200, FHiCut=3000), [5000, -12350]),
    "lsb": (dict(mode=1, FLoCut=-3000, FHiCut=-200), [-43000, 50]),
    "am": (dict(mode=2, FLoCut=-3000, FHiCut=3000), [1000, 33300]),
    "nfm": (dict(mode=3, FLoCut=200, FHiCut=3000), [2500, -20000]),
    "usb_narrow_gains": (dict(mode=0, FLoCut=400, FHiCut=600, rfGainAllBands=6, RFgain=3, audioVolume=55,
                              IQAmpCorrectionFactor=1.02, IQPhaseCorrectionFactor=-0.013), [0, 39950]),
    "usb_cw_sidetone": (dict(mode=0, FLoCut=200, FHiCut=3000, xmtMode=1, CWFreqShift=750,
                             IQPhaseCorrectionFactor=0.021), [650, -30000]),
    # AGC on (DSP_Fn.cpp:504-631): fading signal, attack / fast decay / decay states
    "usb_agc_long": (dict(mode=0, FLoCut=200, FHiCut=3000, AGCMode=1), [5000, -12350]),
    "am_agc_fast": (dict(mode=2, FLoCut=-3000, FHiCut=3000, AGCMode=4), [1000, 33300]),
    "nfm_agc_med": (dict(mode=3, FLoCut=200, FHiCut=3000, AGCMode=3), [2500, -20000]),
    # the synthetic FFT lengths: frames of 4 * fft_length samples
    "lsb_fft1024": (dict(mode=1, FLoCut=-2800, FHiCut=-300, fft_length=1024), [7350, -31000]),
    "usb_fft4096": (dict(mode=0, FLoCut=400, FHiCut=600, fft_length=4096), [12000]),
    # side output (Process.cpp:550-570) and the q15 boundary (Process.cpp:102-111, 936): extra arrays, see main()
    "usb_spectrum": (dict(mode=0, FLoCut=200, FHiCut=3000), [5000, -12350]),
    "usb_q15": (dict(mode=0, FLoCut=200, FHiCut=3000, audioVolume=100), [5000, -12350]),
    # synchronous AM (Demod.cpp:40-139): long enough for the PLL to lock (tests compare the GPU from frame 12 on)
    "sam": (dict(mode=8, FLoCut=-3000, FHiCut=3000), [1000, -20000]),
    # the optional stages behind the demodulator (Process.cpp:841-866, Noise.cpp): Kim, spectral (20 half-blocks of
    # initialisation first), LMS + automatic notch (from power-on the notch is ill-conditioned: the GPU test only
    # requires finite output for it, see tests/test_noise_reduction.py::test_oracle_stage_conditioning)
    "usb_nr_kim": (dict(mode=0, FLoCut=200, FHiCut=3000, nrOptionSelect=1), [5000, -12350]),
    "usb_nr_spectral": (dict(mode=0, FLoCut=200, FHiCut=3000, nrOptionSelect=2), [5000, -12350]),
    "usb_nr_lms_notch": (dict(mode=0, FLoCut=200, FHiCut=3000, nrOptionSelect=3, ANR_notchOn=1), [5000, -12350]),
}
FADE = [(0.3, 2.0), (0.3, 0.05), (0.4, 1.2)]  # AGC cases
NFRAMES_AGC = 10
NFRAMES_SAM = 14
NFRAMES_NR = 16
NFRAMES = 3
L = 2048


def make_inputs(name, kw, nco):
    if kw.get("AGCMode", 0):
        global NFRAMES
        keep, NFRAMES = NFRAMES, NFRAMES_AGC
        try:
            I, Q = make_inputs(name, {k: v for k, v in kw.items() if k != "AGCMode"}, nco)
        finally:
            NFRAMES = keep
        return siggen.fade(I, Q, FADE)
    nch = len(nco)
    seed = 0x5441315F + sum(ord(c) for c in name)
    if kw.get("nrOptionSelect", 0) or kw.get("ANR_notchOn", 0):
        return siggen.make_iq(nch, NFRAMES_NR * L, np.asarray(nco), mode=kw["mode"], seed=seed)
    if kw["mode"] == 8:
        return siggen.make_am_carrier(nch, NFRAMES_SAM * L, np.asarray(nco), seed=seed)
    if kw["mode"] == 3:  # FM-modulated carriers so the discriminator sees a real signal
        rng = np.random.default_rng(seed)
        n = np.arange(NFRAMES * L)
        I = np.empty((nch, n.size), np.float32)
        Q = np.empty((nch, n.size), np.float32)
        for c in range(nch):
            fm = rng.uniform(300, 2500)
            dev = rng.uniform(0.5, 2.5)
            ph = 2 * np.pi * (-48000.0 + nco[c]) / 192000.0 * n + dev * np.sin(2 * np.pi * fm / 192000.0 * n)
            x = 0.3 * np.exp(1j * ph) + 0.003 * (rng.standard_normal(n.size) + 1j * rng.standard_normal(n.size))
            I[c], Q[c] = x.real, x.imag
        return I, Q
    Lf = 4 * kw.get("fft_length", 512)
    band = (450.0, 550.0) if kw.get("FHiCut") == 600 and "fft_length" in kw else (400.0, 2500.0)
    return siggen.make_iq(nch, NFRAMES * Lf, np.asarray(nco), mode=kw["mode"], seed=seed, audio_hz=band)


def main():
    only = set(sys.argv[1:])  # optional: the cases to (re)generate
    for name, (kw, nco) in CASES.items():
        if only and name not in only:
            continue
        I, Q = make_inputs(name, kw, nco)
        p = O.default_params(**kw)
        ob = O.OracleBatch(p, np.asarray(nco, dtype=np.int32))
        extra = {}
        if name == "usb_q15":  # the queues' samples: L queue = Q, R queue = I (Process.cpp:107-108)
            qI = np.clip(np.round(I * 32768.0), -32768, 32767).astype(np.int16)
            qQ = np.clip(np.round(Q * 32768.0), -32768, 32767).astype(np.int16)
            extra = dict(Q_in_L=qQ, Q_in_R=qI, Q_out_L=ob.process_q15(qQ, qI))
            I, Q = qI.astype(np.float32) / np.float32(32768), qQ.astype(np.float32) / np.float32(32768)
            ob.reset()
        if name == "usb_spectrum":  # audioSpectBuffer / audioMaxSquared, AudioMaxIndex, audioMaxSquaredAve per frame
            Lf = ob.frame_len
            nfr = I.shape[1] // Lf
            out = np.empty_like(I)
            sp = np.zeros((len(nco), nfr, 1024), np.float32)
            mx = np.zeros((len(nco), nfr, 3), np.float32)
            for f in range(nfr):
                out[:, f * Lf:(f + 1) * Lf] = ob.process(np.ascontiguousarray(I[:, f * Lf:(f + 1) * Lf]),
                                                         np.ascontiguousarray(Q[:, f * Lf:(f + 1) * Lf]))
                for c in range(len(nco)):
                    sp[c, f] = ob.tap(c, O.TAP_AUDIO_SPECT, 1024)
                    mx[c, f] = ob.tap(c, O.TAP_AUDIO_MAX, 3)
            extra = dict(spect=sp, spect_max=mx)
        else:
            out = ob.process(I, Q)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), I=I, Q=Q, nco=np.asarray(nco, np.int32), audio=out,
                            params=np.array(sorted(kw.items()), dtype=object).astype(str), **extra)
        print(name, out.shape, float(np.abs(out).max()))


if __name__ == "__main__":
    main()

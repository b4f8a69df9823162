"""ctypes binding of oracle/libt41oracle.so (the CPU checker).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package t41_sdr_amd never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

DEMOD_USB, DEMOD_LSB, DEMOD_AM, DEMOD_NFM = 0, 1, 2, 3
DEMOD_SAM = 8
(TAP_POST_NCO_I, TAP_POST_NCO_Q, TAP_DEC_I, TAP_DEC_Q, TAP_IFFT, TAP_DEMOD, TAP_AGC_VOLTS, TAP_AUDIO_SPECT, TAP_AUDIO_MAX,
 TAP_AGC_EDGES, TAP_FFT_SPEC, TAP_FFT_SPEC_OLD) = range(12)
AGC_NAMES = ("attack_mult", "decay_mult", "fast_decay_mult", "fast_backmult", "onemfast_backmult",
             "hang_backmult", "onemhang_backmult", "hang_decay_mult", "out_target", "min_volts",
             "slope_constant", "inv_max_input", "hang_level", "pop_ratio", "hang_count", "attack_buffsize")


class Params(C.Structure):
    _fields_ = [
        ("fft_length", C.c_int32),
        ("mode", C.c_int32),
        ("FLoCut", C.c_int32),
        ("FHiCut", C.c_int32),
        ("rfGainAllBands", C.c_int32),
        ("RFgain", C.c_int32),
        ("IQAmpCorrectionFactor", C.c_float),
        ("IQPhaseCorrectionFactor", C.c_float),
        ("AGCMode", C.c_int32),
        ("audioVolume", C.c_int32),
        ("nfmFilterBW", C.c_int32),
        ("xmtMode", C.c_int32),
        ("CWFreqShift", C.c_int32),
        ("am_lpf_f0", C.c_int32),
        ("AGC_thresh", C.c_int32),
        ("nfm_demod", C.c_int32),
        ("nrOptionSelect", C.c_int32),
        ("ANR_notchOn", C.c_int32),
        ("NR_PSI", C.c_float),
        ("NR_alpha", C.c_float),
        ("NR_beta", C.c_float),
    ]


class Coeffs(C.Structure):
    _fields_ = [
        ("dec1", C.c_float * 28),
        ("dec2", C.c_float * 46),
        ("int1", C.c_float * 48),
        ("int2", C.c_float * 32),
        ("biquad_lowpass1", C.c_float * 5),
        ("agc", C.c_float * 16),
        ("mask", C.c_float * (2 * 4096)),
    ]


_lib = None


def build(native=False):
    target = "libt41oracle_native.so" if native else "libt41oracle.so"
    path = os.path.join(ORACLE_DIR, target)
    deps = [os.path.join(ORACLE_DIR, f) for f in ("t41_oracle.c", "t41_tx_oracle.c", "t41_nr_oracle.c", "t41_oracle.h")]
    if (not os.path.exists(path)) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in deps):
        subprocess.check_call(["make", "-C", ORACLE_DIR, target], stdout=subprocess.DEVNULL)
    return path


def lib(native=False):
    global _lib
    if _lib is not None and not native:
        return _lib
    L = C.CDLL(build(native))
    fp = C.POINTER(C.c_float)
    L.t41o_default_params.argtypes = [C.POINTER(Params)]
    L.t41o_design.argtypes = [C.POINTER(Params), C.POINTER(Coeffs)]
    L.t41o_design.restype = C.c_int
    L.t41o_cfft_f32.argtypes = [fp, C.c_int, C.c_int]
    L.t41o_fir_decimate_f32.argtypes = [fp, C.c_int, C.c_int, fp, fp, fp, C.c_int]
    L.t41o_fir_interpolate_f32.argtypes = [fp, C.c_int, C.c_int, fp, fp, fp, C.c_int]
    L.t41o_biquad_df2T_f32.argtypes = [fp, fp, fp, fp, C.c_int]
    L.t41o_biquad_df1_f32.argtypes = [fp, fp, fp, fp, C.c_int]
    L.t41o_CalcFIRCoeffs.argtypes = [fp, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float]
    L.t41o_CalcCplxFIRCoeffs.argtypes = [fp, fp, C.c_int, C.c_float, C.c_float, C.c_float]
    L.t41o_SetIIRCoeffs.argtypes = [fp, C.c_float, C.c_float, C.c_float, C.c_int]
    L.t41o_Izero.argtypes = [C.c_float]
    L.t41o_Izero.restype = C.c_float
    L.t41o_channel_create.argtypes = [C.c_int]
    L.t41o_channel_create.restype = C.c_void_p
    L.t41o_channel_destroy.argtypes = [C.c_void_p]
    L.t41o_channel_reset.argtypes = [C.c_void_p]
    L.t41o_process_frame.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Coeffs), C.c_long, fp, fp, fp]
    L.t41o_process_frame.restype = C.c_int
    L.t41o_process_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(Params),
                                     C.POINTER(Coeffs), C.POINTER(C.c_int32), fp, fp, fp, C.c_int]
    L.t41o_process_batch.restype = C.c_int
    sp = C.POINTER(C.c_int16)
    L.t41o_q15_to_float.argtypes = [sp, fp, C.c_int]
    L.t41o_float_to_q15.argtypes = [fp, sp, C.c_int]
    L.t41o_process_batch_q15.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(Params),
                                         C.POINTER(Coeffs), C.POINTER(C.c_int32), sp, sp, sp]
    L.t41o_process_batch_q15.restype = C.c_int
    L.t41o_channel_tap.argtypes = [C.c_void_p, C.c_int, fp, C.c_int]
    L.t41o_channel_tap.restype = C.c_int
    L.t41o_channel_set_display.argtypes = [C.c_void_p, C.c_int]
    L.t41o_channel_set_display.restype = C.c_int
    L.t41o_nr_create.restype = C.c_void_p
    L.t41o_nr_destroy.argtypes = [C.c_void_p]
    L.t41o_nr_reset.argtypes = [C.c_void_p]
    L.t41o_nr_block.argtypes = [C.c_void_p, C.POINTER(Params), fp, fp]
    L.t41o_nr_peek.argtypes = [C.c_void_p, C.c_int, fp, C.c_int]
    L.t41o_nr_peek.restype = C.c_int
    L.t41o_nr_supported.argtypes = [C.POINTER(Params)]
    L.t41o_nr_supported.restype = C.c_int
    L.t41o_channel_nr.argtypes = [C.c_void_p]
    L.t41o_channel_nr.restype = C.c_void_p
    L.t41o_tx_create.restype = C.c_void_p
    L.t41o_tx_destroy.argtypes = [C.c_void_p]
    L.t41o_tx_reset.argtypes = [C.c_void_p]
    L.t41o_tx_process_frame.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, sp, sp, sp, sp]
    L.t41o_tx_process_frame.restype = C.c_int
    L.t41o_tx_table.argtypes = [C.c_int, C.POINTER(C.c_int)]
    L.t41o_tx_table.restype = fp
    if not native:
        _lib = L
    return L


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def default_params(**kw):
    p = Params()
    lib().t41o_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def design(p):
    c = Coeffs()
    rc = lib().t41o_design(C.byref(p), C.byref(c))
    if rc:
        raise ValueError("t41o_design rc=%d" % rc)
    return c


def coeff_arrays(c, fft_length):
    return {
        "dec1": np.ctypeslib.as_array(c.dec1).copy(),
        "dec2": np.ctypeslib.as_array(c.dec2).copy(),
        "int1": np.ctypeslib.as_array(c.int1).copy(),
        "int2": np.ctypeslib.as_array(c.int2).copy(),
        "biquad_lowpass1": np.ctypeslib.as_array(c.biquad_lowpass1).copy(),
        "agc": np.ctypeslib.as_array(c.agc).copy(),
        "mask": np.ctypeslib.as_array(c.mask)[: 2 * fft_length].copy(),
    }


class OracleBatch:
    """nchan independent oracle channels run through consecutive frames."""

    def __init__(self, params, nco_freqs, native=False):
        self.L = lib(native)
        self.p = params
        self.c = Coeffs()
        rc = self.L.t41o_design(C.byref(self.p), C.byref(self.c))
        if rc:
            raise ValueError("t41o_design rc=%d" % rc)
        self.nco = np.ascontiguousarray(nco_freqs, dtype=np.int32)
        self.nchan = len(self.nco)
        self.frame_len = 4 * params.fft_length
        self.chs = (C.c_void_p * self.nchan)()
        for i in range(self.nchan):
            self.chs[i] = self.L.t41o_channel_create(params.fft_length)

    def redesign(self):
        rc = self.L.t41o_design(C.byref(self.p), C.byref(self.c))
        if rc:
            raise ValueError("t41o_design rc=%d" % rc)

    def process(self, I, Q, nthreads=1):
        I = np.ascontiguousarray(I, dtype=np.float32)
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        assert I.shape == Q.shape and I.shape[0] == self.nchan and I.shape[1] % self.frame_len == 0
        nframes = I.shape[1] // self.frame_len
        out = np.empty_like(I)
        rc = self.L.t41o_process_batch(self.chs, self.nchan, nframes, C.byref(self.p), C.byref(self.c),
                                       self.nco.ctypes.data_as(C.POINTER(C.c_int32)),
                                       fptr(I), fptr(Q), fptr(out), nthreads)
        if rc:
            raise RuntimeError("t41o_process_batch rc=%d" % rc)
        return out

    def process_q15(self, Q_in_L, Q_in_R):
        a = np.ascontiguousarray(Q_in_L, dtype=np.int16)
        b = np.ascontiguousarray(Q_in_R, dtype=np.int16)
        assert a.shape == b.shape and a.shape[0] == self.nchan and a.shape[1] % self.frame_len == 0
        out = np.empty_like(a)
        sp = C.POINTER(C.c_int16)
        rc = self.L.t41o_process_batch_q15(self.chs, self.nchan, a.shape[1] // self.frame_len, C.byref(self.p),
                                           C.byref(self.c), self.nco.ctypes.data_as(C.POINTER(C.c_int32)),
                                           a.ctypes.data_as(sp), b.ctypes.data_as(sp), out.ctypes.data_as(sp))
        if rc:
            raise RuntimeError("t41o_process_batch_q15 rc=%d" % rc)
        return out

    def tap(self, ch, which, n):
        dst = np.zeros(n, dtype=np.float32)
        got = self.L.t41o_channel_tap(self.chs[ch], which, fptr(dst), n)
        return dst[:got]

    def set_display(self, spectrumZoom):
        """the display FFT side output (FFT.cpp:67-251): spectrumZoom 0..4, -1 = off; ZoomFFTPrep() semantics"""
        for i in range(self.nchan):
            rc = self.L.t41o_channel_set_display(self.chs[i], int(spectrumZoom))
            if rc:
                raise ValueError("t41o_channel_set_display rc=%d" % rc)

    def reset(self):
        for i in range(self.nchan):
            self.L.t41o_channel_reset(self.chs[i])

    def close(self):
        for i in range(self.nchan):
            if self.chs[i]:
                self.L.t41o_channel_destroy(self.chs[i])
                self.chs[i] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TxOracleBatch:
    """nchan independent oracle exciters (oracle/t41_tx_oracle.c) run through consecutive frames"""

    def __init__(self, nchan, mode=DEMOD_USB, amp=1.0, phase=0.0):
        self.L = lib()
        self.nchan, self.mode, self.amp, self.phase = nchan, mode, amp, phase
        self.chs = [self.L.t41o_tx_create() for _ in range(nchan)]

    def process(self, Q_in_L_Ex, Q_in_R_Ex=None):
        a = np.ascontiguousarray(Q_in_L_Ex, dtype=np.int16)
        b = a if Q_in_R_Ex is None else np.ascontiguousarray(Q_in_R_Ex, dtype=np.int16)
        assert a.shape[0] == self.nchan and a.shape[1] % 2048 == 0
        oL, oR = np.empty_like(a), np.empty_like(a)
        sp = C.POINTER(C.c_int16)
        for c in range(self.nchan):
            for f in range(a.shape[1] // 2048):
                sl = slice(f * 2048, (f + 1) * 2048)
                ia, ib = np.ascontiguousarray(a[c, sl]), np.ascontiguousarray(b[c, sl])
                ol, orr = np.empty(2048, np.int16), np.empty(2048, np.int16)
                rc = self.L.t41o_tx_process_frame(self.chs[c], self.mode, self.amp, self.phase, ia.ctypes.data_as(sp),
                                                  ib.ctypes.data_as(sp), ol.ctypes.data_as(sp), orr.ctypes.data_as(sp))
                assert rc == 0
                oL[c, sl], oR[c, sl] = ol, orr
        return oL, oR

    def table(self, which):
        n = C.c_int()
        p = self.L.t41o_tx_table(which, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def close(self):
        for c in self.chs:
            self.L.t41o_tx_destroy(c)
        self.chs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

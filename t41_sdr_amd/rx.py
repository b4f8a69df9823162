"""Host-side mirror of the reference's receive-path interface, over the C ABI.

The reference (tmr4/T41_SDR) drives the path through globals and three functions:
  InitializeDataArrays()  T41_SDR.ino:473   -> RxChain(...)
  SetupMode()/CalcFilters()  Filter.cpp:235-249,341-385  -> RxChain.CalcFilters(**changes)
  ProcessIQData()  Process.cpp:70           -> RxChain.ProcessIQData(float_buffer_L, float_buffer_R)
Names and argument meanings follow the firmware (bands[].FLoCut, NCOFreq, audioVolume ...).
PyTorch is used only as the owner of device memory / streams; all arithmetic happens in
libt41rx.so (HIP).  No CPU fallback exists.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import DEMOD_AM, DEMOD_LSB, DEMOD_NFM, DEMOD_SAM, DEMOD_USB, Params, T41RxError, check  # noqa: F401


def default_params(**overrides):
    """gwv.cpp:14-96 / bands[] defaults (20 m row) with AGCMode = 0."""
    lib = _lib.load()
    p = Params()
    lib.t41rx_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError("t41rx_params has no field %r" % k)
        setattr(p, k, v)
    return p


def design_coeffs(params):
    """CalcFilters() on the host: returns the coefficient blob (bytes-like numpy uint8)."""
    lib = _lib.load()
    n = lib.t41rx_coeff_blob_bytes(params.fft_length)
    if n == 0:
        raise T41RxError(_lib.ERR_ARG, "unsupported fft_length %d" % params.fft_length)
    blob = np.zeros(n, dtype=np.uint8)
    check(lib.t41rx_design_coeffs(C.byref(params), blob.ctypes.data_as(C.c_void_p), n))
    return blob


BLOB_HEADER_WORDS = 32   # magic, abi, fft_length, mode, sizeof(params), 3 reserved | t41rx_params padded to 24 words
STATE_HEADER_BYTES = 32  # checkpoint header: magic, abi, fft_length, n_channels, floats per channel, sections, spectrumZoom, reserved


def blob_params(blob):
    """the t41rx_params a coefficient blob was designed for (what set_coeffs() installs)"""
    p = Params()
    raw = np.ascontiguousarray(blob, dtype=np.uint8)[4 * 8:4 * 8 + C.sizeof(Params)].tobytes()
    C.memmove(C.byref(p), raw, C.sizeof(Params))
    return p


def blob_fields(blob, fft_length):
    """Split a coefficient blob into the reference's arrays (numpy views)."""
    f = np.frombuffer(blob, dtype=np.float32)
    o = BLOB_HEADER_WORDS
    out = {}
    for name, n in (("dec1", 28), ("dec2", 46), ("int1", 48), ("int2", 32), ("biquad_lowpass1", 5),
                    ("scalars", 16), ("agc", 16), ("mask", 2 * fft_length)):
        out[name] = f[o:o + n]
        o += n
    return out


class RxChain:
    """n_channels independent T41 receive channels resident on one MI355X."""

    def __init__(self, n_channels, params=None, device=0, NCOFreq=None):
        self._lib = _lib.load()
        self.params = params if params is not None else default_params()
        self._ctx = C.c_void_p()
        check(self._lib.t41rx_create(C.byref(self._ctx), int(device), int(n_channels), C.byref(self.params)))
        self.n_channels = int(n_channels)
        self.device = int(device)
        self.frame_len = self._lib.t41rx_frame_len(self._ctx)
        self.layout = "channel"
        if NCOFreq is not None:
            self.SetNCOFreq(NCOFreq)

    # -- configuration ---------------------------------------------------------------------
    def CalcFilters(self, **changes):
        """Filter/mode/gain change between two ProcessIQData() calls; state is kept."""
        for k, v in changes.items():
            if not hasattr(self.params, k):
                raise AttributeError("t41rx_params has no field %r" % k)
            setattr(self.params, k, v)
        check(self._lib.t41rx_set_params(self._ctx, C.byref(self.params)))

    SetupMode = CalcFilters

    def SetNCOFreq(self, NCOFreq):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(NCOFreq, dtype=np.int32), (self.n_channels,)))
        check(self._lib.t41rx_set_nco_freq(self._ctx, a.ctypes.data_as(C.POINTER(C.c_int32)), self.n_channels))

    def get_params(self):
        """the parameters the context runs with (after set_coeffs(): the ones the blob was designed for)"""
        p = Params()
        check(self._lib.t41rx_get_params(self._ctx, C.byref(p)))
        return p

    def coeffs(self):
        n = self._lib.t41rx_coeff_blob_bytes(self.params.fft_length)
        blob = np.zeros(n, dtype=np.uint8)
        check(self._lib.t41rx_get_coeffs(self._ctx, blob.ctypes.data_as(C.c_void_p), n))
        return blob

    def set_coeffs(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        check(self._lib.t41rx_set_coeffs(self._ctx, blob.ctypes.data_as(C.c_void_p), blob.size))
        # the context now runs the parameters the blob was designed for: a later CalcFilters(one
        # field) must start from THEM, not from what this object was created with
        self.params = self.get_params()

    def reset(self):
        check(self._lib.t41rx_reset(self._ctx))

    def set_buffer_layout(self, layout):
        """"channel" (default): I / Q / audio are [n_channels, n_frames*frame_len]; "time": [n_frames, n_channels,
        frame_len] -- the [n_channels, frame_len] buffers of consecutive single-frame calls stacked as they arrive
        (t41rx_set_buffer_layout; FFT_LENGTH 512)."""
        code = {"channel": 0, "time": 1}[layout]
        check(self._lib.t41rx_set_buffer_layout(self._ctx, code))
        self.layout = layout

    def get_state(self):
        n = self._lib.t41rx_state_bytes(self._ctx)
        buf = np.zeros(n, dtype=np.uint8)
        check(self._lib.t41rx_get_state(self._ctx, buf.ctypes.data_as(C.c_void_p), n))
        return buf

    def set_state(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        check(self._lib.t41rx_set_state(self._ctx, buf.ctypes.data_as(C.c_void_p), buf.size))

    def state_records(self, buf=None):
        """the per-channel records of a checkpoint (default: a fresh one) as float32 [n_channels, floats]"""
        buf = self.get_state() if buf is None else np.ascontiguousarray(buf, dtype=np.uint8)
        per = int(buf[:STATE_HEADER_BYTES].view(np.int32)[4])  # floats per channel of the path's section (the side stages' follow it)
        return buf[STATE_HEADER_BYTES:STATE_HEADER_BYTES + 4 * per * self.n_channels].view(np.float32).reshape(self.n_channels, per)

    # -- the hot path ----------------------------------------------------------------------
    def ProcessIQData(self, float_buffer_L, float_buffer_R, out=None):
        """One ProcessIQData() per channel (or several consecutive ones).

        torch CUDA tensors [n_channels, n_frames*frame_len] (float32, contiguous) run on the
        current torch stream without synchronising and return a tensor; numpy arrays go
        through the host-pointer entry point and return a numpy array.
        """
        if isinstance(float_buffer_L, np.ndarray):
            I = np.ascontiguousarray(float_buffer_L, dtype=np.float32)
            Q = np.ascontiguousarray(float_buffer_R, dtype=np.float32)
            nfr = self._check_shape(I.shape, Q.shape)
            audio = np.empty_like(I) if out is None else self._check_out_numpy(out, I)
            fp = C.POINTER(C.c_float)
            check(self._lib.t41rx_process_host(self._ctx, I.ctypes.data_as(fp), Q.ctypes.data_as(fp),
                                               audio.ctypes.data_as(fp), nfr))
            return audio
        import torch
        I, Q = float_buffer_L, float_buffer_R
        if not (I.is_cuda and Q.is_cuda and I.dtype == torch.float32 and Q.dtype == torch.float32
                and I.is_contiguous() and Q.is_contiguous()):
            raise ValueError("I/Q must be contiguous float32 CUDA tensors")
        if I.device.index != self.device or Q.device.index != self.device:
            raise ValueError("I/Q live on another device than this RxChain")
        nfr = self._check_shape(tuple(I.shape), tuple(Q.shape))
        audio = torch.empty_like(I) if out is None else self._check_out_torch(out, I)
        stream = torch.cuda.current_stream(I.device).cuda_stream
        check(self._lib.t41rx_process_device(self._ctx, I.data_ptr(), Q.data_ptr(), audio.data_ptr(), nfr,
                                             C.c_void_p(stream)))
        return audio

    def ProcessIQData_q15(self, Q_in_L, Q_in_R, out=None):
        """The same on the firmware's wire format (Process.cpp:102-111, 936-937): int16 (q15) blocks
        of the L and R record queues in -- I is taken from the R queue, Q from the L queue, as the
        firmware does -- and the q15 samples handed to Q_out_L.play() out.
        torch CUDA int16 tensors or numpy int16 arrays, [n_channels, n_frames*frame_len]."""
        if isinstance(Q_in_L, np.ndarray):
            a = np.ascontiguousarray(Q_in_L, dtype=np.int16)
            b = np.ascontiguousarray(Q_in_R, dtype=np.int16)
            nfr = self._check_shape(a.shape, b.shape)
            audio = np.empty_like(a) if out is None else self._check_out_numpy(out, a)
            check(self._lib.t41rx_process_host_q15(self._ctx, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                                   audio.ctypes.data_as(C.c_void_p), nfr))
            return audio
        import torch
        a, b = Q_in_L, Q_in_R
        if not (a.is_cuda and b.is_cuda and a.dtype == torch.int16 and b.dtype == torch.int16
                and a.is_contiguous() and b.is_contiguous()):
            raise ValueError("Q_in_L/Q_in_R must be contiguous int16 CUDA tensors")
        if a.device.index != self.device or b.device.index != self.device:
            raise ValueError("the queues live on another device than this RxChain")
        nfr = self._check_shape(tuple(a.shape), tuple(b.shape))
        audio = torch.empty_like(a) if out is None else self._check_out_torch(out, a)
        stream = torch.cuda.current_stream(a.device).cuda_stream
        check(self._lib.t41rx_process_device_q15(self._ctx, a.data_ptr(), b.data_ptr(), audio.data_ptr(), nfr,
                                                 C.c_void_p(stream)))
        return audio

    def _side_tensor(self, t, per_frame, what):
        """validate a side-output tensor; returns (pointer, frames it has room for)"""
        if t is None:
            return None, None
        import torch
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device.index == self.device):
            raise ValueError("%s must be a contiguous float32 CUDA tensor on device %d" % (what, self.device))
        frames = t.numel() // (self.n_channels * per_frame)
        if frames < 1:
            raise ValueError("%s holds less than one frame (%d floats per channel and frame)" % (what, per_frame))
        return C.c_void_p(t.data_ptr()), frames

    def set_debug_taps(self, post_nco=None, dec=None, demod=None):
        """stage taps (torch CUDA float32 tensors or None): post_nco [n_channels, n_frames*2*frame_len],
        dec [n_channels, n_frames*fft_length], demod [n_channels, n_frames*fft_length/2].  Calls with
        more frames than the smallest of them holds are refused."""
        N = self.params.fft_length
        got = [self._side_tensor(post_nco, 2 * self.frame_len, "post_nco"), self._side_tensor(dec, N, "dec"),
               self._side_tensor(demod, N // 2, "demod")]
        frames = [f for _, f in got if f is not None]
        check(self._lib.t41rx_set_debug_taps(self._ctx, got[0][0], got[1][0], got[2][0], min(frames) if frames else 0))
        self._taps = (post_nco, dec, demod)  # keep alive

    def set_audio_spectrum(self, spect=None, maxima=None):
        """the display by-product of the path (Process.cpp:550-570): torch CUDA float32 tensors
        [n_channels, n_frames, 1024] and [n_channels, n_frames, 3], or None/None to switch it off"""
        ps, fs = self._side_tensor(spect, 1024, "spect")
        pm, fm = self._side_tensor(maxima, 3, "maxima")
        frames = [f for f in (fs, fm) if f is not None]
        check(self._lib.t41rx_set_audio_spectrum(self._ctx, ps, pm, min(frames) if frames else 0))
        self._spect = (spect, maxima)  # keep alive

    def set_display_spectrum(self, spec=None, spec_old=None, spectrumZoom=1):
        """the display FFT (CalcZoom1Magn / ZoomFFTExe, FFT.cpp:67-251): torch CUDA float32 tensors
        [n_channels, n_frames, 512] for FFT_spec and FFT_spec_old, or None/None to switch it off"""
        ps, fs = self._side_tensor(spec, 512, "spec")
        po, fo = self._side_tensor(spec_old, 512, "spec_old")
        frames = [f for f in (fs, fo) if f is not None]
        check(self._lib.t41rx_set_display_spectrum(self._ctx, ps, po, int(spectrumZoom), min(frames) if frames else 0))
        self._disp = (spec, spec_old)  # keep alive

    def _check_out_torch(self, out, like):
        if not (out.is_cuda and out.dtype == like.dtype and out.is_contiguous() and tuple(out.shape) == tuple(like.shape)
                and out.device == like.device):
            raise ValueError("out must be a contiguous %s CUDA tensor of shape %r on %s" % (like.dtype, tuple(like.shape), like.device))
        return out

    @staticmethod
    def _check_out_numpy(out, like):
        if not (isinstance(out, np.ndarray) and out.dtype == like.dtype and out.shape == like.shape
                and out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"]):
            raise ValueError("out must be a writeable C-contiguous %s array of shape %r" % (like.dtype, like.shape))
        return out

    def _check_shape(self, si, sq):
        if self.layout == "time":
            if si != sq or len(si) != 3 or si[0] == 0 or si[1] != self.n_channels or si[2] != self.frame_len:
                raise ValueError("time-major I/Q must be [n_frames, n_channels=%d, frame_len=%d], got %r / %r"
                                 % (self.n_channels, self.frame_len, si, sq))
            return si[0]
        if si != sq or len(si) != 2 or si[0] != self.n_channels or si[1] == 0 or si[1] % self.frame_len:
            raise ValueError("I/Q must be [n_channels=%d, k*frame_len=%d], got %r / %r"
                             % (self.n_channels, self.frame_len, si, sq))
        return si[1] // self.frame_len

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.t41rx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

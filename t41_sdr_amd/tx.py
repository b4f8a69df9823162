"""Host-side mirror of the reference's transmit exciter interface, over the C ABI (include/t41tx.h).

The reference drives the exciter through globals and one function:
  ExciterIQData()   Exciter.cpp:46-169   -> TxChain.ExciterIQData(Q_in_L_Ex, Q_in_R_Ex)
All arithmetic happens in libt41rx.so (HIP); PyTorch only owns device memory and streams.  No CPU
fallback exists.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


class TxParams(C.Structure):
    """struct t41tx_params (include/t41tx.h)"""
    _fields_ = [("mode", C.c_int32), ("IQXAmpCorrectionFactor", C.c_float), ("IQXPhaseCorrectionFactor", C.c_float)]


_vp = C.c_void_p
TX_SYMBOLS = {
    "t41tx_default_params": (None, [C.POINTER(TxParams)]),
    "t41tx_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.POINTER(TxParams)]),
    "t41tx_destroy": (C.c_int, [_vp]),
    "t41tx_set_params": (C.c_int, [_vp, C.POINTER(TxParams)]),
    "t41tx_reset": (C.c_int, [_vp]),
    "t41tx_n_channels": (C.c_int, [_vp]),
    "t41tx_process_device_q15": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "t41tx_process_host_q15": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int]),
}
_bound = False


def _load():
    global _bound
    lib = _lib.load()
    if not _bound:
        for name, (res, args) in TX_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return lib


def default_tx_params(**overrides):
    p = TxParams()
    _load().t41tx_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError("t41tx_params has no field %r" % k)
        setattr(p, k, v)
    return p


class TxChain:
    """n_channels independent T41 SSB exciters resident on one MI355X."""
    FRAME = 2048

    def __init__(self, n_channels, params=None, device=0):
        self._lib = _load()
        self.params = params if params is not None else default_tx_params()
        self._ctx = C.c_void_p()
        check(self._lib.t41tx_create(C.byref(self._ctx), int(device), int(n_channels), C.byref(self.params)))
        self.n_channels = int(n_channels)
        self.device = int(device)

    def set_params(self, **changes):
        for k, v in changes.items():
            if not hasattr(self.params, k):
                raise AttributeError("t41tx_params has no field %r" % k)
            setattr(self.params, k, v)
        check(self._lib.t41tx_set_params(self._ctx, C.byref(self.params)))

    def reset(self):
        check(self._lib.t41tx_reset(self._ctx))

    def ExciterIQData(self, Q_in_L_Ex, Q_in_R_Ex=None):
        """int16 (q15) microphone samples [n_channels, k * 2048] -> (Q_out_L_Ex, Q_out_R_Ex), the I and Q
        drive.  torch CUDA int16 tensors run on the current stream; numpy arrays use the host entry."""
        if isinstance(Q_in_L_Ex, np.ndarray):
            a = np.ascontiguousarray(Q_in_L_Ex, dtype=np.int16)
            nfr = self._frames(a.shape)
            oL, oR = np.empty_like(a), np.empty_like(a)
            p = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
            check(self._lib.t41tx_process_host_q15(self._ctx, p(a), None, p(oL), p(oR), nfr))
            return oL, oR
        import torch
        a = Q_in_L_Ex
        if not (a.is_cuda and a.dtype == torch.int16 and a.is_contiguous() and a.device.index == self.device):
            raise ValueError("Q_in_L_Ex must be a contiguous int16 CUDA tensor on device %d" % self.device)
        nfr = self._frames(tuple(a.shape))
        oL, oR = torch.empty_like(a), torch.empty_like(a)
        stream = torch.cuda.current_stream(a.device).cuda_stream
        check(self._lib.t41tx_process_device_q15(self._ctx, a.data_ptr(), None, oL.data_ptr(), oR.data_ptr(), nfr, C.c_void_p(stream)))
        return oL, oR

    def _frames(self, shape):
        if len(shape) != 2 or shape[0] != self.n_channels or shape[1] == 0 or shape[1] % self.FRAME:
            raise ValueError("samples must be [n_channels=%d, k*2048], got %r" % (self.n_channels, shape))
        return shape[1] // self.FRAME

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.t41tx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

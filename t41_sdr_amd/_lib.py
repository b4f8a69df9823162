"""ctypes loader for libt41rx.so, the C ABI declared in include/t41rx.h.

The library is the product: HIP kernels + C++ host side.  There is no Python or CPU
implementation of the path behind it; if the shared object is missing, loading fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T41RX_LIB", os.path.join(_HERE, "libt41rx.so"))  # override: kernel experiments
if "T41RX_LIB" in os.environ:
    # an experiment build was asked for BY NAME (tools/build_variant.sh -> t41_sdr_amd/abl/): a library whose kernels
    # compute wrong results by construction or carry diagnostics refuses t41rx_create() unless the environment says so
    # (rx_experiments.hpp, rx_host.cpp).  The in-tree product library is never opted in here.
    os.environ.setdefault("T41RX_ALLOW_EXPERIMENT", "1")

T41RX_OK = 0
ERR_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM, ERR_STATE = -1, -2, -3, -4, -5
DEMOD_USB, DEMOD_LSB, DEMOD_AM, DEMOD_NFM = 0, 1, 2, 3
DEMOD_SAM = 8  # synchronous AM (SDT.h:67), fft_length 512
SSB_MODE, CW_MODE, DATA_MODE = 0, 1, 2


class Params(C.Structure):
    """struct t41rx_params (include/t41rx.h)."""
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


# every symbol include/t41rx.h declares: (restype, argtypes)
_vp, _fp = C.c_void_p, C.POINTER(C.c_float)
SYMBOLS = {
    "t41rx_abi_version": (C.c_int, []),
    "t41rx_strerror": (C.c_char_p, [C.c_int]),
    "t41rx_last_error": (C.c_char_p, []),
    "t41rx_supported_fft_length": (C.c_int, [C.c_int]),
    "t41rx_default_params": (None, [C.POINTER(Params)]),
    "t41rx_coeff_blob_bytes": (C.c_size_t, [C.c_int]),
    "t41rx_design_coeffs": (C.c_int, [C.POINTER(Params), _vp, C.c_size_t]),
    "t41rx_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.POINTER(Params)]),
    "t41rx_destroy": (C.c_int, [_vp]),
    "t41rx_set_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "t41rx_get_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "t41rx_get_coeffs": (C.c_int, [_vp, _vp, C.c_size_t]),
    "t41rx_set_coeffs": (C.c_int, [_vp, _vp, C.c_size_t]),
    "t41rx_set_nco_freq": (C.c_int, [_vp, C.POINTER(C.c_int32), C.c_int]),
    "t41rx_reset": (C.c_int, [_vp]),
    "t41rx_n_channels": (C.c_int, [_vp]),
    "t41rx_frame_len": (C.c_int, [_vp]),
    "t41rx_set_buffer_layout": (C.c_int, [_vp, C.c_int]),
    "t41rx_get_buffer_layout": (C.c_int, [_vp]),
    "t41rx_process_device": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp]),
    "t41rx_process_host": (C.c_int, [_vp, _fp, _fp, _fp, C.c_int]),
    "t41rx_set_audio_spectrum": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "t41rx_process_device_q15": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp]),
    "t41rx_process_host_q15": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int]),
    "t41rx_state_bytes": (C.c_size_t, [_vp]),
    "t41rx_get_state": (C.c_int, [_vp, _vp, C.c_size_t]),
    "t41rx_set_state": (C.c_int, [_vp, _vp, C.c_size_t]),
    "t41rx_set_debug_taps": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int]),
    "t41rx_set_display_spectrum": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int]),
}

_lib = None


class T41RxError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__("t41rx status %d%s" % (status, (": " + detail) if detail else ""))


def load():
    """Load libt41rx.so and bind every declared symbol (raises if the build is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `make -C t41_sdr_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback for the RX path." % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).
    # Import it first so this library binds to the SAME runtime instance that owns torch's
    # device memory and streams; two runtimes in one process do not see each other's state.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != T41RX_OK:
        lib = load()
        detail = lib.t41rx_last_error().decode("utf-8", "replace")
        if not detail:
            detail = lib.t41rx_strerror(status).decode()
        raise T41RxError(status, detail)

"""t41_sdr_amd -- MI355X-native T41 receive DSP hot path (ProcessIQData) behind a C ABI.

Only what the path needs lives here: csrc/ (HIP kernels + C++ host side + C ABI) and rx.py (the
host-side mirror of the reference interface).  Importing the package does not need a GPU;
creating an RxChain does.
"""
from ._lib import (DEMOD_AM, DEMOD_LSB, DEMOD_NFM, DEMOD_SAM, DEMOD_USB, LIB_PATH, Params, T41RxError,  # noqa: F401
                   load)
from .rx import RxChain, blob_fields, blob_params, default_params, design_coeffs  # noqa: F401
from .tx import TxChain, TxParams, default_tx_params  # noqa: F401

__all__ = ["RxChain", "Params", "T41RxError", "default_params", "design_coeffs", "blob_fields", "load",
           "DEMOD_USB", "DEMOD_LSB", "DEMOD_AM", "DEMOD_NFM", "DEMOD_SAM", "LIB_PATH"]

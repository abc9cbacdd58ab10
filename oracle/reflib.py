"""TEST INFRASTRUCTURE — ctypes driver for the compiled reference (oracle/_ref/libwhisper_ref.so).

The library is whisper.cpp v1.5.4 + ggml built from the sources under /root/reference by
oracle/Makefile, plus the accessors of oracle/ref_shim.cpp.
"""
from __future__ import annotations

import ctypes as C
import pathlib

import numpy as np

from godot_whisper_amd import abi

HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = HERE / "_ref" / "libwhisper_ref.so"

_SHIM = [
    ("ref_mel_dims", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("ref_mel_copy", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    ("ref_embd_conv", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    ("ref_embd_enc", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    ("ref_kv_copy", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint16), C.c_int]),
    ("ref_set_audio_ctx", None, [C.c_void_p, C.c_int]),
    ("ref_decoder_probs", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]),
    ("ref_timings", None, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    ("ref_process_logits", C.c_int, [C.c_void_p, abi.whisper_full_params, C.POINTER(C.c_float),
                                      C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("ref_sample_draws", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int,
                                    C.POINTER(abi.whisper_token_data)]),
    ("ref_gelu_table", C.c_int, [C.POINTER(C.c_uint16)]),
    ("ref_sizeof_full_params", C.c_size_t, []),
    ("ref_sizeof_token_data", C.c_size_t, []),
]


def available() -> bool:
    return LIB_PATH.exists()


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = abi.bind(C.CDLL(str(LIB_PATH)), abi.WHISPER_API, strict=True)
        abi.bind(_lib, _SHIM, strict=True)
        assert _lib.ref_sizeof_full_params() == C.sizeof(abi.whisper_full_params)
        assert _lib.ref_sizeof_token_data() == C.sizeof(abi.whisper_token_data)
    return _lib


def fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))

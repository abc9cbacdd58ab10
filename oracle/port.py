"""TEST INFRASTRUCTURE — ctypes driver for this repository's CPU restatement (oracle/whisper_port.cpp)."""
from __future__ import annotations

import ctypes as C
import pathlib

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = HERE / "libwhisper_port.so"

_PROTO = [
    ("port_init", C.c_void_p, [C.c_void_p, C.c_size_t, C.c_int]),
    ("port_free", None, [C.c_void_p]),
    ("port_set_threads", None, [C.c_void_p, C.c_int]),
    ("port_pcm_to_mel", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    ("port_set_mel", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]),
    ("port_mel_dims", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("port_encode", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("port_decode", C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(C.c_float)]),
    ("port_get_tensor", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int]),
    ("port_tables", None, [C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]),
    ("port_n_vocab", C.c_int, [C.c_void_p]),
    ("port_hparam", C.c_int, [C.c_void_p, C.c_int]),
    ("port_greedy", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int32, C.c_int, C.POINTER(C.c_int32)]),
]


def available() -> bool:
    return LIB_PATH.exists()


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(LIB_PATH))
        for name, res, args in _PROTO:
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class PortSide:
    """Same driving surface as tests/stage_compare.RefSide, returning tensors in the product's layouts."""

    def __init__(self, model_bytes: bytes, n_threads: int = 8):
        self.lib = lib()
        buf = C.create_string_buffer(model_bytes, len(model_bytes))
        self.ctx = self.lib.port_init(C.cast(buf, C.c_void_p), len(model_bytes), n_threads)
        assert self.ctx, "port: model load failed"
        hp = [self.lib.port_hparam(self.ctx, i) for i in range(10)]
        self.NV, self.n_audio_ctx, self.S, self.H, self.La, self.n_text_ctx, _, _, self.L, self.n_mels = hp
        self.sot = 50257 if self.NV < 51865 else 50258

    def close(self):
        if self.ctx:
            self.lib.port_free(self.ctx)
        self.ctx = None

    def tensor(self, name):
        n = self.lib.port_get_tensor(self.ctx, name.encode(), None, 0)
        out = np.empty(n, np.float32)
        self.lib.port_get_tensor(self.ctx, name.encode(), _fptr(out), n)
        return out

    def mel(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.float32)
        self.lib.port_pcm_to_mel(self.ctx, _fptr(pcm), pcm.size)
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.lib.port_mel_dims(self.ctx, C.byref(a), C.byref(b), C.byref(c))
        return self.tensor("mel").reshape(c.value, a.value), b.value

    def set_mel(self, mel):
        m = np.ascontiguousarray(mel, np.float32)
        assert self.lib.port_set_mel(self.ctx, _fptr(m), m.shape[1], m.shape[0]) == 0

    def encode(self, offset=0, audio_ctx=0):
        self.lib.port_encode(self.ctx, offset, audio_ctx)
        T = audio_ctx if audio_ctx > 0 else self.n_audio_ctx
        S, L = self.S, self.L
        return {"embd_conv": self.tensor("embd_conv").reshape(T, S), "embd_enc": self.tensor("embd_enc").reshape(T, S),
                "cross_k": self.tensor("cross_k").reshape(L, T, S), "cross_v": self.tensor("cross_v").reshape(L, T, S)}

    def decode(self, tokens, n_past):
        t = np.asarray(tokens, np.int32)
        out = np.empty(self.NV, np.float32)
        self.lib.port_decode(self.ctx, t.ctypes.data_as(C.POINTER(C.c_int32)), t.size, n_past, _fptr(out))
        return out

"""Loader for the product library (godot-whisper_amd/libwhisper_mi355.so, built in-tree by
csrc/Makefile).  There is no CPU fallback: a missing library or a box without a HIP device is an
error, never a silent downgrade."""
from __future__ import annotations

import ctypes as C
import os
import pathlib

import numpy as np

from . import abi

LIB_PATH = pathlib.Path(__file__).resolve().parent / "libwhisper_mi355.so"
# A/B builds of the same library (scratch/ probes only; the driver and the tests never set this)
if os.environ.get("WMI_LIB_PATH"):
    LIB_PATH = pathlib.Path(os.environ["WMI_LIB_PATH"]).resolve()

DEVICE_API = [
    ("wmi_device_count", C.c_int, []),
    ("wmi_version", C.c_char_p, []),
    ("wmi_init_from_buffer_on_device", C.c_void_p, [C.c_void_p, C.c_size_t, C.c_int]),
    ("wmi_init_host_only", C.c_void_p, [C.c_void_p, C.c_size_t]),
    ("wmi_pcm_to_mel_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    ("wmi_full_device_pcm", C.c_int, [C.c_void_p, abi.whisper_full_params, C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    ("wmi_full_batch", C.c_int, [C.c_void_p, abi.whisper_full_params, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int]),
    ("wmi_set_lockstep_exact", None, [C.c_int]),
    ("wmi_batch_select", C.c_int, [C.c_void_p, C.c_int]),
    ("wmi_batch_chunk_mode", C.c_int, [C.c_void_p, C.c_int]),
    ("wmi_set_batch_replicas", C.c_int, [C.c_void_p, C.c_int]),
    ("wmi_set_lockstep_groups", C.c_int, [C.c_void_p, C.c_int]),
    ("wmi_selftest_pool", C.c_int64, [C.c_int, C.c_int]),
    ("wmi_selftest_seqsum", C.c_int, [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("wmi_get_batch_timings", None, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    ("wmi_set_audio_ctx", C.c_int, [C.c_void_p, C.c_int]),
    ("wmi_get_tensor", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int]),
    ("wmi_mel_dims", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("wmi_get_timings", None, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    ("wmi_stream", C.c_void_p, [C.c_void_p]),
    ("wmi_process_logits", C.c_int, [C.c_void_p, abi.whisper_full_params, C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                      C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("wmi_sample_draws", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int,
                                    C.POINTER(abi.whisper_token_data)]),
    ("wmi_selftest_proj", C.c_double, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    ("wmi_bench_kernel", C.c_double, [C.c_void_p, C.c_int, C.c_int]),
    ("wmi_reload_knobs", None, []),
    ("wmi_pair_status", C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    ("wmi_step_stamps", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]),
    ("wmi_encoder_gemm_stamps", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int]),
    ("wmi_pool_init", C.c_void_p, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.c_int]),
    ("wmi_pool_free", None, [C.c_void_p]),
    ("wmi_pool_size", C.c_int, [C.c_void_p]),
    ("wmi_pool_context", C.c_void_p, [C.c_void_p, C.c_int]),
    ("wmi_pool_full", C.c_int, [C.c_void_p, abi.whisper_full_params, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]),
    ("wmi_pool_select", C.c_void_p, [C.c_void_p, C.c_int]),
    ("wmi_pool_device_time_us", C.c_int64, [C.c_void_p, C.c_int]),
    ("wmi_downmix_stereo", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    ("wmi_vad", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    ("wmi_selftest_ts_refine", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("wmi_selftest_resample_plan", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                             C.c_void_p]),
    ("wmi_resample", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    ("wmi_model_header", C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    ("wmi_arena_ptr", C.c_void_p, [C.c_void_p]),
    ("wmi_init_from_header", C.c_void_p, [C.c_void_p, C.c_size_t, C.c_int]),
    ("wmi_arena_commit", C.c_int, [C.c_void_p]),
    ("wmi_weights_pending", C.c_int, [C.c_void_p]),
    ("wmi_weights_bytes", C.c_size_t, [C.c_void_p, C.c_int]),
    ("wmi_selftest_quant", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
]

_lib = None


class BackendUnavailable(RuntimeError):
    pass


def load_library() -> C.CDLL:
    """dlopen the HIP library and bind every symbol include/*.h declares (raises if any is missing)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise BackendUnavailable(
                f"{LIB_PATH} not built - run `python __graft_entry__.py build` (hipcc --offload-arch=gfx950); "
                "this backend has no CPU fallback")
        lib = C.CDLL(str(LIB_PATH))
        abi.bind(lib, abi.WHISPER_API, strict=True)
        abi.bind(lib, DEVICE_API, strict=True)
        _lib = lib
    return _lib


def require_gpu() -> C.CDLL:
    lib = load_library()
    if lib.wmi_device_count() <= 0:
        raise BackendUnavailable("no HIP device visible: libwhisper_mi355 needs an AMD GPU (gfx950)")
    return lib


def get_tensor(lib, ctx, name: str) -> np.ndarray:
    n = lib.wmi_get_tensor(ctx, name.encode(), None, 0)
    if n < 0:
        raise KeyError(name)
    out = np.empty(n, dtype=np.float32)
    got = lib.wmi_get_tensor(ctx, name.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), n)
    assert got == n, (name, got, n)
    return out


def silence_logs(lib):
    """Install a no-op log callback (keeps a reference so it is not collected)."""
    cb = abi.ggml_log_callback(lambda lvl, txt, ud: None)
    lib.whisper_log_set(C.cast(cb, C.c_void_p), None)
    lib._wmi_log_cb = cb
    return cb

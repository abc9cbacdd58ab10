"""The drop-in boundary: libwhisper_mi355.so loads without a GPU, exports every symbol include/*.h declares,
agrees with the reference on by-value struct layouts and default parameters, and refuses to run without a device."""
import ctypes as C
import pathlib
import re

import numpy as np
import pytest

from godot_whisper_amd import abi, runtime

ROOT = pathlib.Path(__file__).resolve().parent.parent


def declared_symbols():
    names = []
    for h in ("whisper_mi355.h", "wmi_device.h"):
        text = "\n".join(l for l in (ROOT / "include" / h).read_text().splitlines() if not l.lstrip().startswith("#"))
        names += re.findall(r"WHISPER_API\s+[^;(]*?\b(\w+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = runtime.load_library()
    names = declared_symbols()
    assert len(names) > 70
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    for n in abi.HOST_SYMBOLS:                       # the 11 entry points the Godot host binds
        assert n in names


def test_by_value_struct_layouts():
    # whisper_full_params / whisper_token_data cross the boundary by value (W/whisper.h:433-526, 91-106);
    # 256 / 48 bytes are what the compiled reference reports (oracle/ref_shim.cpp ref_sizeof_*)
    assert C.sizeof(abi.whisper_full_params) == 256
    assert C.sizeof(abi.whisper_token_data) == 48
    assert abi.whisper_token_data.t0.offset == 24 and abi.whisper_token_data.vlen.offset == 40
    assert abi.whisper_full_params.initial_prompt.offset == 64 and abi.whisper_full_params.language.offset == 88
    assert abi.whisper_full_params.grammar_penalty.offset == 248


def test_default_params_match_reference_defaults():
    lib = runtime.load_library()
    p = lib.whisper_full_default_params(abi.WHISPER_SAMPLING_GREEDY)          # W/whisper.cpp:4311-4410
    assert (p.strategy, p.n_max_text_ctx, p.offset_ms, p.duration_ms) == (0, 16384, 0, 0)
    assert (p.translate, p.no_context, p.no_timestamps, p.single_segment) == (False, True, False, False)
    assert (p.token_timestamps, p.max_len, p.split_on_word, p.max_tokens, p.audio_ctx) == (False, 0, False, 0, 0)
    assert p.language == b"en" and not p.detect_language and p.suppress_blank and not p.suppress_non_speech_tokens
    np.testing.assert_allclose([p.thold_pt, p.thold_ptsum, p.temperature, p.max_initial_ts, p.length_penalty, p.temperature_inc,
                                p.entropy_thold, p.logprob_thold, p.no_speech_thold, p.grammar_penalty],
                               [0.01, 0.01, 0.0, 1.0, -1.0, 0.2, 2.4, -1.0, 0.6, 100.0], rtol=1e-6)
    assert p.greedy.best_of == 5 and p.beam_search.beam_size == -1
    b = lib.whisper_full_default_params(abi.WHISPER_SAMPLING_BEAM_SEARCH)
    assert b.greedy.best_of == -1 and b.beam_search.beam_size == 5
    assert lib.whisper_context_default_params().use_gpu is True


def test_reference_defaults_agree(ref_lib):
    lib = runtime.load_library()
    for strat in (0, 1):
        a = lib.whisper_full_default_params(strat); b = ref_lib.whisper_full_default_params(strat)
        for name, _ in abi.whisper_full_params._fields_:
            if name in ("n_threads", "greedy", "beam_search"):
                continue
            va, vb = getattr(a, name), getattr(b, name)
            if name == "prompt_tokens":              # ctypes pointer objects: compare NULL-ness
                va, vb = bool(va), bool(vb)
            assert va == vb, name
        assert (a.greedy.best_of, a.beam_search.beam_size, a.beam_search.patience) == (b.greedy.best_of, b.beam_search.beam_size, b.beam_search.patience)


def test_free_is_null_safe_and_log_callback_installs():
    lib = runtime.load_library()
    lib.whisper_free(None)
    seen = []
    cb = abi.ggml_log_callback(lambda lvl, txt, ud: seen.append((lvl, txt)))
    lib.whisper_log_set(C.cast(cb, C.c_void_p), None)
    bad = C.create_string_buffer(b"not a model", 11)
    assert not lib.whisper_init_from_buffer_with_params(C.cast(bad, C.c_void_p), 11, abi.whisper_context_params(True))
    assert seen and seen[-1][0] == abi.GGML_LOG_LEVEL_ERROR
    runtime.silence_logs(lib)
    assert b"HIP = 1" in lib.whisper_print_system_info() and b"CPU_FALLBACK = 0" in lib.whisper_print_system_info()


def test_no_gpu_means_loud_failure_not_fallback():
    lib = runtime.load_library()
    if lib.wmi_device_count() > 0:
        pytest.skip("a GPU is visible here")
    from godot_whisper_amd import synth
    seen = []
    cb = abi.ggml_log_callback(lambda lvl, txt, ud: seen.append(txt))
    lib.whisper_log_set(C.cast(cb, C.c_void_p), None)
    mb = synth.make_model("micro.en", seed=1)
    buf = C.create_string_buffer(mb, len(mb))
    assert not lib.whisper_init_from_buffer_with_params(C.cast(buf, C.c_void_p), len(mb), abi.whisper_context_params(True))
    assert any(b"requires an AMD GPU" in s for s in seen)
    with pytest.raises(runtime.BackendUnavailable):
        runtime.require_gpu()
    runtime.silence_logs(lib)

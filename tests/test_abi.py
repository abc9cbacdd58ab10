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


def test_every_function_of_the_reference_header_is_declared_and_exported():
    """tests/golden/whisper_h_api.txt lists the 104 functions of W/whisper.h (v1.5.4; made by golden/make_api_list.py):
    a host written against the reference header finds every one of them here."""
    lib = runtime.load_library()
    want = (ROOT / "tests" / "golden" / "whisper_h_api.txt").read_text().split()
    assert len(want) == 104
    declared = set(declared_symbols())
    assert not [n for n in want if n not in declared]
    assert not [n for n in want if not hasattr(lib, n)]
    bound = {n for n, _, _ in abi.WHISPER_API}
    assert not [n for n in want if n not in bound]              # and the ctypes table the tests use covers them all


def test_parameter_helpers_language_names_and_fixed_answers():
    """Entry points that need no device (W/whisper.cpp:3122-3134, 3391-3401, 3558-3567, 4295-4309)."""
    lib = runtime.load_library()
    runtime.silence_logs(lib)
    cp = lib.whisper_context_default_params_by_ref()
    assert cp.contents.use_gpu is True
    lib.whisper_free_context_params(cp)
    for strategy in (abi.WHISPER_SAMPLING_GREEDY, abi.WHISPER_SAMPLING_BEAM_SEARCH):
        fp = lib.whisper_full_default_params_by_ref(strategy)
        val = lib.whisper_full_default_params(strategy)
        assert bytes(fp.contents) == bytes(val)                  # the same defaults, field for field
        lib.whisper_free_params(fp)
    lib.whisper_free_params(None); lib.whisper_free_context_params(None); lib.whisper_free_state(None)
    assert lib.whisper_lang_str_full(0) == b"english" and lib.whisper_lang_str_full(2) == b"german"
    assert lib.whisper_lang_str_full(99) == b"cantonese" and lib.whisper_lang_str_full(100) is None
    assert all(lib.whisper_lang_id(lib.whisper_lang_str_full(i)) == i for i in range(100))     # full names resolve like codes
    assert lib.whisper_ctx_init_openvino_encoder(None, None, None, None) == 1
    pcm = np.zeros(16000, np.float32)
    assert lib.whisper_pcm_to_mel_phase_vocoder(None, pcm.ctypes.data_as(C.POINTER(C.c_float)), pcm.size, 1) == -1


def test_loader_constructors_drain_and_close_the_loader_and_fail_loudly_without_a_gpu():
    """whisper_init*(loader): read callbacks are drained, close is called exactly once, and without a device the result is
    NULL — never a context that would compute on the host (W/whisper.cpp:3253-3269, 3301-3314)."""
    lib = runtime.load_library()
    if lib.wmi_device_count() > 0:
        pytest.skip("a GPU is visible here")
    runtime.silence_logs(lib)
    from godot_whisper_amd import synth
    mb = synth.make_model("micro.en", seed=1)
    for ctor in ("whisper_init", "whisper_init_no_state", "whisper_init_with_params", "whisper_init_with_params_no_state"):
        pos = [0]; closed = [0]
        L = abi.whisper_model_loader
        def rd(_, out, n):
            k = min(n, len(mb) - pos[0]); C.memmove(out, mb[pos[0]:pos[0] + k], k); pos[0] += k; return k
        loader = L(None, L._fields_[1][1](rd), L._fields_[2][1](lambda _: pos[0] >= len(mb)), L._fields_[3][1](lambda _: closed.__setitem__(0, closed[0] + 1)))
        args = [C.cast(C.pointer(loader), C.c_void_p)] + ([abi.whisper_context_params(True)] if "with_params" in ctor else [])
        assert not getattr(lib, ctor)(*args)
        assert pos[0] == len(mb) and closed[0] == 1, ctor
    bad = C.create_string_buffer(mb, len(mb))
    for ctor in ("whisper_init_from_buffer", "whisper_init_from_buffer_no_state"):
        assert not getattr(lib, ctor)(C.cast(bad, C.c_void_p), len(mb))
    assert not lib.whisper_init_from_file(b"/nonexistent/model.bin") and not lib.whisper_init_from_file_no_state(b"/nonexistent/model.bin")


def test_host_only_context_cannot_create_states():
    lib = runtime.load_library()
    runtime.silence_logs(lib)
    from godot_whisper_amd import synth
    mb = synth.make_model("micro.en", seed=1)
    buf = C.create_string_buffer(mb, len(mb))
    ctx = lib.wmi_init_host_only(C.cast(buf, C.c_void_p), len(mb))
    try:
        assert not lib.whisper_init_state(ctx)
        pcm = np.zeros(16000, np.float32); fp = pcm.ctypes.data_as(C.POINTER(C.c_float))
        p = lib.whisper_full_default_params(0)
        assert lib.whisper_full_with_state(ctx, None, p, fp, pcm.size) == -1
        assert lib.whisper_full_parallel(ctx, p, fp, pcm.size, 2) == -1          # cannot allocate the second state
        assert lib.whisper_full_parallel(ctx, p, fp, pcm.size, 1) == -2          # = whisper_full: no compute path
    finally:
        lib.whisper_free(ctx)


def test_host_worker_pool_runs_every_task_exactly_once():
    """pool.cpp (segment emission on a few persistent threads): many small jobs back to back, nested calls inline."""
    lib = runtime.load_library()
    for n, reps in ((1, 50), (3, 2000), (17, 2000), (64, 500)):
        assert lib.wmi_selftest_pool(n, reps) == reps * n * (n + 1) // 2, (n, reps)


def test_sequential_sum_is_exact():
    """csrc/full.cpp: seq_sum_f32 — the token timestamps' window sums are the reference's left-to-right f32 sums
    (W/whisper.cpp:6506-6515); the blocked evaluation (integer additions inside a binade, plain additions around ties and binade
    crossings) must give the same BITS as the one-by-one loop on anything it is handed."""
    lib = runtime.load_library()
    rng = np.random.default_rng(11)
    cases = []
    for n in (0, 1, 7, 63, 64, 65, 127, 128, 1000, 4001, 36000, 100003):
        cases.append(np.abs(rng.standard_normal(n)).astype(np.float32) * np.float32(0.05))             # envelope-like
    cases.append((rng.integers(0, 1 << 12, 50000) / np.float32(1 << 16)).astype(np.float32))           # coarse grid: ties by the thousand
    cases.append((rng.integers(0, 4, 70000) * np.float32(2.0 ** -20)).astype(np.float32))               # tiny steps, long runs of zeros
    cases.append(np.full(40000, 2.0 ** -13, np.float32))                                                # exact half-ulps once the sum passes 1024
    cases.append(np.concatenate([np.full(300, 1e-42, np.float32), np.abs(rng.standard_normal(5000)).astype(np.float32)]))   # denormal start
    cases.append((np.abs(rng.standard_normal(30000)) * 10.0 ** rng.uniform(-8, 6, 30000)).astype(np.float32))       # 14 decades: crossings everywhere
    cases.append(rng.standard_normal(20000).astype(np.float32))                                         # negatives: plain path
    x = np.abs(rng.standard_normal(9000)).astype(np.float32); x[4000] = np.float32(3e38); x[4500] = np.float32(3e38)
    cases.append(x)                                                                                     # overflow to inf
    x = np.abs(rng.standard_normal(3000)).astype(np.float32); x[1234] = np.nan
    cases.append(x)
    for k, x in enumerate(cases):
        for off in (0, 1, 3):                                                                           # unaligned starts
            xs = np.ascontiguousarray(x[off:]) if x.size > off else x
            a, b = C.c_float(), C.c_float()
            assert lib.wmi_selftest_seqsum(xs.ctypes.data_as(C.POINTER(C.c_float)), xs.size, C.byref(a), C.byref(b)) == 0
            ab = np.array([a.value, b.value], np.float32).view(np.uint32)
            assert ab[0] == ab[1] or (np.isnan(a.value) and np.isnan(b.value)), (k, off, a.value, b.value)
            if k < 12 and xs.size:                                                                      # and the loop is what numpy's cumulative sum is
                assert np.float32(b.value) == np.cumsum(xs, dtype=np.float32)[-1]


def test_malformed_model_files_are_rejected_not_trusted():
    """The model buffer is untrusted input (the reference rejects all of these with a load error, W/whisper.cpp:1113-1600):
    truncated payloads, zero / negative / absurd hyper-parameters (a zero head count was a division by zero), lengths that
    would wrap a bounds check, block-quantised rows that are not a whole number of blocks, duplicate tensors."""
    import struct
    from godot_whisper_amd import synth
    lib = runtime.load_library()
    runtime.silence_logs(lib)

    def loads(b):
        buf = C.create_string_buffer(bytes(b), len(b))
        ctx = lib.wmi_init_host_only(C.cast(buf, C.c_void_p), len(b))
        if ctx:
            lib.whisper_free(ctx)
        return bool(ctx)

    good = bytearray(synth.make_model("micro.en", seed=1))
    assert loads(good)
    assert not loads(good[:len(good) - 7])                                  # last tensor truncated
    assert not loads(good[:40])                                             # header truncated
    assert not loads(b"\x00" * 64)                                          # bad magic
    for field, val in ((3, 0), (3, -2), (7, 0), (0, 0), (0, 1 << 20), (1, -5), (5, 0), (9, 0), (2, 1 << 30), (10, 55)):
        bad = bytearray(good); struct.pack_into("<i", bad, 4 + 4 * field, val)
        assert not loads(bad), (field, val)
    bad = bytearray(good); struct.pack_into("<i", bad, 4 + 44, 81)          # n_mel of the filterbank != n_mels
    assert not loads(bad)
    # first tensor record: find it behind the vocabulary
    off = 4 + 44
    n_mel, n_fft = struct.unpack_from("<2i", good, off); off += 8 + 4 * n_mel * n_fft
    (nv,) = struct.unpack_from("<i", good, off); off += 4
    for _ in range(nv):
        (ln,) = struct.unpack_from("<I", good, off); off += 4 + ln
    bad = bytearray(good); struct.pack_into("<I", bad, 4 + 44 + 8 + 4 * n_mel * n_fft + 4, 0xFFFFFFF0)   # a vocabulary length that wraps off + len
    assert not loads(bad)
    nd, nl, tt = struct.unpack_from("<3i", good, off)
    bad = bytearray(good); struct.pack_into("<i", bad, off + 12, 0)         # a dimension of 0
    assert not loads(bad)
    bad = bytearray(good); struct.pack_into("<i", bad, off + 12, -4)        # a negative dimension
    assert not loads(bad)
    bad = bytearray(good); struct.pack_into("<i", bad, off, 9)              # n_dims out of range
    assert not loads(bad)
    bad = bytearray(good); struct.pack_into("<i", bad, off + 8, 7)          # claims q5_1 for a payload of another size
    assert not loads(bad)
    assert not loads(good + good[off:off + 12 + 4 * nd + nl + 64])          # trailing garbage / duplicate record

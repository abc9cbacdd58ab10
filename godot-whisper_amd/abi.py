"""ctypes view of the whisper.h C ABI that the Godot host binds (W/whisper.h:79-619).

Struct field order/types follow the reference header exactly because `whisper_full_params`
and `whisper_token_data` cross the boundary BY VALUE (src/speech_to_text.cpp:403,419,431).
The same prototypes are applied to this repository's `libwhisper_mi355.so` and — in tests —
to the compiled reference checker, so both are driven through identical calls.
"""
from __future__ import annotations

import ctypes as C

WHISPER_SAMPLE_RATE = 16000
WHISPER_N_FFT = 400
WHISPER_HOP_LENGTH = 160
WHISPER_CHUNK_SIZE = 30

WHISPER_SAMPLING_GREEDY = 0
WHISPER_SAMPLING_BEAM_SEARCH = 1

GGML_LOG_LEVEL_ERROR, GGML_LOG_LEVEL_WARN, GGML_LOG_LEVEL_INFO, GGML_LOG_LEVEL_DEBUG = 2, 3, 4, 5


class whisper_context_params(C.Structure):  # W/whisper.h:87-89
    _fields_ = [("use_gpu", C.c_bool)]


class whisper_token_data(C.Structure):  # W/whisper.h:91-106
    _fields_ = [
        ("id", C.c_int32), ("tid", C.c_int32),
        ("p", C.c_float), ("plog", C.c_float), ("pt", C.c_float), ("ptsum", C.c_float),
        ("t0", C.c_int64), ("t1", C.c_int64),
        ("vlen", C.c_float),
    ]


class _greedy(C.Structure):
    _fields_ = [("best_of", C.c_int)]


class _beam(C.Structure):
    _fields_ = [("beam_size", C.c_int), ("patience", C.c_float)]


ggml_log_callback = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_void_p)
whisper_logits_filter_callback = C.CFUNCTYPE(
    None, C.c_void_p, C.c_void_p, C.POINTER(whisper_token_data), C.c_int, C.POINTER(C.c_float), C.c_void_p)


class whisper_full_params(C.Structure):  # W/whisper.h:433-526
    _fields_ = [
        ("strategy", C.c_int),
        ("n_threads", C.c_int), ("n_max_text_ctx", C.c_int), ("offset_ms", C.c_int), ("duration_ms", C.c_int),
        ("translate", C.c_bool), ("no_context", C.c_bool), ("no_timestamps", C.c_bool), ("single_segment", C.c_bool),
        ("print_special", C.c_bool), ("print_progress", C.c_bool), ("print_realtime", C.c_bool),
        ("print_timestamps", C.c_bool),
        ("token_timestamps", C.c_bool), ("thold_pt", C.c_float), ("thold_ptsum", C.c_float),
        ("max_len", C.c_int), ("split_on_word", C.c_bool), ("max_tokens", C.c_int),
        ("speed_up", C.c_bool), ("debug_mode", C.c_bool), ("audio_ctx", C.c_int),
        ("tdrz_enable", C.c_bool),
        ("initial_prompt", C.c_char_p), ("prompt_tokens", C.POINTER(C.c_int32)), ("prompt_n_tokens", C.c_int),
        ("language", C.c_char_p), ("detect_language", C.c_bool),
        ("suppress_blank", C.c_bool), ("suppress_non_speech_tokens", C.c_bool),
        ("temperature", C.c_float), ("max_initial_ts", C.c_float), ("length_penalty", C.c_float),
        ("temperature_inc", C.c_float), ("entropy_thold", C.c_float), ("logprob_thold", C.c_float),
        ("no_speech_thold", C.c_float),
        ("greedy", _greedy), ("beam_search", _beam),
        ("new_segment_callback", C.c_void_p), ("new_segment_callback_user_data", C.c_void_p),
        ("progress_callback", C.c_void_p), ("progress_callback_user_data", C.c_void_p),
        ("encoder_begin_callback", C.c_void_p), ("encoder_begin_callback_user_data", C.c_void_p),
        ("abort_callback", C.c_void_p), ("abort_callback_user_data", C.c_void_p),
        ("logits_filter_callback", C.c_void_p), ("logits_filter_callback_user_data", C.c_void_p),
        ("grammar_rules", C.c_void_p), ("n_grammar_rules", C.c_size_t), ("i_start_rule", C.c_size_t),
        ("grammar_penalty", C.c_float),
    ]


# (name, restype, argtypes) — whisper.h (all 104 functions of v1.5.4) as exported by libwhisper_mi355.so.
# The first block is exactly what the GDExtension host calls.
WHISPER_API = [
    ("whisper_init_from_buffer_with_params", C.c_void_p, [C.c_void_p, C.c_size_t, whisper_context_params]),
    ("whisper_free", None, [C.c_void_p]),
    ("whisper_print_system_info", C.c_char_p, []),
    ("whisper_full_default_params", whisper_full_params, [C.c_int]),
    ("whisper_full", C.c_int, [C.c_void_p, whisper_full_params, C.POINTER(C.c_float), C.c_int]),
    ("whisper_full_n_segments", C.c_int, [C.c_void_p]),
    ("whisper_full_n_tokens", C.c_int, [C.c_void_p, C.c_int]),
    ("whisper_full_get_segment_text", C.c_char_p, [C.c_void_p, C.c_int]),
    ("whisper_full_get_token_text", C.c_char_p, [C.c_void_p, C.c_int, C.c_int]),
    ("whisper_full_get_token_data", whisper_token_data, [C.c_void_p, C.c_int, C.c_int]),
    ("whisper_log_set", None, [C.c_void_p, C.c_void_p]),
    # --- rest of whisper.h used by the comparison harness / bench (W/examples/bench/bench.cpp:64-135)
    ("whisper_init_from_file_with_params", C.c_void_p, [C.c_char_p, whisper_context_params]),
    ("whisper_context_default_params", whisper_context_params, []),
    ("whisper_pcm_to_mel", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]),
    ("whisper_set_mel", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]),
    ("whisper_encode", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("whisper_decode", C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int]),
    ("whisper_tokenize", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32), C.c_int]),
    ("whisper_get_logits", C.POINTER(C.c_float), [C.c_void_p]),
    ("whisper_n_len", C.c_int, [C.c_void_p]),
    ("whisper_n_vocab", C.c_int, [C.c_void_p]),
    ("whisper_n_text_ctx", C.c_int, [C.c_void_p]),
    ("whisper_n_audio_ctx", C.c_int, [C.c_void_p]),
    ("whisper_is_multilingual", C.c_int, [C.c_void_p]),
    ("whisper_model_n_vocab", C.c_int, [C.c_void_p]),
    ("whisper_model_n_audio_ctx", C.c_int, [C.c_void_p]),
    ("whisper_model_n_audio_state", C.c_int, [C.c_void_p]),
    ("whisper_model_n_audio_head", C.c_int, [C.c_void_p]),
    ("whisper_model_n_audio_layer", C.c_int, [C.c_void_p]),
    ("whisper_model_n_text_ctx", C.c_int, [C.c_void_p]),
    ("whisper_model_n_text_state", C.c_int, [C.c_void_p]),
    ("whisper_model_n_text_head", C.c_int, [C.c_void_p]),
    ("whisper_model_n_text_layer", C.c_int, [C.c_void_p]),
    ("whisper_model_n_mels", C.c_int, [C.c_void_p]),
    ("whisper_model_ftype", C.c_int, [C.c_void_p]),
    ("whisper_model_type", C.c_int, [C.c_void_p]),
    ("whisper_model_type_readable", C.c_char_p, [C.c_void_p]),
    ("whisper_token_to_str", C.c_char_p, [C.c_void_p, C.c_int32]),
    ("whisper_token_eot", C.c_int32, [C.c_void_p]),
    ("whisper_token_sot", C.c_int32, [C.c_void_p]),
    ("whisper_token_solm", C.c_int32, [C.c_void_p]),
    ("whisper_token_prev", C.c_int32, [C.c_void_p]),
    ("whisper_token_nosp", C.c_int32, [C.c_void_p]),
    ("whisper_token_not", C.c_int32, [C.c_void_p]),
    ("whisper_token_beg", C.c_int32, [C.c_void_p]),
    ("whisper_token_lang", C.c_int32, [C.c_void_p, C.c_int]),
    ("whisper_token_translate", C.c_int32, [C.c_void_p]),
    ("whisper_token_transcribe", C.c_int32, [C.c_void_p]),
    ("whisper_lang_max_id", C.c_int, []),
    ("whisper_lang_id", C.c_int, [C.c_char_p]),
    ("whisper_lang_str", C.c_char_p, [C.c_int]),
    ("whisper_lang_auto_detect", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    ("whisper_full_lang_id", C.c_int, [C.c_void_p]),
    ("whisper_full_get_segment_t0", C.c_int64, [C.c_void_p, C.c_int]),
    ("whisper_full_get_segment_t1", C.c_int64, [C.c_void_p, C.c_int]),
    ("whisper_full_get_token_id", C.c_int32, [C.c_void_p, C.c_int, C.c_int]),
    ("whisper_full_get_token_p", C.c_float, [C.c_void_p, C.c_int, C.c_int]),
    ("whisper_print_timings", None, [C.c_void_p]),
    ("whisper_reset_timings", None, [C.c_void_p]),
    # --- the remaining constructors, caller-owned states, whisper_full_parallel, bench (W/whisper.h:150-207, 234-367, 529-611)
    ("whisper_init_with_params", C.c_void_p, [C.c_void_p, whisper_context_params]),
    ("whisper_init_from_file_with_params_no_state", C.c_void_p, [C.c_char_p, whisper_context_params]),
    ("whisper_init_from_buffer_with_params_no_state", C.c_void_p, [C.c_void_p, C.c_size_t, whisper_context_params]),
    ("whisper_init_with_params_no_state", C.c_void_p, [C.c_void_p, whisper_context_params]),
    ("whisper_init_from_file", C.c_void_p, [C.c_char_p]),
    ("whisper_init_from_buffer", C.c_void_p, [C.c_void_p, C.c_size_t]),
    ("whisper_init", C.c_void_p, [C.c_void_p]),
    ("whisper_init_from_file_no_state", C.c_void_p, [C.c_char_p]),
    ("whisper_init_from_buffer_no_state", C.c_void_p, [C.c_void_p, C.c_size_t]),
    ("whisper_init_no_state", C.c_void_p, [C.c_void_p]),
    ("whisper_context_default_params_by_ref", C.POINTER(whisper_context_params), []),
    ("whisper_full_default_params_by_ref", C.POINTER(whisper_full_params), [C.c_int]),
    ("whisper_free_params", None, [C.POINTER(whisper_full_params)]),
    ("whisper_free_context_params", None, [C.POINTER(whisper_context_params)]),
    ("whisper_ctx_init_openvino_encoder", C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]),
    ("whisper_init_state", C.c_void_p, [C.c_void_p]),
    ("whisper_free_state", None, [C.c_void_p]),
    ("whisper_pcm_to_mel_with_state", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]),
    ("whisper_set_mel_with_state", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]),
    ("whisper_encode_with_state", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    ("whisper_decode_with_state", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int]),
    ("whisper_lang_auto_detect_with_state", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    ("whisper_full_with_state", C.c_int, [C.c_void_p, C.c_void_p, whisper_full_params, C.POINTER(C.c_float), C.c_int]),
    ("whisper_pcm_to_mel_phase_vocoder", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]),
    ("whisper_pcm_to_mel_phase_vocoder_with_state", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]),
    ("whisper_full_parallel", C.c_int, [C.c_void_p, whisper_full_params, C.POINTER(C.c_float), C.c_int, C.c_int]),
    ("whisper_n_len_from_state", C.c_int, [C.c_void_p]),
    ("whisper_get_logits_from_state", C.POINTER(C.c_float), [C.c_void_p]),
    ("whisper_full_n_segments_from_state", C.c_int, [C.c_void_p]),
    ("whisper_full_lang_id_from_state", C.c_int, [C.c_void_p]),
    ("whisper_full_get_segment_t0_from_state", C.c_int64, [C.c_void_p, C.c_int]),
    ("whisper_full_get_segment_t1_from_state", C.c_int64, [C.c_void_p, C.c_int]),
    ("whisper_full_get_segment_speaker_turn_next", C.c_bool, [C.c_void_p, C.c_int]),
    ("whisper_full_get_segment_speaker_turn_next_from_state", C.c_bool, [C.c_void_p, C.c_int]),
    ("whisper_full_get_segment_text_from_state", C.c_char_p, [C.c_void_p, C.c_int]),
    ("whisper_full_n_tokens_from_state", C.c_int, [C.c_void_p, C.c_int]),
    ("whisper_full_get_token_text_from_state", C.c_char_p, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    ("whisper_full_get_token_id_from_state", C.c_int32, [C.c_void_p, C.c_int, C.c_int]),
    ("whisper_full_get_token_data_from_state", whisper_token_data, [C.c_void_p, C.c_int, C.c_int]),
    ("whisper_full_get_token_p_from_state", C.c_float, [C.c_void_p, C.c_int, C.c_int]),
    ("whisper_lang_str_full", C.c_char_p, [C.c_int]),
    ("whisper_bench_memcpy", C.c_int, [C.c_int]),
    ("whisper_bench_memcpy_str", C.c_char_p, [C.c_int]),
    ("whisper_bench_ggml_mul_mat", C.c_int, [C.c_int]),
    ("whisper_bench_ggml_mul_mat_str", C.c_char_p, [C.c_int]),
]


class whisper_model_loader(C.Structure):  # W/whisper.h:108-114
    _fields_ = [("context", C.c_void_p),
                ("read", C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("eof", C.CFUNCTYPE(C.c_bool, C.c_void_p)),
                ("close", C.CFUNCTYPE(None, C.c_void_p))]

HOST_SYMBOLS = [n for n, _, _ in WHISPER_API[:11]]


def bind(lib: C.CDLL, table=WHISPER_API, strict: bool = True) -> C.CDLL:
    for name, res, args in table:
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if strict:
                raise
            continue
        fn.restype = res
        fn.argtypes = args
    return lib


# grammar-constrained decoding (W/whisper.h:116-145)
GRETYPE_END, GRETYPE_ALT, GRETYPE_RULE_REF, GRETYPE_CHAR, GRETYPE_CHAR_NOT, GRETYPE_CHAR_RNG_UPPER, GRETYPE_CHAR_ALT = range(7)


class whisper_grammar_element(C.Structure):
    _fields_ = [("type", C.c_int), ("value", C.c_uint32)]


def make_grammar(rules):
    """rules: list of rules, each a list of (type, value) without the closing END.  Returns (pointer-array, n_rules, keepalive)
    for whisper_full_params.grammar_rules / n_grammar_rules; keep `keepalive` referenced while the parameters are in use."""
    arrays = []
    for r in rules:
        arr = (whisper_grammar_element * (len(r) + 1))(*[whisper_grammar_element(t, v) for t, v in r], whisper_grammar_element(GRETYPE_END, 0))
        arrays.append(arr)
    ptrs = (C.POINTER(whisper_grammar_element) * len(arrays))(*[C.cast(a, C.POINTER(whisper_grammar_element)) for a in arrays])
    return ptrs, len(arrays), (arrays, ptrs)

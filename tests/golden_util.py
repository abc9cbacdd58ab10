"""Shared by tests/golden/make_goldens.py (writer, needs the compiled reference) and the tests
(readers): case definitions and the compact summaries that are stored instead of full tensors."""
from __future__ import annotations

import pathlib

import numpy as np

GOLDEN = pathlib.Path(__file__).resolve().parent / "golden"

# name: (model shape, model seed, pcm spec, audio_ctx)
# pcm spec: ("synth", seconds, seed) or ("wav", filename)
CASES = {
    "en30": ("micro.en", 1234, ("synth", 30.0, 1234), 0),
    "ml11": ("micro", 4321, ("synth", 11.0, 7), 0),
    "en4_ctx328": ("micro.en", 1234, ("synth", 4.0, 3), 328),
    "jfk": ("micro.en", 99, ("wav", "jfk.wav"), 0),
}

STAGES = ("mel", "embd_conv", "embd_enc", "cross_k", "cross_v")


def case_inputs(name):
    from godot_whisper_amd import synth
    shape, mseed, spec, actx = CASES[name]
    model = synth.make_model(shape, seed=mseed)
    if spec[0] == "synth":
        pcm = synth.make_pcm(spec[1], seed=spec[2])
    else:
        pcm = synth.read_wav_mono16(GOLDEN / spec[1])
    return model, np.ascontiguousarray(pcm, np.float32), actx


def summary(x: np.ndarray, stride: int = 997) -> dict:
    """Order-independent moments + a strided sample: enough to pin a tensor, ~ (n/stride + 3) numbers."""
    x = np.asarray(x, np.float32).ravel()
    x64 = x.astype(np.float64)
    return {"n": np.int64(x.size), "sum": np.float64(x64.sum()), "sumsq": np.float64((x64 * x64).sum()),
            "sample": x[::stride].copy()}


def logits_summary(l: np.ndarray) -> dict:
    l = np.asarray(l, np.float32)
    top = np.argsort(-l, kind="stable")[:32].astype(np.int32)
    return {"top_ids": top, "top_vals": l[top].copy(), "sample": l[::101].copy(), "sum": np.float64(l.astype(np.float64).sum())}


def flatten(prefix: str, d: dict, out: dict):
    for k, v in d.items():
        out[f"{prefix}/{k}"] = np.asarray(v)


def tokens_array(result: list) -> np.ndarray:
    """host.SpeechToText.transcribe() output -> float64 [n][9]: id tid p plog pt ptsum t0 t1 vlen"""
    rows = [[d["id"], d["tid"], d["p"], d["plog"], d["pt"], d["ptsum"], d["t0"], d["t1"], d["vlen"]] for d in result[1:]]
    return np.asarray(rows, np.float64).reshape(-1, 9)


# parameter variants for whisper_full beyond the Godot host's set
def param_variants(node):
    from godot_whisper_amd import abi
    out = {}
    out["host"] = node.full_params("", 0)
    # library defaults (multi-window, multi-segment, no token timestamps) with the temperature fallback switched off:
    # the fallback decision compares avg_logprob with a threshold and is discontinuous in the logits
    p = node.lib.whisper_full_default_params(abi.WHISPER_SAMPLING_GREEDY)
    p.language = b"en"; p.temperature_inc = 0.0; out["default_greedy"] = p
    p = node.lib.whisper_full_default_params(abi.WHISPER_SAMPLING_GREEDY)      # ... and with it on (compared up to the first split)
    p.language = b"en"; out["default_fallback"] = p
    p = node.lib.whisper_full_default_params(abi.WHISPER_SAMPLING_BEAM_SEARCH)
    p.language = b"en"; p.max_tokens = 12; p.single_segment = True; out["beam5"] = p
    p = node.full_params("", 0)
    p.temperature = 0.4; p.temperature_inc = 0.0; p.greedy.best_of = 2; out["sampled_t04"] = p
    p = node.full_params(" Hello, world! It's 42.", 0); out["host_prompt"] = p
    return out


STREAM_INTERVAL = 0.7


def stream_inputs():
    from godot_whisper_amd import synth
    return synth.make_model("micro", seed=77), synth.make_pcm(9.0, seed=21, gate=True)


PROMPTS = ["Hello, world!", " It's 42 degrees; don't panic.", "multi   space\ttab\nnewline", "naïve café — ünïcode ♪", ""]


# ------------------------------------------------------------------------------------------------ test grammars (W/whisper.h:116-145)
from godot_whisper_amd import abi  # noqa: E402  (the package is registered by conftest before this module is imported)


def _lit(s):
    return [(abi.GRETYPE_CHAR, ord(c)) for c in s]


def colour_list_grammar():
    """root ::= " "? item (", " item)* "."     item ::= "red" | "green" | "blue" | [0-9]+      (repetition as recursive rules)"""
    ALT = [(abi.GRETYPE_ALT, 0)]
    ref = lambda i: [(abi.GRETYPE_RULE_REF, i)]
    digit = [(abi.GRETYPE_CHAR, ord("0")), (abi.GRETYPE_CHAR_RNG_UPPER, ord("9"))]
    return [
        ref(3) + ref(1) + ref(2) + _lit("."),                                   # 0 root
        _lit("red") + ALT + _lit("green") + ALT + _lit("blue") + ALT + ref(4),  # 1 item
        _lit(", ") + ref(1) + ref(2) + ALT,                                     # 2 rest (second alternative empty)
        _lit(" ") + ALT,                                                        # 3 optional space
        digit + ref(5),                                                         # 4 digits
        digit + ref(5) + ALT,                                                   # 5 more digits
    ]


def negated_class_grammar():
    """root ::= [^0-9,x-z]+ — a negated class with a range, an extra member and another range: exercises the partial
    UTF-8 rules (most byte-level tokens of the vocabulary end inside a multi-byte sequence)."""
    cls = [(abi.GRETYPE_CHAR_NOT, ord("0")), (abi.GRETYPE_CHAR_RNG_UPPER, ord("9")), (abi.GRETYPE_CHAR_ALT, ord(",")),
           (abi.GRETYPE_CHAR_ALT, ord("x")), (abi.GRETYPE_CHAR_RNG_UPPER, ord("z"))]
    return [[(abi.GRETYPE_RULE_REF, 1)], cls + [(abi.GRETYPE_RULE_REF, 2)], cls + [(abi.GRETYPE_RULE_REF, 2), (abi.GRETYPE_ALT, 0)]]


def unicode_grammar():
    """root ::= ("é" | [α-ω] | "日本")+ "!" — positive classes above U+007F, reached through partial sequences."""
    ALT = [(abi.GRETYPE_ALT, 0)]
    unit = _lit("é") + ALT + [(abi.GRETYPE_CHAR, ord("α")), (abi.GRETYPE_CHAR_RNG_UPPER, ord("ω"))] + ALT + _lit("日本")
    return [[(abi.GRETYPE_RULE_REF, 1), (abi.GRETYPE_RULE_REF, 2)] + _lit("!"), unit, [(abi.GRETYPE_RULE_REF, 1), (abi.GRETYPE_RULE_REF, 2)] + ALT]

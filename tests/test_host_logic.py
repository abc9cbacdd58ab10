"""Host-side decision logic of the product (tokenizer, logit filters, sampling)
on a host-only context — no GPU needed — against the reference's goldens and, where present, live against
the compiled reference fed identical logits."""
import ctypes as C

import numpy as np
import pytest

import golden_util as gu
import stage_compare as sc
from godot_whisper_amd import abi, host, runtime
from oracle import reflib

G = np.load(gu.GOLDEN / "hotpath.npz")


@pytest.fixture(scope="module")
def lib():
    lib = runtime.load_library()
    runtime.silence_logs(lib)
    return lib


def host_ctx(lib, name):
    model, _, _ = gu.case_inputs(name)
    buf = C.create_string_buffer(model, len(model))
    ctx = lib.wmi_init_host_only(C.cast(buf, C.c_void_p), len(model))
    assert ctx
    return ctx


HIST = [([], 0, 3000), ([100, 200], 0, 3000), (["beg+10"], 1, 20), ([300, "beg+10"], 1, 20), (["beg+5", "beg+5"], 1, 10), ([400] * 3, 0, 3000)]


@pytest.mark.parametrize("name", ["en30", "ml11"])
def test_logit_filters_and_sampling_match_reference(lib, name):
    ctx = host_ctx(lib, name)
    node = host.SpeechToText(lib); node.ctx = ctx
    try:
        nv = lib.whisper_n_vocab(ctx); beg = lib.whisper_token_beg(ctx)
        rng = np.random.default_rng(5)
        for ci, (hist, has_ts, sd) in enumerate(HIST):
            hist = [beg + int(h.split("+")[1]) if isinstance(h, str) else h for h in hist]
            raw = (rng.standard_normal(nv) * 6.0).astype(np.float32)
            if ci == 2:
                raw[beg:] += 9.0
            for temp in (0.0, 0.6):
                p = node.full_params("", 0)
                lo, lp, pr = (np.empty(nv, np.float32) for _ in range(3))
                h = np.asarray(hist, np.int32)
                lib.wmi_process_logits(ctx, p, sc._fptr(raw), h.ctypes.data_as(C.POINTER(C.c_int32)), h.size, has_ts, sd,
                                       C.c_float(temp), sc._fptr(lo), sc._fptr(lp), sc._fptr(pr))
                key = f"{name}/filters/{ci}_t{temp}"
                assert int(np.isneginf(lo).sum()) == int(G[key + "/n_neg_inf"])
                assert int(np.flatnonzero(np.isneginf(lo)).astype(np.int64).sum()) == int(G[key + "/neg_inf_hash"])
                fin = np.isfinite(lp)
                np.testing.assert_allclose(lp[fin].astype(np.float64).sum(), G[key + "/logprob_sum"], rtol=1e-6)
                np.testing.assert_allclose(pr.astype(np.float64).sum(), G[key + "/prob_sum"], rtol=1e-6)
                top = np.argsort(-pr, kind="stable")[:8]
                assert list(top) == list(G[key + "/top_ids"])
                np.testing.assert_allclose(pr[top], G[key + "/top_probs"], rtol=1e-6)
                np.testing.assert_allclose(lp[top], G[key + "/top_logprobs"], rtol=1e-6, atol=1e-6)
                if temp > 0:
                    draws = (abi.whisper_token_data * 12)()
                    lib.wmi_sample_draws(ctx, sc._fptr(pr), sc._fptr(lp), 12, 1, draws)
                    assert [[d.id, d.tid] for d in draws] == G[key + "/draws"].tolist()
                    np.testing.assert_allclose([[d.p, d.plog, d.pt, d.ptsum] for d in draws], G[key + "/draw_stats"], rtol=1e-6)
    finally:
        node.ctx = None
        lib.whisper_free(ctx)


def test_tokenizer_matches_reference(lib):
    ctx = host_ctx(lib, "en30")
    try:
        buf = (C.c_int32 * 1024)()
        for i, text in enumerate(gu.PROMPTS):
            n = lib.whisper_tokenize(ctx, text.encode("utf-8"), buf, 1024)
            assert list(buf[:max(n, 0)]) == G[f"tokenize/{i}"].tolist(), text
        assert lib.whisper_tokenize(ctx, b"one two three four", buf, 2) == -1     # too many tokens -> -1 (W/whisper.cpp:3513)
    finally:
        lib.whisper_free(ctx)


def test_special_tokens_and_languages(lib):
    en = host_ctx(lib, "en30"); ml = host_ctx(lib, "ml11")
    try:
        assert [lib.whisper_token_eot(en), lib.whisper_token_sot(en), lib.whisper_token_beg(en)] == [50256, 50257, 50363]
        assert [lib.whisper_token_eot(ml), lib.whisper_token_sot(ml), lib.whisper_token_translate(ml),
                lib.whisper_token_transcribe(ml), lib.whisper_token_beg(ml)] == [50257, 50258, 50358, 50359, 50364]
        assert lib.whisper_is_multilingual(en) == 0 and lib.whisper_is_multilingual(ml) == 1
        assert lib.whisper_lang_id(b"en") == 0 and lib.whisper_lang_id(b"german") == 2 and lib.whisper_lang_id(b"yue") == 99
        assert lib.whisper_lang_id(b"xx") == -1 and lib.whisper_lang_max_id() == 99
        assert lib.whisper_lang_str(7) == b"ja"
        assert lib.whisper_token_to_str(en, 50257) == b"[_SOT_]" and lib.whisper_token_to_str(en, 50364) == b"[_TT_1]"
        assert lib.whisper_token_to_str(ml, 50259) == b"[_LANG_en]"
    finally:
        lib.whisper_free(en); lib.whisper_free(ml)


def test_host_only_context_has_no_compute_path(lib):
    """The product must fail loudly, never fall back to a CPU computation."""
    ctx = host_ctx(lib, "en30")
    try:
        pcm = np.zeros(16000 * 2, np.float32)
        assert lib.whisper_pcm_to_mel(ctx, sc._fptr(pcm), pcm.size, 1) == -1
        assert lib.whisper_encode(ctx, 0, 1) == -1
        tok = (C.c_int32 * 1)(50257)
        assert lib.whisper_decode(ctx, tok, 1, 0, 1) != 0
        p = lib.whisper_full_default_params(0)
        assert lib.whisper_full(ctx, p, sc._fptr(pcm), pcm.size) == -2
        ptrs = (C.c_void_p * 2)(pcm.ctypes.data, pcm.ctypes.data); lens = (C.c_int * 2)(pcm.size, pcm.size)
        assert lib.wmi_full_batch(ctx, p, ptrs, lens, 2, 0) == -2          # lock-step entry point: same loud failure
        assert lib.wmi_batch_select(ctx, 0) == -1 or lib.whisper_full_n_segments(ctx) == 0
    finally:
        lib.whisper_free(ctx)


@pytest.mark.skipif(not reflib.available(), reason="compiled reference absent")
def test_filters_live_against_reference(lib, ref_lib):
    model, _, _ = gu.case_inputs("en30")
    ctx = host_ctx(lib, "en30")
    rnode = host.SpeechToText(ref_lib); rnode.set_language_model(model)
    node = host.SpeechToText(lib); node.ctx = ctx
    try:
        nv = lib.whisper_n_vocab(ctx)
        rng = np.random.default_rng(11)
        for trial in range(6):
            raw = (rng.standard_normal(nv) * 5.0).astype(np.float32)
            hist = rng.integers(0, 50000, size=trial).astype(np.int32)
            outs = []
            for L, c, fn in ((lib, ctx, lib.wmi_process_logits), (ref_lib, rnode.ctx, ref_lib.ref_process_logits)):
                nd = node if L is lib else rnode
                p = nd.full_params("", 0); p.suppress_non_speech_tokens = bool(trial % 2)
                lo, lp, pr = (np.empty(nv, np.float32) for _ in range(3))
                fn(c, p, sc._fptr(raw), hist.ctypes.data_as(C.POINTER(C.c_int32)), hist.size, 0, 3000, C.c_float(0.2 * trial),
                   sc._fptr(lo), sc._fptr(lp), sc._fptr(pr))
                outs.append((lo.copy(), lp.copy(), pr.copy()))
            for a, b in zip(outs[0], outs[1]):
                assert np.array_equal(a, b)
    finally:
        node.ctx = None; lib.whisper_free(ctx); rnode.close()


# ------------------------------------------------------------------------------------------------ grammar-constrained decoding
@pytest.mark.skipif(not reflib.available(), reason="compiled reference absent")
@pytest.mark.parametrize("which", ["colours", "negated", "unicode"])
def test_grammar_penalties_live_against_reference(lib, ref_lib, which):
    """whisper_full_params.grammar_rules (W/whisper.cpp:3876-4290): after the same token history both libraries must
    penalise exactly the same tokens — logits, log-probabilities and probabilities bit-identical on equal raw logits."""
    model, _, _ = gu.case_inputs("en30" if which != "unicode" else "ml11")
    name = "en30" if which != "unicode" else "ml11"
    ctx = host_ctx(lib, name)
    rnode = host.SpeechToText(ref_lib); rnode.set_language_model(model)
    node = host.SpeechToText(lib); node.ctx = ctx
    rules = {"colours": gu.colour_list_grammar, "negated": gu.negated_class_grammar, "unicode": gu.unicode_grammar}[which]()
    texts = {"colours": ["", " red", " red,", " red, 12", " red, 120, gre", " blue.", " purple", " red, green, blue, 7"],
             "negated": ["", " hello", " hello world", " café", " 12", " a,b"],
             "unicode": ["", "é", "éβ", "日", "日本ω", "é!", "abc"]}[which]
    try:
        nv = lib.whisper_n_vocab(ctx)
        beg = lib.whisper_token_beg(ctx)
        rng = np.random.default_rng(5)
        n_penalised = []
        for k, text in enumerate(texts):
            buf = (C.c_int32 * 64)()
            n = lib.whisper_tokenize(ctx, text.encode("utf-8"), buf, 64)
            assert n >= 0
            hist = list(buf[:n])
            if which == "unicode" and text and k % 2 == 1:                       # leave the history inside a multi-byte sequence
                hist = hist[:-1] if len(hist) > 1 else hist
            if k == 3:
                hist = [beg + 5] + hist                                          # special tokens do not move the grammar (:4275)
            hist = np.asarray(hist, np.int32)
            raw = (rng.standard_normal(nv) * 3.0).astype(np.float32)
            raw[beg:] -= 30.0                                                    # keep the timestamp mass below the text tokens
            outs = []
            for L, c, fn in ((lib, ctx, lib.wmi_process_logits), (ref_lib, rnode.ctx, ref_lib.ref_process_logits)):
                nd = node if L is lib else rnode
                p = nd.full_params("", 0)
                ptrs, n_rules, keep = abi.make_grammar(rules)
                p.grammar_rules = C.cast(ptrs, C.c_void_p); p.n_grammar_rules = n_rules; p.i_start_rule = 0; p.grammar_penalty = 50.0 + k
                lo, lp, pr = (np.empty(nv, np.float32) for _ in range(3))
                fn(c, p, sc._fptr(raw), hist.ctypes.data_as(C.POINTER(C.c_int32)), hist.size, 0, 3000, C.c_float(0.0),
                   sc._fptr(lo), sc._fptr(lp), sc._fptr(pr))
                outs.append((lo.copy(), lp.copy(), pr.copy()))
                del keep
            for a, b in zip(outs[0], outs[1]):
                assert np.array_equal(a, b), (which, text)
            n_penalised.append(int(np.sum(np.isfinite(outs[0][0]) & (outs[0][0] < raw - 1.0))))
        # constrained states penalise most of the vocabulary (the negated class: the tokens holding a banned character); histories
        # the grammar cannot parse leave no stack, and then nothing is penalised (:4227)
        assert max(n_penalised) > (5000 if which == "negated" else 40000) and min(n_penalised) == 0, n_penalised
    finally:
        node.ctx = None; lib.whisper_free(ctx); rnode.close()

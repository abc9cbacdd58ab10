"""-m gpu: the HIP path behind the C ABI against the oracle.

Checker = the compiled reference when its prebuilt library travelled with the snapshot
(oracle/_ref/libwhisper_ref.so), else the CPU restatement (oracle/libwhisper_port.so, bit-exact to the
reference in the build container: tests/test_oracle_port.py).  Token streams are additionally compared with
the committed goldens that the reference itself produced.

Tolerances (floating point; the reference is an f16-operand / f32-accumulate machine, SURVEY App. B — every
rounding point is reproduced, what differs is f32 summation order inside dot products, which flips a small
fraction of f16 roundings):
    log-mel                      |d| <= 5e-7                       (same FFT decomposition and tables)
    encoder tensors / cross K,V  rms(d)/rms(ref) <= 1e-3 ; |d| <= 8e-3
    logits                       rms(d)/rms(ref) <= 1e-3 ; |d| <= 3e-2   (logit rms ~ 6)
    token probabilities          |dp| <= 1e-2
(round 5: held to ~2x what a full GPU run measures — mel 1.2e-7, tensors 5.7e-4 / 3.9e-3, logits 5.7e-4 / 1.55e-2,
profiles/r05a_parity_margins.json; every run prints the margins, tests/conftest.py — the round-4 bounds were 3-5x looser and a
3x regression would have passed)
    token ids, timestamps, text  identical on every greedy case below
"""
import ctypes as C
import os

import numpy as np
import pytest

import golden_util as gu
import stage_compare as sc
from godot_whisper_amd import abi, host, runtime, synth
from oracle import port, reflib

pytestmark = pytest.mark.gpu

G = np.load(gu.GOLDEN / "hotpath.npz")
TOL = {"mel": (5e-7, 1.0), "embd_conv": (8e-3, 1e-3), "embd_enc": (8e-3, 1e-3), "cross_k": (8e-3, 1e-3), "cross_v": (8e-3, 1e-3)}
LOGIT_ABS, LOGIT_RMS = 3e-2, 1e-3
P_LOGIT_ABS = 6e-2              # the logit error the token-probability bound of the streaming replay is derived from (unchanged)


def make_checker(model, ref_lib_or_none):
    if ref_lib_or_none is not None:
        return sc.RefSide(ref_lib_or_none, model)
    return port.PortSide(model)


def sot_prompt(chk, prod):
    sot = prod.lib.whisper_token_sot(prod.ctx)
    if prod.lib.whisper_is_multilingual(prod.ctx):
        return [sot, sot + 1, prod.lib.whisper_token_transcribe(prod.ctx)]
    return [sot]


def assert_logits(lp, lr, what):
    st = sc.err_stats(lp, lr)
    sc.hold("logits rms-rel", st["rms_rel"], LOGIT_RMS, (what, st)); sc.hold("logits max |d|", st["max_abs"], LOGIT_ABS, (what, st))
    # arg-max must agree wherever the checker's top-1/top-2 margin exceeds twice the tolerance
    top2 = np.partition(lr, -2)[-2:]
    if top2[1] - top2[0] > 2 * LOGIT_ABS:
        assert int(np.argmax(lp)) == int(np.argmax(lr)), what


@pytest.mark.parametrize("name", list(gu.CASES))
def test_stages_against_checker(product_lib, checker_lib, name):
    model, pcm, actx = gu.case_inputs(name)
    prod = sc.ProductSide(product_lib, model); chk = make_checker(model, checker_lib)
    try:
        mel_r, org_r = chk.mel(pcm); mel_p, org_p = prod.mel(pcm)
        assert mel_r.shape == mel_p.shape and org_r == org_p
        assert list(mel_p.shape) + [org_p] == list(G[f"{name}/mel_shape"])
        sc.hold("log-mel max |d|", float(np.abs(mel_p - mel_r).max()), TOL["mel"][0])
        er = chk.encode(0, actx); ep = prod.encode(0, actx)
        for k in ("embd_conv", "embd_enc", "cross_k", "cross_v"):
            st = sc.err_stats(ep[k], er[k])
            sc.hold_tensor(k, st, TOL[k])
            # and the reference's golden sample of the same tensor
            g = G[f"{name}/{k}/sample"]
            sc.hold(f"{k} max |d| vs the reference's golden sample", float(np.abs(ep[k].ravel()[::997] - g).max()), TOL[k][0], name)
        prompt = sot_prompt(chk, prod)
        lr = chk.decode(prompt, 0); lp = prod.decode(prompt, 0)
        assert_logits(lp, lr, "prompt")
        sc.hold("logits max |d| vs the reference's golden sample", float(np.abs(lp[::101] - G[f"{name}/logits_prompt/sample"]).max()), LOGIT_ABS, name)
        for i, tok in enumerate(G[f"{name}/fed_tokens"]):
            lr = chk.decode([int(tok)], len(prompt) + i); lp = prod.decode([int(tok)], len(prompt) + i)
            assert_logits(lp, lr, f"step{i}")
        many = prompt + [int(x) for x in (np.arange(11) * 997 + 1000)]          # > 8 rows: MFMA GEMM path of the decoder
        assert_logits(prod.decode(many, 0), chk.decode(many, 0), "batch12")
        sc.hold("logits max |d| vs the reference's golden sample", float(np.abs(prod.decode(many, 0)[::101] - G[f"{name}/logits_batch/sample"]).max()), LOGIT_ABS, name)
    finally:
        prod.close(); chk.close()


GREEDY_VARIANTS = ("host", "default_greedy", "host_prompt")


@pytest.mark.parametrize("name", list(gu.CASES))
def test_whisper_full_token_streams_equal_reference_goldens(product_lib, name):
    model, pcm, actx = gu.case_inputs(name)
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    try:
        for vname, p in gu.param_variants(node).items():
            p.audio_ctx = actx
            r = node.transcribe(pcm, params=p)
            want = G[f"{name}/full_{vname}/tokens"]
            assert node.last_ret == int(G[f"{name}/full_{vname}/ret"])
            got = gu.tokens_array(r) if r else np.zeros((0, 9))
            if vname in GREEDY_VARIANTS:
                # Greedy streams must be IDENTICAL to the reference's, except that an arg-max between two candidates
                # whose probabilities differ by less than the stated fp tolerance is a coin toss for any
                # implementation (SURVEY §7 "margin-aware"): the first disagreement, if any, must be such a near-tie
                # (both sides report the same winning probability within 2e-2), and everything before it must match.
                n = min(len(got), len(want))
                same = got[:n, 0] == want[:n, 0]
                first = n if same.all() else int(np.argmin(same))
                if first < n:
                    assert abs(got[first, 2] - want[first, 2]) <= 2e-2, (vname, first, got[first], want[first])
                    assert first >= 17, (vname, "mismatch inside the host-parameter regime", first)
                else:
                    assert got.shape == want.shape, (vname, got.shape, want.shape)
                g, w = got[:first], want[:first]
                # "most probable timestamp" id = arg-max over the timestamp slice; pt is that maximum as a FRACTION of
                # the timestamp mass, so the arg-max is margin-safe (runner-up <= 1 - pt < pt) exactly where pt > 0.5
                sig = w[:, 4] > 0.55
                assert np.array_equal(g[sig, 1], w[sig, 1]), vname
                assert np.abs(g[:, [2, 4, 5]] - w[:, [2, 4, 5]]).max() <= 1e-2, vname             # p, pt, ptsum
                assert np.abs(g[:, 3] - w[:, 3]).max() <= 5e-2, vname                              # plog
                assert np.array_equal(g[:, 8], w[:, 8]), vname                                     # vlen
                if first == n:
                    # t0, t1 — except the LAST token's t1, where the reference reads one element past the end of its
                    # token vector (W/whisper.cpp:6561, `j < ns - 1`) and the golden holds whatever the heap contained
                    assert np.array_equal(got[:, 6], want[:, 6]) and np.array_equal(got[:-1, 7], want[:-1, 7]), (vname, got[:, 6:8], want[:, 6:8])
                    assert bytes(r[0]) == bytes(G[f"{name}/full_{vname}/text"].tobytes())
                    assert product_lib.whisper_full_n_segments(node.ctx) == int(G[f"{name}/full_{vname}/n_segments"])
            else:
                # beam search / t > 0 draw from mt19937 + discrete_distribution on the probabilities; a draw that
                # lands within the fp tolerance of a CDF step may flip.  Identical draws given identical
                # probabilities is pinned by tests/test_host_logic.py; here the stream must agree up to the first flip.
                assert got.shape[0] > 0
                n = min(len(got), len(want))
                same = got[:n, 0] == want[:n, 0]
                first = int(np.argmin(same)) if not same.all() else n
                assert first >= 1, (vname, got[:, 0], want[:, 0])
                assert np.abs(got[:first, 2] - want[:first, 2]).max() <= 1e-2
    finally:
        node.close()


def test_decoder_projection_paths_agree(product_lib):
    """weight-streaming GEMV vs MFMA GEMM on identical inputs, every fused epilogue (pins the hipcc miscompile)."""
    model, _, _ = gu.case_inputs("en30")
    prod = sc.ProductSide(product_lib, model)
    try:
        for op in (0, 1, 2, 4, 5):
            for n in (1, 3, 8):
                d = product_lib.wmi_selftest_proj(prod.ctx, op, n, 0)
                assert 0.0 <= d <= 2e-3, (op, n, d)
    finally:
        prod.close()


def _hip():
    return C.CDLL("libamdhip64.so")


def test_device_resident_pcm_path_equals_host_path(product_lib):
    model, pcm, actx = gu.case_inputs("en30")
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    hip = _hip()
    try:
        want = gu.tokens_array(node.transcribe(pcm, params=node.full_params("", 0)))
        dptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(dptr), C.c_size_t(pcm.nbytes)) == 0
        assert hip.hipMemcpy(dptr, pcm.ctypes.data_as(C.c_void_p), C.c_size_t(pcm.nbytes), 1) == 0
        ret = product_lib.wmi_full_device_pcm(node.ctx, node.full_params("", 0), dptr, pcm.size, None)
        assert ret == 0
        got = gu.tokens_array(node.collect())
        hip.hipFree(dptr)
        assert np.array_equal(got, want)          # including the token-level timestamps (device energy envelope)
    finally:
        node.close()


def test_error_codes_and_edges(product_lib):
    model, pcm, _ = gu.case_inputs("en30")
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    try:
        p = node.full_params("", 0); p.audio_ctx = 1501
        assert product_lib.whisper_full(node.ctx, p, sc._fptr(pcm), pcm.size) == -5           # W/whisper.cpp:5098-5101
        p = node.full_params("", 0); p.speed_up = True
        assert product_lib.whisper_full(node.ctx, p, sc._fptr(pcm), pcm.size) == -1           # :4973-4976
        p = node.full_params("", 0); p.greedy.best_of = 9
        assert product_lib.whisper_full(node.ctx, p, sc._fptr(pcm), pcm.size) == -4           # :5050-5053
        short = pcm[:8000]                                                                     # 0.5 s < 1 s: ok, no segments
        assert product_lib.whisper_full(node.ctx, node.full_params("", 0), sc._fptr(short), short.size) == 0
        assert product_lib.whisper_full_n_segments(node.ctx) == 0
        # two windows: 40 s of audio -> two encoder passes, seek advances by the decoded timestamps / 30 s
        long = np.concatenate([pcm, pcm[:160000]])
        p = product_lib.whisper_full_default_params(0); p.language = b"en"; p.max_tokens = 8
        assert product_lib.whisper_full(node.ctx, p, sc._fptr(long), long.size) == 0
        assert product_lib.whisper_full_n_segments(node.ctx) >= 1
        t6 = (C.c_int64 * 6)(); n5 = (C.c_int32 * 5)()
        product_lib.wmi_get_timings(node.ctx, t6, n5)
        assert n5[0] >= 2
    finally:
        node.close()


def test_base_en_full_size_against_checker(product_lib, checker_lib):
    """BASELINE.json configs[1]: base.en, one 30 s chunk, greedy — the benchmarked configuration."""
    model = synth.make_model("base.en", seed=1234); pcm = synth.make_pcm(30.0, seed=1234)
    prod = sc.ProductSide(product_lib, model); chk = make_checker(model, checker_lib)
    try:
        mel_r, _ = chk.mel(pcm); mel_p, _ = prod.mel(pcm)
        sc.hold("log-mel max |d|", float(np.abs(mel_p - mel_r).max()), TOL["mel"][0])
        er = chk.encode(0, 0); ep = prod.encode(0, 0)
        for k in er:
            st = sc.err_stats(ep[k], er[k])
            sc.hold_tensor(k, st, TOL[k])
        prompt = sot_prompt(chk, prod)
        lr = chk.decode(prompt, 0); lp = prod.decode(prompt, 0)
        assert_logits(lp, lr, "prompt")
        for i in range(8):
            tok = int(np.argmax(lr[:50256]))
            lr = chk.decode([tok], len(prompt) + i); lp = prod.decode([tok], len(prompt) + i)
            assert_logits(lp, lr, f"step{i}")
        # size-independent properties: the same call twice is bit-identical; a 3-token batch equals 3 single steps
        a = prod.decode(prompt + [1000, 2000], 0).copy()
        b = prod.decode(prompt + [1000, 2000], 0).copy()
        assert np.array_equal(a, b)
        prod.decode(prompt, 0); prod.decode([1000], len(prompt)); c = prod.decode([2000], len(prompt) + 1)
        sc.hold("logits max |d|, 3-row batch vs three one-row steps (product vs product)", float(np.abs(a - c).max()), P_LOGIT_ABS)
    finally:
        prod.close(); chk.close()


def test_base_en_transcription_equals_checker_tokens(product_lib, checker_lib):
    if checker_lib is None:
        pytest.skip("token-stream comparison at base.en size needs the compiled reference (host logic is not in the port)")
    model = synth.make_model("base.en", seed=1234); pcm = synth.make_pcm(30.0, seed=1234)
    outs = []
    for L in (product_lib, checker_lib):
        node = host.SpeechToText(L); node.set_language_model(model)
        outs.append(gu.tokens_array(node.transcribe(pcm, "", 0)))
        node.close()
    got, want = outs
    assert got.shape == want.shape and np.array_equal(got[:, 0], want[:, 0])
    assert np.array_equal(got[:, 6], want[:, 6]) and np.array_equal(got[:-1, 7], want[:-1, 7])
    assert np.abs(got[:, 2] - want[:, 2]).max() <= 1e-2


# ------------------------------------------------------------------------------------------------ block-quantised models
# The reference quantises the activation rows of every projection to 8-bit blocks (SURVEY App. B rule 1).  A quantiser is
# discontinuous: an input that moves by one f32 rounding can flip a quant, i.e. move that element by d = amax / 127, so the
# reference's OWN outputs respond to a 1e-6 relative change of the PCM with ~5e-3 rms on the encoder output and ~1e-2 on the
# logits of a 2-layer model, where the f16 models respond with 3e-4 (measured: tests/test_oracle_quants.py::
# test_reference_sensitivity_of_quantised_models).  No implementation with a different f32 summation order anywhere upstream
# (conv, attention, LayerNorm) can agree with the reference more closely than the reference agrees with itself under such a
# perturbation, so that response is the yardstick: every tensor must be within 2x of it (rms) / 3x (max), and never worse
# than the fixed caps below.  What IS exact is pinned separately: identical q8 quants and scales, exact integer block dots,
# bit-exact dequantisation (tests/test_gpu_quant.py).
Q_CAP = {"tensor_rms": 3e-2, "logit_rms": 5e-2}


def _run_stages(side, pcm, actx, prompt, fed, extra_batches=()):
    side.mel(pcm)
    out = dict(side.encode(0, actx))
    out.pop("embd_conv", None)
    lg = [side.decode(prompt, 0)]
    for i, tok in enumerate(fed):
        lg.append(side.decode([int(tok)], len(prompt) + i))
    for bt in extra_batches:
        lg.append(side.decode(list(bt), 0))
    out["logits"] = lg
    return out


def _assert_within_reference_sensitivity(got, ref, ref_pert, what):
    for k in ("embd_enc", "cross_k", "cross_v"):
        e, n = sc.err_stats(got[k], ref[k]), sc.err_stats(ref_pert[k], ref[k])
        sc.hold(f"quantised {k} rms-rel vs max(f16 bound, 2 x reference self-noise)", e["rms_rel"], min(max(TOL[k][1], 2.0 * n["rms_rel"]), Q_CAP["tensor_rms"]), (what, k, e, n))
        sc.hold(f"quantised {k} max |d| vs max(f16 bound, 3 x reference self-noise)", e["max_abs"], max(TOL[k][0], 3.0 * n["max_abs"]), (what, k, e, n))
    for i, (lp, lr, ln) in enumerate(zip(got["logits"], ref["logits"], ref_pert["logits"])):
        e, n = sc.err_stats(lp, lr), sc.err_stats(ln, lr)
        sc.hold("quantised logits rms-rel vs max(f16 bound, 2 x reference self-noise)", e["rms_rel"], min(max(LOGIT_RMS, 2.0 * n["rms_rel"]), Q_CAP["logit_rms"]), (what, "logits", i, e, n))
        sc.hold("quantised logits max |d| vs max(f16 bound, 3 x reference self-noise)", e["max_abs"], max(LOGIT_ABS, 3.0 * n["max_abs"]), (what, "logits", i, e, n))
        top2 = np.partition(lr, -2)[-2:]                     # arg-max wherever the margin exceeds what the reference itself moves by
        if top2[1] - top2[0] > 2 * max(LOGIT_ABS, 3.0 * n["max_abs"]):
            assert int(np.argmax(lp)) == int(np.argmax(lr)), (what, "argmax", i)


def _quantised_case(product_lib, checker_lib, model, pcm, actx, n_steps, extra, what, ref_threads=4):
    prod = sc.ProductSide(product_lib, model)
    refs = [make_checker(model, checker_lib) for _ in range(2)]
    for r in refs:
        r.n_threads = ref_threads
    try:
        prompt = sot_prompt(refs[0], prod)
        # tokens to feed: the reference's own greedy choices
        refs[0].mel(pcm); refs[0].encode(0, actx)
        fed, lr = [], refs[0].decode(prompt, 0)
        for i in range(n_steps):
            fed.append(int(np.argmax(lr[:50256])))
            lr = refs[0].decode([fed[-1]], len(prompt) + i)
        batches = [prompt + list(b) for b in extra]
        ref = _run_stages(refs[0], pcm, actx, prompt, fed, batches)
        pert = _run_stages(refs[1], (pcm * np.float32(1.0 + 1e-6)).astype(np.float32), actx, prompt, fed, batches)
        got = _run_stages(prod, pcm, actx, prompt, fed, batches)
        _assert_within_reference_sensitivity(got, ref, pert, what)
        return got, ref, pert
    finally:
        prod.close()
        for r in refs:
            r.close()


@pytest.mark.parametrize("qtype", ["q5_1", "q8_0", "q4_0", "q4_1", "q5_0"])
def test_quantised_models_against_the_reference(product_lib, checker_lib, qtype):
    """ggml block-quantised files (BASELINE configs[4] uses q5_1): the weights stay quantised in HBM and every projection
    follows the reference's quantised mul_mat — activation rows to q8 blocks, exact integer block dots, per-block f32
    scales (csrc/k_quant.hip)."""
    model = synth.quantize_model(synth.make_model("micro.en", seed=1234), qtype)
    assert synth.QTYPES[qtype][1] == int(np.frombuffer(model[44:48], np.int32)[0]) % 1000
    extra = [[int(x) for x in (np.arange(40) * 997 + 1000)],          # 41 rows: the tiled quantised GEMM
             [1000, 2000, 3000, 4000]]                                 # 5 rows: a beam's step shape
    _quantised_case(product_lib, checker_lib, model, synth.make_pcm(6.0, seed=9), 428, 4, extra, ("micro.en", qtype))


def test_quantised_transcription_equals_reference_tokens(product_lib, checker_lib):
    """whisper_full on a q5_1 model against the compiled reference: identical token ids / timestamps up to the first near-tie
    (the yard-stick above applies to every logit, so a decision with a margin below it may go either way); no temperature
    fallback, whose logprob threshold is one more such decision."""
    if checker_lib is None:
        pytest.skip("token-stream comparison needs the compiled reference (host logic is not in the port)")
    model = synth.quantize_model(synth.make_model("micro.en", seed=1234), "q5_1")
    pcm = synth.make_pcm(11.0, seed=77)
    outs = []
    for L in (product_lib, checker_lib):
        node = host.SpeechToText(L); node.set_language_model(model)
        p = node.full_params("", 0); p.temperature_inc = 0.0
        outs.append(node.transcribe(pcm, params=p))
        node.close()
    g, w = gu.tokens_array(outs[0]), gu.tokens_array(outs[1])
    n = min(len(g), len(w))
    same = g[:n, 0] == w[:n, 0]
    first = n if same.all() else int(np.argmin(same))
    assert first >= 3, (g[:, 0], w[:, 0])
    assert np.abs(g[:first, 2] - w[:first, 2]).max() <= 5e-2            # token probabilities: within the quantised yard-stick
    if first == n:
        assert g.shape == w.shape


def test_large_v3_widths_q5_1_against_checker(product_lib, checker_lib):
    """configs[4]'s widths (1280 state, 20 heads, 128 mels, 51866 tokens) in q5_1 on the 2 + 3 layer slice: K = 1280 / 5120
    block rows, N = 51866 (not a multiple of 32) for the vocabulary projection."""
    import os
    model = synth.quantize_model(synth.make_model("v3-slice", seed=31), "q5_1")
    _quantised_case(product_lib, checker_lib, model, synth.make_pcm(20.0, seed=31), 0, 4, [[1000, 2000, 3000, 4000, 5000, 6000]],
                    "v3-slice q5_1", ref_threads=min(32, os.cpu_count() or 4))


def test_streaming_node_call_pattern_matches_reference_goldens(product_lib):
    """CaptureStreamToText (BASELINE config 3): every call re-transcribes the grown buffer with a different, ragged
    audio_ctx; token streams per call must equal the reference's (same margin rule as above)."""
    model, pcm = gu.stream_inputs()
    node = host.CaptureStreamToText(product_lib, transcribe_interval=gu.STREAM_INTERVAL); node.set_language_model(model)
    try:
        n_calls = 0
        for ci, (fin, text, n_used, actx, toks) in enumerate(node.stream(pcm)):
            meta = G[f"stream/{ci}/meta"]
            assert [int(fin), n_used, actx] == meta.tolist(), (ci, fin, n_used, actx, meta)
            got = gu.tokens_array([b""] + toks); want = G[f"stream/{ci}/tokens"]
            n = min(len(got), len(want))
            same = got[:n, 0] == want[:n, 0]
            first = n if same.all() else int(np.argmin(same))
            if first < n:
                assert abs(got[first, 2] - want[first, 2]) <= 2e-2, (ci, first, got[first], want[first])
                break          # histories differ from here on; the sentence-finish heuristics would too
            assert got.shape == want.shape
            assert np.abs(got[:, 2] - want[:, 2]).max() <= 1e-2
            n_calls += 1
        assert n_calls >= 4
    finally:
        node.close()


# ------------------------------------------------------------------------------------------------ lock-step chunks
def _assert_same_transcription(got, want, what, strict, ref_last_t1=False):
    g, w = gu.tokens_array(got), gu.tokens_array(want)
    if strict:
        # same kernels row for row, bit-identical logits: ids, timestamps and text equal.  The probabilities come out of two forms of
        # the same filter statistics (lock-step rows: 64 block partials of k_filter_stats; one row: one online partial per workgroup
        # of the vocabulary projection's epilogue) — equal up to the f32 order of a 51 864-term sum: a few 1e-7 on p / plog,
        # amplified in pt = p_ts / sum_ts (measured 1.9e-6)
        assert g.shape == w.shape, (what, g.shape, w.shape)
        assert np.array_equal(g[:, [0, 1, 6, 7, 8]], w[:, [0, 1, 6, 7, 8]]), (what, g[:, [0, 1, 6, 7]], w[:, [0, 1, 6, 7]])
        if len(g):
            assert np.abs(g[:, 2:6] - w[:, 2:6]).max() <= 8e-6, what
        assert bytes(got[0]) == bytes(want[0]), what
        return
    # different f32 summation order somewhere upstream (see the callers): identical up to the first near-tie
    n = min(len(g), len(w))
    same = g[:n, 0] == w[:n, 0]
    first = n if same.all() else int(np.argmin(same))
    if first < n:
        assert abs(g[first, 2] - w[first, 2]) <= 2e-2, (what, first, g[first], w[first])
    else:
        assert g.shape == w.shape, (what, g.shape, w.shape)
    if first:
        assert np.abs(g[:first, [2, 4, 5]] - w[:first, [2, 4, 5]]).max() <= 1e-2, what
    if first == n and n:
        if ref_last_t1:        # the reference reads past its token vector for the last token's t1 (W/whisper.cpp:6561), see above
            assert np.array_equal(g[:, [6, 8]], w[:, [6, 8]]) and np.array_equal(g[:-1, 7], w[:-1, 7]), what
        else:
            assert np.array_equal(g[:, [6, 7, 8]], w[:, [6, 7, 8]]), what
        assert bytes(got[0]) == bytes(want[0]), what


@pytest.fixture(params=["mfma", "exact"])
def lockstep_mode(request, product_lib):
    """Lock-step projections: MFMA rows kernel (default) or the VALU kernel that is bit-identical to the one-chunk path."""
    product_lib.wmi_set_lockstep_exact(1 if request.param == "exact" else 0)
    yield request.param
    product_lib.wmi_set_lockstep_exact(0)


@pytest.mark.parametrize("shape,variant", [("micro.en", "host"), ("micro", "host"), ("micro.en", "default_greedy"),
                                           ("micro.en", "default_fallback"), ("micro", "host_prompt")])
def test_lockstep_chunks_equal_one_at_a_time(product_lib, lockstep_mode, shape, variant):
    """wmi_full_batch (several chunks as rows of the same kernels) == whisper_full per chunk on a fresh context:
    ragged lengths (4 s ... 47 s: one and two seek windows, < 1 s: no output), more chunks than rows (11 > 8)."""
    model = synth.make_model(shape, seed=2024)
    secs = [30.0, 11.0, 4.0, 47.0, 30.0, 0.5, 22.5, 30.0, 8.0, 30.0, 15.0]
    pcms = [synth.make_pcm(s, seed=100 + i, gate=(i % 3 == 1)) for i, s in enumerate(secs)]

    def params(node):
        p = gu.param_variants(node)[variant]
        if shape == "micro" and variant == "host":
            node.language = "de"; p = node.full_params("", 0)
        if shape == "micro" or (lockstep_mode == "mfma" and variant == "host"):
            # multi-token prompts are not bit-identical between the two paths (see below), and the temperature
            # fallback is a threshold on avg_logprob — discontinuous in the logits (one of these chunks
            # sits at avg_logprob = -1.00 +- 1e-3).  Token parity is checked with the fallback off, as in the goldens.
            p.temperature_inc = 0.0
        return p

    want = []
    for b in pcms:                                               # fresh context per chunk: decoder RNGs at their seed
        node = host.SpeechToText(product_lib); node.set_language_model(model)
        want.append(node.transcribe(b, params=params(node)))
        assert node.last_ret == 0
        node.close()
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    try:
        got = node.transcribe_batch(pcms, params=params(node))
        assert node.last_ret == 0 and len(got) == len(pcms)
        modes = list(node.last_modes)
        # a chunk leaves the lock-step path when a window asks for the temperature fallback (logprob / entropy
        # thresholds, W/whisper.cpp:5643-5670) and is then run alone; with the fallback switched off none may
        if variant == "default_greedy":
            assert modes == [0] * len(pcms), modes
        assert modes.count(0) >= 2, modes
        for c, (g, w) in enumerate(zip(got, want)):
            # Lock-step rows with a single-token prompt and one window use the same kernels as the one-at-a-time path
            # (bit-identical); chunks run alone are the one-at-a-time path.  Longer prompts (multilingual, initial
            # prompt, second window with context) go through the MFMA GEMM when alone and token by token through the
            # GEMV here, which changes the f32 summation order.
            strict = modes[c] == 1 or (shape == "micro.en" and variant == "host" and lockstep_mode == "exact")
            if variant == "default_fallback" and not strict:
                continue        # context prompts after the first window + live fallback thresholds: covered by default_greedy
            _assert_same_transcription(g, w, (shape, variant, c, modes[c]), strict)
    finally:
        node.close()


def test_lockstep_groups_side_by_side_equal_one_group(product_lib, lockstep_mode):
    """wmi_set_lockstep_groups: the chunks of one wmi_full_batch call as two or three lock-step calls side by side (range 0 on the context,
    the others on replica contexts, host threads of their own) give every chunk the transcription the one-group call gives it — bit for
    bit in the exact mode, and (the encoder attention's key split follows the grid size) margin-aware in the default mode."""
    model = synth.make_model("micro.en", seed=2024)
    secs = [30.0, 11.0, 4.0, 47.0, 30.0, 22.5, 30.0, 8.0, 30.0, 15.0]
    pcms = [synth.make_pcm(s, seed=300 + i, gate=(i % 3 == 1)) for i, s in enumerate(secs)]
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    try:
        p = node.full_params("", 0); p.temperature_inc = 0.0
        assert product_lib.wmi_set_lockstep_groups(node.ctx, 1) >= 0
        want = node.transcribe_batch(pcms, params=p)
        assert node.last_ret == 0 and list(node.last_modes) == [0] * len(pcms)
        for groups in (2, 3):
            product_lib.wmi_set_lockstep_groups(node.ctx, groups)
            got = node.transcribe_batch(pcms, params=p)
            assert node.last_ret == 0 and len(got) == len(pcms) and list(node.last_modes) == [0] * len(pcms)
            for c, (g, w) in enumerate(zip(got, want)):
                _assert_same_transcription(g, w, ("groups", groups, c), lockstep_mode == "exact")
    finally:
        product_lib.wmi_set_lockstep_groups(node.ctx, 0)
        node.close()


@pytest.mark.parametrize("qtype", ["q5_1", "q8_0", "q4_0"])
def test_lockstep_chunks_of_quantised_models_equal_one_at_a_time(product_lib, lockstep_mode, qtype):
    """Block-quantised weights in lock-step: the chunks are rows of the same q8 x block-quantised kernels (per-row arithmetic does
    not depend on the number of rows: exact integer block dots, blocks added in K order).  With the one-group encoder attention
    of the exact mode everything is bit-identical to whisper_full per chunk; the default mode differs in the encoder attention's
    key split only, which the activation quantiser can amplify — compared up to the first near-tie."""
    model = synth.quantize_model(synth.make_model("micro.en", seed=77), qtype)
    secs = [30.0, 12.0, 30.0, 0.5, 21.0, 30.0, 7.0, 30.0, 30.0, 16.0]
    pcms = [synth.make_pcm(s, seed=300 + i, gate=(i % 3 == 2)) for i, s in enumerate(secs)]

    def params(node):
        p = node.full_params("", 0); p.temperature_inc = 0.0
        return p
    want = []
    for b in pcms:
        node = host.SpeechToText(product_lib); node.set_language_model(model)
        want.append(node.transcribe(b, params=params(node)))
        assert node.last_ret == 0
        node.close()
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    try:
        got = node.transcribe_batch(pcms, params=params(node))
        assert node.last_ret == 0 and len(got) == len(pcms)
        modes = list(node.last_modes)
        assert modes == [0] * len(pcms), modes                      # all chunks went through the lock-step kernels
        for c, (g, w) in enumerate(zip(got, want)):
            _assert_same_transcription(g, w, (qtype, c), lockstep_mode == "exact")
    finally:
        node.close()


def test_lockstep_chunks_with_audio_ctx_and_device_pcm(product_lib, lockstep_mode):
    model = synth.make_model("micro.en", seed=5)
    pcms = [synth.make_pcm(6.0, seed=40 + i) for i in range(3)]
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    try:
        want = [node.transcribe(b, "", 428) for b in pcms]
        hip = _hip(); dev = []
        for b in pcms:
            dptr = C.c_void_p()
            assert hip.hipMalloc(C.byref(dptr), C.c_size_t(b.nbytes)) == 0
            assert hip.hipMemcpy(dptr, b.ctypes.data_as(C.c_void_p), C.c_size_t(b.nbytes), 1) == 0
            dev.append(dptr)
        got = node.transcribe_batch([b.size for b in pcms], "", 428, device_ptrs=[d.value for d in dev])
        for d in dev:
            hip.hipFree(d)
        assert node.last_ret == 0
        for c, (g, w) in enumerate(zip(got, want)):
            if node.last_modes[c] == 0:
                _assert_same_transcription(g, w, c, lockstep_mode == "exact")
    finally:
        node.close()


def test_lockstep_base_en_eight_chunks(product_lib, lockstep_mode):
    """BASELINE config 4's per-GPU share: 8 chunks of 30 s on base.en, batch == one at a time."""
    model = synth.make_model("base.en", seed=1234)
    pcms = [synth.make_pcm(30.0, seed=1234 + i) for i in range(8)]
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    try:
        want = [node.transcribe(b, "", 0) for b in pcms]
        got = node.transcribe_batch(pcms, "", 0)
        assert node.last_ret == 0
        if lockstep_mode == "exact":
            assert node.last_modes == [0] * 8
        for c, (g, w) in enumerate(zip(got, want)):
            if node.last_modes[c] == 0:
                _assert_same_transcription(g, w, c, lockstep_mode == "exact")
    finally:
        node.close()


# ------------------------------------------------------------------------------------------------ large-v3 widths
def test_large_v3_widths_against_checker(product_lib, checker_lib):
    """BASELINE.json configs[4]'s shape parameters (1280 state, 20 heads, 128 mel bins, 51866 tokens) on a 2 + 3 layer
    slice: log-mel with the 128-bin bank, encoder, cross K/V, prompt + greedy steps, a 9-token batch (MFMA path)."""
    model = synth.make_model("v3-slice", seed=31); pcm = synth.make_pcm(20.0, seed=31)
    prod = sc.ProductSide(product_lib, model); chk = make_checker(model, checker_lib)
    try:
        mel_r, org_r = chk.mel(pcm); mel_p, org_p = prod.mel(pcm)
        assert mel_r.shape == mel_p.shape == (128, mel_r.shape[1]) and org_r == org_p
        sc.hold("log-mel max |d|", float(np.abs(mel_p - mel_r).max()), TOL["mel"][0])
        er = chk.encode(0, 0); ep = prod.encode(0, 0)
        for k in er:
            st = sc.err_stats(ep[k], er[k])
            sc.hold_tensor(k, st, TOL[k])
        prompt = sot_prompt(chk, prod)
        assert len(prompt) == 3
        lr = chk.decode(prompt, 0); lp = prod.decode(prompt, 0)
        assert_logits(lp, lr, "prompt")
        for i in range(4):
            tok = int(np.argmax(lr[:50256]))
            lr = chk.decode([tok], len(prompt) + i); lp = prod.decode([tok], len(prompt) + i)
            assert_logits(lp, lr, f"step{i}")
        batch = prompt + [1000, 2000, 3000, 4000, 5000, 6000]
        assert_logits(prod.decode(batch, 0), chk.decode(batch, 0), "batch9")
    finally:
        prod.close(); chk.close()


def test_large_v3_widths_transcription_and_lockstep(product_lib, checker_lib):
    model = synth.make_model("v3-slice", seed=31)
    pcms = [synth.make_pcm(12.0, seed=300 + i) for i in range(3)]
    node = host.SpeechToText(product_lib); node.set_language_model(model); node.language = "ja"
    try:
        p = node.full_params("", 0); p.temperature_inc = 0.0
        want = [node.transcribe(b, params=p) for b in pcms]
        assert all(len(w) > 1 for w in want)
        got = node.transcribe_batch(pcms, params=p)
        assert node.last_ret == 0 and node.last_modes == [0, 0, 0]
        for c, (g, w) in enumerate(zip(got, want)):
            _assert_same_transcription(g, w, ("v3-slice", c), False)
        if checker_lib is not None:                              # token stream against the compiled reference
            ref = host.SpeechToText(checker_lib); ref.set_language_model(model); ref.language = "ja"
            pr = ref.full_params("", 0); pr.temperature_inc = 0.0
            r = ref.transcribe(pcms[0], params=pr)
            ref.close()
            _assert_same_transcription(want[0], r, ("v3-slice", "vs reference"), False, ref_last_t1=True)
    finally:
        node.close()


# ------------------------------------------------------------------------------------------------ parameter sweep vs the reference
def _p_variant(node, name):
    """whisper_full_params beyond the goldens' six sets; temperature fallback off so that the streams are comparable
    (its threshold is discontinuous in the logits)."""
    p = node.lib.whisper_full_default_params(abi.WHISPER_SAMPLING_GREEDY)
    p.language = b"en"; p.temperature_inc = 0.0; p.print_progress = False
    if name == "no_timestamps":      p.no_timestamps = True
    elif name == "translate_fr":     p.language = b"fr"; p.translate = True
    elif name == "auto_language":    p.language = b"auto"
    elif name == "offset_duration":  p.offset_ms = 3000; p.duration_ms = 9000
    elif name == "max_len_wrap":     p.token_timestamps = True; p.max_len = 12; p.split_on_word = True
    elif name == "max_len_chars":    p.token_timestamps = True; p.max_len = 7; p.split_on_word = False
    elif name == "no_suppress":      p.suppress_blank = False; p.suppress_non_speech_tokens = False
    elif name == "suppress_nst":     p.suppress_non_speech_tokens = True; p.max_initial_ts = 0.0
    elif name == "short_ctx":        p.n_max_text_ctx = 8; p.no_context = False
    elif name == "single_segment":   p.single_segment = True; p.max_tokens = 24
    elif name == "thold":            p.token_timestamps = True; p.thold_pt = 0.2; p.thold_ptsum = 0.3
    elif name == "beam3":            p.strategy = abi.WHISPER_SAMPLING_BEAM_SEARCH; p.beam_search.beam_size = 3; p.greedy.best_of = 1
    elif name in ("grammar", "grammar_beam"):      # grammar-constrained decoding (W/whisper.cpp:3876-4290): " red, 12, blue."-shaped output
        ptrs, n_rules, keep = abi.make_grammar(gu.colour_list_grammar())
        node._grammar_keep = keep
        p.grammar_rules = C.cast(ptrs, C.c_void_p); p.n_grammar_rules = n_rules; p.i_start_rule = 0; p.grammar_penalty = 100.0
        p.no_timestamps = True; p.single_segment = True; p.max_tokens = 24
        if name == "grammar_beam": p.strategy = abi.WHISPER_SAMPLING_BEAM_SEARCH; p.beam_search.beam_size = 3; p.greedy.best_of = 1
    else: raise KeyError(name)
    return p


def _segments(lib, ctx):
    out = []
    for i in range(lib.whisper_full_n_segments(ctx)):
        out.append((lib.whisper_full_get_segment_t0(ctx, i), lib.whisper_full_get_segment_t1(ctx, i), bytes(lib.whisper_full_get_segment_text(ctx, i)),
                    [lib.whisper_full_get_token_id(ctx, i, j) for j in range(lib.whisper_full_n_tokens(ctx, i))]))
    return out


@pytest.mark.parametrize("variant", ["no_timestamps", "translate_fr", "auto_language", "offset_duration", "max_len_wrap", "max_len_chars",
                                     "no_suppress", "suppress_nst", "short_ctx", "single_segment", "thold", "beam3", "grammar", "grammar_beam"])
def test_parameter_variants_equal_the_compiled_reference(product_lib, checker_lib, variant):
    """Driver branches the Godot host does not take (W/whisper.cpp:4960-5807): both libraries run the same call on the same
    model and 20 s of audio; segments (t0, t1, text, token ids) must be equal up to the first near-tie (|dp| <= 2e-2)."""
    if checker_lib is None:
        pytest.skip("needs the compiled reference")
    multilingual = variant in ("translate_fr", "auto_language")
    model = synth.make_model("micro" if multilingual else "micro.en", seed=55); pcm = synth.make_pcm(20.0, seed=56)
    res = []
    for L in (product_lib, checker_lib):
        node = host.SpeechToText(L); node.set_language_model(model)
        p = _p_variant(node, variant)
        ret = L.whisper_full(node.ctx, p, pcm.ctypes.data_as(C.POINTER(C.c_float)), pcm.size)
        toks = gu.tokens_array([b""] + node.collect()[1:]) if ret == 0 else None
        res.append((ret, _segments(L, node.ctx) if ret == 0 else None, toks, L.whisper_full_lang_id(node.ctx)))
        node.close()
    (rp, sp, tp, lp), (rr, sr, tr, lr) = res
    assert rp == rr == 0
    if variant.startswith("grammar"):             # the constraint took hold: the text is a prefix of a colour list
        import re
        text = b"".join(a[2] for a in sp).decode()
        assert re.fullmatch(r" ?((red|green|blue|[0-9]+)(, (red|green|blue|[0-9]+))*\.?|(red|green|blue|[0-9]+, )*(r|re|g|gr|gre|gree|b|bl|blu)?)?", text) or \
               re.match(r" ?(red|green|blue|[0-9]+)", text), text
    if multilingual:
        assert lp == lr
    n = min(len(tp), len(tr))
    same = tp[:n, 0] == tr[:n, 0]
    first = n if same.all() else int(np.argmin(same))
    if first < n:                                   # a near-tie (or, for beam search, a draw at a CDF step): histories part here
        assert first >= 3, (variant, first, tp[:4, 0], tr[:4, 0])
        if variant not in ("beam3", "grammar_beam"):
            assert abs(tp[first, 2] - tr[first, 2]) <= 2e-2, (variant, first, tp[first], tr[first])
    else:
        assert len(sp) == len(sr), (variant, len(sp), len(sr))
        for a, b in zip(sp, sr):
            assert a[3] == b[3] and a[2] == b[2], (variant, a, b)
            assert a[0] == b[0] and a[1] == b[1], (variant, a[:2], b[:2])
        assert np.abs(tp[:, 2] - tr[:, 2]).max() <= 1e-2
        if variant in ("max_len_wrap", "max_len_chars", "thold"):            # token-level timestamps (last t1: reference UB, see above)
            assert np.array_equal(tp[:, 6], tr[:, 6]) and np.array_equal(tp[:-1, 7], tr[:-1, 7]), variant


# ------------------------------------------------------------------------------------------------ lifecycle
def test_context_lifecycle_releases_device_memory(product_lib):
    """whisper_init / whisper_full / wmi_full_batch / whisper_free in a loop: device memory returns to its level
    (weights arena, state arenas, lock-step work set, per-chunk streams and pinned buffers are all released)."""
    hip = _hip()
    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value
    model = synth.make_model("micro.en", seed=3)
    pcms = [synth.make_pcm(5.0, seed=70 + i) for i in range(3)]
    def cycle():
        node = host.SpeechToText(product_lib); node.set_language_model(model)
        assert node.transcribe(pcms[0], "", 0)
        assert len(node.transcribe_batch(pcms, "", 0)) == 3
        node.close()
    cycle()                                   # first use: module load, lazy runtime allocations
    level = free_bytes()
    for _ in range(8):
        cycle()
    assert level - free_bytes() <= 8 << 20, (level, free_bytes())      # allocator granularity, no per-cycle growth


def test_long_audio_many_windows_equals_the_compiled_reference(product_lib, checker_lib):
    """150 s of audio with the library defaults (multi-segment, context carried from window to window, fallback off):
    the seek loop, prompt_past bookkeeping and segment timestamps over ~6 windows against the compiled reference."""
    if checker_lib is None:
        pytest.skip("needs the compiled reference")
    model = synth.make_model("micro.en", seed=91); pcm = synth.make_pcm(150.0, seed=92, gate=True)
    res = []
    for L in (product_lib, checker_lib):
        node = host.SpeechToText(L); node.set_language_model(model)
        p = L.whisper_full_default_params(abi.WHISPER_SAMPLING_GREEDY)
        p.language = b"en"; p.temperature_inc = 0.0; p.print_progress = False
        assert L.whisper_full(node.ctx, p, pcm.ctypes.data_as(C.POINTER(C.c_float)), pcm.size) == 0
        res.append((_segments(L, node.ctx), gu.tokens_array([b""] + node.collect()[1:])))
        node.close()
    (sp, tp), (sr, tr) = res
    assert len(sr) >= 6
    n = min(len(tp), len(tr))
    same = tp[:n, 0] == tr[:n, 0]
    first = n if same.all() else int(np.argmin(same))
    assert first >= 40, (first, n)                                   # well into the audio before any near-tie
    if first < n:
        # Either a visible near-tie, or the streams part at a window boundary: a window's tokens after its last timestamp
        # are decoded but discarded (W/whisper.cpp:5524-5540, 5682), so a near-tie inside such a tail shows up only as a
        # different window end — the token before the split is then the closing timestamp of a segment on both sides.
        ends_p = np.cumsum([len(a[3]) for a in sp]); ends_r = np.cumsum([len(b[3]) for b in sr])
        at_boundary = first in ends_p or first in ends_r
        assert at_boundary or abs(tp[first, 2] - tr[first, 2]) <= 2e-2, (first, tp[first], tr[first])
    else:
        assert [(a[0], a[1], a[3]) for a in sp] == [(b[0], b[1], b[3]) for b in sr]
    assert np.abs(tp[:first, 2] - tr[:first, 2]).max() <= 1e-2


@pytest.mark.parametrize("case", range(8))
def test_randomised_cases_equal_the_compiled_reference(product_lib, checker_lib, case):
    """Seeded sweep over model seed, audio length (1.2 .. 40 s), gating, audio_ctx (0 or ragged) and prompt: the Godot
    host's parameter set on both libraries; identical token ids / timestamps up to the first near-tie."""
    if checker_lib is None:
        pytest.skip("needs the compiled reference")
    rng = np.random.default_rng(1000 + case)
    shape = "micro.en" if case % 2 == 0 else "micro"
    model = synth.make_model(shape, seed=int(rng.integers(1, 10**6)))
    secs = float(rng.choice([1.2, 2.5, 7.0, 13.0, 29.9, 30.0, 31.0, 40.0]))
    pcm = synth.make_pcm(secs, seed=int(rng.integers(1, 10**6)), gate=bool(rng.integers(0, 2)))
    actx = 0 if rng.integers(0, 2) else min(int(secs * 50 + 128), 1500)
    prompt = "" if rng.integers(0, 2) else " Well, then."
    outs = []
    for L in (product_lib, checker_lib):
        node = host.SpeechToText(L); node.set_language_model(model)
        if shape == "micro":
            node.language = ["en", "de", "ja", "fr"][case % 4]
        p = node.full_params(prompt, actx); p.temperature_inc = 0.0
        r = node.transcribe(pcm, params=p)
        ends = np.cumsum([L.whisper_full_n_tokens(node.ctx, i) for i in range(L.whisper_full_n_segments(node.ctx))]) if r else np.zeros(0, int)
        outs.append((node.last_ret, gu.tokens_array(r) if r else np.zeros((0, 9)), bytes(r[0]) if r else b"", ends))
        node.close()
    (rp, tp, xp, ep), (rr, tr, xr, er) = outs
    assert rp == rr
    n = min(len(tp), len(tr))
    same = tp[:n, 0] == tr[:n, 0]
    first = n if same.all() else int(np.argmin(same))
    if first < n:
        assert abs(tp[first, 2] - tr[first, 2]) <= 2e-2, (case, first, tp[first], tr[first])
    else:
        assert tp.shape == tr.shape and xp == xr and np.array_equal(ep, er), (case, tp.shape, tr.shape)
        if n:
            # t1 of the LAST token of every segment: the reference reads one element past its token vector there
            # (W/whisper.cpp:6561) — exempt, as everywhere in this file
            keep = np.ones(n, bool); keep[ep - 1] = False
            assert np.array_equal(tp[:, 6], tr[:, 6]) and np.array_equal(tp[keep, 7], tr[keep, 7]), case
    if first:
        assert np.abs(tp[:first, [2, 4, 5]] - tr[:first, [2, 4, 5]]).max() <= 1e-2, case


def test_small_multilingual_shape_against_checker(product_lib, checker_lib):
    """BASELINE.json configs[2]'s model shape (small multilingual: 12 + 12 layers, 768 wide, 12 heads) at full size: encoder
    at a ragged audio_ctx as the streaming node uses it, prompt and greedy steps (K = 768: two 512-column chunks per lane
    in k_gemv1, the unfused cross-query path), then a transcription against the reference."""
    model = synth.make_model("small", seed=77); pcm = synth.make_pcm(9.0, seed=21, gate=True)
    prod = sc.ProductSide(product_lib, model); chk = make_checker(model, checker_lib)
    try:
        mel_r, _ = chk.mel(pcm); mel_p, _ = prod.mel(pcm)
        sc.hold("log-mel max |d|", float(np.abs(mel_p - mel_r).max()), TOL["mel"][0])
        er = chk.encode(0, 578); ep = prod.encode(0, 578)
        for k in er:
            st = sc.err_stats(ep[k], er[k])
            sc.hold_tensor(k, st, TOL[k], abs_mul=2.0)     # 12 layers deep: abs bound doubled
        prompt = sot_prompt(chk, prod)
        lr = chk.decode(prompt, 0); lp = prod.decode(prompt, 0)
        assert_logits(lp, lr, "prompt")
        for i in range(4):
            tok = int(np.argmax(lr[:50256]))
            lr = chk.decode([tok], len(prompt) + i); lp = prod.decode([tok], len(prompt) + i)
            assert_logits(lp, lr, f"step{i}")
    finally:
        prod.close(); chk.close()
    if checker_lib is not None:
        outs = []
        for L in (product_lib, checker_lib):
            node = host.SpeechToText(L); node.set_language_model(model); node.language = "de"
            p = node.full_params("", 578); p.temperature_inc = 0.0
            outs.append(node.transcribe(pcm, params=p)); node.close()
        _assert_same_transcription(outs[0], outs[1], "small", False, ref_last_t1=True)


# ------------------------------------------------------------------------------------------------ BASELINE configs[0] shape, batch vs reference, 10-minute stream
def test_tiny_en_shape_against_checker(product_lib, checker_lib):
    """BASELINE.json configs[0]'s model shape (tiny.en: 384 wide, 6 heads, 4 + 4 layers) at full size: stages, and the
    AudioStreamToText call pattern (jfk.wav, host parameters) token for token against the reference."""
    model = synth.make_model("tiny.en", seed=404); pcm = synth.make_pcm(30.0, seed=404)
    prod = sc.ProductSide(product_lib, model); chk = make_checker(model, checker_lib)
    try:
        mel_r, _ = chk.mel(pcm); mel_p, _ = prod.mel(pcm)
        sc.hold("log-mel max |d|", float(np.abs(mel_p - mel_r).max()), TOL["mel"][0])
        er = chk.encode(0, 0); ep = prod.encode(0, 0)
        for k in er:
            st = sc.err_stats(ep[k], er[k])
            sc.hold_tensor(k, st, TOL[k])
        prompt = sot_prompt(chk, prod)
        lr = chk.decode(prompt, 0); lp = prod.decode(prompt, 0)
        assert_logits(lp, lr, "prompt")
        for i in range(6):
            tok = int(np.argmax(lr[:50256]))
            lr = chk.decode([tok], len(prompt) + i); lp = prod.decode([tok], len(prompt) + i)
            assert_logits(lp, lr, f"step{i}")
        batch = prompt + [1000, 2000, 3000, 4000, 5000, 6000, 7000, 8000, 9000]
        assert_logits(prod.decode(batch, 0), chk.decode(batch, 0), "batch10")
    finally:
        prod.close(); chk.close()
    if checker_lib is None:
        return
    jfk = synth.read_wav_mono16(gu.GOLDEN / "jfk.wav")
    outs = []
    for L in (product_lib, checker_lib):
        node = host.AudioStreamToText(L); node.set_language_model(model)
        outs.append((node.transcribe(jfk, "", 0), node.get_text(jfk)))
        node.close()
    _assert_same_transcription(outs[0][0], outs[1][0], "tiny.en jfk.wav", False, ref_last_t1=True)
    first = next((i for i, (a, b) in enumerate(zip(gu.tokens_array(outs[0][0])[:, 0], gu.tokens_array(outs[1][0])[:, 0])) if a != b), None)
    if first is None:
        assert outs[0][1] == outs[1][1]


def test_lockstep_batch_equals_the_compiled_reference(product_lib, checker_lib):
    """wmi_full_batch against the REFERENCE (not against the product's own one-chunk path): every chunk's token stream is what
    whisper_full of the compiled reference returns for it on a fresh context."""
    if checker_lib is None:
        pytest.skip("needs the compiled reference")
    model = synth.make_model("micro.en", seed=1234)
    pcms = [synth.make_pcm(6.0 + 1.5 * i, seed=500 + i) for i in range(5)]
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    ref = host.SpeechToText(checker_lib); ref.set_language_model(model)
    try:
        for variant in ("host", "default_greedy"):
            p = gu.param_variants(node)[variant]; p.temperature_inc = 0.0
            pr = gu.param_variants(ref)[variant]; pr.temperature_inc = 0.0
            got = node.transcribe_batch(pcms, params=p)
            assert node.last_ret == 0 and len(got) == len(pcms)
            for c, b in enumerate(pcms):
                want = ref.transcribe(b, params=pr)
                _assert_same_transcription(got[c], want, ("batch vs reference", variant, c, node.last_modes[c]), False, ref_last_t1=True)
    finally:
        node.close(); ref.close()


def test_lockstep_base_en_eight_chunks_equal_the_compiled_reference(product_lib, checker_lib):
    """BASELINE configs[3]'s per-GPU share at its real size — 8 x 30 s chunks of base.en in one wmi_full_batch call, host
    parameter set — against the REFERENCE's whisper_full of each chunk (round 2 compared this size only with the product's own
    one-chunk path).  4 reference threads x 8 chunks ~ 6 s."""
    if checker_lib is None:
        pytest.skip("needs the compiled reference")
    model = synth.make_model("base.en", seed=1234)
    pcms = [synth.make_pcm(30.0, seed=1234 + i) for i in range(8)]
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    ref = host.SpeechToText(checker_lib); ref.set_language_model(model)
    try:
        p = node.full_params("", 0); p.temperature_inc = 0.0
        pr = ref.full_params("", 0); pr.temperature_inc = 0.0; pr.n_threads = max(4, min(32, os.cpu_count() or 4))
        got = node.transcribe_batch(pcms, params=p)
        assert node.last_ret == 0 and len(got) == 8 and list(node.last_modes) == [0] * 8
        n_tok = 0
        for c, b in enumerate(pcms):
            want = ref.transcribe(b, params=pr)
            assert ref.last_ret == 0
            n_tok += len(want) - 1
            _assert_same_transcription(got[c], want, ("8 x base.en vs reference", c), False, ref_last_t1=True)
        assert n_tok >= 8 * 4
    finally:
        node.close(); ref.close()


def test_ten_minute_stream_small_shape(product_lib, checker_lib):
    """BASELINE.json configs[2]: CaptureStreamToText over 10 minutes of synthetic microphone audio (2 s on / 1 s off) with the
    `small` multilingual shape — the whole grown buffer is transcribed again every 0.3 s with audio_ctx = total_s * 50 + 128.
    At full length: structural properties of every call (the KV cache and ragged encoder lengths are reused ~2000 times); the
    first calls are compared with the compiled reference; a replay of one late call on a fresh context must reproduce it."""
    model = synth.make_model("small", seed=77)
    pcm = synth.make_pcm(600.0, seed=21, gate=True)
    node = host.CaptureStreamToText(product_lib, transcribe_interval=0.3)
    node.language = "de"; node.device_vad = True
    node.set_language_model(model)
    calls = []
    try:
        for fin, text, n_used, actx, toks in node.stream(pcm):
            calls.append((fin, n_used, actx, gu.tokens_array([b""] + toks)))
        assert len(calls) >= 1900
        finals = sum(1 for c in calls if c[0])
        assert finals >= 30                                                  # a sentence at least every 15 s
        for fin, n_used, actx, tk in calls:
            assert actx == min(int(n_used / 16000 * 50 + 128), 1500) and 16000 <= n_used <= 16000 * 15.3 + 1
            assert tk.shape[0] <= 17 and np.all(tk[:, 0] >= 0) and np.all(tk[:, 0] < 51865)
            assert np.all(np.isfinite(tk[:, 2])) and np.all((tk[:, 2] >= 0) & (tk[:, 2] <= 1.0 + 1e-6))
            if tk.shape[0]:
                assert np.all(np.diff(tk[:, 6]) >= 0) and tk[:, 7].max() <= n_used / 160 + 2      # t0 monotone, t1 inside the buffer (10 ms units)
        # determinism across the whole run: replaying a late call on a fresh context gives the same tokens
        fresh = host.SpeechToText(product_lib); fresh.language = "de"; fresh.set_language_model(model)
        sr = 16000; step = int(round(0.3 * sr))
        # reconstruct the buffer of the last call from the recorded sentence boundaries
        start, pos, idx = 0, 0, -1
        while pos < pcm.size:
            pos = min(pos + step, pcm.size)
            if (pos - start) / sr < 1.0:
                continue
            idx += 1
            if idx == len(calls) - 1:
                break
            if calls[idx][0]:
                start = max(pos - int(0.2 * sr), 0)
        buf = pcm[start:pos]
        assert buf.size == calls[-1][1]
        again = gu.tokens_array(fresh.transcribe(buf, "", calls[-1][2]))
        fresh.close()
        assert np.array_equal(again[:, 0], calls[-1][3][:, 0])
    finally:
        node.close()
    if checker_lib is None:
        return
    # the first 64 calls against the compiled reference (same call pattern, reference library behind the same host mirror; the buffer
    # grows through two sentence boundaries, audio_ctx from 178 to ~900: ~0.3-1 s of reference time per call at 32 threads)
    ref = host.CaptureStreamToText(checker_lib, transcribe_interval=0.3); ref.language = "de"; ref.set_language_model(model)
    ref.n_threads = max(4, min(32, os.cpu_count() or 4))
    try:
        n_cmp = 0; near_ties = []; n_tok = 0; n_full = 0; worst_p = 0.0
        for (fin, text, n_used, actx, toks), mine in zip(ref.stream(pcm[: 16000 * 24], max_calls=64), calls):
            if (fin, n_used, actx) != (mine[0], mine[1], mine[2]):
                # the two loops only part ways when a near-tie changed a token count or the last character the sentence rule looks at
                assert near_ties, (n_cmp, fin, n_used, actx, mine[:3])
                print(f"configs[2]: the call patterns part ways at call {n_cmp} (behind the near-tie at call {near_ties[-1][0]})")
                break
            w = gu.tokens_array([b""] + toks); g = mine[3]
            n = min(len(g), len(w)); same = g[:n, 0] == w[:n, 0]
            first = n if same.all() else int(np.argmin(same))
            # margin-aware like every token comparison here: a first disagreement is admitted only as a near-tie (the two picks'
            # probabilities within 1e-2: e.g. call 9, token 1: 0.1462 against 0.1464); they are counted and reported, not capped at one
            if first < n:
                assert abs(g[first, 2] - w[first, 2]) <= 1e-2, (n_cmp, first, g[:, 0], w[:, 0], g[:, 2], w[:, 2])
                near_ties.append((n_cmp, first))
            else:
                assert len(g) == len(w), (n_cmp, len(g), len(w))
                n_full += 1
            if first:
                # p = soft-max probability of the picked token: a logit difference d moves it by p (1 - p) d, i.e. by up to 0.25 x the logit
                # bound (P_LOGIT_ABS = 6e-2, + 20 % for the log-sum-exp) where the distribution is flat (p ~ 0.3-0.5 on these weights); 1e-2 holds where p > 0.9
                dp = np.abs(g[:first, 2] - w[:first, 2]); worst_p = max(worst_p, float(dp.max()))
                assert np.all(dp <= np.maximum(1e-2, 1.2 * P_LOGIT_ABS * w[:first, 2] * (1 - w[:first, 2]) + 2e-3)), (n_cmp, dp, w[:first, 2])
                assert np.array_equal(g[:first, 6], w[:first, 6])                 # token start times of the common prefix
            n_tok += first
            n_cmp += 1
        print(f"configs[2]: {n_cmp} calls compared with the reference, {n_full} identical token streams, {n_tok} common tokens, "
              f"{len(near_ties)} near-ties (call, token): {near_ties}; largest |p difference| on common tokens {worst_p:.3e}")
        assert n_cmp >= 40 and len(near_ties) <= max(2, n_cmp // 8), (n_cmp, near_ties)
    finally:
        ref.close()


# ------------------------------------------------------------------------------------------------ the beam-search driver, pinned
def _synthetic_logits_callback(n_vocab, eot, beg, record=None):
    """A logits_filter_callback that REPLACES the model's logits by a deterministic function of the decoder's token history:
    both libraries then sample from identical distributions, whatever their own numerics — what remains is the driver
    (mt19937 + discrete_distribution draws, candidate ranking, KV sequence bookkeeping, completion rules, scoring)."""
    def cb(ctx, state, tokens, n_tokens, logits, user):
        hist = [tokens[i].id for i in range(n_tokens)]
        seed = (1469598103934665603 ^ len(hist)) & 0xFFFFFFFF
        for t in hist:
            seed = ((seed ^ (t + 1)) * 16777619) & 0xFFFFFFFF
        rng = np.random.default_rng(seed)
        arr = np.ctypeslib.as_array(logits, shape=(n_vocab,))
        keep = np.isneginf(arr)                                            # rules applied before the callback stay
        new = rng.standard_normal(n_vocab).astype(np.float32) * 1.5
        hot = rng.integers(0, 50000, size=6)
        new[hot] += np.float32(7.0) + rng.standard_normal(6).astype(np.float32)
        new[beg:] = np.float32(-14.0) + new[beg:] * np.float32(0.2)       # the 1500 timestamp ids: little mass ...
        new[beg + 2 * (len(hist) + 1): beg + 2 * (len(hist) + 1) + 3] = np.float32(5.0) + rng.standard_normal(3).astype(np.float32)   # ... except a few plausible ones moving forward
        if len(hist) > 9:
            new[eot] += np.float32(8.0)
        new[keep] = -np.inf
        arr[:] = new
        if record is not None:
            record.append(tuple(hist))
    return abi.whisper_logits_filter_callback(cb)


@pytest.mark.parametrize("variant", ["beam5", "beam3_multi_window", "best_of_t04"])
def test_beam_driver_reproduces_the_reference_stream_on_equal_logits(product_lib, checker_lib, variant):
    """beam-mode parity = "identical token stream when fed the oracle's logits".  Both libraries get the same
    synthetic logits through logits_filter_callback; the complete results — ids, probabilities, log-probabilities,
    timestamps statistics, segment boundaries, text — must then be IDENTICAL, for every window."""
    if checker_lib is None:
        pytest.skip("needs the compiled reference")
    model = synth.make_model("micro.en", seed=1234)
    pcm = synth.make_pcm(21.0 if variant == "beam3_multi_window" else 9.0, seed=55)
    outs = []
    for L in (product_lib, checker_lib):
        node = host.SpeechToText(L); node.set_language_model(model)
        nv, eot, beg = L.whisper_n_vocab(node.ctx), L.whisper_token_eot(node.ctx), L.whisper_token_beg(node.ctx)
        cb = _synthetic_logits_callback(nv, eot, beg)
        if variant == "best_of_t04":
            p = node.full_params("", 0); p.temperature = 0.4; p.temperature_inc = 0.0; p.greedy.best_of = 4; p.max_tokens = 24
        else:
            p = L.whisper_full_default_params(abi.WHISPER_SAMPLING_BEAM_SEARCH)
            p.language = b"en"; p.temperature_inc = 0.0
            p.beam_search.beam_size = 5 if variant == "beam5" else 3
            if variant == "beam5":
                p.single_segment = True; p.max_tokens = 20
        p.logits_filter_callback = C.cast(cb, C.c_void_p)
        ret = L.whisper_full(node.ctx, p, pcm.ctypes.data_as(C.POINTER(C.c_float)), pcm.size)
        assert ret == 0
        segs = []
        for s in range(L.whisper_full_n_segments(node.ctx)):
            toks = [L.whisper_full_get_token_data(node.ctx, s, j) for j in range(L.whisper_full_n_tokens(node.ctx, s))]
            segs.append((L.whisper_full_get_segment_t0(node.ctx, s), L.whisper_full_get_segment_t1(node.ctx, s),
                         L.whisper_full_get_segment_text(node.ctx, s),
                         [(t.id, t.tid, t.p, t.plog, t.pt, t.ptsum) for t in toks]))
        outs.append(segs)
        node.close()
    got, want = outs
    assert len(want) >= 1 and sum(len(s[3]) for s in want) >= 8
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[:3] == w[:3]
        assert [t[:2] for t in g[3]] == [t[:2] for t in w[3]]
        assert np.allclose(np.array([t[2:] for t in g[3]]), np.array([t[2:] for t in w[3]]), rtol=0, atol=1e-6)


def test_beam_kv_bookkeeping_gives_each_beam_its_own_history(product_lib):
    """The beams of a step share physical KV cells through sequence-id sets (kv_seq_cp / kv_seq_rm, W/whisper.cpp:1038-1054,
    5402-5417); the mask built from them must give every beam exactly its own history.  Check: the logits the beam step
    produced for a recorded history equal the logits of decoding that history alone on a fresh cache."""
    model = synth.make_model("micro.en", seed=1234)
    pcm = synth.make_pcm(9.0, seed=56)
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    nv = product_lib.whisper_n_vocab(node.ctx)
    beg_id, eot_id = product_lib.whisper_token_beg(node.ctx), product_lib.whisper_token_eot(node.ctx)
    seen = {}
    def cb(ctx, state, tokens, n_tokens, logits, user):
        hist = tuple(tokens[i].id for i in range(n_tokens))
        a = np.ctypeslib.as_array(logits, shape=(nv,))
        if 2 <= len(hist) <= 7 and hist not in seen:
            seen[hist] = a.copy()
        a *= np.float32(0.02)                      # flatten what the sampler sees: the draws of the beams diverge ...
        a[beg_id:] = -np.inf; a[eot_id] = -np.inf   # ... and keep them running (no timestamps, no end of text)
    cbk = abi.whisper_logits_filter_callback(cb)
    try:
        p = product_lib.whisper_full_default_params(abi.WHISPER_SAMPLING_BEAM_SEARCH)
        p.language = b"en"; p.temperature_inc = 0.0; p.beam_search.beam_size = 5; p.single_segment = True; p.max_tokens = 10
        p.logits_filter_callback = C.cast(cbk, C.c_void_p)
        assert product_lib.whisper_full(node.ctx, p, pcm.ctypes.data_as(C.POINTER(C.c_float)), pcm.size) == 0
        assert len(seen) >= 10 and len({h[:2] for h in seen}) >= 2            # beams did diverge
        sot = product_lib.whisper_token_sot(node.ctx)
        checked = 0
        for hist, lg in list(seen.items())[:24]:
            seq = np.asarray([sot] + list(hist), np.int32)
            assert product_lib.whisper_decode(node.ctx, seq.ctypes.data_as(C.POINTER(C.c_int32)), seq.size, 0, 4) == 0
            lp = np.ctypeslib.as_array(product_lib.whisper_get_logits(node.ctx), shape=(seq.size * nv,)).reshape(seq.size, nv)[-1]
            ok = ~np.isneginf(lg)                                               # entries the filters had not yet suppressed
            assert np.abs(lp[ok] - lg[ok]).max() <= 2e-2, hist                  # batch-of-n vs one-row kernels: f32 summation order
            checked += 1
        assert checked >= 10
    finally:
        node.close()


@pytest.mark.parametrize("variant", ["beam5", "beam3_default", "best_of_t04", "fallback"])
@pytest.mark.parametrize("shape", ["micro.en", "micro"])
def test_device_draws_equal_host_draws(product_lib, variant, shape):
    """beam-search candidates and t > 0 samples are drawn on the device (k (id, p, plog) per row cross PCIe instead
    of 207 KB of logits).  The generators stay on the host, so the device path must pick the same ids as the host's
    std::discrete_distribution (bit-exact vs the reference: tests/test_host_logic.py) — the only difference is the order of the
    f32 / f64 sums behind the CDF (a draw within ~1e-6 of a CDF step could flip; none does on these inputs)."""
    import os
    model = synth.make_model(shape, seed=1234)
    pcm = synth.make_pcm(12.0, seed=61)
    res = {}
    for mode in ("device", "host"):
        if mode == "host":
            os.environ["WMI_HOST_DRAWS"] = "1"
        else:
            os.environ.pop("WMI_HOST_DRAWS", None)
        product_lib.wmi_reload_knobs()                      # (the switches are read once per process)
        node = host.SpeechToText(product_lib); node.set_language_model(model)
        if shape == "micro":
            node.language = "fr"
        try:
            if variant == "best_of_t04":
                p = node.full_params("", 0); p.temperature = 0.4; p.temperature_inc = 0.0; p.greedy.best_of = 4
            elif variant == "fallback":
                p = node.full_params("", 0); p.logprob_thold = 10.0; p.greedy.best_of = 3           # every temperature fails: walks 0.0 .. 1.0
            else:
                p = product_lib.whisper_full_default_params(abi.WHISPER_SAMPLING_BEAM_SEARCH)
                p.language = node.language.encode(); p.temperature_inc = 0.0
                node._keep = [p.language]
                p.beam_search.beam_size = 5 if variant == "beam5" else 3
                if variant == "beam5":
                    p.single_segment = True; p.max_tokens = 16
            res[mode] = gu.tokens_array(node.transcribe(pcm, params=p))
            t6 = (C.c_int64 * 6)(); n5 = (C.c_int32 * 5)(); product_lib.wmi_get_timings(node.ctx, t6, n5)
            res[mode + "_sample_us"] = t6[5]
        finally:
            node.close()
    os.environ.pop("WMI_HOST_DRAWS", None)
    product_lib.wmi_reload_knobs()
    g, w = res["device"], res["host"]
    assert len(w) >= 4
    assert g.shape == w.shape and np.array_equal(g[:, [0, 1, 6, 7]], w[:, [0, 1, 6, 7]]), (g[:, 0], w[:, 0])
    assert np.abs(g[:, 2:6] - w[:, 2:6]).max() <= 1e-4              # log-sum-exp of 51 864 terms: f32 tree vs sequential order

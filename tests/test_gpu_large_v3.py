"""-m gpu: BASELINE.json configs[4]'s model at FULL depth — large-v3, 32 + 32 layers, 1280 wide, 128 mel bins, 51866 tokens —
as f16 and as q5_1, against the compiled reference on the same seeded synthetic weights (real weights are not available
offline).  Stages: encoder output, cross K/V of all 32 decoder layers, prompt + 4 greedy steps.

f16: the usual bounds (encoder / cross rms-rel 2e-3, logits rms-rel 2e-3).
q5_1: within the reference's own response to a 1e-6 relative change of the PCM (see tests/test_gpu_parity.py, block-quantised
section: an 8-bit activation quantiser in front of every projection makes the reference itself that sensitive), and the
weights must occupy their quantised size in HBM (~1.1 GB, not the 3.1 GB of an f16 expansion)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import stage_compare as sc
from godot_whisper_amd import synth
from oracle import reflib

import test_gpu_parity as tp

pytestmark = pytest.mark.gpu

_CACHE = {}


def _ref_quantize_model(ref, model: bytes, qtype: str) -> bytes:
    """synth.quantize_model with the reference's own block quantiser doing the arithmetic (byte-identical, pinned in
    tests/test_synth_and_shard.py; ~20x faster than the numpy restatement on 1.5 G weights)."""
    gtype, ftype = synth.QTYPES[qtype]
    fn = getattr(ref, f"ggml_quantize_{qtype}")
    fn.restype = C.c_size_t; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    bb = {"q4_0": 18, "q4_1": 20, "q5_0": 22, "q5_1": 24, "q8_0": 34}[qtype]
    hist = (C.c_int64 * 16)()
    b = model
    hp = list(struct.unpack_from("<11i", b, 4))
    off = 4 + 44
    n_mel, n_fft = struct.unpack_from("<2i", b, off); off += 8 + 4 * n_mel * n_fft
    (nv,) = struct.unpack_from("<i", b, off); off += 4
    for _ in range(nv):
        (ln,) = struct.unpack_from("<I", b, off); off += 4 + ln
    hp[10] = 2 * 1000 + ftype
    out = [b[:4], struct.pack("<11i", *hp), b[48:off]]
    while off < len(b):
        nd, nl, tt = struct.unpack_from("<3i", b, off); off += 12
        ne = struct.unpack_from(f"<{nd}i", b, off); off += 4 * nd
        name = b[off:off + nl]; off += nl
        n = int(np.prod(ne)); nbytes = n * (2 if tt == 1 else 4)
        raw = b[off:off + nbytes]; off += nbytes
        if nd == 2 and name.decode() not in synth._SKIP:
            x = np.frombuffer(raw, np.float16 if tt == 1 else np.float32).astype(np.float32)
            dst = np.empty(n // 32 * bb, np.uint8)
            got = fn(x.ctypes.data, dst.ctypes.data, n, int(ne[0]), C.cast(hist, C.c_void_p))
            assert got == dst.size
            out += [struct.pack("<3i", nd, nl, gtype), struct.pack(f"<{nd}i", *ne), name, dst.tobytes()]
        else:
            out += [struct.pack("<3i", nd, nl, tt), struct.pack(f"<{nd}i", *ne), name, raw]
    return b"".join(out)


def _model(kind, ref):
    if "f16" not in _CACHE:
        _CACHE["f16"] = synth.make_model("large-v3", seed=2024)
    if kind == "f16":
        return _CACHE["f16"]
    if "q5_1" not in _CACHE:
        _CACHE["q5_1"] = _ref_quantize_model(ref, _CACHE["f16"], "q5_1") if ref is not None else synth.quantize_model(_CACHE["f16"], "q5_1")
    return _CACHE["q5_1"]


def _threads():
    return max(4, min(64, os.cpu_count() or 4))


def _free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    assert tp._hip().hipMemGetInfo(C.byref(f), C.byref(t)) == 0          # hipMemGetInfo: what the device itself reports
    return f.value


def test_large_v3_full_depth_f16(product_lib, checker_lib):
    model = _model("f16", checker_lib); pcm = synth.make_pcm(30.0, seed=2024)
    prod = sc.ProductSide(product_lib, model); chk = tp.make_checker(model, checker_lib)
    chk.n_threads = _threads()
    try:
        mel_r, _ = chk.mel(pcm); mel_p, _ = prod.mel(pcm)
        assert np.abs(mel_p - mel_r).max() <= tp.TOL["mel"][0]
        er = chk.encode(0, 0); ep = prod.encode(0, 0)
        stats = {k: sc.err_stats(ep[k], er[k]) for k in er}
        print("large-v3 f16 encoder:", {k: (round(v["rms_rel"], 6), round(v["max_abs"], 5)) for k, v in stats.items()})
        for k, st in stats.items():
            # 32 layers: the absolute bound of the 6-layer models is doubled (residual stream rms grows with depth), rms-rel stays
            assert st["rms_rel"] <= tp.TOL[k][1] and st["max_abs"] <= 2 * tp.TOL[k][0], (k, st)
        prompt = tp.sot_prompt(chk, prod)
        lr = chk.decode(prompt, 0); lp = prod.decode(prompt, 0)
        ls = [sc.err_stats(lp, lr)]
        for i in range(4):
            tok = int(np.argmax(lr[:50256]))
            lr = chk.decode([tok], len(prompt) + i); lp = prod.decode([tok], len(prompt) + i)
            ls.append(sc.err_stats(lp, lr))
            top2 = np.partition(lr, -2)[-2:]
            if top2[1] - top2[0] > 4 * tp.LOGIT_ABS:
                assert int(np.argmax(lp)) == int(np.argmax(lr)), i
        print("large-v3 f16 logits:", [(round(s["rms_rel"], 6), round(s["max_abs"], 5)) for s in ls])
        for s in ls:
            assert s["rms_rel"] <= tp.LOGIT_RMS and s["max_abs"] <= 2 * tp.LOGIT_ABS, s
    finally:
        prod.close(); chk.close()


def test_large_v3_full_depth_q5_1(product_lib, checker_lib):
    model = _model("q5_1", checker_lib); pcm = synth.make_pcm(30.0, seed=2024)
    # HBM footprint of the weights: quantised size, not an f16 expansion
    assert product_lib.wmi_device_count() > 0
    free0 = _free_bytes()
    probe = sc.ProductSide(product_lib, model)
    try:
        arena = product_lib.wmi_weights_bytes(probe.ctx, 0); mats = product_lib.wmi_weights_bytes(probe.ctx, 1)
        assert product_lib.wmi_weights_bytes(probe.ctx, 2) == 7                     # q5_1 blocks in HBM
        used = free0 - _free_bytes()
        print(f"large-v3 q5_1: file {len(model) / 1e6:.0f} MB, weight arena {arena / 1e6:.0f} MB (matrices {mats / 1e6:.0f} MB), context total {used / 1e6:.0f} MB")
        assert 1.0e9 < arena < 1.3e9 and mats < 1.2e9, (arena, mats)
        assert used < arena + 1.2e9                                                   # + KV caches, activations, q8 rows, logits
    finally:
        probe.close()
    got, ref, pert = tp._quantised_case(product_lib, checker_lib, model, pcm, 0, 4, [], "large-v3 q5_1", ref_threads=_threads())
    for k in ("embd_enc", "cross_k", "cross_v"):
        print("large-v3 q5_1", k, "product vs reference", sc.err_stats(got[k], ref[k])["rms_rel"], "| reference vs itself (PCM x (1 + 1e-6))", sc.err_stats(pert[k], ref[k])["rms_rel"])
    print("large-v3 q5_1 logits rms-rel product:", [round(sc.err_stats(a, b)["rms_rel"], 5) for a, b in zip(got["logits"], ref["logits"])],
          "reference self:", [round(sc.err_stats(a, b)["rms_rel"], 5) for a, b in zip(pert["logits"], ref["logits"])])


def _beam_params(node, beam_size=5):
    from godot_whisper_amd import abi
    q = node.full_params("", 0)
    p = node.lib.whisper_full_default_params(abi.WHISPER_SAMPLING_BEAM_SEARCH)
    for f in ("language", "audio_ctx", "split_on_word", "token_timestamps", "suppress_non_speech_tokens", "single_segment",
              "max_tokens", "entropy_thold", "initial_prompt"):
        setattr(p, f, getattr(q, f))
    p.beam_search.beam_size = beam_size
    p.temperature_inc = 0.0          # the fallback threshold is discontinuous in the logits; beams are what is under test
    return p


def _common_prefix(a, b):
    n = min(len(a), len(b))
    same = a[:n, 0] == b[:n, 0]
    return n if same.all() else int(np.argmin(same))


@pytest.mark.parametrize("kind", ["f16", "q5_1"])
def test_large_v3_beam5_transcription_vs_reference(product_lib, checker_lib, kind):
    """BASELINE configs[4] as it is benchmarked: whisper_full with beam_size = 5 (host parameter set on the reference's beam-search
    defaults) on large-v3 at full depth, one 30 s chunk, against the compiled reference's token stream.

    Beam candidates are DRAWN (whisper_sample_token_topk: std::discrete_distribution over the filtered probabilities, W/whisper.cpp:
    4834-4909), so a stream follows the reference exactly as long as every uniform number lands in the same CDF cell.  f16: the
    product's logits are within 5e-4 of the reference's — the streams must agree up to the first near-tie like every greedy case.
    q5_1: the reference's OWN logits move by ~1e-2 when the PCM is scaled by (1 + 1e-6) (8-bit activation quantiser in front of
    every projection, module doc), and on the flat distributions of random weights (p_max ~ 0.2) that re-cells draws from the
    first token on: the yardstick is therefore the reference against itself — the product must follow the reference at least as
    far as the reference follows its own perturbed run, and its transcription must be well-formed."""
    if checker_lib is None:
        pytest.skip("needs the compiled reference")
    from godot_whisper_amd import host
    model = _model(kind, checker_lib); pcm = synth.make_pcm(30.0, seed=4321)
    node = host.SpeechToText(product_lib); node.set_language_model(model); node.language = "en"
    ref = host.SpeechToText(checker_lib); ref.set_language_model(model); ref.language = "en"
    try:
        pr = _beam_params(ref); pr.n_threads = min(32, _threads())
        want = ref.transcribe(pcm, params=pr)
        assert ref.last_ret == 0 and len(want) > 1
        got = node.transcribe(pcm, params=_beam_params(node))
        assert node.last_ret == 0 and len(got) > 1
        g, w = tp.gu.tokens_array(got), tp.gu.tokens_array(want)
        first = _common_prefix(g, w)
        print(f"large-v3 {kind} beam 5: product {len(g)} / reference {len(w)} tokens, identical ids up to {first};",
              "ids", g[:6, 0].astype(int).tolist(), "vs", w[:6, 0].astype(int).tolist(),
              "p", np.round(g[:4, 2], 3).tolist(), "vs", np.round(w[:4, 2], 3).tolist())
        # well-formed whatever the draws: token ids in range, monotone token times inside the chunk, probabilities in (0, 1]
        assert len(g) >= 4 and np.all(g[:, 0] >= 0) and np.all(g[:, 0] < 51866) and np.all(g[:, 2] > 0) and np.all(g[:, 2] <= 1.0 + 1e-6)
        assert np.all(g[:, 6] <= g[:, 7] + 1) and g[:, 7].max() <= 3000
        if kind == "q5_1":
            pert = ref.transcribe((pcm.astype(np.float64) * (1.0 + 1e-6)).astype(np.float32), params=pr)
            pa = tp.gu.tokens_array(pert)
            self_first = _common_prefix(pa, w)
            # ptsum (the probability mass on the timestamp tokens) of the FIRST step is the same quantity in both runs whatever
            # token was drawn afterwards: the reference's own response to the perturbation, in the units of the bound below
            self_noise = float(abs(pa[0, 5] - w[0, 5]))
            print(f"large-v3 q5_1 beam 5: the reference follows its own (1 + 1e-6)-scaled run for {self_first} tokens;",
                  f"its first-step ptsum moves by {self_noise:.3e}")
            assert first >= min(self_first, 3), (first, self_first)
            n = first
        else:
            assert first >= 3, (first, g[:4, :3], w[:4, :3])
            if first < min(len(g), len(w)):
                assert abs(g[first, 2] - w[first, 2]) <= 5e-2, (first, g[first], w[first])
            n = first
        if n:
            dmax = float(np.abs(g[:n, [2, 4, 5]] - w[:n, [2, 4, 5]]).max())
            print(f"large-v3 {kind} beam 5: max |p, pt, ptsum difference| over the {n} common tokens {dmax:.3e}")
            assert dmax <= (5e-2 if kind == "q5_1" else 1e-2)
        if n == len(g) == len(w):
            assert bytes(got[0]) == bytes(want[0])
            if kind == "q5_1":
                # tid is an arg-max over ~1500 timestamp logits that random weights leave nearly flat (pt ~ 1e-3): under the 8-bit
                # activation quantiser a near-tie may resolve differently (the pt of both picks is within the bound just asserted);
                # where the timestamp token agrees, the times derived from it must agree too
                same = g[:, 1] == w[:, 1]
                print(f"large-v3 q5_1 beam 5: timestamp token equal on {int(same.sum())} of {len(g)} tokens")
                assert np.array_equal(g[same][:, [6, 8]], w[same][:, [6, 8]])
            else:
                assert np.array_equal(g[:, [1, 6, 8]], w[:, [1, 6, 8]]) and np.array_equal(g[:-1, 7], w[:-1, 7])
    finally:
        node.close(); ref.close()

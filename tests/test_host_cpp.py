"""The host side in the reference's own language (godot-whisper_amd/host_cpp: SpeechToText / AudioStreamToText /
CaptureStreamToText over the C ABI) against the Python mirror the parity tests use — both must make the same calls
and return the same token dictionaries."""
import json
import pathlib
import struct
import subprocess

import numpy as np
import pytest

import golden_util as gu
from godot_whisper_amd import host, synth

DEMO = pathlib.Path(__file__).resolve().parent.parent / "godot-whisper_amd" / "wmi_host_demo"


def run_demo(tmp_path, model: bytes | None, pcm_bytes: bytes, mode: str, *args):
    assert DEMO.exists(), "build it: python __graft_entry__.py build"
    mp = tmp_path / "model.bin"; mp.write_bytes(model or b"")
    pp = tmp_path / "pcm.f32"; pp.write_bytes(pcm_bytes)
    out = subprocess.run([str(DEMO), str(mp), str(pp), mode, *[str(a) for a in args]], capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-2000:]
    return json.loads(out.stdout.decode("latin-1"))


def test_vad_matches_python_mirror(tmp_path):
    rng = np.random.default_rng(3)
    sr = 16000
    cases = [synth.make_pcm(4.0, seed=1), synth.make_pcm(4.0, seed=2, gate=True), np.zeros(4 * sr, np.float32),
             (rng.standard_normal(4 * sr) * 1e-5).astype(np.float32), synth.make_pcm(2.0, seed=3)]
    tail_quiet = synth.make_pcm(4.0, seed=5).copy(); tail_quiet[-sr // 2:] *= 1e-6; cases.append(tail_quiet)
    node = host.SpeechToText(lib=None)
    for pcm in cases:
        want = node.voice_activity_detection(pcm.copy())
        got = run_demo(tmp_path, None, pcm.astype("<f4").tobytes(), "vad")["vad"]
        assert got == want


def _as_rows(tr):
    return np.asarray([[t["id"], t["tid"], t["p"], t["plog"], t["pt"], t["ptsum"], t["t0"], t["t1"], t["vlen"]] for t in tr["tokens"]],
                      np.float64).reshape(-1, 9)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,language,prompt,actx", [("micro.en", 1, "", 0), ("micro", 4, " Hello, world!", 0), ("micro.en", 1, "", 328)])
def test_transcribe_equals_python_mirror(product_lib, tmp_path, shape, language, prompt, actx):
    model = synth.make_model(shape, seed=8); pcm = synth.make_pcm(9.0, seed=8)
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    node.language = product_lib.whisper_lang_str(language - 1).decode()          # enum index -> code, as the C++ side does
    try:
        want = node.transcribe(pcm, prompt, actx)
    finally:
        node.close()
    got = run_demo(tmp_path, model, pcm.astype("<f4").tobytes(), "transcribe", language, prompt, actx)
    assert got["ok"] and got["full_text"].encode("latin-1") == bytes(want[0])
    w = gu.tokens_array(want); g = _as_rows(got)
    assert g.shape == w.shape and np.array_equal(g[:, [0, 1, 6, 7]], w[:, [0, 1, 6, 7]])
    assert np.abs(g - w).max() <= 1e-6                      # same library, same kernels: float text round trip only
    assert [t["text"].encode("latin-1") for t in got["tokens"]] == [d["text"] for d in want[1:]]


@pytest.mark.gpu
def test_stream_and_batch_equal_python_mirror(product_lib, tmp_path):
    model, pcm = gu.stream_inputs()
    node = host.CaptureStreamToText(product_lib, transcribe_interval=gu.STREAM_INTERVAL); node.set_language_model(model)
    try:
        want = list(node.stream(pcm, max_calls=8))
    finally:
        node.close()
    got = run_demo(tmp_path, model, pcm.astype("<f4").tobytes(), "stream", 1)
    assert len(got) == len(want)
    for g, (finish, text, n, actx, toks) in zip(got, want):
        assert (g["finish"], g["n_samples"], g["audio_ctx"]) == (finish, n, actx)
        assert g["ids"] == [t["id"] for t in toks]
        assert g["text"].encode("latin-1").decode("utf-8", errors="replace") == text

    bufs = [synth.make_pcm(s, seed=60 + i) for i, s in enumerate((6.0, 11.0, 3.0))]
    node = host.SpeechToText(product_lib); node.set_language_model(model)
    try:
        want_b = node.transcribe_batch(bufs, "", 0)
    finally:
        node.close()
    blob = struct.pack("<i", len(bufs)) + b"".join(struct.pack("<i", b.size) for b in bufs) + b"".join(b.astype("<f4").tobytes() for b in bufs)
    got_b = run_demo(tmp_path, model, blob, "batch", 1)
    assert len(got_b) == len(want_b)
    for g, w in zip(got_b, want_b):
        assert np.array_equal(_as_rows(g)[:, [0, 6, 7]], gu.tokens_array(w)[:, [0, 6, 7]])


@pytest.mark.gpu
@pytest.mark.parametrize("mix_rate,interp", [(48000, 2), (44100, 2), (44100, 1)])
def test_resample_of_the_cpp_host_equals_the_converter(product_lib, tmp_path, mix_rate, interp):
    """SpeechToText::resample (src/speech_to_text.cpp:353-376) in the C++ mirror: fold + libsamplerate's src_simple on the device.
    The frames must be the sequential converter's (oracle/host_dsp.c) bit for bit — compared through a hash of their bits — and the
    "size differ" message must appear exactly where the reference prints it (44.1 kHz: the converter stops one frame early)."""
    import ctypes as C
    root = pathlib.Path(__file__).resolve().parent.parent
    so = root / "oracle" / "liboracle_dsp.so"
    if not so.exists():
        pytest.skip("oracle/liboracle_dsp.so not built")
    dsp = C.CDLL(str(so))
    dsp.oracle_resample_audio_buffer.restype = C.c_uint32
    dsp.oracle_resample_audio_buffer.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    dsp.oracle_downmix_stereo.restype = None
    dsp.oracle_downmix_stereo.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
    raw = (root / "godot-whisper_amd" / "csrc" / "data" / ("sinc_fastest.bin" if interp == 2 else "sinc_medium.bin")).read_bytes()
    inc, cnt = struct.unpack("<ii", raw[:8]); tab = np.frombuffer(raw[8:], "<f4", cnt).copy()
    n = mix_rate * 3
    rng = np.random.default_rng(mix_rate + interp)
    fr = (0.3 * rng.standard_normal((n, 2))).astype(np.float32)
    mono = np.zeros(n, np.float32); dsp.oracle_downmix_stereo(n, fr.ctypes.data, mono.ctypes.data)
    want = np.zeros(n, np.float32)
    got_n = dsp.oracle_resample_audio_buffer(mono.ctypes.data, n, mix_rate, 16000, tab.ctypes.data, cnt, inc, want.ctypes.data)
    want = want[:got_n]
    h = 1469598103934665603
    for b in want.tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    model = synth.make_model("micro.en", seed=8)
    got = run_demo(tmp_path, model, fr.astype("<f4").tobytes(), "resample", mix_rate, interp)
    expected = n * 16000 // mix_rate
    assert got["frames"] == got_n and got["fnv1a"] == "%016x" % h
    assert got["error"] == ("" if got_n == expected else f"size differ exp: {expected} res: {got_n}")

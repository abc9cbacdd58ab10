"""One tiny end-to-end invocation of the hot path on cuda:0, checked against the oracle."""
from __future__ import annotations

import numpy as np


def run():
    import __graft_entry__ as entry
    entry.load_oracle()
    from godot_whisper_amd import host, runtime, synth
    from oracle import port, reflib
    import sys, pathlib
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "tests"))
    import stage_compare as sc

    lib = runtime.require_gpu()            # raises if the HIP library or a GPU is missing: no fallback
    runtime.silence_logs(lib)
    model = synth.make_model("micro.en", seed=1234)
    pcm = synth.make_pcm(4.0, seed=3)
    prod = sc.ProductSide(lib, model)
    chk = sc.RefSide(_quiet(reflib.lib()), model) if reflib.available() else port.PortSide(model)
    mel_r, _ = chk.mel(pcm); mel_p, _ = prod.mel(pcm)
    assert np.abs(mel_r - mel_p).max() <= 2e-6
    er = chk.encode(0, 328); ep = prod.encode(0, 328)
    for k in er:
        st = sc.err_stats(ep[k], er[k])
        assert st["rms_rel"] <= 2e-3, (k, st)
    sot = lib.whisper_token_sot(prod.ctx)
    lr = chk.decode([sot], 0); lp = prod.decode([sot], 0)
    for i in range(4):
        assert np.abs(lr - lp).max() <= 6e-2
        tok = int(np.argmax(lr[:50256]))
        assert tok == int(np.argmax(lp[:50256]))
        lr = chk.decode([tok], 1 + i); lp = prod.decode([tok], 1 + i)
    prod.close(); chk.close()
    node = host.AudioStreamToText(lib); node.set_language_model(model)
    text = node.get_text(pcm)
    dsp = _resampler_check(lib, node)
    node.close()
    print("smoke ok: micro.en 4 s chunk, logits within 6e-2 of the oracle, greedy tokens identical; text =", repr(text[:40]) + dsp)


def _resampler_check(lib, node) -> str:
    """0.5 s of 44.1 kHz capture frames through the node's device resampler against oracle/host_dsp.c (bit for bit), when that checker is built."""
    import ctypes as C, pathlib, struct
    root = pathlib.Path(__file__).resolve().parent.parent
    so = root / "oracle" / "liboracle_dsp.so"
    if not so.exists():
        return ""
    raw = (root / "godot-whisper_amd" / "csrc" / "data" / "sinc_fastest.bin").read_bytes()
    inc, cnt = struct.unpack("<ii", raw[:8]); tab = np.frombuffer(raw[8:], "<f4", cnt).copy()
    dsp = C.CDLL(str(so))
    dsp.oracle_resample_audio_buffer.restype = C.c_uint32
    dsp.oracle_resample_audio_buffer.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    n = 22050
    x = (0.3 * np.random.default_rng(5).standard_normal(n)).astype(np.float32)
    want = np.zeros(n, np.float32)
    nw = dsp.oracle_resample_audio_buffer(x.ctypes.data, n, 44100, 16000, tab.ctypes.data, cnt, inc, want.ctypes.data)
    got = np.zeros(n, np.float32)
    ng = lib.wmi_resample(node.ctx, x.ctypes.data_as(C.c_void_p), n, 44100, 16000, 2, 0, got.ctypes.data_as(C.c_void_p), n)
    assert ng == nw and got[:ng].tobytes() == want[:nw].tobytes(), (ng, nw)
    return f"; resampler 44.1 -> 16 kHz: {ng} frames identical to the sequential converter"


def _quiet(lib):
    import ctypes as C
    from godot_whisper_amd import abi
    cb = abi.ggml_log_callback(lambda lvl, txt, ud: None)
    lib.whisper_log_set(C.cast(cb, C.c_void_p), None)
    lib._cb = cb
    return lib

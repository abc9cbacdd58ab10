#!/usr/bin/env python3
"""Beam search on several chunks of ONE GPU: same-device replica contexts (wmi_pool_init with a device listed k times) run their
chunks concurrently on their own streams.  Prints ms per chunk for k = 1, 2, 4, 8 (base.en f16 and large-v3 q5_1)."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import abi, host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)

def run(name, model, n_chunks, ks, beam=5):
    pcms = [synth.make_pcm(30.0, seed=7000 + i) for i in range(n_chunks)]
    buf = C.create_string_buffer(model, len(model))
    one = host.SpeechToText(lib); one.language = "en"
    q = one.full_params("", 0)
    p = lib.whisper_full_default_params(abi.WHISPER_SAMPLING_BEAM_SEARCH if beam > 1 else abi.WHISPER_SAMPLING_GREEDY)
    for f in ("language", "audio_ctx", "split_on_word", "token_timestamps", "suppress_non_speech_tokens", "single_segment", "max_tokens", "entropy_thold", "initial_prompt"):
        setattr(p, f, getattr(q, f))
    if beam > 1: p.beam_search.beam_size = beam
    p.temperature_inc = 0.0
    ptrs = (C.c_void_p * n_chunks)(*[b.ctypes.data for b in pcms]); lens = (C.c_int * n_chunks)(*[b.size for b in pcms])
    base = None
    for k in ks:
        devs = (C.c_int * k)(*([0] * k))
        pool = lib.wmi_pool_init(C.cast(buf, C.c_void_p), len(model), devs, k)
        assert pool
        assert lib.wmi_pool_full(pool, p, ptrs, lens, n_chunks) == 0
        t0 = time.perf_counter(); reps = 2
        for _ in range(reps):
            assert lib.wmi_pool_full(pool, p, ptrs, lens, n_chunks) == 0
        dt = (time.perf_counter() - t0) / reps
        toks = []
        for c in range(n_chunks):
            ctx = lib.wmi_pool_select(pool, c)
            node = host.SpeechToText(lib); node.ctx = ctx; got = node.collect(); node.ctx = None
            toks.append(tuple(t["id"] for t in got[1:]))
        if base is None: base = toks
        print(f"{name}: beam {beam}, {n_chunks} chunks, {k} replica context(s) on one GPU: {dt * 1e3:8.2f} ms per call = {dt * 1e3 / n_chunks:7.2f} ms per chunk, "
              f"{n_chunks * 30.0 / dt:8.1f} x realtime; results equal the 1-context run: {toks == base}", flush=True)
        lib.wmi_pool_free(pool)

which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("base", "both"):
    run("base.en f16", synth.make_model("base.en", seed=1234), 8, (1, 2, 4, 8))
if which in ("large", "both"):
    m = synth.quantize_model(synth.make_model("large-v3", seed=2024), "q5_1")
    run("large-v3 q5_1", m, 8, (1, 2, 4, 8))

#!/usr/bin/env python3
"""Beam-5 chunks through wmi_full_batch with 0 / 1 / 3 replica contexts, and through a 4-context same-device pool (own arena copies)."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import abi, host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
which = sys.argv[1] if len(sys.argv) > 1 else "base"
model = synth.make_model("base.en", seed=1234) if which == "base" else synth.quantize_model(synth.make_model("large-v3", seed=2024), "q5_1")
n = 8
pcms = [synth.make_pcm(30.0, seed=7000 + i) for i in range(n)]
node = host.SpeechToText(lib); node.set_language_model(model); node.language = "en"
q = node.full_params("", 0)
p = lib.whisper_full_default_params(abi.WHISPER_SAMPLING_BEAM_SEARCH)
for f in ("language", "audio_ctx", "split_on_word", "token_timestamps", "suppress_non_speech_tokens", "single_segment", "max_tokens", "entropy_thold", "initial_prompt"):
    setattr(p, f, getattr(q, f))
p.beam_search.beam_size = 5; p.temperature_inc = 0.0
for n_rep in (0, 1, 3, 7, 3, 0):
    lib.wmi_set_batch_replicas(node.ctx, n_rep)
    node.transcribe_batch(pcms, params=p)
    t0 = time.perf_counter(); reps = 2
    for _ in range(reps): node.transcribe_batch(pcms, params=p)
    dt = (time.perf_counter() - t0) / reps
    print(f"{which}: wmi_full_batch beam 5 x {n} chunks, {n_rep} replicas: {dt*1e3:.1f} ms per call = {dt*1e3/n:.2f} ms per chunk", flush=True)
buf = C.create_string_buffer(model, len(model))
for k in (1, 4):
    devs = (C.c_int * k)(*([0] * k))
    pool = lib.wmi_pool_init(C.cast(buf, C.c_void_p), len(model), devs, k)
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in pcms]); lens = (C.c_int * n)(*[b.size for b in pcms])
    lib.wmi_pool_full(pool, p, ptrs, lens, n)
    t0 = time.perf_counter()
    for _ in range(2): lib.wmi_pool_full(pool, p, ptrs, lens, n)
    dt = (time.perf_counter() - t0) / 2
    print(f"{which}: pool of {k} contexts (arena copies): {dt*1e3:.1f} ms per call = {dt*1e3/n:.2f} ms per chunk", flush=True)
    lib.wmi_pool_free(pool)

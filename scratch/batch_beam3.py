#!/usr/bin/env python3
import ctypes as C, sys, time, os
sys.path.insert(0, ".")
import __graft_entry__ as entry
entry.load_package()
mode = sys.argv[1]
if "torch" in mode:
    import torch; torch.cuda.init(); x = torch.zeros(4, device="cuda")
from godot_whisper_amd import abi, host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
if "base" in mode:                      # an earlier context with 16 lock-step lanes, freed again (what bench.py has done by then)
    n0 = host.SpeechToText(lib); n0.set_language_model(synth.make_model("base.en", seed=1234)); n0.language = "en"
    g0 = n0.full_params("", 0); g0.temperature_inc = 0.0
    n0.transcribe_batch([synth.make_pcm(30.0, seed=10 + i) for i in range(16)], params=g0)
    n0.close()
model = synth.quantize_model(synth.make_model("large-v3", seed=2024), "q5_1")
n = 8
pcms = [synth.make_pcm(30.0, seed=5000 + i) for i in range(n)]
node = host.SpeechToText(lib); node.set_language_model(model); node.language = "en"
if "eager" in mode: lib.wmi_set_batch_replicas(node.ctx, 3)
q = node.full_params("", 0)
if "single" in mode:
    for _ in range(3): node.transcribe(pcms[0], params=q)
if "lanes" in mode:
    g = node.full_params("", 0); g.temperature_inc = 0.0
    node.transcribe_batch(pcms, params=g)
p = lib.whisper_full_default_params(abi.WHISPER_SAMPLING_BEAM_SEARCH)
for f in ("language", "audio_ctx", "split_on_word", "token_timestamps", "suppress_non_speech_tokens", "single_segment", "max_tokens", "entropy_thold", "initial_prompt"):
    setattr(p, f, getattr(q, f))
p.beam_search.beam_size = 5; p.temperature_inc = 0.0
seq = tuple(int(x) for x in os.environ.get('SEQ', '0,3,3').split(','))
for n_rep in seq:
    lib.wmi_set_batch_replicas(node.ctx, n_rep)
    node.transcribe_batch(pcms, params=p)
    t0 = time.perf_counter(); reps = 2
    for _ in range(reps): node.transcribe_batch(pcms, params=p)
    dt = (time.perf_counter() - t0) / reps
    print(f"[{mode}] large-v3 q5_1 beam 5 x {n} chunks, {n_rep} replicas: {dt*1e3/n:.2f} ms per chunk", flush=True)

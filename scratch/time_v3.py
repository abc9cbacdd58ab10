"""Scratch: timing of the large-v3 q5_1 path (BASELINE configs[4] per-GPU share: one 30 s chunk, greedy and beam 5)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
entry.load_package(); entry.load_oracle()
from godot_whisper_amd import abi, host, runtime, synth
from oracle import reflib
import test_gpu_large_v3 as tl

shape = os.environ.get("SHAPE", "large-v3"); qt = os.environ.get("QT", "q5_1")
lib = runtime.require_gpu(); runtime.silence_logs(lib)
t0 = time.time()
m = synth.make_model(shape, seed=2024)
if qt != "f16":
    m = tl._ref_quantize_model(reflib.lib(), m, qt) if reflib.available() else synth.quantize_model(m, qt)
print("model", shape, qt, len(m) / 1e6, "MB in", round(time.time() - t0, 1), "s", flush=True)
node = host.SpeechToText(lib); node.set_language_model(m); node.language = "en"
pcm = synth.make_pcm(30.0, seed=7)
cases = (("greedy mt16", 0, 1, 16), ("beam5 mt16", 1, 5, 16), ("greedy mt0", 0, 1, 0), ("beam5 mt0", 1, 5, 0))[: int(os.environ.get("ONLY", "4"))]
if os.environ.get("CASE"): cases = [c for c in cases if c[0].startswith(os.environ["CASE"])]
for name, strat, bs, mt in cases:
    p = lib.whisper_full_default_params(strat)
    q = node.full_params("", 0)
    for f in ("language", "audio_ctx", "split_on_word", "token_timestamps", "suppress_non_speech_tokens", "single_segment", "entropy_thold", "initial_prompt"):
        setattr(p, f, getattr(q, f))
    p.max_tokens = mt; p.temperature_inc = 0.0
    if strat == 1: p.beam_search.beam_size = bs
    node.transcribe(pcm, params=p)
    lib.whisper_reset_timings(node.ctx)
    t0 = time.perf_counter(); n = 3
    for _ in range(n): r = node.transcribe(pcm, params=p)
    dt = (time.perf_counter() - t0) / n
    t6 = (C.c_int64 * 6)(); n5 = (C.c_int32 * 5)(); lib.wmi_get_timings(node.ctx, t6, n5)
    print(f"{name}: {dt*1e3:.1f} ms/chunk ({30/dt:.0f}x) tokens {len(r)-1 if r else 0} | mel {t6[0]/n/1e3:.2f} enc {t6[1]/n/1e3:.2f} dec {t6[2]/n/1e3:.2f} ({n5[1]//n} calls) batchd {t6[3]/n/1e3:.2f} ({n5[2]//n} tok) prompt {t6[4]/n/1e3:.2f} sample {t6[5]/n/1e3:.2f}", flush=True)
if os.environ.get("BATCH"):
    nb = int(os.environ["BATCH"])
    pcms = [synth.make_pcm(30.0, seed=50 + i) for i in range(nb)]
    p = node.full_params("", 0); p.temperature_inc = 0.0
    node.transcribe_batch(pcms, params=p)
    t0 = time.perf_counter(); r = node.transcribe_batch(pcms, params=p); dt = time.perf_counter() - t0
    t4 = (C.c_int64 * 4)(); ns = C.c_int32(); lib.wmi_get_batch_timings(node.ctx, t4, C.byref(ns))
    print(f"lock-step {nb} chunks: {dt*1e3:.1f} ms per call ({nb*30/dt:.0f}x) modes {list(node.last_modes)} | mel {t4[0]/1e3:.2f} enc {t4[1]/1e3:.2f} dec {t4[2]/1e3:.2f} ({ns.value} steps) emit {t4[3]/1e3:.2f}", flush=True)
print('greedy step chain on the GPU, us per step:', lib.wmi_bench_kernel(node.ctx, 20, 20), '| vocabulary projection (rotating copies), us:', lib.wmi_bench_kernel(node.ctx, 6, 100), flush=True)
node.close()

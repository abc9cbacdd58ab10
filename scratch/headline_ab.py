"""Scratch: headline call (base.en, one chunk, host params) — wall time and decode time per token; prints one line (A/B by environment)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model(os.environ.get("SHAPE", "base.en"), seed=1234))
pcm = [synth.make_pcm(30.0, seed=1234 + i) for i in range(8)]
p = node.full_params("", 0)
for i in range(100): lib.whisper_full(node.ctx, p, pcm[i % 8].ctypes.data_as(C.POINTER(C.c_float)), pcm[i % 8].size)
toks = [lib.whisper_full_get_token_id(node.ctx, 0, t) for t in range(lib.whisper_full_n_tokens(node.ctx, 0))]
best = 1e9
for rep in range(3):
    lib.whisper_reset_timings(node.ctx)
    n = 400; t0 = time.perf_counter()
    for i in range(n): lib.whisper_full(node.ctx, p, pcm[i % 8].ctypes.data_as(C.POINTER(C.c_float)), pcm[i % 8].size)
    dt = (time.perf_counter() - t0) / n
    t6 = (C.c_int64 * 6)(); n5 = (C.c_int32 * 5)(); lib.wmi_get_timings(node.ctx, t6, n5)
    best = min(best, dt)
    print(f"{os.environ.get('TAGNAME', 'default'):18s} {dt*1e3:7.4f} ms | enc {t6[1]/n/1e3:.3f} dec {t6[2]/n/1e3:.3f} = {t6[2]/max(n5[1],1):.2f} us per token ({n5[1]//n} calls) | tokens {toks[:6]}..{sum(toks)}", flush=True)
us = lib.wmi_bench_kernel(node.ctx, 20, 200)
print(f"{os.environ.get('TAGNAME', 'default'):18s} step chain on the GPU {us:.2f} us")
node.close()

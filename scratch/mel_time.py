"""Scratch: log-mel of one 30 s chunk, device-resident PCM, ms per call (stream-synchronised wall)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
import numpy as np
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model("micro.en", seed=1))
pcm = synth.make_pcm(30.0, seed=3)
f = pcm.ctypes.data_as(C.POINTER(C.c_float))
for _ in range(5): lib.whisper_pcm_to_mel(node.ctx, f, pcm.size, 1)
t0 = time.perf_counter(); n = 200
for _ in range(n): lib.whisper_pcm_to_mel(node.ctx, f, pcm.size, 1)
print("pcm_to_mel (host PCM): %.1f us per call" % ((time.perf_counter() - t0) / n * 1e6))
node.close()

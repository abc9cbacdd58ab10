"""Scratch: a few encoder passes of large-v3 q5_1 (for PMC collection)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
entry.load_package(); entry.load_oracle()
from godot_whisper_amd import host, runtime, synth
from oracle import reflib
import test_gpu_large_v3 as tl
lib = runtime.require_gpu(); runtime.silence_logs(lib)
shape = os.environ.get("SHAPE", "large-v3"); qt = os.environ.get("QT", "q5_1")
m = synth.make_model(shape, seed=2024)
if qt != "f16":
    m = tl._ref_quantize_model(reflib.lib(), m, qt) if reflib.available() else synth.quantize_model(m, qt)
node = host.SpeechToText(lib); node.set_language_model(m)
pcm = synth.make_pcm(30.0, seed=7)
assert lib.whisper_pcm_to_mel(node.ctx, pcm.ctypes.data_as(C.POINTER(C.c_float)), pcm.size, 4) == 0
for _ in range(int(os.environ.get("REPS", "2"))):
    assert lib.whisper_encode(node.ctx, 0, 4) == 0
node.close()

import sys, os, time, numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as e
e.load_package()
from godot_whisper_amd import runtime, synth, host
lib = runtime.require_gpu()
model = synth.make_model("base.en", seed=1234); pcm = synth.make_pcm(30.0, seed=1234)
node = host.SpeechToText(lib); node.set_language_model(model)
p = node.full_params("", 0)
for i in range(4):
    t0 = time.perf_counter(); r = node.transcribe(pcm, params=p); print("py wall %.3f ms" % ((time.perf_counter()-t0)*1e3))

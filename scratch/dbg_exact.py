import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from godot_whisper_amd import host, runtime, synth
import golden_util as gu
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model = synth.make_model("micro.en", seed=2024)
pcm = synth.make_pcm(30.0, seed=100)
node = host.SpeechToText(lib); node.set_language_model(model)
w = gu.tokens_array(node.transcribe(pcm, "", 0))
lib.wmi_set_lockstep_exact(1)
g = gu.tokens_array(node.transcribe_batch([pcm, pcm], "", 0)[1])
print("max |dp| per token:", np.abs(g[:, 2] - w[:, 2]).round(7))

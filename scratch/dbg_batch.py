import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from godot_whisper_amd import host, runtime, synth
import golden_util as gu
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model = synth.make_model("micro", seed=2024)
secs = [30.0, 11.0, 4.0, 47.0, 30.0, 0.5, 22.5, 30.0, 8.0, 30.0, 15.0]
pcms = [synth.make_pcm(s, seed=100 + i, gate=(i % 3 == 1)) for i, s in enumerate(secs)]
def P(node): return gu.param_variants(node)["host_prompt"]
node = host.SpeechToText(lib); node.set_language_model(model)
def show(tag, r):
    a = gu.tokens_array(r)
    print(tag, [(int(x[0]), round(x[2], 4)) for x in a[:6]])
w = node.transcribe(pcms[10], params=P(node)); show("alone     ", w)
for combo in ([10, 10], [8, 9, 10], [10, 8], [0, 10]):
    g = node.transcribe_batch([pcms[i] for i in combo], params=P(node))
    for i, c in enumerate(combo):
        show(f"batch{combo}[{c}] mode={node.last_modes[i]}", g[i])
w0 = node.transcribe(pcms[0], params=P(node)); show("alone c0  ", w0)
w8 = node.transcribe(pcms[8], params=P(node)); show("alone c8  ", w8)

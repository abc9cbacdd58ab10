import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from godot_whisper_amd import host, runtime, synth
import golden_util as gu
lib = runtime.require_gpu(); runtime.silence_logs(lib)
def show(tag, r):
    a = gu.tokens_array(r)
    print(tag, [(int(x[0]), round(float(x[2]), 4)) for x in a[:6]])
for shape, prompt, lang in (("micro", " Hello, world! It's 42.", "en"), ("micro.en", " Hello, world! It's 42.", "en"), ("micro", "", "de")):
    model = synth.make_model(shape, seed=2024)
    pcm = synth.make_pcm(15.0, seed=110, gate=True)
    node = host.SpeechToText(lib); node.set_language_model(model); node.language = lang
    w = node.transcribe(pcm, prompt, 0); show(f"{shape} prompt={bool(prompt)} alone ", w)
    g = node.transcribe_batch([pcm, pcm], prompt, 0); show(f"{shape} prompt={bool(prompt)} batch mode={node.last_modes}", g[1])
    node.close()

import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as e
e.load_package(); e.load_oracle()
from godot_whisper_amd import runtime, synth, host
import golden_util as gu
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model, pcm, actx = gu.case_inputs("en30")
node = host.SpeechToText(lib); node.set_language_model(model)
r = node.transcribe(pcm, "", 0)
print("ret", node.last_ret, [(d["id"], round(d["p"],4)) for d in r[1:]])

import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import ctypes as C
import __graft_entry__ as ge
ge.load_package()
from godot_whisper_amd import host, synth, abi
from oracle import reflib
import golden_util as gu
lib = reflib.lib()
cb = abi.ggml_log_callback(lambda lvl, txt, ud: None); lib.whisper_log_set(C.cast(cb, C.c_void_p), None)
model = synth.make_model("micro", seed=2024)
pcm = synth.make_pcm(15.0, seed=110, gate=True)
node = host.SpeechToText(lib); node.set_language_model(model)
r = node.transcribe(pcm, " Hello, world! It's 42.", 0)
a = gu.tokens_array(r)
print("ref", [(int(x[0]), round(float(x[2]), 4)) for x in a[:6]])

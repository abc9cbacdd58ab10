"""Scratch: WMI_DEBUG_EMIT / WMI_DEBUG_TIMING lines of a few headline calls (base.en, host params)."""
import ctypes as C, os, sys
os.environ["WMI_DEBUG_EMIT"] = "1"; os.environ["WMI_DEBUG_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu()
node = host.SpeechToText(lib); node.set_language_model(synth.make_model("base.en", seed=1234))
pcm = synth.make_pcm(30.0, seed=1234)
p = node.full_params("", 0)
for i in range(40): lib.whisper_full(node.ctx, p, pcm.ctypes.data_as(C.POINTER(C.c_float)), pcm.size)
node.close()

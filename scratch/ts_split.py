"""Scratch: WMI_DEBUG_EMIT=1 split of the token-timestamp host time (window sums / walks) on the headline chunk."""
import os, sys
os.environ["WMI_DEBUG_EMIT"] = "1"
sys.path.insert(0, ".")
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu()
node = host.SpeechToText(lib); node.set_language_model(synth.make_model("base.en", seed=1234))
pcm = synth.make_pcm(30.0, seed=1234)
for _ in range(6): node.transcribe(pcm, "", 0)
node.close()

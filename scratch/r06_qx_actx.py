"""Scratch: the quantised cross-attention forms (one launch / two) over ragged audio_ctx values; prints RESULT json (compare across processes)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
out = {}
for shape, qt in (("base.en", "q5_1"), ("small", "q4_0")):
    node = host.SpeechToText(lib); node.set_language_model(synth.quantize_model(synth.make_model(shape, seed=77), qt))
    node.language = "en" if shape.endswith(".en") else "de"
    for actx in (64, 100, 183, 192, 193, 400, 777, 1201, 1500):
        pcm = synth.make_pcm(min(30.0, actx / 50.0), seed=actx)
        p = node.full_params("", actx); p.temperature_inc = 0.0; p.max_tokens = 12
        r = node.transcribe(pcm, params=p)
        out["%s:%s:%d" % (shape, qt, actx)] = [[int(t["id"]), float(t["p"]), float(t["plog"])] for t in r[1:]]
    node.close()
print("RESULT" + json.dumps(out))

"""Scratch: the first 12 streaming calls of the configs[2] test on the product and the reference, ids + p of every call."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as entry
entry.load_package(); entry.load_oracle()
from godot_whisper_amd import host, runtime, synth
from oracle import reflib
import golden_util as gu
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model = synth.make_model("small", seed=77)
pcm = synth.make_pcm(600.0, seed=21, gate=True)[: 16000 * 12]
def run(L):
    node = host.CaptureStreamToText(L, transcribe_interval=0.3); node.language = "de"
    if L is lib: node.device_vad = True
    node.set_language_model(model)
    out = []
    for fin, text, n_used, actx, toks in node.stream(pcm, max_calls=12):
        out.append((n_used, actx, gu.tokens_array([b""] + toks)))
    node.close()
    return out
mine = run(lib)
if os.environ.get("NOREF") == "1":
    for i, (n, a, g) in enumerate(mine): print(i, n, a, g[:6, 0].astype(int).tolist(), np.round(g[:6, 2], 4).tolist())
else:
    rl = reflib.lib()
    ref = run(rl)
    for i, ((n, a, g), (n2, a2, w)) in enumerate(zip(mine, ref)):
        k = min(len(g), len(w)); same = g[:k, 0] == w[:k, 0]; first = k if same.all() else int(np.argmin(same))
        print(i, n, a, "first", first, "ids", g[:5, 0].astype(int).tolist(), "vs", w[:5, 0].astype(int).tolist(), "p", np.round(g[:5, 2], 4).tolist(), "vs", np.round(w[:5, 2], 4).tolist())

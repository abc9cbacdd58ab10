import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from godot_whisper_amd import host, runtime, synth
import golden_util as gu
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model = synth.make_model("micro.en", seed=2024)
secs = [30.0, 11.0, 4.0, 47.0, 30.0, 0.5, 22.5, 30.0]
if os.environ.get("SECS"): secs = [float(x) for x in os.environ["SECS"].split(",")]
pcms = [synth.make_pcm(s, seed=100 + i, gate=(i % 3 == 1)) for i, s in enumerate(secs)]
want = []
for b in pcms:
    node = host.SpeechToText(lib); node.set_language_model(model)
    want.append(gu.tokens_array(node.transcribe(b, "", 0))); node.close()
node = host.SpeechToText(lib); node.set_language_model(model)
lib.wmi_set_lockstep_exact(1)
got = node.transcribe_batch(pcms, "", 0)
for c, (g, w) in enumerate(zip(got, want)):
    g = gu.tokens_array(g)
    if g.shape != w.shape: print(c, "shape", g.shape, w.shape); continue
    print(c, secs[c], node.last_modes[c], "max dp", float(np.abs(g[:, 2] - w[:, 2]).max()) if len(g) else None)

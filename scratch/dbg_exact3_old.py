import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from godot_whisper_amd import host, runtime, synth
import golden_util as gu
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model = synth.make_model("micro.en", seed=2024)
secs = [12.0] * 3
pcms = [synth.make_pcm(s, seed=100 + i, gate=(i % 3 == 1)) for i, s in enumerate(secs)]
singles = []
for b in pcms:
    node = host.SpeechToText(lib); node.set_language_model(model)
    w = gu.tokens_array(node.transcribe(b, "", 0))
    ck = runtime.get_tensor(lib, node.ctx, "cross_k"); x = runtime.get_tensor(lib, node.ctx, "enc_x")
    singles.append((w, ck.copy(), x.copy())); node.close()
node = host.SpeechToText(lib); node.set_language_model(model)
lib.wmi_set_lockstep_exact(1)
got = node.transcribe_batch(pcms, "", 0)
bk = None
nb = len(pcms)
for c in range(nb):
    w, ck, x = singles[c]
    g = gu.tokens_array(got[c])
    print(c, "dp", float(np.abs(g[:, 2] - w[:, 2]).max()))

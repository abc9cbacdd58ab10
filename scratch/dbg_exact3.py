import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from godot_whisper_amd import host, runtime, synth
import golden_util as gu
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model = synth.make_model("micro.en", seed=2024)
secs = [12.0] * int(os.environ.get("NCH", "3"))
pcms = [synth.make_pcm(s, seed=100 + i, gate=(i % 3 == 1)) for i, s in enumerate(secs)]
singles = []
for b in pcms:
    node = host.SpeechToText(lib); node.set_language_model(model)
    w = gu.tokens_array(node.transcribe(b, "", 0))
    ck = runtime.get_tensor(lib, node.ctx, "cross_k"); cv = runtime.get_tensor(lib, node.ctx, "cross_v"); x = runtime.get_tensor(lib, node.ctx, "enc_x")
    singles.append((w, ck.copy(), cv.copy(), x.copy())); node.close()
node = host.SpeechToText(lib); node.set_language_model(model)
lib.wmi_set_lockstep_exact(1)
got = node.transcribe_batch(pcms, "", 0)
bk = runtime.get_tensor(lib, node.ctx, "batch_cross_k"); bv = runtime.get_tensor(lib, node.ctx, "batch_cross_v"); bx = runtime.get_tensor(lib, node.ctx, "batch_enc_x")
L, T, S, nb = 3, 1500, 128, len(pcms)
bk = bk.reshape(L, nb, T, S); bv = bv.reshape(L, nb, T, S); bx = bx.reshape(nb, T, S)
for c in range(nb):
    w, ck, cv, x = singles[c]
    g = gu.tokens_array(got[c])
    dk = np.abs(bk[:, c] - ck.reshape(L, T, S)); dv = np.abs(bv[:, c] - cv.reshape(L, T, S)); dx = np.abs(bx[c] - x.reshape(T, S))
    print(c, "dp %.2e" % float(np.abs(g[:, 2] - w[:, 2]).max()), "| cross_k max %.2e n %d t %s" % (dk.max(), (dk > 0).sum(), np.argwhere(dk.max(axis=(0, 2)) > 0)[:4].ravel().tolist()),
          "| cross_v max %.2e n %d" % (dv.max(), (dv > 0).sum()), "| enc_x max %.2e n %d" % (dx.max(), (dx > 0).sum()))

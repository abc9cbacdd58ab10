import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as e
e.load_package(); e.load_oracle()
from godot_whisper_amd import runtime, synth, host
from oracle import port
import golden_util as gu, stage_compare as sc
lib = runtime.require_gpu(); runtime.silence_logs(lib)
G = np.load(gu.GOLDEN / "hotpath.npz")
want = G["en30/full_default_greedy/tokens"]
model, pcm, actx = gu.case_inputs("en30")
prod = sc.ProductSide(lib, model); chk = port.PortSide(model)
prod.mel(pcm); chk.mel(pcm); chk.encode(0, 0); prod.encode(0, 0)
lr = chk.decode([50257], 0); lp = prod.decode([50257], 0)
worst = []
for i in range(0, 112):
    tok = int(want[i, 0])
    lr = chk.decode([tok], 1 + i); lp = prod.decode([tok], 1 + i)
    st = sc.err_stats(lp, lr)
    worst.append((st["rms_rel"], i))
    if i >= 100: print(i, tok, "rms_rel %.5f max %.4f" % (st["rms_rel"], st["max_abs"]), "argmax", int(np.argmax(lr)), int(np.argmax(lp)))
print("worst", sorted(worst)[-3:])

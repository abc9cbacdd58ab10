import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as e
e.load_package(); e.load_oracle()
from godot_whisper_amd import runtime, synth, host
from oracle import port
import golden_util as gu, stage_compare as sc
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model, pcm, actx = gu.case_inputs("en30")
prod = sc.ProductSide(lib, model); chk = port.PortSide(model)
prod.mel(pcm); chk.mel(pcm)
for off in (0, 2400, 2890):
    er = chk.encode(off, 0); ep = prod.encode(off, 0)
    print("offset", off, {k: round(sc.err_stats(ep[k], er[k])["rms_rel"], 6) for k in er})
    sot = 50257
    lr = chk.decode([sot], 0); lp = prod.decode([sot], 0)
    st = sc.err_stats(lp, lr); print("   prompt logits", st["max_abs"], st["rms_rel"], int(np.argmax(lr)), int(np.argmax(lp)))
    tok = int(np.argmax(lr[:50256]))
    lr = chk.decode([tok], 1); lp = prod.decode([tok], 1)
    st = sc.err_stats(lp, lr); print("   step logits", st["max_abs"], st["rms_rel"])

import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import ctypes as C
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from godot_whisper_amd import host, runtime, synth
from oracle import port
import stage_compare as sc
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model = synth.make_model("micro", seed=2024)
gate = os.environ.get("GATE", "1") == "1"
pcm = synth.make_pcm(15.0, seed=110, gate=gate)
prod = sc.ProductSide(lib, model); chk = port.PortSide(model)
buf = (C.c_int32 * 64)()
n = lib.whisper_tokenize(prod.ctx, b" Hello, world! It's 42.", buf, 64)
sot = lib.whisper_token_sot(prod.ctx)
toks = [lib.whisper_token_prev(prod.ctx)] + list(buf[:n]) + [sot, sot + 1, lib.whisper_token_transcribe(prod.ctx)]
print("prompt", len(toks), toks)
prod.mel(pcm); chk.mel(pcm)
ep = prod.encode(0, 0); er = chk.encode(0, 0)
for k in ("embd_enc", "cross_k", "cross_v"):
    print(k, sc.err_stats(ep[k], er[k]))
lr = chk.decode(toks, 0)
lp = prod.decode(toks, 0)
print("batch13 vs port:", sc.err_stats(lp, lr), int(np.argmax(lp)), int(np.argmax(lr)))
for nn in (2, 4, 8, 9, 10, 12):
    lr2 = chk.decode(toks[:nn], 0); lp2 = prod.decode(toks[:nn], 0)
    print(f"batch{nn} vs port:", sc.err_stats(lp2, lr2))
# token by token on the product
for i, t in enumerate(toks):
    l1 = prod.decode([t], i)
print("stepwise vs port:", sc.err_stats(l1, lr))

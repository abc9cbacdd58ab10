import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import ctypes as C
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from godot_whisper_amd import host, runtime, synth
from oracle import port
import stage_compare as sc
import golden_util as gu
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model = synth.make_model("micro", seed=2024)
pcm = synth.make_pcm(15.0, seed=110, gate=True)
def show(tag, r):
    a = gu.tokens_array(r)
    print(tag, [(int(x[0]), round(float(x[2]), 4)) for x in a[:4]])
node = host.SpeechToText(lib); node.set_language_model(model)
for i in range(3):
    show(f"alone call {i}", node.transcribe(pcm, " Hello, world! It's 42.", 0))
node.close()
if os.environ.get("STAGES"):
    prod = sc.ProductSide(lib, model); chk = port.PortSide(model)
    buf = (C.c_int32 * 64)()
    n = lib.whisper_tokenize(prod.ctx, b" Hello, world! It's 42.", buf, 64)
    sot = lib.whisper_token_sot(prod.ctx)
    toks = [lib.whisper_token_prev(prod.ctx)] + list(buf[:n]) + [sot, sot + 1, lib.whisper_token_transcribe(prod.ctx)]
    prod.mel(pcm); chk.mel(pcm); prod.encode(0, 0); chk.encode(0, 0)
    lr = chk.decode(toks, 0)
    for split in (11, 10, 9, 8, 5):
        prod.decode(toks[:split], 0)
        for i in range(split, len(toks)):
            lp = prod.decode([toks[i]], i)
        print(f"split {split}+1s vs port:", sc.err_stats(lp, lr))

import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as e
e.load_package(); e.load_oracle()
from godot_whisper_amd import runtime, synth
from oracle import reflib
import stage_compare as sc
lib = runtime.require_gpu(); runtime.silence_logs(lib)
rl = reflib.lib()
mb = synth.make_model("micro.en", seed=1234); pcm = synth.make_pcm(30.0, seed=1234)
prod = sc.ProductSide(lib, mb); ref = sc.RefSide(rl, mb)
ref.mel(pcm); prod.mel(pcm); ref.encode(); prod.encode()
sot = rl.whisper_token_sot(ref.ctx)
for n in (1, 5):
    toks = [sot] + [int(x) for x in (np.arange(n - 1) * 997 + 1000)]
    a = ref.decode(toks, 0); b = prod.decode(toks, 0)
    print("mask", os.environ.get("WMI_GEMM_MASK"), n, sc.err_stats(b, a)["rms_rel"])

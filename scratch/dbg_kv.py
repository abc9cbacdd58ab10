import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as e
e.load_package(); e.load_oracle()
from godot_whisper_amd import runtime, synth
import stage_compare as sc
lib = runtime.require_gpu(); runtime.silence_logs(lib)
mb = synth.make_model("micro.en", seed=1234); pcm = synth.make_pcm(30.0, seed=1234)
prod = sc.ProductSide(lib, mb)
prod.mel(pcm); prod.encode()
sot = lib.whisper_token_sot(prod.ctx)
prod.decode([sot, 1000, 2000], 0)
S = prod.S; L = prod.L
k = prod.tensor("self_k").reshape(L, -1, S); v = prod.tensor("self_v").reshape(L, -1, S)
np.save("/root/repo/gpurun_out/k_%s.npy" % os.environ.get("WMI_GEMM_MASK", "0"), k[:, :4])
np.save("/root/repo/gpurun_out/v_%s.npy" % os.environ.get("WMI_GEMM_MASK", "0"), v[:, :4])
print("k layer0 cell0[:6]", k[0, 0, :6], "v", v[0, 0, :6])
print("k layer0 cell1[:6]", k[0, 1, :6], "v", v[0, 1, :6])
print("k l0 c0 [128-6:]", k[0, 0, -6:], "v", v[0, 0, -6:])

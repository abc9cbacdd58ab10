import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
tag = sys.argv[1]
model = synth.make_model("micro.en", seed=2024)
pcms = [synth.make_pcm(12.0, seed=100 + i, gate=(i % 3 == 1)) for i in range(3)]
node = host.SpeechToText(lib); node.set_language_model(model)
node.transcribe(pcms[2], "", 0)
out = {}
for nm in ("embd_conv", "enc_x", "embd_enc", "cross_k", "cross_v"):
    out[nm] = runtime.get_tensor(lib, node.ctx, nm)
np.savez(f"gpurun_out/dump_{tag}.npz", **out)
print("saved", tag)

import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as e
e.load_package(); e.load_oracle()
from godot_whisper_amd import runtime, synth
import stage_compare as sc
lib = runtime.require_gpu(); runtime.silence_logs(lib)
mb = synth.make_model("micro.en", seed=1234)
prod = sc.ProductSide(lib, mb)
for op in (0, 1, 2, 4, 5):
    for n in (1, 3, 8):
        print("op", op, "n", n, "maxdiff", lib.wmi_selftest_proj(prod.ctx, op, n, 0))

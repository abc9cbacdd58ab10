"""Scratch: two 8-chunk lock-step calls of base.en (for PMC collection on the M = 12 000 encoder GEMMs and attention)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model("base.en", seed=1234))
params = node.full_params("", 0)
nb = int(os.environ.get("NB", "8"))
pcm = [synth.make_pcm(30.0, seed=100 + i) for i in range(nb)]
ptrs = (C.c_void_p * nb)(*[p.ctypes.data for p in pcm]); lens = (C.c_int * nb)(*[p.size for p in pcm])
for _ in range(int(os.environ.get("REPS", "2"))):
    assert lib.wmi_full_batch(node.ctx, params, ptrs, lens, nb, 0) == 0
node.close()

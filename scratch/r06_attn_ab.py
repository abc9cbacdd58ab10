"""Scratch (round 6): encoder attention, one layer, over the lock-step buffers (wmi_bench_kernel 5) and over one chunk (2): microseconds."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model("base.en", seed=1234))
params = node.full_params("", 0)
nb = 8
pcm = [synth.make_pcm(30.0, seed=100 + i) for i in range(nb)]
ptrs = (C.c_void_p * nb)(*[p.ctypes.data for p in pcm]); lens = (C.c_int * nb)(*[p.size for p in pcm])
for _ in range(3): assert lib.wmi_full_batch(node.ctx, params, ptrs, lens, nb, 0) == 0
node.transcribe(pcm[0], "", 0)
r8 = [lib.wmi_bench_kernel(node.ctx, 5, 200) for _ in range(5)]
r1 = [lib.wmi_bench_kernel(node.ctx, 2, 400) for _ in range(5)]
print(os.environ.get("TAG", ""), "attention layer, 8 chunks:", " ".join("%.2f" % v for v in r8), "us | one chunk:", " ".join("%.2f" % v for v in r1), "us")

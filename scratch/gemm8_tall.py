"""Scratch: mlp.0 at M = 12 000 (8 lock-step chunks of base.en) under the current WMI_GEMM_* knobs; encoder time of the batch."""
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
for _ in range(3):
    assert lib.wmi_full_batch(node.ctx, params, ptrs, lens, nb, 0) == 0
t4 = (C.c_int64 * 4)(); ns = C.c_int32(); lib.wmi_get_batch_timings(node.ctx, t4, C.byref(ns))
us = [lib.wmi_bench_kernel(node.ctx, 4, 200) for _ in range(3)]
print("TALL", os.environ.get("WMI_GEMM_TALL"), "STAGGER", os.environ.get("WMI_GEMM_STAGGER"), "mlp.0 x 8 us", [round(u, 2) for u in us], "TFLOP/s", round(2 * 12000 * 2048 * 512 / min(us) / 1e6, 1), "batch encode ms (3 calls)", round(t4[1] / 1e3, 3), flush=True)
node.close()

"""Scratch: encoder attention / GEMM probes (one chunk and 8 lock-step chunks), us per launch."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model(os.environ.get("SHAPE", "base.en"), seed=1234))
params = node.full_params("", 0)
nb = 8
pcm = [synth.make_pcm(30.0, seed=100 + i) for i in range(nb)]
ptrs = (C.c_void_p * nb)(*[p.ctypes.data for p in pcm]); lens = (C.c_int * nb)(*[p.size for p in pcm])
node.transcribe(pcm[0], "", 0)
assert lib.wmi_full_batch(node.ctx, params, ptrs, lens, nb, 0) == 0
lib.wmi_bench_kernel.restype = C.c_double
for which, name, it in ((2, "attention layer, 1 chunk", 100), (5, "attention layer, 8 chunks", 50), (0, "mlp.0 GEMM, 1 chunk", 200), (4, "mlp.0 GEMM, 8 chunks", 100)):
    print("%-28s %8.2f us" % (name, lib.wmi_bench_kernel(node.ctx, which, it)))
lib.wmi_bench_kernel(node.ctx, 7, 1); lib.wmi_bench_kernel(node.ctx, 8, 1)
t6 = (C.c_int64 * 6)(); n5 = (C.c_int32 * 5)()
lib.whisper_reset_timings(node.ctx)
for _ in range(20): node.transcribe(pcm[0], "", 0)
lib.wmi_get_timings(node.ctx, t6, n5)
print("encode ms (1 chunk): %.4f" % (t6[1] / 1e3 / n5[0]))
node.close()

"""Lock-step decode step on the GPU alone, whole and one kernel kind at a time (bench kernel 20 + rows, WMI_STEP_MASK)."""
import ctypes as C, os, sys
sys.path.insert(0, ".")
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
NB = int(os.environ.get("NB", "8"))
model = synth.make_model("base.en", seed=1234)
node = host.SpeechToText(lib); node.set_language_model(model)
pcms = [synth.make_pcm(30.0, seed=1234 + i) for i in range(NB)]
for _ in range(2):
    node.transcribe_batch(pcms, "", 0)
lib.wmi_bench_kernel.restype = C.c_double
libc = C.CDLL(None)
def chain(mask, it=60):
    libc.setenv(b"WMI_STEP_MASK", str(mask).encode(), 1)
    return lib.wmi_bench_kernel(node.ctx, 20 + NB, it)
print("rows = %d; whole step: %.1f us" % (NB, chain(0x7ff)))
kinds = [("embed", 1, 1), ("qkv (LN)", 2, 6), ("self-attention rows", 4, 6), ("out", 8, 6), ("cross scores + P.V", 16, 12), ("combine", 32, 6),
         ("cross out", 64, 6), ("mlp.0 (LN, GELU)", 128, 6), ("mlp.2 (K = 4S)", 256, 6), ("logits", 512, 1), ("filters (2 kernels)", 1024, 2)]
tot = 0.0
for name, m, n in kinds:
    t = chain(m); tot += t
    print("%-22s %6.1f us per step = %5.2f us per launch (%d launches)" % (name, t, t / n, n))
print("sum of kinds: %.1f us" % tot)

"""Decode step on the GPU alone (no host round trip), whole and one kernel kind at a time (bench kernel 20 + WMI_STEP_MASK)."""
import ctypes as C, os, sys, time
sys.path.insert(0, ".")
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
model = synth.make_model(os.environ.get("SHAPE", "base.en"), seed=1234)
node = host.SpeechToText(lib); node.set_language_model(model)
pcm = synth.make_pcm(30.0, seed=1234)
for _ in range(3):
    node.transcribe(pcm, "", 0)
lib.wmi_bench_kernel.restype = C.c_double
libc = C.CDLL(None)
def chain(mask, it=100):
    libc.setenv(b"WMI_STEP_MASK", str(mask).encode(), 1)
    return lib.wmi_bench_kernel(node.ctx, 20, it)
full = chain(0x1ff)
print("whole step: %.1f us" % full)
kinds = [("embed", 1, 1), ("qkv (LN)", 2, 6), ("self-attn + out", 4, 6), ("cross-attention (fused)", 8, 6),
         ("combine + cross out", 16, 6), ("mlp.0 (LN, GELU)", 32, 6), ("mlp.2 (K = 4S)", 64, 6), ("logits", 128, 1), ("filters (2 kernels)", 256, 2)]
tot = 0.0
for name, m, n in kinds:
    t = chain(m)
    print("%-22s %6.1f us per step = %5.2f us per launch (%d launches)" % (name, t, t / n, n))
    if not name.startswith("  "): tot += t
print("sum of kinds: %.1f us" % tot)
for w in (10, 11, 12):
    print("touch chain %d: %.2f us" % (w, lib.wmi_bench_kernel(node.ctx, w, 500)))

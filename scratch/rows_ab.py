"""Scratch: the lock-step step chain at 8 / 16 rows (wmi_bench_kernel 20 + rows) with knobs toggled inside ONE process."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
import torch
lib = runtime.require_gpu(); runtime.silence_logs(lib)
libc = C.CDLL(None)
node = host.SpeechToText(lib); node.set_language_model(synth.make_model("base.en", seed=1234))
params = node.full_params("", 0)
knob, val = (sys.argv[1].split("=") + ["1"])[:2]
for nb in (8, 16):
    pcm = [torch.from_numpy(synth.make_pcm(30.0, seed=1234 + i)).cuda() for i in range(nb)]
    ptrs = (C.c_void_p * nb)(*[t.data_ptr() for t in pcm]); lens = (C.c_int * nb)(*[t.numel() for t in pcm])
    for _ in range(3): assert lib.wmi_full_batch(node.ctx, params, ptrs, lens, nb, 1) == 0
    a, b = [], []
    for rep in range(4):
        libc.unsetenv(knob.encode()); a.append(lib.wmi_bench_kernel(node.ctx, 20 + nb, 100))
        libc.setenv(knob.encode(), val.encode(), 1); b.append(lib.wmi_bench_kernel(node.ctx, 20 + nb, 100)); libc.unsetenv(knob.encode())
    print(nb, "rows step chain us | default:", " ".join("%.1f" % v for v in a), "| %s=%s:" % (knob, val), " ".join("%.1f" % v for v in b), flush=True)
node.close()

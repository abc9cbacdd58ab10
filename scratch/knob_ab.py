"""Scratch: the one-row step chain on the GPU (wmi_bench_kernel 20) with an environment knob off / on, alternating inside ONE process.
   python scratch/knob_ab.py KNOB[=VALUE] [shape ...]      (knobs that the library reads per enqueue)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
libc = C.CDLL(None)
knob, val = (sys.argv[1].split("=") + ["1"])[:2]
for shape in (sys.argv[2:] or ["base.en"]):
    node = host.SpeechToText(lib); node.set_language_model(synth.make_model(shape, seed=1234))
    pcm = synth.make_pcm(30.0, seed=1234)
    for _ in range(6): node.transcribe(pcm, "", 0)
    libc.setenv(b"WMI_STEP_MASK", b"0x1ff", 1)
    a, b = [], []
    for rep in range(5):
        libc.unsetenv(knob.encode()); a.append(lib.wmi_bench_kernel(node.ctx, 20, 200))
        libc.setenv(knob.encode(), val.encode(), 1); b.append(lib.wmi_bench_kernel(node.ctx, 20, 200)); libc.unsetenv(knob.encode())
    print(shape, "step chain us | default:", " ".join("%.2f" % v for v in a), "| %s=%s:" % (knob, val), " ".join("%.2f" % v for v in b), flush=True)
    node.close()

"""Scratch: the one-row step chain on the GPU (wmi_bench_kernel 20) with the MLP as one launch vs two, alternating inside ONE process."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
libc = C.CDLL(None)
for shape in os.environ.get("SHAPES", "base.en,tiny.en").split(","):
    node = host.SpeechToText(lib); node.set_language_model(synth.make_model(shape, seed=1234))
    pcm = synth.make_pcm(30.0, seed=1234)
    for _ in range(6): node.transcribe(pcm, "", 0)
    libc.setenv(b"WMI_STEP_MASK", b"0x1ff", 1)
    res = {"x8": [], "x4": []}
    for rep in range(5):
        libc.unsetenv(b"WMI_XATTN_WPB"); res["x8"].append(lib.wmi_bench_kernel(node.ctx, 20, 300))
        libc.setenv(b"WMI_XATTN_WPB", b"4", 1); res["x4"].append(lib.wmi_bench_kernel(node.ctx, 20, 300)); libc.unsetenv(b"WMI_XATTN_WPB")
    print(shape, "step chain us | cross-attention, two slices per 8-wavefront workgroup:", " ".join("%.2f" % v for v in res["x8"]), "| one slice per workgroup:", " ".join("%.2f" % v for v in res["x4"]), flush=True)
    res = {"sa8": [], "sa4": []}
    for rep in range(5):
        libc.unsetenv(b"WMI_SA_WPB"); res["sa8"].append(lib.wmi_bench_kernel(node.ctx, 20, 300))
        libc.setenv(b"WMI_SA_WPB", b"4", 1); res["sa4"].append(lib.wmi_bench_kernel(node.ctx, 20, 300)); libc.unsetenv(b"WMI_SA_WPB")
    print(shape, "step chain us | self-attention + out on 8 wavefronts (one head each):", " ".join("%.2f" % v for v in res["sa8"]), "| 4 wavefronts:", " ".join("%.2f" % v for v in res["sa4"]), flush=True)
    res = {"pair8": [], "pair4": [], "two": []}
    for rep in range(5):
        libc.unsetenv(b"WMI_NO_MLP_PAIR"); libc.unsetenv(b"WMI_PAIR_WPB"); res["pair8"].append(lib.wmi_bench_kernel(node.ctx, 20, 300))
        libc.setenv(b"WMI_PAIR_WPB", b"4", 1); res["pair4"].append(lib.wmi_bench_kernel(node.ctx, 20, 300)); libc.unsetenv(b"WMI_PAIR_WPB")
        libc.setenv(b"WMI_NO_MLP_PAIR", b"1", 1); res["two"].append(lib.wmi_bench_kernel(node.ctx, 20, 300))
    libc.unsetenv(b"WMI_NO_MLP_PAIR")
    print(shape, "step chain us | one launch per MLP, 8-wavefront workgroups:", " ".join("%.2f" % v for v in res["pair8"]), "| 4-wavefront:", " ".join("%.2f" % v for v in res["pair4"]), "| two launches:", " ".join("%.2f" % v for v in res["two"]), flush=True)
    node.close()

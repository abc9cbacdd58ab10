"""Scratch: encoder time of large-v3 q5_1 under the current environment knobs."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
entry.load_package(); entry.load_oracle()
from godot_whisper_amd import host, runtime, synth
from oracle import reflib
import test_gpu_large_v3 as tl
lib = runtime.require_gpu(); runtime.silence_logs(lib)
cache = "/tmp/v3q.bin"
if os.path.exists(cache): m = open(cache, "rb").read()
else:
    m = tl._ref_quantize_model(reflib.lib(), synth.make_model("large-v3", seed=2024), "q5_1"); open(cache, "wb").write(m)
node = host.SpeechToText(lib); node.set_language_model(m)
pcm = synth.make_pcm(30.0, seed=7)
assert lib.whisper_pcm_to_mel(node.ctx, pcm.ctypes.data_as(C.POINTER(C.c_float)), pcm.size, 4) == 0
lib.whisper_encode(node.ctx, 0, 4)
t0 = time.perf_counter(); n = 5
for _ in range(n): lib.whisper_encode(node.ctx, 0, 4)
print("BM", os.environ.get("WMI_QGEMM_BM"), "NST", os.environ.get("WMI_QGEMM_NST"), "F16_ROWS", os.environ.get("WMI_QGEMM_F16_ROWS"), "encode ms", round((time.perf_counter() - t0) / n * 1e3, 2), "fc1 us", round(lib.wmi_bench_kernel(node.ctx, 0, 50), 1), flush=True)
node.close()
